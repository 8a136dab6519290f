"""Shared test helpers: golden fixtures, config building, state comparison."""
import glob
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

INT_FIELDS = ["stone", "wood", "house_owner", "loc_r", "loc_c", "inv_res", "esc_res",
              "cda_n_bids", "cda_n_asks", "cda_n_orders", "cda_bid_hist", "cda_ask_hist",
              "tax_cycle_pos", "tax_last_completions", "tax_rate_idx", "timestep", "completions", "auto_warmup", "mt_pos",
              "labor_first_step"]
F64_FIELDS = ["inv_coin", "esc_coin", "labor", "build_payment", "build_skill",
              "bonus_gather_prob", "util", "cda_price_history", "tax_last_coin",
              "tax_last_income", "tax_last_marginal_rate", "tax_total_collected", "skill",
              "production"]


def golden_names():
    """Gather-trade-build / one-step-economy fixtures (the COVID ones have their own format)."""
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz")))
    return [n for n in names if not n.startswith("c4_covid")]


def covid_golden_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "c4_covid*.npz")))


def load_covid_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        g = {k: z[k] for k in z.files}
    g["cfg"] = json.loads(str(g["config_json"]))
    g["cfg"]["components"] = [tuple(c) for c in g["cfg"]["components"]]
    return g


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        g = {k: z[k] for k in z.files}
    g["cfg"] = json.loads(str(g["cfg_json"]))
    if "construction_seed" in g:
        # scenarios that draw from the GLOBAL NumPy stream in their constructor (SplitLayout's
        # ranked skills): make_env re-seeds the global stream the way the generator did
        g["cfg"]["_construction_seed"] = int(g["construction_seed"])
    if "s0_skill" in g:
        # SimpleLabor estimates its skills from the GLOBAL NumPy stream at construction
        # (simple_labor.py:66-74); the fixture carries the values the reference drew.
        for comp in g["cfg"]["components"]:
            if comp[0] == "SimpleLabor":
                comp[1]["skills"] = [float(x) for x in g["s0_skill"]]
    return g


def make_env(cfg, n_envs=1, **extra):
    """Host env (no device work happens until reset/step)."""
    from ai_economist_amd import foundation

    kw = dict(cfg)
    cseed = kw.pop("_construction_seed", None)
    if cseed is not None:
        np.random.seed(cseed)
    scenario = kw.pop("scenario_name")
    kw["components"] = [tuple(c) for c in kw["components"]]
    kw.update(extra)
    return foundation.make_env_instance(scenario, n_envs=n_envs, **kw)


def state_from_golden(g, prefix, t=None):
    out = {}
    for k, v in g.items():
        if k.startswith(prefix):
            out[k[len(prefix):]] = v if t is None else v[t]
    return out


def book_equal(n_a, book_a, n_b, book_b):
    """Order books: only the first n entries of each [R, M] row are meaningful."""
    for r in range(2):
        if int(n_a[r]) != int(n_b[r]):
            return False
        if not np.array_equal(book_a[r, : n_a[r]], book_b[r, : n_b[r]]):
            return False
    return True


def compare_state(got, want, where="", f64_tol=1e-9):
    """got/want: {field: array} for ONE replica.  Integer state bit-exact, f64 within tol."""
    for k in INT_FIELDS:
        if k in want and k in got:
            assert np.array_equal(np.asarray(got[k]), np.asarray(want[k])), (
                "%s: integer field %s differs\n got=%s\nwant=%s" % (where, k, got[k], want[k]))
    if "cda_bids" in want and "cda_bids" in got:
        assert book_equal(got["cda_n_bids"], got["cda_bids"], want["cda_n_bids"], want["cda_bids"]), (
            "%s: bid book differs" % where)
        assert book_equal(got["cda_n_asks"], got["cda_asks"], want["cda_n_asks"], want["cda_asks"]), (
            "%s: ask book differs" % where)
    for k in F64_FIELDS:
        if k in want and k in got:
            np.testing.assert_allclose(np.asarray(got[k]), np.asarray(want[k]), rtol=f64_tol,
                                       atol=f64_tol, err_msg="%s: f64 field %s" % (where, k))


def oracle_host_pre_reset(env, oracle, which=None):
    """What BaseEnvironment.host_pre_reset does on the device, on an OracleEnv: scenarios
    with a host-side reset part (uniform/...: a fresh layout from the replica's own MT19937
    stream) run it here before oracle.reset()."""
    if not hasattr(env, "generate_layout"):
        return
    which = range(oracle.E) if which is None else which
    rs = np.random.RandomState()
    for e in which:
        rs.set_state(("MT19937", oracle.t["mt"][e].copy(), int(oracle.t["mt_pos"][e]),
                      int(oracle.t["mt_has_gauss"][e]), float(oracle.t["mt_gauss"][e])))
        oracle.t["cell_flags"][e] = env.generate_layout_flags(rs)
        st = rs.get_state()
        oracle.t["mt"][e] = st[1]
        oracle.t["mt_pos"][e] = st[2]
        oracle.t["mt_has_gauss"][e] = st[3]
        oracle.t["mt_gauss"][e] = st[4]
