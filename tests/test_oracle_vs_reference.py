"""CPU tests that need the live reference (/root/reference, skipped elsewhere): the C
restatement is stepped side by side with the UNMODIFIED reference env on configurations
the committed fixtures do not cover (multi-action agents, single-action planner, no
observation scaling, other planner reward types, other component orders, fixed tax
models, disabled taxes...)."""
import numpy as np
import pytest

from helpers import compare_state, make_env

pytestmark = pytest.mark.reference

BASE = dict(scenario_name="layout_from_file/simple_wood_and_stone", n_agents=4, world_size=[25, 25],
            episode_length=80, starting_agent_coin=12, resource_regen_prob=0.08,
            env_layout_file="uniform_25x25_25each_65clump.txt")
GTB = [["Build", {}], ["ContinuousDoubleAuction", {"max_num_orders": 4, "order_duration": 15}],
       ["Gather", {}], ["PeriodicBracketTax", {"period": 20}]]

VARIANTS = {
    "multi_action_agents": dict(components=GTB, multi_action_mode_agents=True),
    "single_action_planner": dict(components=GTB, multi_action_mode_planner=False),
    "no_obs_scaling": dict(components=GTB, allow_observation_scaling=False),
    "inv_income_utility": dict(components=GTB, planner_reward_type="inv_income_weighted_utility"),
    "inv_income_coin": dict(components=GTB, planner_reward_type="inv_income_weighted_coin_endowments",
                            mixing_weight_gini_vs_coin=0.3),
    "order_gather_first": dict(components=[GTB[2], GTB[1], GTB[0], GTB[3]]),
    "us_federal_fixed": dict(components=GTB[:3] + [["PeriodicBracketTax", {
        "period": 10, "tax_model": "us-federal-single-filer-2018-scaled"}]]),
    "fixed_bracket_linear": dict(components=GTB[:3] + [["PeriodicBracketTax", {
        "period": 10, "tax_model": "fixed-bracket-rates", "bracket_spacing": "linear", "n_brackets": 4,
        "top_bracket_cutoff": 30, "fixed_bracket_rates": [0.0, 0.1, 0.3, 0.6]}]]),
    # tax annealing (the paper's phase-2 setting): planner rate actions unmasked as episodes
    # complete; 80-step episodes => the limit moves twice within the test
    "annealed_wrapper": dict(components=GTB[:3] + [["PeriodicBracketTax", {
        "period": 10, "rate_disc": 0.1, "tax_annealing_schedule": [-1, 0.35]}]]),
    "annealed_wrapper_single_planner": dict(multi_action_mode_planner=False, components=GTB[:3] + [[
        "PeriodicBracketTax", {"period": 10, "n_brackets": 3, "tax_annealing_schedule": [0, 0.4]}]]),
    "annealed_us_federal": dict(components=GTB[:3] + [["PeriodicBracketTax", {
        "period": 10, "tax_model": "us-federal-single-filer-2018-scaled", "tax_annealing_schedule": [-1, 0.3]}]]),
    "full_observability": dict(components=GTB, full_observability=True, world_size=[25, 25]),
    "full_observability_no_tax_planner_blind": dict(components=GTB[:3], full_observability=True,
                                                    planner_gets_spatial_info=False),
    "taxes_disabled": dict(components=GTB[:3] + [["PeriodicBracketTax", {"period": 10, "disable_taxes": True}]]),
    "log_brackets_wrapper": dict(components=GTB[:3] + [["PeriodicBracketTax", {
        "period": 10, "bracket_spacing": "log", "n_brackets": 5, "top_bracket_cutoff": 40, "rate_disc": 0.1}]]),
    "no_cda_obs_range3": dict(components=[GTB[0], GTB[2], GTB[3]], mobile_agent_observation_range=3),
    "energy_decay_warmup": dict(components=GTB, energy_warmup_constant=3, energy_warmup_method="decay",
                                 isoelastic_eta=0.5),
    "wealth_redistribution": dict(components=GTB[:3] + [["WealthRedistribution", {}]]),
    "wealth_redistribution_then_tax": dict(components=GTB[:3] + [["WealthRedistribution", {}], GTB[3]],
                                           n_agents=9),
    "six_agents_40x40": dict(components=GTB, n_agents=6, world_size=[40, 40],
                             env_layout_file="quadrant_40x40_50each.txt"),
}


def _ref_env(cfg):
    from ref_harness import load_reference_foundation

    foundation = load_reference_foundation()
    kw = dict(cfg)
    name = kw.pop("scenario_name")
    kw["components"] = [tuple(c) for c in kw["components"]]
    return foundation.make_env_instance(name, **kw)


def _random_actions(env, rng, multi_a, multi_p):
    ag = env.world.agents[0]
    acts = {}
    n = env.n_agents
    if multi_a:
        dims = [ag.action_dim[nm] for nm in ag._action_names]
        arr = np.zeros((n, len(dims)), np.int32)
        for i in range(n):
            # mostly one sub-action at a time, sometimes several (incl. Buy and Sell together)
            for s, d in enumerate(dims):
                if rng.rand() < 0.3:
                    arr[i, s] = rng.randint(0, d)
            acts[str(i)] = [int(x) for x in arr[i]]
    else:
        A = ag.action_spaces
        arr = np.zeros((n, 1), np.int32)
        for i in range(n):
            u = rng.rand()
            a = rng.randint(A - 4, A) if (u < 0.45 and "Gather" in ag._action_names) else rng.randint(0, A)
            arr[i, 0] = a
            acts[str(i)] = int(a)
    pl = env.world.planner
    names = [nm for nm in pl._action_names if nm != "PassiveAgentPlaceholder"]
    if names:
        d = pl.action_dim[names[0]]
        if multi_p:
            pa = rng.randint(0, d, size=len(names)).astype(np.int32)
            acts["p"] = [int(x) for x in pa]
        else:
            pa = np.array([rng.randint(0, pl.action_spaces)], np.int32)
            acts["p"] = int(pa[0])
    else:
        pa = np.zeros(1, np.int32)
    return acts, arr, pa


def check_metrics(ref, host, o, where):
    """env.metrics: the host-side formulas (ai_economist_amd.foundation.metrics) on the oracle's
    state + episode accumulators against the reference's scenario + component metrics."""
    import warnings

    from ai_economist_amd.foundation.metrics import env_metrics

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # np.mean of the reference's still-empty lists
        want = ref.metrics
    got = env_metrics(host, dict(o.t))
    assert sorted(got) == sorted(want), (where, sorted(set(got) ^ set(want)))
    for k, v in want.items():
        g = float(got[k][0])
        if v is None or (isinstance(v, float) and np.isnan(v)) or np.isnan(float(v)):
            assert np.isnan(g), "%s: metric %s = %r, reference NaN" % (where, k, g)
        else:
            np.testing.assert_allclose(g, float(v), rtol=1e-9, atol=1e-12, err_msg="%s: metric %s" % (where, k))


@pytest.mark.parametrize("variant", sorted(VARIANTS))
def test_oracle_tracks_live_reference(variant):
    from oracle_lib import OracleEnv
    from ref_extract import extract_obs, extract_state, rewards_array

    cfg = dict(BASE)
    cfg.update(VARIANTS[variant])
    ref = _ref_env(cfg)
    host = make_env(cfg)
    o = OracleEnv(host.build_config(), host.layout_planes())
    np.random.seed(31)
    st = np.random.get_state()
    o.t["mt"][0] = st[1]
    o.t["mt_pos"][0] = st[2]
    obs = ref.reset()
    o.reset()
    rng = np.random.RandomState(5)
    multi_a = bool(cfg.get("multi_action_mode_agents", False))
    multi_p = bool(cfg.get("multi_action_mode_planner", True))

    def check(where, obs, rew=None):
        compare_state({k: v[0] for k, v in o.t.items()}, extract_state(ref), where=where, f64_tol=1e-9)
        assert np.array_equal(o.t["mt"][0], np.random.get_state()[1]), where + ": MT19937 state"
        for k, want in extract_obs(ref, obs).items():
            got = o.t[k][0]
            if want.dtype.kind in "iu":
                assert np.array_equal(got, want), "%s: obs %s" % (where, k)
            else:
                np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-6, err_msg="%s: obs %s" % (where, k))
        if rew is not None:
            got = np.concatenate([o.t["rewards_a"][0], o.t["rewards_p"][[0]]])
            np.testing.assert_allclose(got, rewards_array(ref, rew), rtol=0, atol=1e-5, err_msg=where)
        check_metrics(ref, host, o, where)

    check(variant + " reset", obs)
    for t in range(170):
        acts, aa, pa = _random_actions(ref, rng, multi_a, multi_p)
        obs, rew, done, _ = ref.step(acts)
        o.step(aa[None], pa[None])
        check("%s step %d" % (variant, t + 1), obs, rew)
        assert bool(o.t["done"][0]) == bool(done["__all__"])
        if done["__all__"]:
            obs = ref.reset()
            o.reset()
            check("%s reset after step %d" % (variant, t + 1), obs)


@pytest.mark.parametrize("seed", range(24))
def test_oracle_tracks_live_reference_random_configs(seed):
    """Randomly drawn valid configurations (helpers.random_gtb_config): same side-by-side check."""
    from helpers import oracle_host_pre_reset, random_gtb_config
    from oracle_lib import OracleEnv
    from ref_extract import extract_obs, extract_state, rewards_array

    cfg = random_gtb_config(seed)
    np.random.seed(500 + seed)  # pareto/lognormal skill draws etc. come from the global stream
    ref = _ref_env(cfg)
    host = make_env(cfg)
    o = OracleEnv(host.build_config(), host.layout_planes())
    np.random.seed(31 + seed)
    st = np.random.get_state()
    o.t["mt"][0] = st[1]
    o.t["mt_pos"][0] = st[2]
    obs = ref.reset()
    oracle_host_pre_reset(host, o)
    o.reset()
    rng = np.random.RandomState(5)
    multi_a, multi_p = cfg["multi_action_mode_agents"], cfg["multi_action_mode_planner"]

    def check(where, obs, rew=None):
        compare_state({k: v[0] for k, v in o.t.items()}, extract_state(ref), where=where, f64_tol=1e-9)
        assert np.array_equal(o.t["mt"][0], np.random.get_state()[1]), where + ": MT19937 state"
        for k, want in extract_obs(ref, obs).items():
            got = o.t[k][0]
            if want.dtype.kind in "iu":
                assert np.array_equal(got, want), "%s: obs %s" % (where, k)
            else:
                np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-6, err_msg="%s: obs %s" % (where, k))
        if rew is not None:
            got = np.concatenate([o.t["rewards_a"][0], o.t["rewards_p"][[0]]])
            np.testing.assert_allclose(got, rewards_array(ref, rew), rtol=0, atol=1e-5, err_msg=where)
        check_metrics(ref, host, o, where)

    where0 = "random config %d %r" % (seed, cfg)
    check(where0 + " reset", obs)
    for t in range(2 * cfg["episode_length"] + 7):
        acts, aa, pa = _random_actions(ref, rng, multi_a, multi_p)
        obs, rew, done, _ = ref.step(acts)
        o.step(aa[None], pa[None])
        check("%s step %d" % (where0, t + 1), obs, rew)
        assert bool(o.t["done"][0]) == bool(done["__all__"])
        if done["__all__"]:
            obs = ref.reset()
            oracle_host_pre_reset(host, o)
            o.reset()
            check("%s reset after step %d" % (where0, t + 1), obs)


OSE_VARIANTS = {
    "c5_default": dict(n_agents=100),
    "coin_eq_40": dict(n_agents=40, planner_reward_type="coin_eq_times_productivity", mixing_weight_gini_vs_coin=0.25),
    "isoelastic_12_coin_eq": dict(n_agents=12, agent_reward_type="isoelastic_coin_minus_labor",
                                  planner_reward_type="coin_eq_times_productivity", isoelastic_eta=0.4),
    "single_action_planner_25": dict(n_agents=25, multi_action_mode_planner=False, labor_cost=0.5, labor_exponent=1.5),
    "wealth_redistribution_20": dict(n_agents=20, extra_components=[["WealthRedistribution", {}]]),
    "no_first_step_mask": dict(n_agents=30, labor_kw=dict(mask_first_step=False), episode_length=3),
}


@pytest.mark.parametrize("variant", sorted(OSE_VARIANTS))
def test_oracle_tracks_live_reference_one_step_economy(variant):
    from oracle_lib import OracleEnv
    from ref_extract import extract_obs, extract_state, rewards_array

    kw = dict(OSE_VARIANTS[variant])
    labor_kw = kw.pop("labor_kw", {})
    extra = kw.pop("extra_components", [])
    cfg = dict(scenario_name="one-step-economy", world_size=[1, 1], episode_length=kw.pop("episode_length", 2),
               components=[["SimpleLabor", dict(labor_kw)]] + extra +
                          [["PeriodicBracketTax", {"bracket_spacing": "us-federal", "period": 1,
                                                   "tax_model": "model_wrapper"}]], **kw)
    np.random.seed(77)
    ref = _ref_env(cfg)
    cfg["components"][0][1]["skills"] = [float(x) for x in ref.get_component("SimpleLabor").skills]
    host = make_env(cfg)
    o = OracleEnv(host.build_config(), host.layout_planes())
    np.random.seed(5)
    st = np.random.get_state()
    o.t["mt"][0] = st[1]
    o.t["mt_pos"][0] = st[2]
    obs = ref.reset()
    o.reset()
    rng = np.random.RandomState(8)
    n = cfg["n_agents"]
    multi_p = bool(cfg.get("multi_action_mode_planner", True))

    def check(where, obs, rew=None):
        compare_state({k: v[0] for k, v in o.t.items()}, extract_state(ref), where=where, f64_tol=1e-9)
        assert np.array_equal(o.t["mt"][0], np.random.get_state()[1]), where + ": MT19937 state"
        for k, want in extract_obs(ref, obs).items():
            np.testing.assert_allclose(o.t[k][0], want, rtol=2e-6, atol=2e-6, err_msg="%s: obs %s" % (where, k))
        if rew is not None:
            got = np.concatenate([o.t["rewards_a"][0], o.t["rewards_p"][[0]]])
            np.testing.assert_allclose(got, rewards_array(ref, rew), rtol=2e-7, atol=1e-5, err_msg=where)
        check_metrics(ref, host, o, where)

    check(variant + " reset", obs)
    for t in range(9):
        aa = rng.randint(0, 101, size=(n, 1)).astype(np.int32)
        acts = {str(i): int(aa[i, 0]) for i in range(n)}
        if multi_p:
            pa = rng.randint(0, 22, size=7).astype(np.int32)
            acts["p"] = [int(x) for x in pa]
        else:
            pa = np.array([rng.randint(0, ref.world.planner.action_spaces)], np.int32)
            acts["p"] = int(pa[0])
        obs, rew, done, _ = ref.step(acts)
        o.step(aa[None], pa[None])
        check("%s step %d" % (variant, t + 1), obs, rew)
        if done["__all__"]:
            obs = ref.reset()
            o.reset()
            check("%s reset after step %d" % (variant, t + 1), obs)


@pytest.mark.parametrize("seed", range(12))
def test_oracle_tracks_live_reference_random_one_step_economy(seed):
    from helpers import random_ose_config
    from oracle_lib import OracleEnv
    from ref_extract import extract_obs, extract_state, rewards_array

    cfg = random_ose_config(seed)
    np.random.seed(77 + seed)
    ref = _ref_env(cfg)
    for comp in cfg["components"]:  # SimpleLabor's skills come from the global stream at construction
        if comp[0] == "SimpleLabor":
            comp[1]["skills"] = [float(x) for x in ref.get_component("SimpleLabor").skills]
    host = make_env(cfg)
    o = OracleEnv(host.build_config(), host.layout_planes())
    np.random.seed(5)
    st = np.random.get_state()
    o.t["mt"][0] = st[1]
    o.t["mt_pos"][0] = st[2]
    obs = ref.reset()
    o.reset()
    rng = np.random.RandomState(8)
    n = cfg["n_agents"]
    pl = ref.world.planner
    names = [nm for nm in pl._action_names if nm != "PassiveAgentPlaceholder"]

    def check(where, obs, rew=None):
        compare_state({k: v[0] for k, v in o.t.items()}, extract_state(ref), where=where, f64_tol=1e-9)
        assert np.array_equal(o.t["mt"][0], np.random.get_state()[1]), where + ": MT19937 state"
        for k, want in extract_obs(ref, obs).items():
            np.testing.assert_allclose(o.t[k][0], want, rtol=2e-6, atol=2e-6, err_msg="%s: obs %s" % (where, k))
        if rew is not None:
            got = np.concatenate([o.t["rewards_a"][0], o.t["rewards_p"][[0]]])
            np.testing.assert_allclose(got, rewards_array(ref, rew), rtol=2e-7, atol=1e-5, err_msg=where)
        check_metrics(ref, host, o, where)

    where0 = "random one-step-economy %d %r" % (seed, cfg)
    check(where0 + " reset", obs)
    for t in range(3 * cfg["episode_length"] + 1):
        aa = rng.randint(0, 101, size=(n, 1)).astype(np.int32)
        acts = {str(i): int(aa[i, 0]) for i in range(n)}
        if not names:
            pa = np.zeros(1, np.int32)
        elif cfg["multi_action_mode_planner"]:
            pa = rng.randint(0, pl.action_dim[names[0]], size=len(names)).astype(np.int32)
            acts["p"] = [int(x) for x in pa]
        else:
            pa = np.array([rng.randint(0, pl.action_spaces)], np.int32)
            acts["p"] = int(pa[0])
        obs, rew, done, _ = ref.step(acts)
        o.step(aa[None], pa[None])
        check("%s step %d" % (where0, t + 1), obs, rew)
        if done["__all__"]:
            obs = ref.reset()
            o.reset()
            check("%s reset after step %d" % (where0, t + 1), obs)
