"""PeriodicBracketTax tax_model="saez" on the device (csrc/aie_kernels_saez.hip + the hooks in the step /
reset kernels) against the C restatement, which tests/test_oracle_vs_reference.py pins to the live
reference (random-rate phase side by side; the formula as a function of the reference's own state).

The formula is checked here the same way -- as a function of an injected state -- because a side-by-side
trajectory past the first formula period amplifies last-bit differences ("Saez")."""
import numpy as np
import pytest
from helpers import make_env


def _cfg(case):
    from test_oracle_vs_reference import _saez_cfg

    return _saez_cfg(case)


def _cases():
    from test_oracle_vs_reference import SAEZ_CASES

    return sorted(SAEZ_CASES)


def _compare_saez_buffer(be, oracle, where):
    n = oracle.t["saez_buffer_len"]
    got = be.tensors["saez_buffer"].cpu().numpy()
    msk = np.arange(got.shape[1])[None, :] < n[:, None]
    np.testing.assert_allclose(got[msk], oracle.t["saez_buffer"][msk], rtol=1e-9, atol=1e-9, err_msg=where)


def test_saez_host_kwargs_and_layout():
    """CPU: the host accepts the reference's saez kwargs and the C-ABI sizes the per-replica Saez block."""
    import ctypes

    from ai_economist_amd import _cabi
    from test_cabi_symbols import _lib_path

    cfg, _ = _cfg("uniform_weights_fixed_elas")
    env = make_env(cfg, n_envs=8)
    tax = env.get_component("PeriodicBracketTax")
    assert tax.tax_model == "saez" and tax.get_n_actions("BasicPlanner") == 0
    c = env.build_config()
    assert c.tax_model == _cabi.TAX_MODEL["saez"] and c.saez_buffer_size == 500
    assert c.saez_pareto_weight_uniform == 1 and c.saez_fixed_elas_given == 1 and c.saez_fixed_elas == 0.4
    lib = _cabi.bind(ctypes.CDLL(_lib_path()))
    base = lib.aie_arena_bytes(ctypes.byref(c))
    c.saez_buffer_size = 1500
    assert lib.aie_arena_bytes(ctypes.byref(c)) >= base + 8 * 1000 * 16
    c.saez_buffer_size = 0
    assert lib.aie_arena_bytes(ctypes.byref(c)) == _cabi.E_INVALID
    with pytest.raises(AssertionError):
        make_env(dict(cfg, components=cfg["components"][:3] + [["PeriodicBracketTax", {
            "tax_model": "saez", "pareto_weight_type": "nope"}]]))


@pytest.mark.gpu
@pytest.mark.parametrize("case", _cases())
def test_hip_saez_random_rate_phase_matches_oracle(case):
    """Period starts draw np.random.uniform rates from each replica's stream; tax days fill the buffer."""
    import torch
    from oracle_lib import OracleEnv
    from test_gpu_parity import _compare_all

    cfg, _ = _cfg(case)
    cfg["episode_length"] = 40
    E = 48
    env = make_env(cfg, n_envs=E, device="cuda:0")
    env.seed(21)
    env.reset()
    be = env.backend
    oracle = OracleEnv(env.build_config(), env.layout_planes())
    oracle.seed(21)
    oracle.reset()
    _compare_all(be, oracle, case + " reset")
    for t in range(100):
        a, p = be.sample_random_actions(seed=3)
        env.step({"a": a, "p": p})
        torch.cuda.synchronize()
        oracle.step(a.cpu().numpy(), p.cpu().numpy(), nthreads=4)
        _compare_all(be, oracle, "%s step %d" % (case, t + 1))
        _compare_saez_buffer(be, oracle, "%s step %d buffer" % (case, t + 1))
        if bool(be.tensors["done"][0]):
            env.reset(be.tensors["done"])
            oracle.reset(oracle.t["done"].copy())
            _compare_all(be, oracle, "%s reset after step %d" % (case, t + 1))
    assert oracle.t["saez_buffer_len"].min() > 30 and not oracle.t["saez_reached_min_samples"].any()


@pytest.mark.gpu
@pytest.mark.parametrize("case", _cases())
def test_hip_saez_formula_matches_oracle(case):
    """aie_saez_kernel as a function of an injected state: per replica a different sample buffer (negative,
    zero, rounding-residue, in-range and above-top incomes; clustered and spread marginal rates), elasticity
    estimates and running averages; one step at a period start; bracket rates, estimates, running average and
    the observed rates compared with the restatement (1e-9)."""
    import torch
    from oracle_lib import OracleEnv
    from test_oracle_vs_reference import synthetic_saez_buffer

    cfg, size = _cfg(case)
    E = 96
    host = make_env(cfg, n_envs=E, device="cuda:0")
    host.get_component("PeriodicBracketTax")._buffer_size = size
    host.seed(5)
    host.reset()
    be = host.backend
    oracle = OracleEnv(host.build_config(), host.layout_planes())
    oracle.seed(5)
    oracle.reset()
    rs = np.random.RandomState(17)
    top = float(host.get_component("PeriodicBracketTax").bracket_cutoffs[-1])
    cap = oracle.t["saez_buffer"].shape[1]
    for rnd in range(3):
        buf = np.zeros((E, cap, 2))
        lens = np.zeros(E, np.int32)
        for e in range(E):
            m = size if e % 5 else int(rs.randint(max(1, size // 3), size))  # some replicas still short of samples
            zt = synthetic_saez_buffer(rs, m, top, e)
            z, tau = zt[:, 0], zt[:, 1]
            buf[e, :m, 0], buf[e, :m, 1] = z, tau
            lens[e] = m
        state = {"saez_buffer": buf, "saez_buffer_len": lens,
                 "saez_reached_min_samples": (rs.rand(E) < 0.3).astype(np.int32),
                 "saez_elas": np.stack([rs.rand(E) * 2, rs.rand(E), rs.randn(E), rs.randn(E)], 1),
                 "saez_running_avg_tax_rates": rs.rand(*oracle.t["saez_running_avg_tax_rates"].shape) * 0.5,
                 "tax_cycle_pos": np.ones(E, np.int32)}
        for k, v in state.items():
            oracle.t[k][...] = v
            be.tensors[k].copy_(torch.from_numpy(np.ascontiguousarray(v)).to(be.tensors[k].device))
        a = torch.zeros_like(be.sample_random_actions(seed=1)[0])
        p = torch.zeros_like(be.sample_random_actions(seed=1)[1])
        host.step({"a": a, "p": p})
        torch.cuda.synchronize()
        oracle.step(a.cpu().numpy(), p.cpu().numpy())
        reached = oracle.t["saez_reached_min_samples"].astype(bool)
        assert np.array_equal(be.tensors["saez_reached_min_samples"].cpu().numpy(), oracle.t["saez_reached_min_samples"])
        assert reached.sum() > E // 2 and (~reached).sum() > 3
        assert np.array_equal(be.tensors["mt"].cpu().numpy().view(np.uint32), oracle.t["mt"])  # random rates elsewhere
        for k in ("tax_saez_bracket_rates", "tax_saez_observed_rates", "saez_elas", "saez_running_avg_tax_rates"):
            np.testing.assert_allclose(be.tensors[k].cpu().numpy(), oracle.t[k], rtol=1e-9, atol=1e-9,
                                       err_msg="%s round %d: %s" % (case, rnd, k))
        np.testing.assert_allclose(be.tensors["saez_next_rates"].cpu().numpy()[reached], oracle.t["saez_next_rates"][reached],
                                   rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(be.tensors["obs_p_flat"].cpu().numpy(), oracle.t["obs_p_flat"], rtol=2e-6, atol=2e-6)


@pytest.mark.gpu
def test_hip_saez_runs_into_the_formula_phase():
    """End to end with a small buffer: the buffer fills, the formula takes over, rates stay within
    [rate_min, rate_max] and move the running average; the first formula period agrees with the restatement."""
    import torch
    from oracle_lib import OracleEnv

    cfg, size = _cfg("uniform_weights_fixed_elas")
    E = 32
    env = make_env(cfg, n_envs=E, device="cuda:0")
    env.get_component("PeriodicBracketTax")._buffer_size = size
    env.seed(2)
    env.reset()
    be = env.backend
    oracle = OracleEnv(env.build_config(), env.layout_planes())
    oracle.seed(2)
    oracle.reset()
    first_formula = None
    for t in range(150):
        a, p = be.sample_random_actions(seed=9)
        env.step({"a": a, "p": p})
        if first_formula is None:
            oracle.step(a.cpu().numpy(), p.cpu().numpy(), nthreads=4)
            if oracle.t["saez_reached_min_samples"].all():
                first_formula = t
                np.testing.assert_allclose(be.tensors["tax_saez_bracket_rates"].cpu().numpy(),
                                           oracle.t["tax_saez_bracket_rates"], rtol=1e-9, atol=1e-9)
                np.testing.assert_allclose(be.tensors["saez_elas"].cpu().numpy(), oracle.t["saez_elas"], rtol=1e-9)
        if bool(be.tensors["done"][0]):
            env.reset(be.tensors["done"])
            if first_formula is None:
                oracle.reset(oracle.t["done"].copy())
    assert first_formula is not None
    rates = be.tensors["tax_saez_bracket_rates"].cpu().numpy()
    assert (rates >= 0.05 - 1e-12).all() and (rates <= 0.8 + 1e-12).all()
    assert (be.tensors["saez_running_avg_tax_rates"].cpu().numpy() > 0).all()
    assert be.tensors["saez_reached_min_samples"].cpu().numpy().all()
    m = env.metrics
    assert np.isfinite(m["PeriodicTax/saez/estimated_elasticity"]).all()


@pytest.mark.gpu
def test_hip_saez_global_buffer_union_matches_oracle():
    """set_global_saez_buffer on the device (aie_set_global_saez_buffer): replicas fill their local buffers, the union
    of all of them (sharding.accumulate_and_broadcast_saez_buffers, the reference's trainer-side exchange,
    tutorials/rllib/utils/remote.py:56-73) becomes the global buffer, and the following period starts -- global + each
    replica's own new samples -- agree with the restatement (which tests/test_oracle_vs_reference.py pins to the live
    reference) to 1e-9."""
    import torch
    from oracle_lib import OracleEnv

    from ai_economist_amd.sharding import accumulate_and_broadcast_saez_buffers

    cfg, size = _cfg("inverse_income")
    cfg["episode_length"] = 30
    E = 24
    host = make_env(cfg, n_envs=E, device="cuda:0")
    tax = host.get_component("PeriodicBracketTax")
    tax._buffer_size = size
    host.seed(5)
    host.reset()
    be = host.backend
    oracle = OracleEnv(host.build_config(), host.layout_planes())
    oracle.seed(5)
    oracle.reset()
    assert be.tensors["saez_global_buffer"].shape[1] == E * size

    def run(steps):
        for _ in range(steps):
            a, p = be.sample_random_actions(seed=3)
            host.step({"a": a, "p": p})
            torch.cuda.synchronize()
            oracle.step(a.cpu().numpy(), p.cpu().numpy(), nthreads=4)
            if bool(be.tensors["done"][0]):
                host.reset(be.tensors["done"])
                oracle.reset(oracle.t["done"].copy())

    run(30)  # random-rate phase: every replica collects samples
    n_loc = be.tensors["saez_buffer_len"].cpu().numpy()
    assert n_loc.min() > 0 and np.array_equal(n_loc, oracle.t["saez_buffer_len"])
    glob = accumulate_and_broadcast_saez_buffers(host)
    assert glob.shape == (int(n_loc.sum()), 2) and int(be.tensors["saez_global_len"][0]) == int(n_loc.sum())
    want = np.concatenate([oracle.t["saez_buffer"][e, : n_loc[e]] for e in range(E)])
    np.testing.assert_allclose(glob.cpu().numpy(), want, rtol=1e-9, atol=1e-9)
    oracle.set_global_saez_buffer(want)
    assert np.array_equal(be.tensors["saez_additions"].cpu().numpy(), oracle.t["saez_additions"])
    # the global buffer alone holds far more than _buffer_size samples: the very next period start runs the formula
    for k in range(12):
        run(1)
        for name in ("tax_saez_bracket_rates", "saez_elas", "saez_running_avg_tax_rates"):
            np.testing.assert_allclose(be.tensors[name].cpu().numpy(), oracle.t[name], rtol=1e-9, atol=1e-9,
                                       err_msg="step %d after the union: %s" % (k + 1, name))
        assert np.array_equal(be.tensors["saez_reached_min_samples"].cpu().numpy(), oracle.t["saez_reached_min_samples"])
    assert int(be.tensors["saez_reached_min_samples"].min()) == 1
    # reset_saez_buffers empties local and global buffers: random rates again
    tax.reset_saez_buffers()
    assert int(be.tensors["saez_global_len"][0]) == 0 and int(be.tensors["saez_additions"].abs().sum()) == 0
    with pytest.raises(ValueError):
        be._check(be.lib.aie_set_global_saez_buffer(be.handle, glob.data_ptr(), E * size + 1))


@pytest.mark.gpu
def test_hip_saez_trajectory_without_auction_matches_oracle():
    """The no-auction Saez configuration whose TRAJECTORY the restatement follows beside the live reference through
    the formula phase (tests/test_oracle_vs_reference.py::test_oracle_tracks_live_reference_saez_trajectory_without_auction):
    the device against the restatement, 32 replicas with their own streams, three episodes -- random-rate phase, then
    formula periods; float state 1e-9, integer state and the MT19937 stream exact every step."""
    import torch
    from oracle_lib import OracleEnv
    from test_gpu_parity import _compare_all
    from test_oracle_vs_reference import SAEZ_NO_AUCTION, SAEZ_NO_AUCTION_BUFFER

    cfg, size = dict(SAEZ_NO_AUCTION), SAEZ_NO_AUCTION_BUFFER
    E = 32
    env = make_env(cfg, n_envs=E, device="cuda:0")
    env.get_component("PeriodicBracketTax")._buffer_size = size
    env.seed(13)
    env.reset()
    be = env.backend
    oracle = OracleEnv(env.build_config(), env.layout_planes())
    oracle.seed(13)
    oracle.reset()
    _compare_all(be, oracle, "reset")
    rs = np.random.RandomState(2)
    T = 3 * cfg["episode_length"]
    for t in range(T):
        a, p = be.sample_random_actions(seed=3)
        an = a.cpu().numpy().copy()
        an[rs.rand(*an.shape) < 0.35] = 1  # builds: incomes for the formula
        a = torch.as_tensor(an, device=a.device)
        env.step({"a": a, "p": p})
        torch.cuda.synchronize()
        oracle.step(an, p.cpu().numpy(), nthreads=4)
        _compare_all(be, oracle, "step %d" % (t + 1))
        _compare_saez_buffer(be, oracle, "step %d buffer" % (t + 1))
        if bool(be.tensors["done"][0]):
            env.reset(be.tensors["done"])
            oracle.reset(oracle.t["done"].copy())
            _compare_all(be, oracle, "reset after step %d" % (t + 1))
    assert oracle.t["saez_reached_min_samples"].all(), "every replica should be in the formula phase by now"
    assert np.abs(oracle.t["tax_saez_bracket_rates"]).max() > 0
