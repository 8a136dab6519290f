"""Neighbourhood regeneration (`regen_halfwidth > 0`, dynamic_layout.py:446-463) with `max_health > 1`: the plane the
reference convolves, max(map, source blocks), then changes with the map, so the regeneration probability of a source
block is a true d x d window sum and scipy's accumulation order shows in its last bit.

* the order the restatement assumes (kernel row-major: input rows r0+hw .. r0-hw, columns c0+hw .. c0-hw, zeros
  outside) against scipy.signal.convolve2d itself, bit for bit, on random planes and kernels;
* the restatement stepped side by side with the live reference on configurations that use it;
* (GPU) the HIP path against the restatement."""
import numpy as np
import pytest

from helpers import compare_state, make_env, oracle_host_pre_reset

CONFIGS = {
    "uniform_hw2_health3": dict(scenario_name="uniform/simple_wood_and_stone", world_size=[14, 14], n_agents=4,
                                wood_regen_halfwidth=2, wood_max_health=3, wood_regen_weight=0.5,
                                stone_regen_halfwidth=1, stone_max_health=2, stone_regen_weight=0.7,
                                starting_wood_coverage=0.1, starting_stone_coverage=0.1),
    "quadrant_hw3_mixed": dict(scenario_name="quadrant/simple_wood_and_stone", world_size=[16, 16], n_agents=5,
                               wood_regen_halfwidth=3, wood_max_health=2, wood_regen_weight=0.9,
                               stone_regen_halfwidth=0, stone_max_health=3, stone_regen_weight=0.2,
                               starting_wood_coverage=0.08, starting_stone_coverage=0.08),
    "uniform_hw1_health1_and_hw2_health4": dict(scenario_name="uniform/simple_wood_and_stone", world_size=[12, 12],
                                                n_agents=3, wood_regen_halfwidth=1, wood_max_health=1,
                                                wood_regen_weight=0.6, stone_regen_halfwidth=2, stone_max_health=4,
                                                stone_regen_weight=1.0, starting_wood_coverage=0.1,
                                                starting_stone_coverage=0.1),
}
COMPONENTS = [["Build", {}], ["ContinuousDoubleAuction", {"max_num_orders": 3}], ["Gather", {}],
              ["PeriodicBracketTax", {"period": 10}]]


def _cfg(name):
    return dict(CONFIGS[name], components=COMPONENTS, episode_length=40, starting_agent_coin=10)


def test_scipy_convolve2d_accumulation_order():
    from scipy import signal

    rng = np.random.RandomState(0)

    def window_sums(a, k):
        H, W = a.shape
        kh, kw = k.shape
        ch, cw = (kh - 1) // 2, (kw - 1) // 2
        out = np.zeros((H, W))
        for m in range(H):
            for n in range(W):
                s = 0.0
                for j in range(kh):
                    for kk in range(kw):
                        i0, i1 = m + ch - j, n + cw - kk
                        if 0 <= i0 < H and 0 <= i1 < W:
                            s = s + k[j, kk] * a[i0, i1]
                out[m, n] = s
        return out

    for trial in range(40):
        H, W = rng.randint(5, 14), rng.randint(5, 14)
        d = int(rng.choice([3, 5, 7]))
        if trial % 2:  # the regeneration's own shape: small integers times one weight
            a = rng.randint(0, 5, size=(H, W)).astype(np.float64)
            k = np.full((d, d), float(rng.rand())) / (d ** 2)
        else:
            a = rng.randn(H, W) * 10.0 ** rng.randint(-3, 4, size=(H, W))
            k = rng.rand(d, d)
        want = signal.convolve2d(a, k, "same")
        assert np.array_equal(want.view(np.uint64), window_sums(a, k).view(np.uint64)), trial


@pytest.mark.reference
@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_oracle_tracks_live_reference(name):
    from oracle_lib import OracleEnv
    from ref_extract import extract_state
    from test_oracle_vs_reference import _random_actions, _ref_env

    cfg = _cfg(name)
    np.random.seed(3)
    ref = _ref_env(cfg)
    host = make_env(cfg)
    o = OracleEnv(host.build_config(), host.layout_planes())
    np.random.seed(17)
    st = np.random.get_state()
    o.t["mt"][0], o.t["mt_pos"][0] = st[1], st[2]
    rng = np.random.RandomState(5)
    grown = 0
    for ep in range(2):
        ref.reset()
        oracle_host_pre_reset(host, o)
        o.reset()
        compare_state({k: v[0] for k, v in o.t.items()}, extract_state(ref), where="%s reset %d" % (name, ep))
        for t in range(cfg["episode_length"]):
            acts, a, p = _random_actions(ref, rng, False, True)
            ref.step(acts)
            o.step(a[None], p[None])
            got = {k: v[0] for k, v in o.t.items()}
            compare_state(got, extract_state(ref), where="%s episode %d step %d" % (name, ep, t + 1))
            assert np.array_equal(got["mt"], np.random.get_state()[1])
            grown = max(grown, int(max(got["wood"].max(), got["stone"].max())))
    assert grown >= 2, "no tile ever grew past health 1: the configuration does not exercise max_health > 1"


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_hip_matches_oracle(name):
    import torch
    from oracle_lib import OracleEnv
    from test_gpu_parity import _compare_all

    cfg = _cfg(name)
    np.random.seed(11)
    E = 48
    env = make_env(cfg, n_envs=E, device="cuda:0")
    env.seed(7)
    be = env.backend
    oracle = OracleEnv(env.build_config(), env.layout_planes())
    oracle.seed(7)
    env.reset()
    oracle_host_pre_reset(env, oracle)
    oracle.reset()
    _compare_all(be, oracle, name + " reset")
    for t in range(2 * cfg["episode_length"] + 3):
        a, p = be.sample_random_actions(seed=17)
        env.step({"a": a, "p": p})
        torch.cuda.synchronize()
        oracle.step(a.cpu().numpy(), p.cpu().numpy(), nthreads=4)
        if (t + 1) % 5 == 0:
            _compare_all(be, oracle, "%s step %d" % (name, t + 1))
        if bool(be.tensors["done"][0]):
            env.reset(be.tensors["done"])
            oracle_host_pre_reset(env, oracle)
            oracle.reset(oracle.t["done"].copy())
            _compare_all(be, oracle, "%s reset after step %d" % (name, t + 1))
    assert int(be.tensors["wood"].max()) >= 2 or int(be.tensors["stone"].max()) >= 2
