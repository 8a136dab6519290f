"""GPU parity tests (pytest -m gpu): the HIP path, called through the C ABI, against
(1) the committed reference golden vectors, (2) the CPU oracle on seeded random
rollouts, (3) size-independent properties at the full BASELINE batch size."""
import zlib

import os

import numpy as np
import pytest

from helpers import (F64_FIELDS, FAMILY_CASES, INT_FIELDS, compare_state, dev_library, golden_names, load_golden, make_env,
                     state_from_golden)

pytestmark = pytest.mark.gpu

# f64 state that goes through the Saez formula (OLS, histogram interpolation: libm-independent but summed in wave order)
F64_TOLERANT = {"saez_elas", "saez_running_avg_tax_rates", "tax_saez_bracket_rates"}
OBS_TOL = 2e-6   # f32 observations (f64 in the reference, rounded once to f32)
REW_TOL = 1e-5   # BASELINE.json north_star: coin-utility reward floats within 1e-5

GTB = [["Build", {}], ["ContinuousDoubleAuction", {"max_num_orders": 5}], ["Gather", {}],
       ["PeriodicBracketTax", {}]]
C2 = dict(scenario_name="layout_from_file/simple_wood_and_stone", n_agents=4, world_size=[25, 25],
          episode_length=1000, components=GTB, starting_agent_coin=10,
          env_layout_file="quadrant_25x25_20each_30clump.txt")


def _replica(be, e):
    out = {}
    for k, t in be.tensors.items():
        if t.shape[0] != be.E:  # tensors shared by all replicas (the global Saez buffer)
            continue
        v = t[e].cpu().numpy()
        if k == "mt":
            v = v.view(np.uint32)
        out[k] = v
    return out


def _obs_check(be, g, k, where, e=0):
    for name in [x for x in g.keys() if x.startswith("ob_")]:
        t = name[3:]
        want = g[name][k]
        got = be.tensors[t][e].cpu().numpy()
        if want.dtype.kind in "iu":
            assert np.array_equal(got, want), "%s: obs %s differs" % (where, t)
        else:
            np.testing.assert_allclose(got, want, rtol=OBS_TOL, atol=OBS_TOL, err_msg="%s: obs %s" % (where, t))


@pytest.mark.parametrize("name", golden_names())
def test_hip_step_matches_reference_golden(name):
    import torch

    g = load_golden(name)
    E = 3  # replicas 0 and 2 carry the golden state; replica 1 runs something else
    env = make_env(g["cfg"], n_envs=E, device="cuda:0")
    env.seed(77)
    env.reset()
    be = env.backend
    s0 = state_from_golden(g, "s0_")
    be.load_state(s0, e=0)
    be.load_state(s0, e=2)
    T = g["actions_a"].shape[0]
    obs_steps = list(g["obs_steps"])
    resets = {int(t): i for i, t in enumerate(g.get("reset_at", []))}
    n = env.n_agents
    for t in range(T):
        a = np.zeros((E, n), np.int32)
        a[0] = a[2] = g["actions_a"][t]
        a[1] = (g["actions_a"][t] * 7 + t) % 6
        act = {"a": torch.as_tensor(a, device="cuda:0")}
        if g["actions_p"].shape[1]:
            p = np.zeros((E, g["actions_p"].shape[1]), np.int32)
            p[0] = p[2] = g["actions_p"][t]
            act["p"] = torch.as_tensor(p, device="cuda:0")
        env.step(act)
        want = state_from_golden(g, "st_", t)
        for e in (0, 2):
            got = _replica(be, e)
            compare_state(got, want, where="%s step %d replica %d" % (name, t + 1, e))
            assert zlib.crc32(got["mt"].tobytes()) == int(g["st_mt_crc"][t])
            rew = np.concatenate([got["rewards_a"], got["rewards_p"][None]])
            np.testing.assert_allclose(rew, g["rew"][t], rtol=2e-7, atol=REW_TOL)  # f32 storage
            assert int(got["done"]) == int(g["done"][t])
        if (t + 1) in obs_steps:
            _obs_check(be, g, obs_steps.index(t + 1), "%s step %d" % (name, t + 1), e=2)
        if (t + 1) in resets:
            env.reset(be.tensors["done"])
            compare_state(_replica(be, 0), state_from_golden(g, "rs_", resets[t + 1]),
                          where="%s reset after step %d" % (name, t + 1))
    assert np.array_equal(_replica(be, 0)["mt"], g["final_mt"])


@pytest.mark.parametrize("name", golden_names())
def test_hip_reset_matches_reference_golden(name):
    g = load_golden(name)
    E = 2
    env = make_env(g["cfg"], n_envs=E, device="cuda:0")
    be = env.backend
    keys = np.stack([g["pre_reset_mt"]] * E)
    be.set_rng_state(keys, np.full(E, int(g["pre_reset_pos"]), np.int32))
    env.reset()
    want = state_from_golden(g, "s0_")
    for e in range(E):
        got = _replica(be, e)
        compare_state(got, want, where="%s reset replica %d" % (name, e))
        assert np.array_equal(got["mt"], want["mt"])
    if 0 in list(g["obs_steps"]):
        _obs_check(be, g, list(g["obs_steps"]).index(0), name + " reset obs", e=1)


def _state_keys(tensors):
    """Every named tensor but the sampler's draw index: `sample_t` (a record field since round 5, so that a captured
    launch can be replayed) counts the action draws an environment has made -- tests that draw from one environment and
    step two with the same actions differ there and nowhere else."""
    return [k for k in tensors if k != "sample_t"]


def _compare_all(be, oracle, where, sl=None):
    """Every state field, observation, reward and accumulator of the replicas in `sl` (default: all): integers,
    books and the MT19937 key bit-exact -- including `auto_warmup`, the integer the sign of a ~1e-16 mean
    reward decides (the device computes pow / exp exactly as libm does, csrc/aie_glibc_math.h)."""
    sl = slice(None) if sl is None else sl

    def dev(k):
        v = be.tensors[k][sl].cpu().numpy()
        return v.view(np.uint32) if k == "mt" else v

    for k in INT_FIELDS + ["mt"]:
        if k in be.tensors and k in oracle.t:
            assert np.array_equal(dev(k), oracle.t[k][sl]), "%s: %s differs" % (where, k)
    for r in range(2):
        for side in ("bids", "asks"):
            if "cda_" + side not in be.tensors:
                continue
            nn = oracle.t["cda_n_" + side][sl][:, r]
            got = dev("cda_" + side)[:, r]
            want = oracle.t["cda_" + side][sl][:, r]
            msk = np.arange(got.shape[1])[None, :] < nn[:, None]
            assert np.array_equal(got[msk], want[msk]), "%s: %s book differs" % (where, side)
    # tax_model "saez": the bracket rates come out of the formula kernel (OLS + histogram sums in wave order, 1e-9),
    # and every coin they touch inherits that; all other configurations compare f64 state bit for bit
    saez = "saez_elas" in be.tensors
    for k in F64_FIELDS:
        if k in be.tensors and k in oracle.t:
            if saez or k in F64_TOLERANT:
                np.testing.assert_allclose(dev(k), oracle.t[k][sl], rtol=1e-9, atol=1e-9, err_msg="%s: %s" % (where, k))
            else:  # coin, labor, skills, utilities, tax trackers, price histories: the same doubles, bit for bit
                got, want = dev(k), oracle.t[k][sl]
                same = got.view(np.uint64) == want.view(np.uint64)
                assert same.all(), "%s: f64 field %s differs in %d of %d values (max |diff| %g)" % (
                    where, k, (~same).sum(), same.size, np.abs(got - want).max())
    # episode accumulators behind env.metrics: the counts and -- since coin is bit-exact (round 2) and every
    # accumulator adds its terms in the oracle's order -- the f64 sums too, bit for bit (tax_model "saez" keeps 1e-9,
    # see above)
    for k in be.tensors:
        if k.startswith("metrics_"):
            got, want = dev(k), oracle.t[k][sl]
            if got.dtype.kind in "iu" or not saez:
                assert np.array_equal(got, want), "%s: %s differs (max |diff| %g)" % (
                    where, k, np.abs(got.astype(np.float64) - want.astype(np.float64)).max())
            else:
                np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-9, err_msg="%s: %s" % (where, k))
    for k in be.tensors:
        if k.startswith("obs_") or k.startswith("rewards") or k == "done":
            got, want = dev(k), oracle.t[k][sl]
            if got.dtype.kind in "iu":
                assert np.array_equal(got, want), "%s: %s differs" % (where, k)
            else:
                np.testing.assert_allclose(got, want, rtol=OBS_TOL, atol=OBS_TOL, err_msg="%s: %s" % (where, k))


def _compare_metrics(env, oracle, where, require_trade_tax=True):
    """env.metrics (scenario + component metrics, one array over replicas per key) from the
    device state against the same formulas on the oracle's state."""
    from ai_economist_amd.foundation.metrics import env_metrics

    got = env.metrics
    want = env_metrics(env, dict(oracle.t))
    assert sorted(got) == sorted(want)
    if require_trade_tax:
        assert any(k.startswith("Trade/") for k in got) and "PeriodicTax/avg_effective_tax_rate" in got
    for k, v in want.items():
        if k.startswith("PeriodicTax/avg_tax_rate/"):
            # argmin / argmax over coin endowments: agents that tie (or differ in the last bit between the
            # device and the oracle) may swap; the value itself is checked wherever the same agent is picked
            g, w = np.asarray(got[k], np.float64), np.asarray(v, np.float64)
            assert np.isclose(g, w, rtol=1e-7, atol=1e-9, equal_nan=True).mean() >= 0.9, "%s: metric %s" % (where, k)
            continue
        np.testing.assert_allclose(np.asarray(got[k], np.float64), np.asarray(v, np.float64), rtol=1e-7, atol=1e-9,
                                   equal_nan=True, err_msg="%s: metric %s" % (where, k))
    one = env.metrics_of(3)
    assert set(one) == set(got) and all(np.isscalar(x) for x in one.values())


@pytest.mark.parametrize("n_agents,E,T", [(4, 256, 230), (10, 128, 120)])
def test_hip_matches_oracle_on_random_rollouts(n_agents, E, T):
    """Seeded uniform-random rollouts (the bench's policy), every replica, every field,
    every 10 steps, across a tax day and an episode boundary."""
    import torch
    from oracle_lib import OracleEnv

    cfg = dict(C2, n_agents=n_agents, episode_length=200, starting_agent_coin=15,
               resource_regen_prob=0.05, env_layout_file="uniform_25x25_25each_65clump.txt")
    env = make_env(cfg, n_envs=E, device="cuda:0")
    env.seed(5)
    env.reset()
    be = env.backend
    oracle = OracleEnv(env.build_config(), env.layout_planes())
    oracle.seed(5)
    oracle.reset()
    _compare_all(be, oracle, "after reset")
    for t in range(T):
        a, p = be.sample_random_actions(seed=99)
        env.step({"a": a, "p": p})
        torch.cuda.synchronize()
        oracle.step(a.cpu().numpy(), p.cpu().numpy(), nthreads=4)
        if (t + 1) % 10 == 0 or t + 1 == 200:
            _compare_all(be, oracle, "step %d" % (t + 1))
        if (t + 1) % 100 == 0:
            _compare_metrics(env, oracle, "step %d" % (t + 1))
        if t + 1 == 200:
            assert bool(be.tensors["done"].all())
            env.reset(be.tensors["done"])
            oracle.reset(oracle.t["done"].copy())
            _compare_all(be, oracle, "after episode reset")


@pytest.mark.parametrize("case", ["gather_trade_build", "one_step_economy", "covid"])
def test_step_sample_next_equals_two_launches(case):
    """aie_step_sample_next == aie_step followed by aie_sample_random_actions: same actions, same
    trajectory, bit for bit -- in every scenario family."""
    import torch

    cfg = dict(C2, episode_length=50) if case == "gather_trade_build" else _reward_log_cases()[case]
    T = cfg["episode_length"]
    envs = [make_env(cfg, n_envs=96, device="cuda:0", env_offset=640) for _ in range(2)]
    for env in envs:
        if case != "covid":
            env.seed(9)
        env.reset()
    b0, b1 = envs[0].backend, envs[1].backend
    cur = b1.sample_random_actions(seed=4242, env_offset=640, slot=0)
    slot = 0
    for t in range(min(60, 2 * T + 3)):
        a, p = b0.sample_random_actions(seed=4242, env_offset=640)
        assert torch.equal(a, cur[0]) and torch.equal(p, cur[1]), "actions differ at step %d" % t
        b0.step(a, p)
        cur = b1.step_sample_next(cur[0], cur[1], seed=4242, env_offset=640, next_slot=slot ^ 1)
        slot ^= 1
        if (t + 1) % T == 0:
            b0.reset(b0.tensors["done"])
            b1.reset(b1.tensors["done"])
    b0.sample_random_actions(seed=4242, env_offset=640)  # (b1 has drawn the next step's actions already: the draw index
    torch.cuda.synchronize()                             # is a record field)
    assert torch.equal(b0.tensors["sample_t"], b1.tensors["sample_t"])
    assert torch.equal(b0.arena, b1.arena)
    with pytest.raises(ValueError):
        b1._check(b1.lib.aie_step_sample_next(b1.handle, cur[0].data_ptr(), cur[1].data_ptr(), 1, 0,
                                              cur[0].data_ptr(), cur[1].data_ptr(), None))


def test_full_batch_properties_c2_4096():
    """BASELINE configs[1] at full size (4096 replicas): determinism, shard invariance
    (a replica's trajectory depends only on its global id), coin conservation."""
    import torch

    E, T = 4096, 120
    env = make_env(dict(C2, resource_regen_prob=0.05, env_layout_file="uniform_25x25_25each_65clump.txt"),
                   n_envs=E, device="cuda:0")
    env.seed(11)
    env.reset()
    be = env.backend
    for t in range(T):
        a, p = be.sample_random_actions(seed=7)
        env.step({"a": a, "p": p})
    torch.cuda.synchronize()
    snap = be.arena.clone()
    # determinism
    env2 = make_env(dict(C2, resource_regen_prob=0.05, env_layout_file="uniform_25x25_25each_65clump.txt"),
                    n_envs=E, device="cuda:0")
    env2.seed(11)
    env2.reset()
    for t in range(T):
        a, p = env2.backend.sample_random_actions(seed=7)
        env2.step({"a": a, "p": p})
    torch.cuda.synchronize()
    assert torch.equal(snap, env2.backend.arena)
    # shard invariance: replicas [1024, 1024+64) run alone as a 64-replica shard
    off, Es = 1024, 64
    env3 = make_env(dict(C2, resource_regen_prob=0.05, env_layout_file="uniform_25x25_25each_65clump.txt"),
                    n_envs=Es, device="cuda:0", env_offset=off)
    env3.seed(11)
    env3.reset()
    for t in range(T):
        a, p = env3.backend.sample_random_actions(seed=7, env_offset=off)
        env3.step({"a": a, "p": p})
    torch.cuda.synchronize()
    for k in ("cells", "loc_r", "loc_c", "inv_res", "inv_coin", "labor", "mt", "obs_a_flat", "rewards_a"):
        assert torch.equal(be.tensors[k][off:off + Es], env3.backend.tensors[k]), k
    # coin conservation: trades and taxes move coin around, only building mints it
    coin = (be.tensors["inv_coin"] + be.tensors["esc_coin"]).sum(dim=1).cpu().numpy()
    houses = (be.tensors["house_owner"] >= 0).sum(dim=(1, 2)).cpu().numpy()
    np.testing.assert_allclose(coin, 4 * 10.0 + 10.0 * houses, rtol=0, atol=1e-9)
    assert houses.sum() > 0
    # resource bookkeeping never goes negative; order books stay within quota
    assert int(be.tensors["inv_res"].min()) >= 0 and int(be.tensors["esc_res"].min()) >= 0
    assert int(be.tensors["cda_n_orders"].max()) <= 5
    assert int(be.tensors["timestep"].min()) == T and int(be.tensors["timestep"].max()) == T


def test_error_behaviour_through_cabi():
    import ctypes

    from ai_economist_amd import _cabi

    env = make_env(C2, n_envs=4, device="cuda:0")
    be = env.backend
    with pytest.raises(KeyError):
        be.download("no_such_tensor")
    buf = np.zeros(3, np.int32)
    rc = be.lib.aie_upload(be.handle, b"loc_r", buf.ctypes.data, buf.nbytes)  # wrong size
    assert rc == _cabi.E_INVALID and b"expected" in be.lib.aie_last_error(be.handle)
    d = _cabi.AieTensorDesc()
    assert be.lib.aie_get_tensor(be.handle, b"obs_a_world-map", ctypes.byref(d)) == 0
    assert tuple(d.shape[:5]) == (4, 4, 7, 11, 11) and d.dtype == 5


def _variant_names():
    from test_oracle_vs_reference import VARIANTS

    return sorted(VARIANTS)


@pytest.mark.parametrize("variant", _variant_names())
def test_hip_matches_oracle_on_config_variants(variant):
    """Configuration variants that tests/test_oracle_vs_reference.py pins against the live
    reference (multi-action agents, single-action planner, no observation scaling, other
    planner rewards, component orders, tax models, 6 agents on 40x40): HIP vs oracle."""
    import torch
    from oracle_lib import OracleEnv
    from test_oracle_vs_reference import BASE, VARIANTS

    cfg = dict(BASE)
    cfg.update(VARIANTS[variant])
    E, T = 48, 170
    env = make_env(cfg, n_envs=E, device="cuda:0")
    env.seed(3)
    env.reset()
    be = env.backend
    oracle = OracleEnv(env.build_config(), env.layout_planes())
    oracle.seed(3)
    oracle.reset()
    _compare_all(be, oracle, variant + " reset")
    for t in range(T):
        a, p = be.sample_random_actions(seed=17)
        env.step({"a": a, "p": p})
        torch.cuda.synchronize()
        oracle.step(a.cpu().numpy(), p.cpu().numpy(), nthreads=4)
        if (t + 1) % 10 == 0:
            _compare_all(be, oracle, "%s step %d" % (variant, t + 1))
        if bool(be.tensors["done"][0]):
            env.reset(be.tensors["done"])
            oracle.reset(oracle.t["done"].copy())
            _compare_all(be, oracle, "%s reset after step %d" % (variant, t + 1))


OSE_VARIANTS = {
    "c5_default_100ag": dict(n_agents=100),
    "coin_eq_40ag": dict(n_agents=40, planner_reward_type="coin_eq_times_productivity",
                         mixing_weight_gini_vs_coin=0.25),
    "isoelastic_12ag_coin_eq": dict(n_agents=12, agent_reward_type="isoelastic_coin_minus_labor",
                                     planner_reward_type="coin_eq_times_productivity", isoelastic_eta=0.4),
    "no_first_step_mask_128ag": dict(n_agents=128, labor_kw=dict(mask_first_step=False)),
    "wealth_redistribution_77ag": dict(n_agents=77, extra_components=[["WealthRedistribution", {}]]),
    # tax_model "saez": 45 samples per step => the 500-sample buffer fills during the test (random rates, then the
    # formula kernel's first periods; compared with the restatement every step)
    "saez_45ag": dict(n_agents=45, steps=16, tax_kw={"tax_model": "saez", "rate_max": 0.8}),
}


@pytest.mark.parametrize("seed", range(40))
def test_hip_matches_oracle_on_random_configs(seed):
    """helpers.random_gtb_config: the same randomly drawn configurations the reference-marked
    CPU test pins against the live reference (seeds 0..23 there), HIP vs oracle, two episodes."""
    import torch
    from helpers import oracle_host_pre_reset, random_gtb_config
    from oracle_lib import OracleEnv

    cfg = random_gtb_config(seed)
    np.random.seed(500 + seed)
    E = 24
    env = make_env(cfg, n_envs=E, device="cuda:0")
    env.seed(3)
    be = env.backend
    oracle = OracleEnv(env.build_config(), env.layout_planes())
    oracle.seed(3)
    env.reset()
    oracle_host_pre_reset(env, oracle)
    oracle.reset()
    where0 = "random config %d" % seed
    _compare_all(be, oracle, where0 + " reset")
    T = cfg["episode_length"]
    for t in range(2 * T + 5):
        a, p = be.sample_random_actions(seed=17)
        env.step({"a": a, "p": p})
        torch.cuda.synchronize()
        oracle.step(a.cpu().numpy(), p.cpu().numpy(), nthreads=4)
        if (t + 1) % 9 == 0 or (t + 1) % T == 0:
            _compare_all(be, oracle, "%s step %d" % (where0, t + 1))
        if (t + 1) % T == 0:
            assert bool(be.tensors["done"].all())
            _compare_metrics(env, oracle, "%s step %d" % (where0, t + 1), require_trade_tax=False)
            env.reset(be.tensors["done"])
            oracle_host_pre_reset(env, oracle)
            oracle.reset(oracle.t["done"].copy())
            _compare_all(be, oracle, "%s reset after step %d" % (where0, t + 1))


def test_maximum_agent_count_matches_oracle():
    """62 mobile agents (the spatial scenarios' limit: one lane each + the planner's bookkeeping) on the 40 x 40
    quadrant map with all four components: 310-slot order books (the LDS book path), a 496-word draw window, a
    record that leaves room for 3-4 workgroups per CU; 63 agents are refused at construction."""
    import torch
    from oracle_lib import OracleEnv
    from test_oracle_vs_reference import BASE, GTB

    cfg = dict(BASE, components=GTB, n_agents=62, world_size=[40, 40], env_layout_file="quadrant_40x40_50each.txt",
               episode_length=30)
    E = 6
    env = make_env(cfg, n_envs=E, device="cuda:0")
    env.seed(9)
    env.reset()
    be = env.backend
    oracle = OracleEnv(env.build_config(), env.layout_planes())
    oracle.seed(9)
    oracle.reset()
    _compare_all(be, oracle, "62 agents reset")
    for t in range(45):
        a, p = be.sample_random_actions(seed=31)
        env.step({"a": a, "p": p})
        torch.cuda.synchronize()
        oracle.step(a.cpu().numpy(), p.cpu().numpy(), nthreads=4)
        if t % 5 == 4 or t == 29:
            _compare_all(be, oracle, "62 agents step %d" % (t + 1))
        if t == 29:
            env.reset(be.tensors["done"])
            oracle.reset(oracle.t["done"].copy())
            _compare_all(be, oracle, "62 agents second reset")
    with pytest.raises(Exception, match="n_agents"):
        make_env(dict(cfg, n_agents=63), n_envs=2, device="cuda:0").reset()


@pytest.mark.parametrize("n_agents,size", [(24, 8), (12, 6), (9, 12)])
def test_crowded_world_move_conflicts_match_oracle(n_agents, size):
    """Many agents on a small map, four of five actions a move: agents keep targeting tiles that another agent of the
    same step's random order has just left or entered -- the case in which the look-ahead Gather (from 8 agents on:
    every lane resolves its own agent's tiles before the serial loop) must fall back to the LDS path for that agent.
    Every field of every replica against the oracle after every step."""
    import torch
    from helpers import oracle_host_pre_reset
    from oracle_lib import OracleEnv

    cfg = dict(scenario_name="uniform/simple_wood_and_stone", n_agents=n_agents, world_size=[size, size],
               episode_length=60, components=[["Build", {}], ["Gather", {}]], starting_agent_coin=10,
               starting_stone_coverage=0.15, starting_wood_coverage=0.15)
    E = 48
    env = make_env(cfg, n_envs=E, device="cuda:0")
    env.seed(21)
    env.reset()
    be = env.backend
    oracle = OracleEnv(env.build_config(), env.layout_planes())
    oracle.seed(21)
    oracle_host_pre_reset(env, oracle)
    oracle.reset()
    _compare_all(be, oracle, "crowded reset")
    g = torch.Generator(device="cpu").manual_seed(7)
    moved = 0
    for t in range(70):
        a = torch.randint(0, 6, (E, n_agents), generator=g, dtype=torch.int32)  # 0 NO-OP, 1 Build, 2..5 moves
        a = torch.where(a == 0, torch.full_like(a, 2 + t % 4), a).to("cuda:0")  # (hardly any NO-OP: more traffic)
        _, p = be.sample_random_actions(seed=3)
        before = be.tensors["loc_r"].clone(), be.tensors["loc_c"].clone()
        env.step({"a": a, "p": p})
        torch.cuda.synchronize()
        moved += int(((be.tensors["loc_r"] != before[0]) | (be.tensors["loc_c"] != before[1])).sum())
        oracle.step(a.cpu().numpy(), p.cpu().numpy(), nthreads=4)
        _compare_all(be, oracle, "crowded step %d" % (t + 1))
        if t == 59:
            env.reset(be.tensors["done"])
            oracle_host_pre_reset(env, oracle)
            oracle.reset(oracle.t["done"].copy())
            _compare_all(be, oracle, "crowded second reset")
    assert moved > 10 * E, "the policy is supposed to move agents around"


@pytest.mark.parametrize("side,on_device", [(52, True), (66, False)])
def test_large_uniform_worlds(side, on_device):
    """uniform/ on 52 x 52: the reset kernel draws the layouts itself (its planes + the record image take 70 KB of the
    workgroup's LDS -- gfx950 allows 160 KB; round 2 stopped at 64 KB / 48 x 48), the 10 816-word regeneration sweep
    crosses 17 generator windows per step.  66 x 66 (more than the 4096 cells the device path takes): the layouts come
    from the host-side procedure (dynamic_layout.py: generate_layout, each replica's own stream).  HIP vs oracle over
    an episode boundary."""
    import torch
    from helpers import oracle_host_pre_reset
    from oracle_lib import OracleEnv

    cfg = dict(scenario_name="uniform/simple_wood_and_stone", n_agents=3, world_size=[side, side], episode_length=8,
               components=[["Build", {}], ["Gather", {}]], starting_agent_coin=3, starting_stone_coverage=0.03,
               starting_wood_coverage=0.03)
    E = 3
    np.random.seed(77)
    env = make_env(cfg, n_envs=E, device="cuda:0")
    assert bool(env.layouts_on_device) == on_device
    env.seed(6)
    be = env.backend
    oracle = OracleEnv(env.build_config(), env.layout_planes())
    oracle.seed(6)
    env.reset()
    oracle_host_pre_reset(env, oracle)
    oracle.reset()
    where = "%dx%d" % (side, side)
    _compare_all(be, oracle, where + " reset")
    for t in range(12):
        a, p = be.sample_random_actions(seed=2)
        env.step({"a": a, "p": p})
        torch.cuda.synchronize()
        oracle.step(a.cpu().numpy(), p.cpu().numpy(), nthreads=3)
        _compare_all(be, oracle, where + " step %d" % (t + 1))
        if t == 7:
            env.reset(be.tensors["done"])
            oracle_host_pre_reset(env, oracle)
            oracle.reset(oracle.t["done"].copy())
            _compare_all(be, oracle, where + " second reset")


def test_dense_source_layouts_take_the_row_by_row_regeneration():
    """More than 128 source doubles per replica (here: uniform layouts drawn with 22 % coverage per resource on
    20 x 20) leave the sparse gather regeneration for the row-by-row fallback (aie_kernels.hip:
    scenario_step_regen_rows): same stream, same respawns as the oracle."""
    import torch
    from helpers import oracle_host_pre_reset
    from oracle_lib import OracleEnv

    cfg = dict(scenario_name="uniform/simple_wood_and_stone", n_agents=5, world_size=[20, 20], episode_length=25,
               components=[["Build", {}], ["Gather", {}]], starting_agent_coin=3, starting_stone_coverage=0.22,
               starting_wood_coverage=0.22, wood_regen_weight=0.3, stone_regen_weight=0.2)
    E = 16
    env = make_env(cfg, n_envs=E, device="cuda:0")
    env.seed(4)
    be = env.backend
    oracle = OracleEnv(env.build_config(), env.layout_planes())
    oracle.seed(4)
    env.reset()
    oracle_host_pre_reset(env, oracle)
    oracle.reset()
    _compare_all(be, oracle, "dense layout reset")
    flags = be.tensors["cell_flags"].reshape(E, -1).cpu().numpy()
    n_src = ((flags & 2) != 0).sum(axis=1) + ((flags & 4) != 0).sum(axis=1)
    assert (n_src > 128).all(), n_src  # every replica is on the fallback
    for t in range(40):
        a, p = be.sample_random_actions(seed=23)
        env.step({"a": a, "p": p})
        torch.cuda.synchronize()
        oracle.step(a.cpu().numpy(), p.cpu().numpy(), nthreads=4)
        if t % 6 == 5 or t == 24:
            _compare_all(be, oracle, "dense layout step %d" % (t + 1))
        if t == 24:
            env.reset(be.tensors["done"])
            oracle_host_pre_reset(env, oracle)
            oracle.reset(oracle.t["done"].copy())
            _compare_all(be, oracle, "dense layout second reset")


def _expected_src_list(flags_row, cap):
    """Source doubles of one replica from its flag bytes: Wood cells ascending, then Stone cells (+ H W)."""
    hw = flags_row.size
    d = np.concatenate([np.flatnonzero(flags_row & 4), hw + np.flatnonzero(flags_row & 2)])
    lst = np.zeros(cap, np.int64)
    lst[: min(cap, d.size)] = d[:cap]
    return d.size, lst


@pytest.mark.parametrize("case", ["fixed_shared", "uniform_per_replica", "dense"])
def test_record_source_list_follows_the_flags(case):
    """Round 6: the regeneration's source doubles are a record field (`regen_src_n`, `regen_src_list`) the reset kernel
    derives from the cells' flag bytes -- and aie_set_layout / aie_upload / load_state wherever the flags change without
    a reset -- instead of a scan inside every step.  The list equals the flags' after create, after resets with fresh
    layouts and after a state injection; a step leaves it alone."""
    import torch

    cfg = {"fixed_shared": dict(C2),
           "uniform_per_replica": dict(scenario_name="uniform/simple_wood_and_stone", n_agents=4, world_size=[15, 15],
                                       episode_length=6, components=[["Build", {}], ["Gather", {}]],
                                       starting_stone_coverage=0.1, starting_wood_coverage=0.1),
           "dense": dict(scenario_name="uniform/simple_wood_and_stone", n_agents=5, world_size=[20, 20], episode_length=6,
                         components=[["Build", {}], ["Gather", {}]], starting_stone_coverage=0.22,
                         starting_wood_coverage=0.22)}[case]
    E = 12
    env = make_env(cfg, n_envs=E, device="cuda:0")
    be = env.backend
    cap = int(be.tensors["regen_src_list"].shape[-1])

    def check(where):
        torch.cuda.synchronize()
        flags = be.tensors["cell_flags"].reshape(E, -1).cpu().numpy()
        got_n = be.tensors["regen_src_n"].cpu().numpy()
        got = be.tensors["regen_src_list"].cpu().numpy().view(np.uint16).astype(np.int64)
        for e in range(E):
            n, lst = _expected_src_list(flags[e], cap)
            assert got_n[e] == n, (where, e, got_n[e], n)
            assert np.array_equal(got[e], lst), (where, e)
        return got_n

    if case == "fixed_shared":
        check("after create (aie_set_layout)")
    env.seed(3)
    env.reset()
    n0 = check("after reset")
    assert (n0 > 0).all() and ((n0 > cap).all() if case == "dense" else (n0 <= cap).all())
    for t in range(8):  # crosses an episode end for the generated layouts (auto reset off: explicit masked reset)
        a, p = be.sample_random_actions(seed=9)
        env.step({"a": a, "p": p})
        if t == 5 and case != "fixed_shared":
            env.reset(be.tensors["done"])
    check("after steps and a second reset")
    if case == "fixed_shared":
        # a state injection with another layout into one replica (load_state derives the list on the host)
        st = {k: v[0].cpu().numpy() for k, v in be.tensors.items() if k in ("stone", "wood")}
        rs = np.random.RandomState(1)
        st["stone_src"] = (rs.rand(25, 25) < 0.04).astype(np.uint8)
        st["wood_src"] = (rs.rand(25, 25) < 0.04).astype(np.uint8)
        st["water"] = np.zeros((25, 25), np.uint8)
        be.load_state(st, e=2)
        check("after load_state")


@pytest.mark.parametrize("variant", sorted(OSE_VARIANTS))
def test_hip_matches_oracle_one_step_economy(variant):
    """BASELINE configs[4] family: one-step-economy + SimpleLabor + PeriodicBracketTax."""
    import torch
    from oracle_lib import OracleEnv

    kw = dict(OSE_VARIANTS[variant])
    labor_kw = kw.pop("labor_kw", {})
    extra = kw.pop("extra_components", [])
    tax_kw = dict({"bracket_spacing": "us-federal", "period": 1, "tax_model": "model_wrapper"}, **kw.pop("tax_kw", {}))
    rs = np.random.RandomState(4)
    n = kw["n_agents"]
    labor_kw["skills"] = [float(x) for x in np.sort(1 + rs.rand(n) * 2)]
    cfg = dict(scenario_name="one-step-economy", world_size=[1, 1], episode_length=2,
               components=[["SimpleLabor", labor_kw]] + extra + [["PeriodicBracketTax", tax_kw]], **kw)
    E, T = 96, kw.pop("steps", 12)
    cfg.pop("steps", None)
    env = make_env(cfg, n_envs=E, device="cuda:0")
    env.seed(9)
    env.reset()
    be = env.backend
    oracle = OracleEnv(env.build_config(), env.layout_planes())
    oracle.seed(9)
    oracle.reset()
    _compare_all(be, oracle, variant + " reset")
    for t in range(T):
        a, p = be.sample_random_actions(seed=23)
        env.step({"a": a, "p": p})
        torch.cuda.synchronize()
        oracle.step(a.cpu().numpy(), p.cpu().numpy(), nthreads=4)
        _compare_all(be, oracle, "%s step %d" % (variant, t + 1))
        if bool(be.tensors["done"][0]):
            env.reset(be.tensors["done"])
            oracle.reset(oracle.t["done"].copy())
            _compare_all(be, oracle, "%s reset after step %d" % (variant, t + 1))


@pytest.mark.parametrize("seed", [1, 3, 4, 9])
def test_partial_resets_mid_episode(seed):
    """reset(env_mask) of an arbitrary subset of replicas in the middle of an episode (what a
    vectorised trainer does with `done`, F/env_wrapper.py:341-353), incl. scenarios whose reset has
    a host-side part: the untouched replicas must not notice, the reset ones restart exactly."""
    import torch
    from helpers import oracle_host_pre_reset, random_gtb_config
    from oracle_lib import OracleEnv

    cfg = random_gtb_config(seed)
    np.random.seed(500 + seed)
    E = 20
    env = make_env(cfg, n_envs=E, device="cuda:0")
    env.seed(11)
    be = env.backend
    oracle = OracleEnv(env.build_config(), env.layout_planes())
    oracle.seed(11)
    env.reset()
    oracle_host_pre_reset(env, oracle)
    oracle.reset()
    rng = np.random.RandomState(seed)
    for t in range(36):
        a, p = be.sample_random_actions(seed=5)
        env.step({"a": a, "p": p})
        torch.cuda.synchronize()
        oracle.step(a.cpu().numpy(), p.cpu().numpy(), nthreads=4)
        if t in (7, 19, 20):
            mask = (rng.rand(E) < 0.4).astype(np.uint8)
            env.reset(torch.as_tensor(mask, device="cuda"))
            oracle_host_pre_reset(env, oracle, which=np.nonzero(mask)[0])
            oracle.reset(mask)
            _compare_all(be, oracle, "partial reset after step %d" % (t + 1))
    _compare_all(be, oracle, "end")
    ts = be.tensors["timestep"].cpu().numpy()
    assert len(set(ts.tolist())) > 1  # replicas really are at different points of their episodes


@pytest.mark.parametrize("seed", range(16))
def test_hip_matches_oracle_on_random_one_step_economy_configs(seed):
    """helpers.random_ose_config (seeds 0..11 are pinned against the live reference on CPU)."""
    import torch
    from helpers import random_ose_config
    from oracle_lib import OracleEnv

    cfg = random_ose_config(seed)
    np.random.seed(77 + seed)  # SimpleLabor's Monte-Carlo skills (global stream at construction)
    env = make_env(cfg, n_envs=40, device="cuda:0")
    env.seed(9)
    env.reset()
    be = env.backend
    oracle = OracleEnv(env.build_config(), env.layout_planes())
    oracle.seed(9)
    oracle.reset()
    where0 = "random one-step-economy %d" % seed
    _compare_all(be, oracle, where0 + " reset")
    for t in range(3 * cfg["episode_length"] + 1):
        a, p = be.sample_random_actions(seed=23)
        env.step({"a": a, "p": p})
        torch.cuda.synchronize()
        oracle.step(a.cpu().numpy(), p.cpu().numpy(), nthreads=4)
        _compare_all(be, oracle, "%s step %d" % (where0, t + 1))
        if bool(be.tensors["done"][0]):
            _compare_metrics(env, oracle, "%s step %d" % (where0, t + 1), require_trade_tax=False)
            env.reset(be.tensors["done"])
            oracle.reset(oracle.t["done"].copy())
            _compare_all(be, oracle, "%s reset after step %d" % (where0, t + 1))


@pytest.mark.parametrize("multi", [False, True])
def test_masked_action_sampler_respects_masks(multi):
    """aie_sample_masked_actions: every sampled sub-action is allowed by the current mask,
    all allowed entries get sampled, and a masked rollout still matches the oracle."""
    import torch
    from oracle_lib import OracleEnv

    cfg = dict(C2, episode_length=120, resource_regen_prob=0.05, multi_action_mode_agents=multi,
               env_layout_file="uniform_25x25_25each_65clump.txt")
    E = 256
    env = make_env(cfg, n_envs=E, device="cuda:0")
    env.seed(2)
    env.reset()
    be = env.backend
    oracle = OracleEnv(env.build_config(), env.layout_planes())
    oracle.seed(2)
    oracle.reset()
    seen_move = np.zeros(5, bool)
    for t in range(60):
        a, p = be.sample_masked_actions(seed=5)
        torch.cuda.synchronize()
        ma = be.tensors["obs_a_action_mask"].cpu().numpy()
        mp = be.tensors["obs_p_action_mask"].cpu().numpy()
        an, pn = a.cpu().numpy(), p.cpu().numpy()
        if multi:
            off = 0
            dims = [1, 11, 11, 11, 11, 4]
            for s, d in enumerate(dims):
                sel = np.take_along_axis(ma[:, :, off:off + d + 1], an[:, :, s:s + 1], axis=2)
                assert np.all(sel == 1.0), "masked sub-action sampled (subspace %d)" % s
                off += d + 1
            seen_move[np.unique(an[:, :, 5])] = True
        else:
            sel = np.take_along_axis(ma, an[:, :, :1], axis=2)
            assert np.all(sel == 1.0), "masked action sampled"
        for b in range(7):
            sel = np.take_along_axis(mp[:, b * 22:(b + 1) * 22], pn[:, b:b + 1], axis=1)
            assert np.all(sel == 1.0), "masked planner action sampled"
        env.step({"a": a, "p": p})
        torch.cuda.synchronize()
        oracle.step(an, pn, nthreads=4)
    _compare_all(be, oracle, "masked rollout")
    if multi:
        assert seen_move.all()


def _reward_log_cases():
    from helpers import load_covid_golden
    from test_oracle_vs_reference import BASE, GTB

    c = _auto_reset_cases()
    covid = dict(load_covid_golden("c4_covid_variant")["cfg"], scenario_name="CovidAndEconomySimulation")
    return {"gather_trade_build": dict(BASE, components=GTB, episode_length=5), "one_step_economy": c["one_step_economy"],
            "covid": covid}


@pytest.mark.parametrize("case", ["gather_trade_build", "one_step_economy", "covid"])
def test_reward_log_slots_follow_the_steps(case):
    """aie_set_reward_log: every step also writes (rewards, done) into the next slot of the caller's log -- in every
    scenario family (the multi-GPU exchange ships whole blocks of slots, sharding.RewardLogGather)."""
    import torch

    cfg = _reward_log_cases()[case]
    E = 40
    env = make_env(cfg, n_envs=E, device="cuda:0")
    if case != "covid":
        env.seed(4)
    env.reset()
    be = env.backend
    n = be.n
    log = be.set_reward_log(3)
    assert tuple(log.shape) == (3, E, n + 2)
    cur = be.sample_random_actions(7, 0, slot=0)
    slot = 0
    for t in range(8):
        if t % 2 or case != "gather_trade_build":  # both step entry points fill the log
            env.step({"a": cur[0], "p": cur[1]})
            cur = be.sample_random_actions(7, 0, slot=slot)
        else:
            cur = be.step_sample_next(cur[0], cur[1], 7, 0, next_slot=slot ^ 1)
            slot ^= 1
        row = log[t % 3].cpu().numpy()
        assert np.array_equal(row[:, :n], be.tensors["rewards_a"].cpu().numpy()), t
        assert np.array_equal(row[:, n], be.tensors["rewards_p"].cpu().numpy()), t
        assert np.array_equal(row[:, n + 1] > 0.5, be.tensors["done"].cpu().numpy().astype(bool)), t
        if bool(be.tensors["done"][0]):
            env.reset(be.tensors["done"])
    before = log.clone()
    be.set_reward_log(0)
    env.step({"a": cur[0], "p": cur[1]})
    assert torch.equal(log, before)


def test_covid_replica_past_its_episode_end_keeps_its_call_counters_with_the_batch():
    """a COVID replica stepped past its episode's end without a reset does nothing -- but its reward-log slot
    and its draw index still advance with every launch, so that after its reset it writes the slot (and draws with the
    index) its peers do."""
    cfg = _reward_log_cases()["covid"]
    E = 6
    env = make_env(cfg, n_envs=E, device="cuda:0")
    env.reset()
    be = env.backend
    be.set_reward_log(4)
    T = int(env.episode_length)
    # put replicas 0..2 at the end of their episode (the state a rollout reaches after T steps), leave 3..5 at the start
    be.tensors["timestep"][:3] = T
    cur = be.sample_random_actions(7, 0, slot=0)
    slot = 0
    for t in range(3):
        cur = be.step_sample_next(cur[0], cur[1], 7, 0, next_slot=slot ^ 1)
        slot ^= 1
    assert be.tensors["timestep"].cpu().tolist() == [T] * 3 + [3] * 3  # the first three sat the launches out ...
    assert be.tensors["rew_log_slot"].cpu().tolist() == [3] * E      # ... and kept their slot
    assert be.tensors["sample_t"].cpu().tolist() == [int(be.tensors["sample_t"][5])] * E  # ... and their draw index


def test_captured_step_follows_a_later_set_reward_log_call():
    """the reward log's address, slot count and restart live in the device-side parameter block, so a step
    launch captured in a hipGraph writes to the log that the LATEST aie_set_reward_log call named (RewardLogGather.finish
    rewinds the log after the capture): replays after a rewind restart at slot 0 and equal an eager twin's log."""
    import torch

    E = 24
    envs = []
    for _ in range(2):
        env = make_env(C2, n_envs=E, device="cuda:0")
        env.seed(5)
        env.reset()
        envs.append(env)
    (eg, ee) = envs
    bg, bee = eg.backend, ee.backend
    n = bg.n
    log_g, log_e = bg.set_reward_log(4), bee.set_reward_log(4)
    acts = [bg.sample_random_actions(31 + k, 0, slot=0) for k in range(1)][0]
    a, p = acts[0].clone(), acts[1].clone()
    for be in (bg, bee):  # two steps each: the slot counters stand at 2
        be.step(a, p)
        be.step(a, p)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        bg.step(a, p)
    torch.cuda.current_stream().wait_stream(side)
    bee.step(a, p)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        bg.step(a, p)
    # after the capture: rewind both logs (a new epoch: the next step fills slot 0) -- the graph must follow
    bg.rewind_reward_log()
    bee.rewind_reward_log()
    log_g.zero_()
    log_e.zero_()
    for _ in range(3):
        graph.replay()
        bee.step(a, p)
    torch.cuda.synchronize()
    assert bg.tensors["rew_log_slot"].cpu().tolist() == [3] * E
    assert torch.equal(log_g, log_e)
    assert float(log_g[:3].abs().sum()) > 0 and float(log_g[3].abs().sum()) == 0
    # and a log somewhere else entirely
    new_g, new_e = bg.set_reward_log(2), bee.set_reward_log(2)
    graph.replay()
    bee.step(a, p)
    torch.cuda.synchronize()
    assert torch.equal(new_g, new_e) and float(new_g[0].abs().sum()) > 0


def test_reward_log_gather_over_rccl():
    """sharding.RewardLogGather on the real collective backend: an RCCL ("nccl") process group over whatever GPUs
    this process sees (one rank here; the multi-rank logic is covered by the gloo tests): blocks of 4 steps travel
    through dist.gather and arrive with exactly the rewards / done flags the steps produced."""
    import socket

    import torch
    import torch.distributed as dist

    from ai_economist_amd.sharding import RewardLogGather

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        env = make_env(dict(C2, episode_length=50), n_envs=64, device="cuda:0")
        env.seed(2)
        env.reset()
        be = env.backend
        g = RewardLogGather(be, steps_per_gather=4, keep=True, force_collective=True)
        want = []
        for t in range(12):
            a, p = be.sample_random_actions(seed=9)
            be.step(a, p)
            want.append(torch.cat([be.tensors["rewards_a"], be.tensors["rewards_p"][:, None],
                                   be.tensors["done"].to(torch.float32)[:, None]], dim=1).clone())
            g.after_step()
        g.finish()
        assert g.n_collectives == 3 and len(g.received) == 3
        got = torch.cat([blk[0] for blk in g.received], dim=0)  # rank 0's slots, [12, E, n + 2]
        assert torch.equal(got, torch.stack(want))
    finally:
        dist.destroy_process_group()


def test_bench_launch_path_under_torchrun_on_one_gpu():
    """`bench.py --gpus N` (N > 1) re-executes itself under torch.distributed.run, initialises the RCCL group with a
    device id, shards the replicas by rank and gathers (reward, done) and the per-rank times -- code that otherwise runs
    for the first time on the day an 8-GPU node shows up.  `--launcher torchrun --force-gather` takes exactly that path
    with one rank on this box: the JSON line must come back with the exchange done and accounted for."""
    import json
    import subprocess
    import sys

    from helpers import ROOT

    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--launcher", "torchrun", "--force-gather",
           "--steps", "150", "--warmup", "10", "--envs-per-gpu", "256", "--no-cpu-baseline", "--no-workloads"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert res.returncode == 0, res.stderr[-2000:]
    line = [ln for ln in res.stdout.strip().splitlines() if ln.startswith("{")][-1]
    assert line == res.stdout.strip().splitlines()[-1], "the JSON line must be the last line on stdout"
    assert len(line) < 4096, "the driver keeps an 8 KB tail of stdout: the last line has to stay short"
    d = json.loads(line)
    assert d["n_gpus"] == 1 and len(d["per_rank_seconds"]) == 1
    assert d["gather"]["collectives"] >= 2 and d["gather"]["bytes_per_collective"] == 64 * 256 * (4 + 2) * 4
    assert d["config"]["envs_per_gpu"] == 256 and d["value"] > 0 and d["steps"] == 150
    assert d["config"]["global_envs"] == d["n_gpus"] * d["config"]["envs_per_gpu"]
    assert d["roofline"]["frac"] > 0 and d["roofline"]["kernel"].startswith("aie_step_kernel")
    # the full result went out on an earlier line (and to bench_detail.json): the exchange is accounted for there
    full = json.loads([ln for ln in res.stdout.strip().splitlines() if ln.startswith("{")][-2])
    assert full["exchange_ok"] is True and full["gather"]["collectives"] == d["gather"]["collectives"]  # (the compact line rounds)
    # asking for more GPUs than the node has is refused before anything is launched
    import torch

    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(torch.cuda.device_count() + 1),
                          "--steps", "5", "--warmup", "1"], capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and "visible" in (res.stderr + res.stdout)


def test_error_flags_for_what_the_reference_raises():
    """Out-of-range action indices are NO-OPs on the device and a ValueError in the reference (move.py:133-134,
    build.py:158-159): the replica's `error_flags` records them, env.check_errors() raises, reset clears."""
    import torch

    env = make_env(dict(C2, episode_length=20), n_envs=6, device="cuda:0")
    env.seed(2)
    env.reset()
    be = env.backend
    a = torch.zeros((6, 4), dtype=torch.int32, device="cuda:0")
    p = torch.zeros((6, 7), dtype=torch.int32, device="cuda:0")
    env.step({"a": a, "p": p})
    env.check_errors()
    assert int(be.tensors["error_flags"].abs().sum()) == 0
    a[1, 2] = 50          # one past the last action (A = 50)
    a[4, 0] = -3
    p[3, 6] = 22          # bracket rates are 0..21
    before = be.tensors["inv_coin"].clone()
    env.step({"a": a, "p": p})
    assert be.tensors["error_flags"].cpu().tolist() == [0, 1, 0, 2, 1, 0]
    assert torch.equal(before, be.tensors["inv_coin"])  # the bad entries acted as NO-OPs
    with pytest.raises(ValueError):
        env.check_errors()
    env.step({"a": torch.zeros_like(a), "p": torch.zeros_like(p)})
    assert be.tensors["error_flags"].cpu().tolist() == [0, 1, 0, 2, 1, 0]  # sticky
    env.reset(torch.tensor([0, 1, 0, 0, 0, 0], dtype=torch.uint8, device="cuda:0"))
    assert be.tensors["error_flags"].cpu().tolist() == [0, 0, 0, 2, 1, 0]
    # wrongly shaped action buffers never reach the kernel
    with pytest.raises(ValueError):
        env.step({"a": a[:, :3], "p": p})
    with pytest.raises(ValueError):
        env.step({"a": a, "p": p[0]})
    with pytest.raises(ValueError):
        env.step({"0": 3, "1": 0, "p": [0] * 7})


C1_INSTANCE = dict(scenario_name="uniform/simple_wood_and_stone", n_agents=4, world_size=[15, 15], episode_length=1000,
                   components=[["Build", {}], ["Gather", {}]], starting_agent_coin=10, starting_stone_coverage=0.10,
                   starting_wood_coverage=0.10)


@pytest.mark.parametrize("case", sorted(FAMILY_CASES))
def test_compile_time_instance_equals_generic_kernel(case):
    """BASELINE configs[0] / [1] / [2] AND every configuration that differs from them in scalars only run on a
    compile-time instance of the step and reset kernels (aie_spec_generated.h: the code-shaping part of the parameter
    block folded into the code, the scalars read from the run-time block) without any call by the user; the generic
    kernel on the same replicas must produce the same arena, bit for bit, across episode ends and masked resets."""
    import ctypes

    import torch

    cfg, in_family = FAMILY_CASES[case]
    with dev_library():  # aie_dev_lds_bytes below is a development hook
        env_s = make_env(cfg, n_envs=8, device="cuda:0")
        env_s.backend
    k_inst = env_s.backend.lib.aie_step_kernel_instance(env_s.backend.handle)
    if not in_family:
        assert k_inst == -1, "%s must not match an instance's family" % case
        return
    assert k_inst >= 0, "no compile-time instance selected for %s" % case
    # the occupancy the instance is compiled for (_specs.py) is what its LDS footprint lets a CU hold
    from ai_economist_amd import _specs

    lds = (ctypes.c_int64 * 6)()
    env_s.backend.lib.aie_dev_lds_bytes.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    assert env_s.backend.lib.aie_dev_lds_bytes(env_s.backend.handle, lds) == 0
    assert -(-2 * lds[5] // 4) == _specs.SPECS[k_inst][4], "instance %d: %d B of LDS -> %d workgroups per CU" % (k_inst, lds[0], lds[5])
    pair = [make_env(cfg, n_envs=512, device="cuda:0") for _ in range(2)]
    b_spec, b_ref = pair[0].backend, pair[1].backend
    assert b_ref.lib.aie_select_step_kernel(b_ref.handle, 1) == 0  # AIE_KERNEL_GENERIC (step and reset)
    assert b_spec.lib.aie_step_kernel_instance(b_spec.handle) >= 0 and b_ref.lib.aie_step_kernel_instance(b_ref.handle) == -1
    for env in pair:
        env.seed(21)
        env.reset()
    T = int(cfg["episode_length"])
    cur_s = b_spec.sample_random_actions(seed=4, slot=0)
    slot = 0
    steps = 230 if T > 200 else 2 * T + 25
    checks = {0, 57, 101, steps - 1, T - 1, T, 2 * T - 1}
    for t in range(steps):
        a, p = b_ref.sample_random_actions(seed=4)
        b_ref.step(a, p)
        if t % 3 == 0:  # both entry points of the instance
            cur_s = b_spec.step_sample_next(cur_s[0], cur_s[1], seed=4, next_slot=slot ^ 1)
            slot ^= 1
        else:
            b_spec.step(cur_s[0], cur_s[1])
            cur_s = b_spec.sample_random_actions(seed=4, slot=slot)
        if t in checks:
            torch.cuda.synchronize()
            for k in _state_keys(b_ref.tensors):
                assert torch.equal(b_ref.tensors[k], b_spec.tensors[k]), "%s step %d: %s differs" % (case, t + 1, k)
        if (t + 1) % T == 0:  # episode end: the instance's reset kernel against the generic one
            for b in (b_ref, b_spec):
                b.reset(b.tensors["done"])
        elif t % 40 == 39:  # and a masked reset mid-episode
            mask = (torch.arange(512, device="cuda") % 5 == (t // 40) % 5).to(torch.uint8)
            for b in (b_ref, b_spec):
                b.reset(mask)


def test_dense_log_switch_keeps_state_identical():
    """aie_set_dense_log_active: while no episode is being logged the dense-log replica steps with the rest of the batch
    on the fast kernel; state, observations and rewards are the same as with the switch on -- only the event rows stop."""
    import torch

    cfg, _ = FAMILY_CASES["phase2_yaml"]
    cfg = dict(cfg, episode_length=50)
    on, off = [make_env(cfg, n_envs=96, device="cuda:0") for _ in range(2)]
    for env in (on, off):
        env.seed(3)
        env.backend.reset(None)
    off.backend.set_dense_log_active(False)
    assert "log_events" in on.backend.tensors and on.backend.lib.aie_step_kernel_instance(on.backend.handle) >= 0
    for t in range(120):
        a, p = on.backend.sample_random_actions(seed=2)
        for env in (on, off):
            env.backend.step(a, p)
        if (t + 1) % 50 == 0:
            for env in (on, off):
                env.backend.reset(env.backend.tensors["done"])
        if t in (0, 1, 49, 50, 119):
            torch.cuda.synchronize()
            for k in _state_keys(on.backend.tensors):
                if not k.startswith("log_event"):
                    assert torch.equal(on.backend.tensors[k], off.backend.tensors[k]), "step %d: %s differs" % (t + 1, k)
    # back on: the very next step records rows again, identical to an environment that never switched
    off.backend.set_dense_log_active(True)
    a, p = on.backend.sample_random_actions(seed=2)
    for env in (on, off):
        env.backend.step(a, p)
    torch.cuda.synchronize()
    for k in _state_keys(on.backend.tensors):
        if k != "log_events":
            assert torch.equal(on.backend.tensors[k], off.backend.tensors[k]), "after switching back on: %s differs" % k
    for e in range(on.backend.tensors["log_events"].shape[0]):  # (rows behind the count are leftovers of earlier steps)
        cnt = int(on.backend.tensors["log_event_count"][e])
        assert torch.equal(on.backend.tensors["log_events"][e, :cnt], off.backend.tensors["log_events"][e, :cnt])


def test_instance_family_member_matches_oracle():
    """One member of C2's family that is not BASELINE's configuration, on the instance, against the CPU oracle (the
    other family tests compare with the generic kernel, which the variant tests compare with the oracle)."""
    import torch
    from oracle_lib import OracleEnv

    cfg, _ = FAMILY_CASES["c2_every_scalar"]
    E = 48
    env = make_env(cfg, n_envs=E, device="cuda:0")
    env.seed(5)
    env.reset()
    be = env.backend
    assert be.lib.aie_step_kernel_instance(be.handle) >= 0
    oracle = OracleEnv(env.build_config(), env.layout_planes())
    oracle.seed(5)
    oracle.reset()
    for t in range(150):
        a, p = be.sample_random_actions(seed=12)
        be.step(a, p)
        torch.cuda.synchronize()
        oracle.step(a.cpu().numpy(), p.cpu().numpy())
        if (t + 1) % 70 == 0:
            be.reset(be.tensors["done"])
            torch.cuda.synchronize()
            oracle.reset(np.ones(E, dtype=np.uint8))
        if t in (0, 19, 20, 69, 70, 149):
            _compare_all(be, oracle, "family member step %d" % (t + 1))


JIT_CASES = {
    # the reference's phase-2 training setting (tutorials/rllib/phase2/config.yaml: the planner sees no maps)
    "phase2_planner_blind": dict(C2, planner_gets_spatial_info=False),
    "uniform_layout_6_agents": dict(C2, n_agents=6, env_layout_file="uniform_25x25_25each_65clump.txt", episode_length=150),
    "build_gather_halfwidth": dict(scenario_name="uniform/simple_wood_and_stone", n_agents=5, world_size=[15, 15],
                                   episode_length=90, components=[["Build", {}], ["Gather", {}]], starting_agent_coin=10,
                                   starting_stone_coverage=0.10, starting_wood_coverage=0.10),
    # a one-step-economy of another size than BASELINE configs[4]'s (which has the build's instance)
    "one_step_economy_37_agents": dict(scenario_name="one-step-economy", n_agents=37, world_size=[1, 1], episode_length=3,
                                       components=[["SimpleLabor", {"skills": [1.0 + 0.05 * i for i in range(37)]}],
                                                   ["PeriodicBracketTax", {"bracket_spacing": "us-federal", "period": 1,
                                                                           "tax_model": "model_wrapper"}]]),
}


@pytest.mark.parametrize("case", sorted(JIT_CASES))
def test_runtime_specialisation_equals_generic_kernel(case):
    """aie_specialize: configurations without a compile-time instance get step / reset kernels compiled for them at
    run time (hiprtc, the parameter block as a constant image).  The specialised environment and one on the generic
    kernel, same seed and actions: every tensor bit for bit across masked resets and an episode end."""
    import torch

    cfg = dict(JIT_CASES[case])
    pair = [make_env(cfg, n_envs=384, device="cuda:0") for _ in range(2)]
    b_jit, b_ref = pair[0].backend, pair[1].backend
    assert b_jit.lib.aie_step_kernel_instance(b_jit.handle) == -1, "the case must not have a compile-time instance"
    assert b_ref.lib.aie_select_step_kernel(b_ref.handle, 1) == 0  # pinned: the background specialisation must not swap in
    assert pair[0].specialize(required=True)
    assert b_jit.lib.aie_step_kernel_instance(b_jit.handle) == 1000 and b_ref.lib.aie_step_kernel_instance(b_ref.handle) == -1
    for env in pair:
        env.seed(4)
        env.reset()
    T = int(cfg["episode_length"]) + 20 if cfg["episode_length"] <= 200 else 160
    for t in range(T):
        a, p = b_ref.sample_random_actions(seed=8)
        b_ref.step(a, p)
        b_jit.step(a, p)
        if t % 40 == 39:
            mask = (torch.arange(384, device="cuda") % 7 == (t // 40) % 7).to(torch.uint8) | b_ref.tensors["done"]
            b_ref.reset(mask)
            b_jit.reset(mask)
        if t in (0, 1, 39, 40, T - 1) or bool(b_ref.tensors["done"][0]):
            torch.cuda.synchronize()
            for k in _state_keys(b_ref.tensors):
                assert torch.equal(b_ref.tensors[k], b_jit.tensors[k]), "%s step %d: %s differs" % (case, t + 1, k)
    # the switch works both ways, and a second environment of the same configuration finds the cached code object
    assert b_jit.lib.aie_select_step_kernel(b_jit.handle, 1) == 0 and b_jit.lib.aie_step_kernel_instance(b_jit.handle) == -1
    assert b_jit.lib.aie_select_step_kernel(b_jit.handle, 0) == 0 and b_jit.lib.aie_step_kernel_instance(b_jit.handle) == 1000
    assert b_ref.lib.aie_select_step_kernel(b_ref.handle, 0) == 0
    assert pair[1].specialize(required=True) and b_ref.lib.aie_step_kernel_instance(b_ref.handle) == 1000


@pytest.mark.parametrize("case", ["gtb_7_agents", "gtb_7_agents_fast_rng", "one_step_economy_37_agents", "uniform_layout_5_agents"])
def test_background_specialisation_swaps_in_without_a_call(monkeypatch, case):
    """A configuration outside every compile-time instance's family starts on the generic kernel; aie_create has its
    kernels compiled in the background (hiprtc, cached) and a later aie_reset adopts them -- no call by the user.
    Before, across and after the switch every tensor equals a twin that is pinned to the generic kernel, bit for bit.
    The product's default path (tests/conftest.py switches it off for the rest of the suite) in every kernel family: the
    gather-trade-build step / reset pair with either generator, the one-step-economy kernel, a scenario whose reset
    draws layouts."""
    import time

    import torch

    monkeypatch.setenv("AIE_JIT_AUTO", "1")  # the product's default (tests/conftest.py switches it off for the suite)
    extra = {}
    if case.startswith("gtb_7_agents"):
        cfg = dict(C2, n_agents=7, episode_length=40, starting_agent_coin=2)
        if case.endswith("fast_rng"):
            extra = dict(rng_mode="fast")
    elif case == "one_step_economy_37_agents":
        cfg = dict(JIT_CASES["one_step_economy_37_agents"], episode_length=40)
    else:
        cfg = dict(JIT_CASES["build_gather_halfwidth"], episode_length=40)
    env, twin = [make_env(cfg, n_envs=256, device="cuda:0", **extra) for _ in range(2)]
    be, bt = env.backend, twin.backend
    assert bt.lib.aie_select_step_kernel(bt.handle, 1) == 0
    assert be.lib.aie_step_kernel_instance(be.handle) == -1 and bt.lib.aie_step_kernel_instance(bt.handle) == -1
    for e in (env, twin):
        e.seed(3)
        e.reset()
    t0, steps, swapped_at = time.time(), 0, None
    while steps < 60 or (swapped_at is None and time.time() - t0 < 120) or (swapped_at is not None and steps < swapped_at + 90):
        a, p = bt.sample_random_actions(seed=6)
        for b in (be, bt):
            b.step(a, p)
        steps += 1
        if steps % 40 == 0:
            for b in (be, bt):
                b.reset(b.tensors["done"])
        if swapped_at is None and be.lib.aie_step_kernel_instance(be.handle) == 1000:
            swapped_at = steps
        if steps % 20 == 0 or steps == swapped_at:
            torch.cuda.synchronize()
            for k in _state_keys(bt.tensors):
                assert torch.equal(bt.tensors[k], be.tensors[k]), "step %d (switch at %s): %s differs" % (steps, swapped_at, k)
            if swapped_at is None:
                time.sleep(0.05)  # (the compiler needs a second or two the first time; nothing to do with the device)
    assert swapped_at is not None, "the background specialisation never arrived (hiprtc missing?)"
    assert bt.lib.aie_step_kernel_instance(bt.handle) == -1


def test_reset_is_deterministic_across_environments():
    """Three environments of BASELINE configs[0]'s scenario (every reset draws its source layouts on the device, four
    wavefronts per replica), same seed: every tensor of the three arenas is identical after reset, and again after a
    second reset and a few steps.  (Found a store-data hazard in round 3: a 16-byte buffer store with a scalar offset
    register followed directly by a write of its data registers stored the NEW value in a quarter of the lanes.)"""
    import torch

    envs = [make_env(dict(C1_INSTANCE), n_envs=512, device="cuda:0") for _ in range(3)]
    for rep in range(2):
        for env in envs:
            env.seed(21 + rep)
            env.reset()
        for t in range(3 * rep):
            for env in envs:
                a, p = env.backend.sample_random_actions(seed=6)
                env.backend.step(a, p)
        torch.cuda.synchronize()
        a = envs[0].backend
        for other in envs[1:]:
            for k in _state_keys(a.tensors):
                assert torch.equal(a.tensors[k], other.backend.tensors[k]), "round %d: %s differs between identical environments" % (rep, k)


@pytest.mark.parametrize("words", [8, 64])
def test_draw_window_refills_do_not_change_the_stream(words):
    """The components draw from a small LDS window of tempered MT19937 words; running past it refills it from the
    state in HBM (with a twist when word 623 is passed).  With the window cut down to `words` every step of a 10-agent
    environment refills several times: the arena must stay bit-identical to the default window's."""
    import ctypes

    import torch

    cfg = dict(C2, n_agents=10, episode_length=60)
    with dev_library():  # aie_dev_set_draw_window is a development hook
        ref, small = [make_env(cfg, n_envs=192, device="cuda:0") for _ in range(2)]
        for env in (ref, small):
            env.seed(8)
            env.reset()
    lib = small.backend.lib
    lib.aie_dev_set_draw_window.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert lib.aie_dev_set_draw_window(small.backend.handle, words) == 0
    for t in range(130):  # two episode ends
        a, p = ref.backend.sample_random_actions(seed=9)
        ref.backend.step(a, p)
        small.backend.step(a, p)
        if t % 60 == 59:
            ref.reset(ref.backend.tensors["done"])
            small.reset(small.backend.tensors["done"])
        if t in (0, 1, 2, 29, 59, 60, 129):
            torch.cuda.synchronize()
            for k in _state_keys(ref.backend.tensors):
                assert torch.equal(ref.backend.tensors[k], small.backend.tensors[k]), "step %d: %s differs" % (t + 1, k)


def test_compile_time_instance_equals_generic_kernel_one_step_economy():
    """BASELINE configs[4] (one-step-economy, 100 agents) has a compile-time instance too (SimpleLabor's skills are
    run-time data): instance vs generic kernel, bit for bit, across episode ends with and without auto-reset."""
    import ctypes

    import torch

    rs = np.random.RandomState(11)
    cfg = dict(scenario_name="one-step-economy", n_agents=100, world_size=[1, 1], episode_length=2,
               components=[["SimpleLabor", {"skills": [float(x) for x in np.sort(1 + rs.rand(100) * 2)]}],
                           ["PeriodicBracketTax", {"bracket_spacing": "us-federal", "period": 1,
                                                   "tax_model": "model_wrapper"}]])
    pair = [make_env(cfg, n_envs=256, device="cuda:0") for _ in range(2)]
    for env in pair:
        env.seed(5)
        env.reset()
    b_spec, b_ref = pair[0].backend, pair[1].backend
    assert b_ref.lib.aie_select_step_kernel(b_ref.handle, 1) == 0  # AIE_KERNEL_GENERIC
    assert b_spec.lib.aie_step_kernel_instance(b_spec.handle) >= 0 and b_ref.lib.aie_step_kernel_instance(b_ref.handle) == -1
    for t in range(9):
        if t == 4:
            b_spec.set_auto_reset(True)
            b_ref.set_auto_reset(True)
        a, p = b_ref.sample_random_actions(seed=4)
        b_ref.step(a, p)
        b_spec.step(a, p)
        if t < 4 and bool(b_ref.tensors["done"][0]):
            b_ref.reset(b_ref.tensors["done"])
            b_spec.reset(b_spec.tensors["done"])
        torch.cuda.synchronize()
        for k in _state_keys(b_ref.tensors):
            assert torch.equal(b_ref.tensors[k], b_spec.tensors[k]), "step %d: %s differs" % (t + 1, k)


def _auto_reset_cases():
    rs = np.random.RandomState(4)
    ose = dict(scenario_name="one-step-economy", n_agents=12, world_size=[1, 1], episode_length=3,
               components=[["SimpleLabor", {"skills": [float(x) for x in np.sort(1 + rs.rand(12) * 2)]}],
                           ["PeriodicBracketTax", {"bracket_spacing": "us-federal", "period": 1,
                                                   "tax_model": "model_wrapper"}]])
    gtb = dict(C2, episode_length=7, starting_agent_coin=15, resource_regen_prob=0.05,
               env_layout_file="uniform_25x25_25each_65clump.txt")
    return {"one_step_economy": ose, "gather_trade_build": gtb}


@pytest.mark.parametrize("case", ["one_step_economy", "gather_trade_build"])
def test_auto_reset_equals_step_then_masked_reset(case):
    """aie_set_auto_reset: a replica that ends its episode restarts inside (one-step-economy) / right behind (others) the
    step -- the same state as step + reset(done) from the host, with the terminal step's rewards and `done` kept."""
    import torch

    cfg = _auto_reset_cases()[case]
    envs = [make_env(cfg, n_envs=32, device="cuda:0") for _ in range(2)]
    for env in envs:
        env.seed(6)
        env.reset()
    b_auto, b_ref = envs[0].backend, envs[1].backend
    # de-phase: half of the replicas restart one step in
    a, p = b_ref.sample_random_actions(seed=3)
    for b in (b_auto, b_ref):
        b.step(a, p)
        b.reset(torch.as_tensor((np.arange(32) % 2).astype(np.uint8), device="cuda:0"))
    b_auto.set_auto_reset(True)
    T = cfg["episode_length"]
    for t in range(3 * T + 2):
        a, p = b_ref.sample_random_actions(seed=3)
        b_auto.step(a, p)
        b_ref.step(a, p)
        done = b_ref.tensors["done"].clone()
        rew_a, rew_p = b_ref.tensors["rewards_a"].clone(), b_ref.tensors["rewards_p"].clone()
        if bool(done.any()):
            b_ref.reset(done)
        torch.cuda.synchronize()
        assert torch.equal(b_auto.tensors["done"], done), "step %d: done" % (t + 1)
        assert torch.equal(b_auto.tensors["rewards_a"], rew_a) and torch.equal(b_auto.tensors["rewards_p"], rew_p)
        for k, v in b_ref.tensors.items():
            if k in ("done", "rewards_a", "rewards_p", "sample_t") or k.startswith("metrics_"):
                continue
            assert torch.equal(v, b_auto.tensors[k]), "step %d: %s differs" % (t + 1, k)
    assert int(b_auto.tensors["completions"].min()) >= 2


# ---- hipGraph replay (a policy in the loop without the host between the steps) -----------------------
def _graph_env(cfg, E, seed, reward_log):
    env = make_env(cfg, n_envs=E, device="cuda:0")
    env.seed(seed)
    env.reset()
    log = env.backend.set_reward_log(5) if reward_log else None
    return env, log


@pytest.mark.parametrize("case", ["gtb_c2", "gtb_multi_action", "one_step_economy", "rows_22_12", "rows_6_52", "rows_6_12_five_agents"])
def test_policy_sampler_equals_its_cpu_restatement(case):
    """aie_sample_policy_actions (inverse-CDF sampling from the caller's logits under the action masks, one launch) against
    oracle/'s restatement: the same sub-action in every slot of every replica, over steps whose masks change (inventory-
    dependent trades, the planner's tax days), with NaN logits, ties and fully masked rows thrown in; the draw index
    advances once per call.  The cases cover the kernel's instances: rows of 50 + 22 entries (BASELINE configs[1]: 64- and
    32-lane segments), multi-action agents and a one-row planner of 148 entries / the one-step economy (the generic kernel:
    rows of different lengths, chunks of 64), and rows of 22 + 12, 6 + 52 and 6 + 12 entries (32 + 16, 16 + 64 and 16 + 16
    lanes per row; five agents: a last item with one row of four)."""
    import torch
    from oracle_lib import OracleEnv

    if case == "one_step_economy":
        cfg = _auto_reset_cases()["one_step_economy"]
    elif case.startswith("rows_"):
        def tax(disc):
            return ["PeriodicBracketTax", {"rate_disc": disc, "period": 10}]
        comps = {"rows_22_12": [["Build", {}], ["ContinuousDoubleAuction", {"max_num_orders": 5, "max_bid_ask": 3}], ["Gather", {}], tax(0.1)],
                 "rows_6_52": [["Build", {}], ["Gather", {}], tax(0.02)],
                 "rows_6_12_five_agents": [["Build", {}], ["Gather", {}], tax(0.1)]}[case]
        cfg = dict(C2, episode_length=30, components=comps, n_agents=5 if case.endswith("five_agents") else 4)
    else:
        cfg = dict(C2, episode_length=30, multi_action_mode_agents=(case == "gtb_multi_action"),
                   multi_action_mode_planner=(case != "gtb_multi_action"))
    E = 64
    env = make_env(cfg, n_envs=E, device="cuda:0", env_offset=1000)
    env.seed(3)
    env.reset()
    be = env.backend
    oracle = OracleEnv(env.build_config(), env.layout_planes())
    oracle.seed(3 + 1000)
    oracle.reset()
    a_buf, p_buf = be._action_buffers(0)
    MA, MP = be.tensors["obs_a_action_mask"].shape[-1], be.tensors["obs_p_action_mask"].shape[-1]
    g = torch.Generator(device="cpu").manual_seed(11)
    seen = set()
    for t in range(36):
        la = (torch.randn(E, be.n, MA, generator=g) * 3).float()
        lp = (torch.randn(E, MP, generator=g) * 3).float()
        la[t % E, 0, 1:4] = float("nan")           # NaN logits are masked entries
        la[(t + 1) % E, 1, :] = 0.25                # all-equal logits: the noise alone decides
        lp[(t + 2) % E, :] = -1e30                  # hopeless logits still yield an allowed entry
        la[(t + 3) % E, 2, :] = float("nan")        # a fully NaN row: NO-OP
        a, p = be.sample_policy_actions(la.to("cuda:0"), lp.to("cuda:0"), seed=77, env_offset=1000)
        torch.cuda.synchronize()
        assert np.array_equal(be.tensors["obs_a_action_mask"].cpu().numpy(), oracle.t["obs_a_action_mask"])
        wa, wp = oracle.sample_policy_actions(la.numpy(), lp.numpy(), 77, 1000, width_a=a.shape[-1], width_p=p.shape[-1])
        assert np.array_equal(a.cpu().numpy(), wa), "step %d: agents' sub-actions differ" % t
        assert np.array_equal(p.cpu().numpy(), wp), "step %d: planner's sub-actions differ" % t
        assert np.array_equal(be.tensors["sample_t"].cpu().numpy(), oracle.t["sample_t"]) and int(oracle.t["sample_t"][0]) == t + 1
        assert (wa[(t + 3) % E, 2] == 0).all()
        seen.update(np.unique(wa).tolist())
        env.step({"a": a, "p": p})
        oracle.step(wa.reshape(E, -1), wp, nthreads=4)
        if (t + 1) % int(cfg["episode_length"]) == 0:
            env.reset(be.tensors["done"])
            oracle.reset(oracle.t["done"].copy())
    assert int(be.tensors["error_flags"].abs().sum()) == 0  # every choice was an allowed one
    assert len(seen) > 5
    # NULL pairs: only the planner / only the agents
    before = a.clone()
    be._check(be.lib.aie_sample_policy_actions(be.handle, None, lp.to("cuda:0").data_ptr(), 77, 1000, None, p.data_ptr(), None))
    torch.cuda.synchronize()
    assert torch.equal(a, before)
    assert be.lib.aie_sample_policy_actions(be.handle, None, None, 77, 1000, a.data_ptr(), None, None) != 0


@pytest.mark.parametrize("E", [1, 5])
def test_policy_sampler_odd_batches(E):
    """A workgroup of the sampler holds two replicas: batches of one and of five replicas (a half-empty last workgroup)
    against the oracle, agents' and planner's rows, the draw index advancing."""
    import torch
    from oracle_lib import OracleEnv

    env = make_env(dict(C2, episode_length=30), n_envs=E, device="cuda:0", env_offset=7)
    env.seed(4)
    env.reset()
    be = env.backend
    oracle = OracleEnv(env.build_config(), env.layout_planes())
    oracle.seed(4 + 7)
    oracle.reset()
    MA, MP = be.tensors["obs_a_action_mask"].shape[-1], be.tensors["obs_p_action_mask"].shape[-1]
    g = torch.Generator(device="cpu").manual_seed(2)
    for t in range(6):
        la = (torch.randn(E, be.n, MA, generator=g) * 2).float()
        lp = (torch.randn(E, MP, generator=g) * 2).float()
        a, p = be.sample_policy_actions(la.to("cuda:0"), lp.to("cuda:0"), seed=9, env_offset=7)
        torch.cuda.synchronize()
        wa, wp = oracle.sample_policy_actions(la.numpy(), lp.numpy(), 9, 7, width_a=a.shape[-1], width_p=p.shape[-1])
        assert np.array_equal(a.cpu().numpy(), wa) and np.array_equal(p.cpu().numpy(), wp), t
        assert np.array_equal(be.tensors["sample_t"].cpu().numpy(), oracle.t["sample_t"])
        env.step({"a": a, "p": p})
        oracle.step(wa.reshape(E, -1), wp, nthreads=1)


def test_backend_lifecycle_free_now_and_failed_construction(monkeypatch):
    """A library-owned arena (forced here for a small environment) goes back to the device at free_now(), not when the
    garbage collector finds the last view; a constructor that fails after aie_create destroys the handle it made."""
    import gc

    import torch
    from ai_economist_amd.env import DeviceBackend

    monkeypatch.setenv("AIE_ARENA_VMM_MIN_MB", "1")
    env = make_env(C2, n_envs=512, device="cuda:0")
    env.seed(2)
    env.reset()
    be = env.backend
    info = be.arena_info()
    assert info["allocator"] == "vmm" and info["piece_mib"] == 64
    a, p = be.sample_random_actions(seed=1)
    be.step(a, p)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    del a, p
    be.free_now()
    assert be.handle is None and torch.cuda.mem_get_info()[0] >= free0 + (32 << 20)
    with pytest.raises(Exception):
        DeviceBackend(env.build_config(), None, device="cuda:0")  # fails behind aie_create (no layout planes)
    gc.collect()
    free1 = torch.cuda.mem_get_info()[0]
    for _ in range(3):
        with pytest.raises(Exception):
            DeviceBackend(env.build_config(), None, device="cuda:0")
    gc.collect()
    assert torch.cuda.mem_get_info()[0] >= free1 - (8 << 20), "failed constructions leak their arenas"


def test_policy_sampler_covid_collated_masks():
    """aie_sample_policy_actions on the COVID scenario's collated masks ([1 + levels][states] rows per replica, stride n):
    against the Python transcription of the sampler (tests/helpers.py: counter hash, fixed-operation exp, the prefix sums'
    order) -- every state's and the planner's sub-action, over steps whose masks change with the cool-downs."""
    import torch
    from helpers import counter_rng, load_covid_golden, sampler_entry_rng, sampler_pick_row

    cfg = dict(load_covid_golden("c4_covid_variant")["cfg"], scenario_name="CovidAndEconomySimulation")
    E = 6
    env = make_env(cfg, n_envs=E, device="cuda:0", env_offset=50)
    env.reset()
    be = env.backend
    n = be.n
    ma = be.tensors["obs_a_action_mask"]   # [E, 1 + levels, n]
    mp = be.tensors["obs_p_action_mask"]   # [E, MP]
    NL1, MP = ma.shape[1], mp.shape[-1]
    per_env = n + 1

    def pick(logits, mask, e, t, j):
        return sampler_pick_row(logits, mask, sampler_entry_rng(counter_rng(31, 50 + e, t, per_env), j))

    g = torch.Generator(device="cpu").manual_seed(5)
    changed = 0
    for t in range(12):
        la = (torch.randn(E, n, NL1, generator=g) * 2).float()
        lp = (torch.randn(E, MP, generator=g) * 2).float()
        a, p = be.sample_policy_actions(la.to("cuda:0"), lp.to("cuda:0"), seed=31, env_offset=50)
        torch.cuda.synchronize()
        mah, mph = ma.cpu().numpy(), mp.cpu().numpy()
        assert int(be.tensors["sample_t"][0]) == t + 1
        for e in range(E):
            for i in range(n):
                assert int(a[e, i, 0]) == pick(la[e, i].numpy(), mah[e, :, i], e, t, i), (t, e, i)
            assert int(p[e, 0]) == pick(lp[e].numpy(), mph[e], e, t, n), (t, e)
        changed += int((a != 0).sum())
        env.step({"a": a, "p": p})
    assert changed > 0


@pytest.mark.parametrize("reward_log", [False, True])
def test_step_is_hipgraph_replayable(reward_log):
    """policy -> aie_step (+ the auto-reset launch) captured ONCE on the caller's stream and replayed 64 times equals
    the same loop issued call by call -- the whole arena, bit for bit, and the reward log with its slot counters --
    and the eager twin equals the oracle stepped with the actions the policy chose.  Nothing aie_step needs travels by
    value from a host-side counter (include/aie.h), so a replay is a step, not a repetition of the captured one."""
    import torch

    from ai_economist_amd.rollout import GraphedStep, MaskedMLPPolicy
    from oracle_lib import OracleEnv

    cfg = dict(C2, episode_length=40, starting_agent_coin=12)  # 64 replays cross an episode end (auto-reset inside the graph)
    E, WARM, N = 48, 3, 64
    env_g, log_g = _graph_env(cfg, E, 5, reward_log)
    env_e, log_e = _graph_env(cfg, E, 5, reward_log)
    oracle = OracleEnv(env_e.build_config(), env_e.layout_planes())
    oracle.seed(5)
    oracle.reset()
    pol_g, pol_e = MaskedMLPPolicy(env_g.backend, seed=3), MaskedMLPPolicy(env_e.backend, seed=3)
    gs = GraphedStep(env_g, pol_g, auto_reset=True, warmup=WARM)  # (runs WARM eager steps, then captures one more: not executed)
    be_e = env_e.backend
    be_e.set_auto_reset(True)
    a_e, p_e = be_e._action_buffers(0)
    T = int(cfg["episode_length"])
    for t in range(WARM + N):
        pol_e(be_e.tensors, a_e, p_e)
        a_h, p_h = a_e.cpu().numpy().copy(), p_e.cpu().numpy().copy()
        be_e.step(a_e, p_e)
        oracle.step(a_h.reshape(E, -1), p_h, nthreads=4)
        if (t + 1) % T == 0:  # auto-reset: rewards / done of the terminal step stay, state and observations restart
            rew_a, rew_p, done = (oracle.t[k].copy() for k in ("rewards_a", "rewards_p", "done"))
            assert done.all()
            oracle.reset(done)
            oracle.t["rewards_a"][...], oracle.t["rewards_p"][...], oracle.t["done"][...] = rew_a, rew_p, done
    gs.replay(N)
    torch.cuda.synchronize()
    assert int(be_e.tensors["sample_t"][0]) == WARM + N  # (the policy's sampler advanced the draw index once per step)
    assert torch.equal(env_g.backend.arena, be_e.arena), "replayed loop != eager loop"
    if reward_log:
        assert torch.equal(log_g, log_e)
        assert int(be_e.tensors["rew_log_slot"][0]) == (WARM + N) % 5
    assert int(be_e.tensors["timestep"][0]) == (WARM + N) % T
    _compare_all(be_e, oracle, "eager twin of the replayed loop")
    assert np.array_equal(be_e.tensors["obs_a_action_mask"].cpu().numpy(), oracle.t["obs_a_action_mask"])
    # the policy's choices respected the masks it was shown: no replica raised an action error
    assert int(be_e.tensors["error_flags"].abs().sum()) == 0


@pytest.mark.parametrize("case", ["gather_trade_build", "one_step_economy", "covid"])
def test_fused_random_policy_is_hipgraph_replayable(case):
    """aie_step_sample_next (the bench's one-launch-per-step rollout) captured as a pair of launches -- the two action
    buffers swap roles every step -- and replayed equals the same calls issued one by one: the draw index is the
    replicas' own record field, not a host counter.  All three scenario families (three kernels)."""
    import torch

    cfg = dict(C2, episode_length=50) if case == "gather_trade_build" else _reward_log_cases()[case]
    T = int(cfg["episode_length"])
    envs = [make_env(cfg, n_envs=64, device="cuda:0", env_offset=128) for _ in range(2)]
    for env in envs:
        if case != "covid":
            env.seed(9)
        env.reset()
        env.backend.set_auto_reset(True)
    b0, b1 = envs[0].backend, envs[1].backend
    logs = [b.set_reward_log(4) for b in (b0, b1)]
    for b in (b0, b1):
        b.sample_random_actions(seed=77, env_offset=128, slot=0)

    def pair(b):
        a0, p0 = b._action_buffers(0)
        a1, p1 = b._action_buffers(1)
        b.step_sample_next(a0, p0, 77, 128, next_slot=1)
        b.step_sample_next(a1, p1, 77, 128, next_slot=0)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        pair(b1)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        pair(b1)
    n_pairs = min(30, T + 3)
    for _ in range(n_pairs):
        g.replay()
    for _ in range(1 + n_pairs):
        pair(b0)
    torch.cuda.synchronize()
    assert int(b0.tensors["sample_t"][0]) == 1 + 2 * (1 + n_pairs)
    assert torch.equal(b0.arena, b1.arena)
    assert torch.equal(logs[0], logs[1])
    for s in (0, 1):
        assert torch.equal(b0._action_buffers(s)[0], b1._action_buffers(s)[0])
        assert torch.equal(b0._action_buffers(s)[1], b1._action_buffers(s)[1])
