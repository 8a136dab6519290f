"""The HIP path against the oracle AT THE BASELINE SIZES (BASELINE.json configs[1..4]): every replica, every field.

The smaller rollouts of test_gpu_parity.py never fill the chip; here all workgroups of a launch are resident at
once (C2/C3: 4096 replicas = 16 workgroups per CU, the XCD-aware replica mapping in `replica_of_block`), C5 runs its
65 536 replicas in several rounds of workgroups and C4 its 8192.  Integer state, order books and the MT19937 keys are
compared bit for bit, floats within the tolerances of test_gpu_parity.py / test_covid_golden.py."""
import os

import numpy as np
import pytest

from helpers import make_env
from test_gpu_parity import C2, _compare_all

pytestmark = pytest.mark.gpu

NTHREADS = max(1, min(32, len(os.sched_getaffinity(0))))


def _chunks(E, size):
    return [slice(lo, min(E, lo + size)) for lo in range(0, E, size)]


@pytest.mark.parametrize("n_agents", [4, 10])
def test_c2_c3_full_batch_matches_oracle(n_agents):
    """BASELINE configs[1] (4 agents) and one GPU's share of configs[2] (10 agents): 4096 replicas of the
    benchmark workload itself, 130 steps = across the first tax day (period 100), the first order expiries
    (order_duration 50) and with books filling up; every field of every replica every 10 steps."""
    import torch
    from oracle_lib import OracleEnv

    E, T = 4096, 130
    cfg = dict(C2, n_agents=n_agents)
    env = make_env(cfg, n_envs=E, device="cuda:0")
    env.seed(1)
    env.reset()
    be = env.backend
    oracle = OracleEnv(env.build_config(), env.layout_planes())
    oracle.seed(1)
    oracle.reset()
    _compare_all(be, oracle, "C2/C3 n=%d reset" % n_agents)
    for t in range(T):
        a, p = be.sample_random_actions(seed=1234)
        env.step({"a": a, "p": p})
        torch.cuda.synchronize()
        oracle.step(a.cpu().numpy(), p.cpu().numpy(), nthreads=NTHREADS)
        if (t + 1) % 10 == 0 or t + 1 in (51, 52, 100, 101):
            _compare_all(be, oracle, "n=%d step %d" % (n_agents, t + 1))
    trades = int(oracle.t["metrics_cda_trades"][..., 0].sum())
    assert int(oracle.t["metrics_tax_days"].min()) == 1 and int(be.tensors["timestep"].min()) == T
    assert trades > 0, "no trade in 4096 replicas x %d steps?" % T


def test_c2_dense_full_batch_matches_oracle_with_autowarmup():
    """4096 replicas of a busier C2 (dense layout, regeneration, energy warm-up "auto"): builds, trades, gathers in
    every step, an episode boundary with reset, and the integer auto_warmup counter exact in all of them."""
    import torch
    from oracle_lib import OracleEnv

    E, T = 4096, 140
    cfg = dict(C2, episode_length=120, starting_agent_coin=15, resource_regen_prob=0.05,
               env_layout_file="uniform_25x25_25each_65clump.txt", energy_warmup_constant=50.0,
               energy_warmup_method="auto", isoelastic_eta=0.23)
    env = make_env(cfg, n_envs=E, device="cuda:0")
    env.seed(3)
    env.reset()
    be = env.backend
    oracle = OracleEnv(env.build_config(), env.layout_planes())
    oracle.seed(3)
    oracle.reset()
    for t in range(T):
        a, p = be.sample_random_actions(seed=77)
        env.step({"a": a, "p": p})
        torch.cuda.synchronize()
        oracle.step(a.cpu().numpy(), p.cpu().numpy(), nthreads=NTHREADS)
        if (t + 1) % 20 == 0:
            _compare_all(be, oracle, "dense C2 step %d" % (t + 1))
        if t + 1 == 120:
            assert bool(be.tensors["done"].all())
            env.reset(be.tensors["done"])
            oracle.reset(oracle.t["done"].copy())
            _compare_all(be, oracle, "dense C2 episode reset")
    aw = be.tensors["auto_warmup"].cpu().numpy()
    assert np.array_equal(aw, oracle.t["auto_warmup"]) and aw.max() > 0


def test_c5_full_batch_matches_oracle():
    """BASELINE configs[4]: one-step-economy, 100 agents, 65 536 replicas, one 2-step episode + reset + one more
    step, compared in blocks of 4096 replicas (the observation tensors alone are 5.8 GB)."""
    import torch
    from oracle_lib import OracleEnv

    E = 65536
    rs = np.random.RandomState(4)
    cfg = dict(scenario_name="one-step-economy", n_agents=100, world_size=[1, 1], episode_length=2,
               components=[["SimpleLabor", {"skills": [float(x) for x in np.sort(1 + rs.rand(100) * 2)]}],
                           ["PeriodicBracketTax", {"bracket_spacing": "us-federal", "period": 1,
                                                   "tax_model": "model_wrapper"}]])
    env = make_env(cfg, n_envs=E, device="cuda:0")
    env.seed(9)
    env.reset()
    be = env.backend
    oracle = OracleEnv(env.build_config(), env.layout_planes())
    oracle.seed(9)
    oracle.reset()

    def compare(where):
        for sl in _chunks(E, 4096):
            _compare_all(be, oracle, "%s replicas %d.." % (where, sl.start), sl=sl)

    compare("C5 reset")
    for t in range(3):
        a, p = be.sample_random_actions(seed=23)
        env.step({"a": a, "p": p})
        torch.cuda.synchronize()
        oracle.step(a.cpu().numpy(), p.cpu().numpy(), nthreads=NTHREADS)
        compare("C5 step %d" % (t + 1))
        if t == 1:
            assert bool(be.tensors["done"].all())
            env.reset(be.tensors["done"])
            oracle.reset(oracle.t["done"].copy())
            compare("C5 episode reset")


_C4_ORACLE = {}


@pytest.mark.parametrize("recurrence", [False, True], ids=["window-sums", "recurrence"])
def test_c4_full_batch_matches_oracle(recurrence):
    """BASELINE configs[3]: COVID, 51 states + planner, 8192 replicas, 64 days, every replica its own action
    stream; both instantiations of the step kernel (the default window sums and the O(1) recurrence that bench.py's
    C4 line runs).  The NumPy oracle materialises the reference's [n, filters, 600] signal tensor per replica, so it
    runs in cache-sized blocks of 8 replicas on worker subprocesses (tests/covid_pool.py), once for both."""
    import torch
    from covid_pool import run_blocks
    from helpers import load_covid_golden
    from test_covid_golden import STATE_TOL, hip_env

    E, T, B = 8192, 64, 8
    cfg = load_covid_golden("c4_covid_51ag")["cfg"]
    ns = dict(cfg["components"])["FederalGovernmentSubsidy"]["num_subsidy_levels"]
    env = hip_env(cfg, n_envs=E, filter_recurrence=recurrence)
    env.reset()
    t = env.tensors
    rng = np.random.RandomState(8)
    acts_a = rng.randint(0, 11, size=(T, E, 51)).astype(np.int32)
    acts_a[rng.rand(T, E, 51) < 0.5] = 0
    acts_p = rng.randint(0, ns + 1, size=(T, E)).astype(np.int32)
    check_at = (1, 30, T)
    if "want" not in _C4_ORACLE:
        _C4_ORACLE["want"] = run_blocks(cfg, acts_a, acts_p, B, check_at, workers=min(16, NTHREADS))
    want = _C4_ORACLE["want"]
    err_a, err_p = [], []
    for k in range(1, T + 1):
        env.step({"a": torch.as_tensor(acts_a[k - 1], device="cuda"),
                  "p": torch.as_tensor(acts_p[k - 1][:, None], device="cuda")})
        np.testing.assert_allclose(t["rewards_a"].cpu().numpy(), want["rew_a"][k - 1], rtol=0, atol=1e-5,
                                   err_msg="C4 day %d agent rewards" % k)
        np.testing.assert_allclose(t["rewards_p"].cpu().numpy(), want["rew_p"][k - 1], rtol=0, atol=1e-5,
                                   err_msg="C4 day %d planner reward" % k)
        err_a.append(np.abs(t["rewards_a"].cpu().numpy().astype(np.float64) - want["rew_a"][k - 1]).ravel())
        err_p.append(np.abs(t["rewards_p"].cpu().numpy().astype(np.float64) - want["rew_p"][k - 1]).ravel())
        if k in check_at:
            st = want["state"][k]
            for name, tol in STATE_TOL.items():
                np.testing.assert_allclose(t[name].cpu().numpy().astype(np.float64), st[name], rtol=tol, atol=1e-3,
                                           err_msg="C4 day %d %s" % (k, name))
            assert np.array_equal(t["cooldown_until"].cpu().numpy(), st["cooldown_until"]), "day %d" % k
            assert np.array_equal(t["subsidy_level"].cpu().numpy(), st["subsidy_level"]), "day %d" % k
            for name, v in want["obs"][k].items():
                np.testing.assert_allclose(t[name].cpu().numpy().reshape(v.shape), v, rtol=1e-5, atol=1e-6,
                                           err_msg="C4 day %d %s" % (k, name))
    assert not bool(t["done"].any()) and int(t["timestep"].min()) == T
    # the reward-error distribution behind the 1e-5 tolerance (rewards are min-max normalised differences of nearly
    # equal float32 numbers; the reference's own CPU<->CUDA tolerance is not in its repository)
    ea, ep = np.concatenate(err_a), np.concatenate(err_p)
    q = lambda x: [float(np.quantile(x, p)) for p in (0.5, 0.99, 0.9999, 1.0)]  # noqa: E731
    print("C4 (%s) reward |error| (median, p99, p99.99, max): agents %s of %d, planner %s of %d" % (
        "recurrence" if recurrence else "window sums", ["%.2e" % v for v in q(ea)], ea.size,
        ["%.2e" % v for v in q(ep)], ep.size))


def test_c1_uniform_layouts_drawn_on_device_full_batch():
    """BASELINE configs[0]'s scenario (uniform 15x15, Build + Gather) at 4096 replicas: every replica draws its own
    source layout INSIDE the reset kernel (no host round trip: aie_kernels.hip layout_generate), bit-identical to
    the restatement's layouts (which the CPU tests pin to the live reference); two episodes with auto-reset."""
    import time

    import torch
    from oracle_lib import OracleEnv

    E = 4096
    cfg = dict(scenario_name="uniform/simple_wood_and_stone", n_agents=4, world_size=[15, 15], episode_length=12,
               components=[["Build", {}], ["Gather", {}]], starting_agent_coin=10, starting_stone_coverage=0.10,
               starting_wood_coverage=0.10)
    env = make_env(cfg, n_envs=E, device="cuda:0")
    env.seed(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    env.reset()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    be = env.backend
    oracle = OracleEnv(env.build_config(), env.layout_planes())
    oracle.seed(1)
    oracle.reset()
    _compare_all(be, oracle, "C1 reset")
    flags = be.tensors["cell_flags"].reshape(E, -1)
    assert len({bytes(f.tobytes()) for f in flags[:64].cpu().numpy()}) == 64  # every replica its own layout
    assert dt < 2.0, "a 4096-replica reset took %.2f s: is the layout generation back on the host?" % dt
    be.set_auto_reset(True)
    for t in range(2 * cfg["episode_length"] + 2):
        a, p = be.sample_random_actions(seed=5)
        env.step({"a": a, "p": p})
        torch.cuda.synchronize()
        oracle.step(a.cpu().numpy(), p.cpu().numpy(), nthreads=NTHREADS)
        done = oracle.t["done"].copy()
        rew, rew_p = oracle.t["rewards_a"].copy(), oracle.t["rewards_p"].copy()
        if done.any():
            oracle.reset(done)
            oracle.t["done"][...] = done      # auto-reset keeps the terminal step's done / rewards
            oracle.t["rewards_a"][...] = rew
            oracle.t["rewards_p"][...] = rew_p
        if (t + 1) % 6 == 0:
            _compare_all(be, oracle, "C1 step %d" % (t + 1))
    assert int(be.tensors["completions"].min()) == 2


def _drive_rollout_against_oracle(wl, E, timed_steps, compare_every):
    """bench.Rollout -- the very object bench.py times -- with its backend calls mirrored on the CPU oracle: every action
    pair a step launch consumed (the fused draws of aie_step_sample_next included), every masked block reset, in order."""
    import sys

    import torch

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import bench
    from oracle_lib import OracleEnv
    from test_gpu_parity import _compare_all

    W = bench.WORKLOADS[wl]
    env = bench.make_env(W["cfg"](), n_envs=E, device="cuda:0")
    env.seed(bench.ENV_SEED)
    env.reset()
    be = env.backend
    oracle = OracleEnv(env.build_config(), env.layout_planes())
    oracle.seed(bench.ENV_SEED)
    oracle.reset()
    roll = bench.Rollout(wl, env, 0)
    calls = []
    real_step, real_reset = be.step_sample_next, be.reset

    def spy_step(a, p, *args, **kw):
        torch.cuda.synchronize()
        calls.append(("step", a.cpu().numpy().copy(), p.cpu().numpy().copy()))
        return real_step(a, p, *args, **kw)

    def spy_reset(mask=None):
        calls.append(("reset", None if mask is None else mask.cpu().numpy().astype(np.uint8).copy()))
        return real_reset(mask)

    be.step_sample_next, be.reset = spy_step, spy_reset
    auto = roll.auto_reset

    def replay():
        for kind, x, *rest in calls:
            if kind == "step":
                oracle.step(x, rest[0], nthreads=8)
                if auto:  # aie_set_auto_reset: the replicas a step finishes restart behind it
                    d = oracle.t["done"].astype(np.uint8).copy()
                    if d.any():
                        rew_a, rew_p = oracle.t["rewards_a"].copy(), oracle.t["rewards_p"].copy()
                        oracle.reset(d)
                        oracle.t["done"][:] = d  # (the terminal step's rewards and done stay)
                        oracle.t["rewards_a"][:] = rew_a
                        oracle.t["rewards_p"][:] = rew_p
            else:
                oracle.reset(x)
        calls.clear()

    n_prologue = roll.prologue()
    replay()
    torch.cuda.synchronize()
    _compare_all(be, oracle, "%s after the de-phasing prologue (%d steps)" % (wl, n_prologue))
    if not auto:
        roll.warm_reset_path()
    resets_seen = 0
    for k in range(1, timed_steps + 1):
        roll.step(timed=True)
        resets_seen += sum(1 for c in calls if c[0] == "reset" and c[1] is not None and c[1].any())
        if k % compare_every == 0 or k == timed_steps:
            replay()
            torch.cuda.synchronize()
            _compare_all(be, oracle, "%s timed path, step %d" % (wl, k))
    return roll, resets_seen


def test_bench_rollout_path_matches_oracle_c2():
    """the timed path of bench.py as a whole -- staggered prologue, masked block resets on the host-known
    schedule, one aie_step_sample_next launch per step -- against the oracle: BASELINE configs[1], 512 replicas, the
    1000-step prologue plus 60 steps of the timed loop (three block resets), every tensor every 10 steps."""
    roll, resets = _drive_rollout_against_oracle("C2", 512, 60, 10)
    assert roll.G == 50 and roll.fused and resets >= 3


def test_bench_rollout_path_matches_oracle_c5():
    """The same for BASELINE configs[4] (one-step-economy, auto-reset: a replica restarts inside the step launch that
    ends its 2-step episode)."""
    roll, _ = _drive_rollout_against_oracle("C5", 256, 12, 2)
    assert roll.auto_reset and roll.G == 1


def test_c2_full_batch_crosses_an_episode_end_on_the_compile_time_instance():
    """4096 replicas of BASELINE configs[1] with a shortened episode (60 steps: the instance's family
    covers it, episode_length is read from the run-time block) across two episode ends, reset by the `done` mask --
    step and reset kernels of the compile-time instance against the oracle, every field of every replica."""
    import torch
    from oracle_lib import OracleEnv

    E, T, EP = 4096, 130, 60
    cfg = dict(C2, episode_length=EP)
    env = make_env(cfg, n_envs=E, device="cuda:0")
    env.seed(2)
    env.reset()
    be = env.backend
    assert be.lib.aie_step_kernel_instance(be.handle) >= 0, "the shortened episode must stay in C2's instance family"
    oracle = OracleEnv(env.build_config(), env.layout_planes())
    oracle.seed(2)
    oracle.reset()
    for t in range(T):
        a, p = be.sample_random_actions(seed=99)
        be.step(a, p)
        torch.cuda.synchronize()
        oracle.step(a.cpu().numpy(), p.cpu().numpy(), nthreads=NTHREADS)
        if (t + 1) % EP == 0:
            _compare_all(be, oracle, "terminal step %d" % (t + 1))
            assert bool(be.tensors["done"].all())
            be.reset(be.tensors["done"])
            oracle.reset(np.ones(E, np.uint8))
            torch.cuda.synchronize()
            _compare_all(be, oracle, "reset after step %d" % (t + 1))
        elif (t + 1) % 10 == 0 or (t + 1) % EP == 1:
            _compare_all(be, oracle, "step %d" % (t + 1))
    assert int(be.tensors["completions"].min()) == 2


@pytest.mark.parametrize("n_agents,rng_mode", [(4, "numpy"), (10, "numpy"), (4, "fast"), (10, "fast")])
def test_c2_c3_full_batch_whole_episode_and_its_end(n_agents, rng_mode):
    """BASELINE configs[1] / one GPU's share of configs[2] exactly as benchmarked -- 4096 replicas, 1000-step episodes -- through a WHOLE episode, its
    end and the reset behind it: ten tax days, twenty order-expiry horizons, every field of every replica every 100
    steps, at the terminal step, after the reset and five steps into the second episode.  "fast": the C2f / C3f
    workloads (counter-based stream) against the oracle's restatement of the same generator."""
    import torch
    from oracle_lib import OracleEnv

    E, EP = 4096, 1000
    env = make_env(dict(C2, n_agents=n_agents), n_envs=E, device="cuda:0", rng_mode=rng_mode)
    env.seed(4)
    env.reset()
    be = env.backend
    oracle = OracleEnv(env.build_config(), env.layout_planes())
    oracle.seed(4)
    oracle.reset()
    for t in range(EP + 5):
        a, p = be.sample_random_actions(seed=31)
        be.step(a, p)
        torch.cuda.synchronize()
        oracle.step(a.cpu().numpy(), p.cpu().numpy(), nthreads=NTHREADS)
        if t + 1 == EP:
            _compare_all(be, oracle, "terminal step")
            assert bool(be.tensors["done"].all())
            be.reset(be.tensors["done"])
            oracle.reset(np.ones(E, np.uint8))
            torch.cuda.synchronize()
            _compare_all(be, oracle, "reset behind the first episode")
        elif (t + 1) % 100 == 0 or t + 1 > EP:
            _compare_all(be, oracle, "step %d" % (t + 1))
    assert int(oracle.t["metrics_tax_days"].min()) == 0 and int(be.tensors["completions"].min()) == 1
