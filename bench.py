#!/usr/bin/env python
"""bench.py -- random-policy rollout throughput of the batched Foundation env.step().

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], "C2"): layout_from_file/simple_wood_and_stone, 25x25
quadrant layout, 4 mobile agents + planner, components Build + ContinuousDoubleAuction
(max_num_orders 5) + Gather + PeriodicBracketTax (model_wrapper, us-federal, period 100),
starting_agent_coin 10, episode_length 1000, 4096 env replicas PER GPU (weak scaling),
uniform random actions from a counter RNG keyed (seed, global replica, t, slot).

One "step" = sample actions on device + one env.step() of every replica (+ a batched
reset whenever an episode ends; with N > 1 also the per-step (reward, done) gather to
rank 0, the path's only exchange).  Inputs are resident in HBM before the timed region.

Prints ONE JSON line (rank 0) with the contract fields plus:
  roofline     dominant kernel (aie_step_kernel): algorithmic bytes per launch / average
               launch duration measured live with HIP events on the launch stream
  cpu_baseline the C restatement of the reference step (oracle/, kind "port") timed on
               this host's cores on a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("OMP_WAIT_POLICY", "passive")
ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s achievable

WORKLOAD = dict(
    scenario_name="layout_from_file/simple_wood_and_stone", n_agents=4, world_size=[25, 25],
    episode_length=1000,
    components=[["Build", {}], ["ContinuousDoubleAuction", {"max_num_orders": 5}],
                ["Gather", {}], ["PeriodicBracketTax", {}]],
    starting_agent_coin=10, env_layout_file="quadrant_25x25_20each_30clump.txt")
WORKLOAD_NAME = "C2"  # BASELINE.json configs[1]; tools/bench_c3.py reuses this file for configs[2]
ENVS_PER_GPU = 4096
ACTION_SEED = 1234
ENV_SEED = 1


def algorithmic_bytes_per_env_step(be):
    """B_alg of SURVEY.md 8(d), recomputed from the final layouts: observation tensors in
    the reference's own format + state record read+write + actions + rewards/done."""
    import torch

    obs = 0
    for k, t in be.tensors.items():
        if k.startswith("obs_"):
            obs += t[0].numel() * t.element_size()
    # per-replica record bytes = stride of any record field along the env axis
    rec = be.descs["cells"][2][0]
    n = be.n
    act = n * 4 + be._act_p_width() * 4
    rew = (n + 1) * 4 + 1
    return dict(obs=obs, state_rw=2 * rec, act=act, rew_done=rew, total=obs + 2 * rec + act + rew)


def measured_traffic(envs_per_gpu):
    """HBM bytes per aie_step_kernel launch from the committed rocprofv3 PMC summary
    (profiles/*_pmc.json, made by tools/pmc_summary.py from separate FETCH_SIZE / WRITE_SIZE
    passes of this same command).  Only valid for the default batch size."""
    import glob

    import re

    def version(path):  # r01_v10_pmc.json after r01_v9_pmc.json
        return [int(x) for x in re.findall(r"\d+", os.path.basename(path))]

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")), key=version)
    if not files or envs_per_gpu != ENVS_PER_GPU or WORKLOAD_NAME != "C2":
        return None, None
    d = json.load(open(files[-1]))
    return d["hbm_bytes_per_launch"], os.path.relpath(files[-1], ROOT)


def usable_cores():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:  # cgroup v2 CPU quota
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def cpu_baseline(seconds_target=12.0):
    """Times the CPU restatement (oracle/aie_oracle.c, OpenMP over replicas) on a bounded
    sample of the same workload.  The thread count that gives the best throughput among
    {1, 8, 32, all usable cores} is reported together with its core count."""
    import numpy as np

    from helpers import make_env
    from oracle_lib import OracleEnv

    E = 1024
    env = make_env(WORKLOAD, n_envs=E)
    o = OracleEnv(env.build_config(), env.layout_planes())
    o.seed(ENV_SEED)
    o.reset()
    rng = np.random.RandomState(ACTION_SEED)
    n = env.n_agents
    acts = rng.randint(0, 50, size=(20, E, n)).astype(np.int32)
    acts_p = rng.randint(0, 22, size=(20, E, 7)).astype(np.int32)
    ncores = usable_cores()
    cands = sorted({c for c in (1, 8, 32, ncores) if c <= ncores})
    best = None
    per = seconds_target / len(cands)
    for th in cands:
        for t in range(2):
            o.step(acts[t], acts_p[t], nthreads=th)
        steps = 0
        t0 = time.perf_counter()
        while True:
            for t in range(20):
                o.step(acts[t], acts_p[t], nthreads=th)
            steps += 20
            if time.perf_counter() - t0 > per:
                break
        dt = time.perf_counter() - t0
        rate = E * n * steps / dt
        if best is None or rate > best[0]:
            best = (rate, th, steps, dt)
    rate, th, steps, dt = best
    return dict(value=rate, unit="agent-steps/s", cores=th, kind="port",
                sample="%d replicas x %d steps of the same C2 workload (%.1f s), C restatement of the reference "
                       "step, OpenMP over replicas; best of thread counts %s (host reports %d usable cores)"
                       % (E, steps, dt, cands, ncores))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-gather", action="store_true",
                    help="development: run the N > 1 reward-log gather in a 1-rank group (under torch.distributed.run)")
    args = ap.parse_args()

    import torch

    from ai_economist_amd.sharding import RewardLogGather, dist_info
    from helpers import make_env

    rank, local_rank, world = dist_info()
    if args.gpus > 1 or world > 1 or args.force_gather:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
        assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    E = args.envs_per_gpu
    env_offset = rank * E
    env = make_env(WORKLOAD, n_envs=E, device=device, env_offset=env_offset)
    env.seed(ENV_SEED)
    env.reset()
    be = env.backend
    n = env.n_agents
    T_ep = env.episode_length
    # N > 1: (reward, done) of every step travel to the learner rank, 64 steps per collective, straight from the
    # log the step kernel fills (aie_set_reward_log) -- no per-step launches or collectives beside the step
    gather, gather_note = None, None
    if world > 1 or args.force_gather:
        try:
            gather = RewardLogGather(be, steps_per_gather=64, force_collective=args.force_gather)
        except Exception as exc:  # keep the scaling run alive; the JSON line says what happened
            gather_note = "reward gather disabled: %r" % (exc,)
    t_in_ep = 0
    # uniform random policy: the actions of step t+1 are drawn inside the launch of step t
    # (aie_step_sample_next: the replica's second wavefront is idle during the serial dynamics),
    # so a rollout step is ONE launch; the first draw is a launch of its own.
    cur = be.sample_random_actions(ACTION_SEED, env_offset, slot=0)
    slot = 0

    def one_step():
        nonlocal t_in_ep, cur, slot, gather, gather_note
        cur = be.step_sample_next(cur[0], cur[1], ACTION_SEED, env_offset, next_slot=slot ^ 1)
        slot ^= 1
        t_in_ep += 1
        if gather is not None:
            try:
                gather.after_step()
            except Exception as exc:
                gather_note = "reward gather disabled after an error: %r" % (exc,)
                gather = None
        if t_in_ep == T_ep:  # all replicas are in lock-step: every one is done now
            be.reset(be.tensors["done"])
            t_in_ep = 0

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    barrier()
    # one pair of HIP events on the launch stream around the whole timed region: K back-to-back
    # aie_step_kernel launches (one per step, see aie_step_sample_next) -> average launch period.
    # (Bracketing every single launch with its own event pair adds ~4 us of queue packets per step.)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        one_step()
    ev1.record()
    if gather is not None:
        gather.finish()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- roofline of the dominant kernel ----
    roof = None
    if rank == 0:
        avg_ms = ev0.elapsed_time(ev1) / args.steps
        b = algorithmic_bytes_per_env_step(be)
        bytes_per_launch = b["total"] * E
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        traffic, traffic_src = measured_traffic(E)
        roof = dict(bound="hbm", kernel="aie_step_kernel", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=achieved / HBM_PEAK_GBS, traffic=traffic, traffic_source=traffic_src,
                    algorithmic_bytes_per_launch=bytes_per_launch,
                    algorithmic_bytes_per_agent_step=b["total"] / n, bytes_breakdown_per_env_step=b,
                    avg_launch_ms=avg_ms, launches_timed=args.steps,
                    note="achieved = bytes of the reference-format observations + state a step produces / consumes "
                         "(algorithmic) per launch time; the kernel keeps the map observations in place and rewrites "
                         "only what a step changes, so the measured HBM traffic is lower than the algorithmic bytes")

    if rank == 0:
        agent_steps = world * E * n * args.steps
        value = agent_steps / elapsed
        out = {
            "metric": "agent-steps/sec, gather-trade-build 25x25 %d-agent batched envs" % n,
            "value": value, "unit": "agent-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/i32 state + f64 coin/utility (f32 observations)", "data": "synthetic",
            "config": {
                "workload": WORKLOAD_NAME + ": layout_from_file/simple_wood_and_stone 25x25 quadrant layout, %d agents + planner, " % n +
                            "Build+ContinuousDoubleAuction(max_num_orders=5)+Gather+PeriodicBracketTax, "
                            "episode_length 1000, uniform random policy, mobile agents counted (planner excluded)",
                "envs_per_gpu": E, "global_envs": world * E, "n_agents": n,
                "rng": "per-replica NumPy-legacy MT19937 (parity mode)",
                "policy": "uniform random (counter RNG); the draw for step t+1 happens inside the launch of step t "
                          "(aie_step_sample_next), one launch per step",
                "parallelism": "replica sharding, %d rank(s); (reward, done) of every step gathered to rank 0 over RCCL, "
                               "64 steps per collective, overlapped with the steps" % world
                if world > 1 else "single GPU",
            },
            "roofline": roof,
        }
        if gather_note:
            out["config"]["exchange_note"] = gather_note
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
