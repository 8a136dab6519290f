"""Development: where does the time of a 20-step timed window go (the driver's `--steps 20 --warmup 5`)?
wall clock between the two synchronisations vs HIP-event time of the same launches, with / without the masked reset."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

env = bench.make_env(dict(bench.C2_CFG), n_envs=4096, device="cuda:0")
env.seed(1)
env.reset()
roll = bench.Rollout("C2", env, 0)
roll.prologue()
for rep in range(6):
    for _ in range(5):
        roll.step()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(20):
        roll.step(timed=(rep % 2 == 0))
    ev1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("rep %d timed_resets=%s: wall %.1f us, issue %.1f us, events %.1f us, resets in window %d" % (
        rep, rep % 2 == 0, (t2 - t0) * 1e6, (t1 - t0) * 1e6, ev0.elapsed_time(ev1) * 1e3, len(roll.reset_events)))
    roll.reset_events.clear()
