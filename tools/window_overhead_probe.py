#!/usr/bin/env python
"""Where the wall time of a 20-step bench window goes beyond the kernels: host timestamps around the pieces of
bench.py's timed region (same Rollout object, same calls)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

W = bench.WORKLOADS["C2"]
env = bench.make_env(W["cfg"](), n_envs=W["envs"], device="cuda:0")
env.seed(bench.ENV_SEED)
env.reset()
roll = bench.Rollout("C2", env, 0)
roll.prologue()
roll.warm_reset_path()
for _ in range(5):
    roll.step(timed=True)
torch.cuda.synchronize()
for rep in range(4):
    roll.reset_events.clear()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    T = []
    t0 = time.perf_counter()
    ev0.record()
    T.append(("ev0.record", time.perf_counter() - t0))
    for k in range(20):
        roll.step(timed=True)
        if k in (0, 1, 19):
            T.append(("step %d issued" % k, time.perf_counter() - t0))
    ev1.record()
    T.append(("ev1.record", time.perf_counter() - t0))
    n = 0
    while not ev1.query():
        n += 1
    T.append(("ev1 seen (%d polls)" % n, time.perf_counter() - t0))
    torch.cuda.synchronize()
    T.append(("synchronize", time.perf_counter() - t0))
    region = ev0.elapsed_time(ev1)
    print("rep %d: GPU region %.1f us | " % (rep, region * 1e3) + " | ".join("%s %.1f" % (a, b * 1e6) for a, b in T))
