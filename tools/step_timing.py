#!/usr/bin/env python
"""Development: microseconds per launch of the gather-trade-build step on configs[1] (4096 replicas) through three
paths -- aie_step with fixed actions, aie_step_sample_next, and the traced twin of the instance (AIE_DEV_LIB=1) --
200 back-to-back launches each, HIP events.   [AIE_HIP_LIBRARY=...] python tools/step_timing.py [n_agents] [E]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
from helpers import make_env  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
E = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
env = make_env(dict(bench.C2_CFG, n_agents=n), n_envs=E, device="cuda:0")
env.seed(1)
env.reset()
be = env.backend
for _ in range(300):
    a, p = be.sample_random_actions(1234)
    be.step(a, p)
torch.cuda.synchronize()


def timed(fn, reps=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / reps


a, p = be.sample_random_actions(1234)
snap = be.arena.clone()
res = {"step(fixed actions)": timed(lambda: be.step(a, p))}
be.arena.copy_(snap)
cur = [be.sample_random_actions(bench.ACTION_SEED, 0, slot=0)]
slot = [0]


def fused():
    cur[0] = be.step_sample_next(cur[0][0], cur[0][1], bench.ACTION_SEED, 0, next_slot=slot[0] ^ 1)
    slot[0] ^= 1


res["step_sample_next"] = timed(fused)
print("instance", be.lib.aie_step_kernel_instance(be.handle), " ".join("%s %.2f us" % kv for kv in res.items()), flush=True)
