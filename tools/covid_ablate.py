#!/usr/bin/env python
"""Development tool: launch time of the COVID step kernel (BASELINE configs[3], recurrence instantiation) with parts of
its memory traffic switched off (CV_SKIP in csrc/aie_kernels_covid.hip, -DAIE_DEV build).  GPU only."""
import ctypes
import os

os.environ["AIE_DEV_LIB"] = "1"
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402

E = 8192
env = bench.make_env(bench._c4_cfg(), n_envs=E, device="cuda:0")
env.reset()
be = env.backend
be.lib.aie_dev_set_skip_mask.argtypes = [ctypes.c_void_p, ctypes.c_int]
cur = [be.sample_random_actions(1234, 0, slot=0), 0]


def step():
    cur[0] = be.step_sample_next(cur[0][0], cur[0][1], 1234, 0, next_slot=cur[1] ^ 1)
    cur[1] ^= 1


def timed(n):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(n):
        step()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / n * 1e3


NAMES = {1: "today's history byte store", 2: "history byte loads", 4: "observation stores", 8: "episode sums (RMW)",
         16: "state row stores"}
for _ in range(20):
    step()
base = None
for mask in (0, 1, 2, 3, 4, 8, 16, 31, 0):
    be.lib.aie_dev_set_skip_mask(be.handle, mask)
    for _ in range(5):
        step()
    env.reset()
    us = min(timed(100) for _ in range(3))
    base = base or us
    print("skip %-3d %-80s %.2f us (%+.2f)" % (mask, " + ".join(v for k, v in NAMES.items() if mask & k) or "nothing", us, us - base))
