#!/usr/bin/env python
"""Development tool: launch time of the one-step-economy step kernel (BASELINE configs[4]) with parts of its memory
traffic switched off (OSE_SKIP in csrc/aie_kernels_ose.hip, -DAIE_DEV build): what does each stream cost?  GPU only."""
import ctypes
import os

os.environ["AIE_DEV_LIB"] = "1"  # the aie_dev_* hooks live in libaie_hip_dev.so (-DAIE_DEV) only
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
from helpers import make_env  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = make_env(bench._c5_cfg(), n_envs=E, device="cuda:0")
env.seed(1)
env.reset()
be = env.backend
be.set_auto_reset(True)
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
    return ev0.elapsed_time(ev1) / n


NAMES = {1: "flat rows (45.2 KB)", 2: "mask rows (40.4 KB)", 4: "metrics atomics (300)", 8: "small observation tensors",
         16: "record store (5.7 KB)", 32: "record load (5.7 KB)"}
for _ in range(10):
    step()
base = None
for mask in (0, 1, 2, 3, 4, 8, 16, 32, 4 | 8 | 16 | 32, 1 | 2 | 8, 63, 0):
    be.lib.aie_dev_set_skip_mask(be.handle, mask)
    for _ in range(4):
        step()
    ms = timed(40)
    if base is None:
        base = ms
    what = " + ".join(v for k, v in NAMES.items() if mask & k) or "nothing"
    print("skip %-3d %-90s %.3f ms  (%+.3f)" % (mask, what, ms, ms - base))

