#!/usr/bin/env python
"""Development tool: runs 20 launches of aie_step_kernel per dev skip mask so that a
`rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU ...` run can attribute dynamic instruction
counts to phases (parse with tools/phase_insts_report.py)."""
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

MASKS = [0, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 63]
env = make_env(bench.C2_CFG, n_envs=4096, device="cuda:0")
env.seed(1)
env.reset()
be = env.backend
be.lib.aie_dev_set_skip_mask.argtypes = [ctypes.c_void_p, ctypes.c_int]
for _ in range(300):
    a, p = be.sample_random_actions(1234)
    be.step(a, p)
torch.cuda.synchronize()
snap = be.arena.clone()
a, p = be.sample_random_actions(1234)
for m in MASKS:
    be.arena.copy_(snap)
    be.lib.aie_dev_set_skip_mask(be.handle, m)
    for _ in range(20):
        be.step(a, p)
    torch.cuda.synchronize()
print("masks", MASKS)
