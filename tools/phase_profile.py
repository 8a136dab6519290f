#!/usr/bin/env python
"""Development tool: times aie_step_kernel with individual phases skipped
(aie_dev_set_skip_mask) to see where a launch spends its time.  GPU only."""
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

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = dict(bench.C2_CFG)
if len(sys.argv) > 2:
    cfg["n_agents"] = int(sys.argv[2])
env = make_env(cfg, n_envs=E, device="cuda:0")
env.seed(1)
env.reset()
be = env.backend
be.lib.aie_dev_set_skip_mask.argtypes = [ctypes.c_void_p, ctypes.c_int]
for _ in range(300):
    a, p = be.sample_random_actions(1234)
    be.step(a, p)
torch.cuda.synchronize()
snap = be.arena.clone()


def timeit(mask, n=200):
    be.arena.copy_(snap)
    be.lib.aie_dev_set_skip_mask(be.handle, mask)
    a, p = be.sample_random_actions(1234)
    for _ in range(20):
        be.step(a, p)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(n):
        be.step(a, p)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


names = {0: "full", 1: "-serial components", 2: "-regen", 4: "-spatial obs", 8: "-flat obs+masks",
         16: "-rewards", 32: "-record store", 63: "only load+decode+locmap+decay",
         62: "only serial", 61: "only regen", 59: "only spatial", 55: "only flat", 47: "only rewards"}
names.update({64: "-flat stageA", 128: "-flat cda fill", 256: "-flat tax fill", 512: "-flat masks",
              1024: "-flat copy-out", 2048: "-build", 4096: "-cda", 8192: "-gather", 16384: "-tax"})
full = timeit(0)
for m, nm in names.items():
    t = timeit(m)
    print("%-34s %8.1f us   (delta vs full %+7.1f)" % (nm, t, t - full))
