#!/usr/bin/env python
"""Development tool: per-workgroup clock stamps of aie_step_kernel (start, dynamics done, end)
to see launch stagger / tail effects.  GPU only."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from helpers import make_env  # noqa: E402

E = 4096
env = make_env(dict(bench.WORKLOAD), n_envs=E, device="cuda:0")
env.seed(1)
env.reset()
be = env.backend
for _ in range(300):
    a, p = be.sample_random_actions(1234)
    be.step(a, p)
buf = torch.zeros(8 * E, dtype=torch.int64, device="cuda")
be.lib.aie_dev_set_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
be.lib.aie_dev_set_trace(be.handle, ctypes.c_void_p(buf.data_ptr()))
names = ["start", "loaded", "build", "cda", "gather", "tax", "regen", "end"]
for rep in range(2):
    a, p = be.sample_random_actions(1234)
    torch.cuda.synchronize()
    be.step(a, p)
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(E, 8).astype(np.float64)
    t = (t - t[:, 0].min()) / 100.0  # wall_clock64 ticks at 100 MHz -> us
    q = lambda x: " ".join("%6.1f" % v for v in np.percentile(x, [0, 10, 50, 90, 99, 100]))  # noqa: E731
    print("absolute (us)             p0    p10    p50    p90    p99   p100")
    for k, nm in enumerate(names):
        print("  %-10s            %s" % (nm, q(t[:, k])))
    print("phase durations")
    for k in range(1, 8):
        print("  %-10s            %s" % (names[k], q(t[:, k] - t[:, k - 1])))
    slow = np.argsort(t[:, 6])[-200:]
    print("slowest 200 blocks, mean phase durations:", " ".join("%s=%.1f" % (names[k], (t[slow, k] - t[slow, k - 1]).mean()) for k in range(1, 8)))
    rec = be.tensors
    nb = rec["cda_n_bids"].cpu().numpy().sum(axis=1) + rec["cda_n_asks"].cpu().numpy().sum(axis=1)
    e_of_block = np.arange(E)
    e_of_block = (e_of_block & 7) * (E >> 3) + (e_of_block >> 3)
    print("orders in book: all %.1f, slowest blocks %.1f" % (nb.mean(), nb[e_of_block[slow]].mean()))
