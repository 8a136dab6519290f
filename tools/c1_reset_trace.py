#!/usr/bin/env python
"""Development tool: where a replica's reset-time layout generation (BASELINE configs[0]'s scenario, four wavefronts per
replica) spends its time: per-replica phase clocks through the -DAIE_DEV build.  GPU only."""
import ctypes
import os

os.environ["AIE_DEV_LIB"] = "1"
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from helpers import make_env  # noqa: E402

E = 4096
env = make_env(dict(bench.C1_CFG), n_envs=E, device="cuda:0")
env.seed(1)
env.reset()
be = env.backend
buf = torch.zeros(12 * E, dtype=torch.int64, device="cuda")
be.lib.aie_dev_set_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
be.lib.aie_dev_set_trace(be.handle, ctypes.c_void_p(buf.data_ptr()))
part = (torch.arange(E, device="cuda") % 50 == 0).to(torch.uint8)
for rep in range(3):
    buf.zero_()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    be.reset(part)
    ev1.record()
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(E, 12).astype(np.float64)
    t = t[t[:, 0] > 0]
    q = lambda x: " ".join("%7.1f" % v for v in np.percentile(x, [0, 50, 90, 100]))  # noqa: E731
    print("masked reset of %d replicas: %.1f us" % (len(t), ev0.elapsed_time(ev1) * 1e3))
    print("                              min     p50     p90     max")
    for k, nm in ((0, "layout_generate (us)"), (1, "  rand planes"), (2, "  threshold search"), (3, "  gauss requests"),
                  (4, "  convolution+count")):
        print("  %-22s %s" % (nm, q(t[:, k] / 100.0)))
    for k, nm in ((5, "tries"), (6, "growth passes"), (7, "threshold blocks")):
        print("  %-22s %s" % (nm, q(t[:, k])))
    tot = t[:, 0].sum()
    print("  share of the total: rand %.2f threshold %.2f gauss %.2f conv %.2f" % tuple(t[:, k].sum() / tot for k in (1, 2, 3, 4)))
    print("  per growth pass: gauss %.1f us, conv %.1f us; per threshold block %.1f us" % (
        t[:, 3].sum() / t[:, 6].sum() / 100, t[:, 4].sum() / t[:, 6].sum() / 100, t[:, 2].sum() / max(1, t[:, 7].sum()) / 100))
