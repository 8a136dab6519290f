#!/usr/bin/env python
"""Development tool: per-workgroup clock stamps of the one-step-economy step kernel's phases (BASELINE configs[4])
through the -DAIE_DEV build (OSE_STAMP in csrc/aie_kernels_ose.hip).  GPU only.
  python tools/ose_trace.py [n_envs] [auto|noauto]"""
import ctypes
import os

os.environ["AIE_DEV_LIB"] = "1"  # the aie_dev_* hooks live in libaie_hip_dev.so (-DAIE_DEV) only
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from helpers import make_env  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
auto = not (len(sys.argv) > 2 and sys.argv[2] == "noauto")
env = make_env(bench._c5_cfg(), n_envs=E, device="cuda:0")
env.seed(1)
env.reset()
be = env.backend
be.set_auto_reset(auto)
print("replicas", E, "auto-reset", auto, "step kernel instance", be.lib.aie_step_kernel_instance(be.handle))
cur = be.sample_random_actions(1234, 0, slot=0)
slot = 0
for _ in range(10):
    cur = be.step_sample_next(cur[0], cur[1], 1234, 0, next_slot=slot ^ 1)
    slot ^= 1
buf = torch.zeros(12 * E, dtype=torch.int64, device="cuda")
be.lib.aie_dev_set_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
be.lib.aie_dev_set_trace(be.handle, ctypes.c_void_p(buf.data_ptr()))
names = ["start", "loaded+parsed", "labor(perm)", "tax", "obs:sort+tmpl", "obs:gini", "obs:flat rows", "obs:masks", "obs done",
         "metrics", "rewards", "end"]
for rep in range(3):
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    cur = be.step_sample_next(cur[0], cur[1], 1234, 0, next_slot=slot ^ 1)
    slot ^= 1
    ev1.record()
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(E, 12).astype(np.float64)
    ok = t[:, 4] > 0  # replicas that restarted in this launch write their observations inside the reset body
    t = (t - t[:, 0].min()) / 100.0  # wall_clock64 ticks at 100 MHz -> us
    q = lambda x: " ".join("%7.1f" % v for v in np.percentile(x, [0, 10, 50, 90, 99, 100]))  # noqa: E731
    print("launch %d: %.3f ms; workgroups with step observations: %d" % (rep, ev0.elapsed_time(ev1), ok.sum()))
    print("  start (us)              %s" % q(t[:, 0]))
    print("  end (us)                %s" % q(t[:, 11]))
    print("  life (us)               %s" % q(t[:, 11] - t[:, 0]))
    print("phase durations (us), in the order the stamps were taken      p0     p10     p50     p90     p99    p100")
    order = np.argsort(np.median(t, axis=0), kind="stable")
    for prev, k in zip(order[:-1], order[1:]):
        print("  %-16s (after %-16s)    %s" % (names[k], names[prev], q(t[:, k] - t[:, prev])))
    buf.zero_()
