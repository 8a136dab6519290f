#!/usr/bin/env python
"""Development tool: per-workgroup clock stamps of aie_step_kernel (start, dynamics done, end)
to see launch stagger / tail effects.  GPU only."""
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

E = 4096
N_AGENTS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = dict(bench.C2_CFG, n_agents=N_AGENTS)
if os.environ.get("TRACE_LAYOUT_FILE"):  # e.g. uniform_25x25_25each_65clump.txt
    cfg["env_layout_file"] = os.environ["TRACE_LAYOUT_FILE"]
if os.environ.get("TRACE_RNG_MODE"):  # "fast": the counter-based stream (C2f / C3f)
    cfg["rng_mode"] = os.environ["TRACE_RNG_MODE"]
env = make_env(cfg, n_envs=E, device="cuda:0")
env.seed(1)
env.reset()
be = env.backend
if len(sys.argv) > 2 and sys.argv[2] == "generic":
    be.lib.aie_select_step_kernel(be.handle, 1)
print("n_agents", N_AGENTS, "step kernel instance", be.lib.aie_step_kernel_instance(be.handle))
lds = (ctypes.c_int64 * 6)()
be.lib.aie_dev_lds_bytes.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
be.lib.aie_dev_lds_bytes(be.handle, lds)
print("LDS bytes per workgroup %d: record %d, location map %d, f64 scratch %d, staging %d -> %d workgroups per CU" % tuple(lds))
for _ in range(300):
    a, p = be.sample_random_actions(1234)
    be.step(a, p)
buf = torch.zeros(12 * E, dtype=torch.int64, device="cuda")
be.lib.aie_dev_set_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
be.lib.aie_dev_set_trace(be.handle, ctypes.c_void_p(buf.data_ptr()))
names = ["start", "loaded", "build", "cda", "gather", "tax", "regen", "end", "rec_in_lds", "srcn_zeroed", "flat_done(w0)", "spatial+masks(w1)"]
for rep in range(2):
    a, p = be.sample_random_actions(1234)
    torch.cuda.synchronize()
    tr0 = be.tensors["metrics_cda_trades"].cpu().numpy()[:, 0, :, :, 0].sum(axis=(1, 2))
    no0 = be.tensors["cda_n_orders"].cpu().numpy().sum(axis=(1, 2))
    for _ in range(20 if "b2b" in sys.argv else 1):  # b2b: stamps of the last of 20 back-to-back launches
        be.step(a, p)
    torch.cuda.synchronize()
    tr1 = be.tensors["metrics_cda_trades"].cpu().numpy()[:, 0, :, :, 0].sum(axis=(1, 2))
    no1 = be.tensors["cda_n_orders"].cpu().numpy().sum(axis=(1, 2))
    acts = a.cpu().numpy().reshape(E, -1)
    n_order_acts = ((acts >= 2) & (acts <= 45)).sum(axis=1)
    t = buf.cpu().numpy().reshape(E, 12).astype(np.float64)
    t = (t - t[:, 0].min()) / 100.0  # wall_clock64 ticks at 100 MHz -> us
    q = lambda x: " ".join("%6.1f" % v for v in np.percentile(x, [0, 10, 50, 90, 99, 100]))  # noqa: E731
    print("absolute (us)             p0    p10    p50    p90    p99   p100")
    for k, nm in enumerate(names):
        print("  %-10s            %s" % (nm, q(t[:, k])))
    print("  pre-dynamics: srcn zero+barrier %.2f | record load+decode+barrier %.2f | locmap/agents_load/decay+barrier %.2f (medians)" % (
        np.median(t[:, 9] - t[:, 0]), np.median(t[:, 8] - t[:, 9]), np.median(t[:, 1] - t[:, 8])))
    print("  post-dynamics: flat %.2f | rewards+done+wait %.2f | wave1 spatial+masks %.2f (medians, from regen end)" % (
        np.median(t[:, 10] - t[:, 6]), np.median(t[:, 7] - t[:, 10]), np.median(t[:, 11] - t[:, 6])))
    h, _ = np.histogram(t[:, 0], bins=np.arange(0, t[:, 0].max() + 1.0, 0.5))
    print("  start histogram (0.5 us bins):", " ".join(str(v) for v in h))
    print("  mean start by blockIdx/256:", " ".join("%.1f" % t[k * 256:(k + 1) * 256, 0].mean() for k in range(E // 256)))
    print("  mean start by blockIdx%8 (XCD):", " ".join("%.1f" % t[k::8, 0].mean() for k in range(8)))
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
    cda_t = t[:, 3] - t[:, 2]
    trades = (tr1 - tr0)[e_of_block]
    removed = (no0 + n_order_acts - no1)[e_of_block] - 2 * trades  # expiries (upper bound: refused orders count too)
    for nm, x in (("trades", trades), ("order actions", n_order_acts[e_of_block]), ("expired/refused", removed)):
        print("  cda time by %-16s" % nm, " ".join("%d:%.1f(%d)" % (k, cda_t[x == k].mean(), (x == k).sum()) for k in sorted(set(x.tolist()))[:9]))
    print("  corr(cda time, start time) = %.2f" % np.corrcoef(cda_t, t[:, 0])[0, 1])
