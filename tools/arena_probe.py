#!/usr/bin/env python
"""Development probe: the one-step-economy kernel runs at two speeds (1.45 vs 1.67 ms per launch) with the same binary
ON THE SAME BOX IN THE SAME PROCESS, depending on where its 7 GB arena landed: every environment built here gets a
fresh arena from torch's allocator; the launch time is printed next to the arena's address.  (Measured: ~half of the
placements are slow; all of them are 2 MiB aligned; hipExtMallocWithFlags(hipDeviceMallocContiguous) gives both
speeds as well; the box's fill roof is the same for both.)  GPU only.   python tools/arena_probe.py [count]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
from helpers import make_env  # noqa: E402

WL = os.environ.get("PROBE_WL", "C5")
keep = []  # (holding the previous arenas makes the allocator hand out new addresses)


def run(launches=40, warm=15):
    W = bench.WORKLOADS[WL]
    env = make_env(W["cfg"](), n_envs=W["envs"], device="cuda:0")
    env.seed(1)
    env.reset()
    be = env.backend
    if WL == "C5":
        be.lib.aie_set_auto_reset(be.handle, 1)
    for _ in range(warm):
        a, p = be.sample_random_actions(1234)
        be.step(a, p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a, p = be.sample_random_actions(1234)
    e0.record()
    for _ in range(launches):
        be.step(a, p)
    e1.record()
    torch.cuda.synchronize()
    print("%s arena at %#x  %.4f ms per launch" % (WL, be.arena.data_ptr(), e0.elapsed_time(e1) / launches), flush=True)
    keep.append(env)


for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    run()
