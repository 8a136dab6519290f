#!/usr/bin/env python
"""Development tool: launch time of the gather-trade-build step kernel's COMPILE-TIME instance with one phase switched
off at a time (aie_dev_set_skip_mask through the -DAIE_DEV build: the traced instances honour the mask), from one
arena snapshot.  Run under `rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU ...` to get the dynamic
instruction counts of the same launches (parse with tools/spec_ablate_report.py).  GPU only.

   python tools/spec_ablate.py [n_agents ...]        (default: 4 10)
"""
import ctypes
import os

os.environ["AIE_DEV_LIB"] = "1"
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
from helpers import make_env  # noqa: E402

SK = 63 | 512  # the skeleton: record in, decode, occupancy map (nothing else, no record store)
MASKS = [0, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 8 | 16, 2 | 4 | 512, 31 | 512, SK,
         SK | 1 << 16, SK | 1 << 17, SK | 1 << 18, SK | 1 << 19, SK | 15 << 16, 0]
if os.environ.get("ABLATE_SKELETON"):
    MASKS = [0, 31 | 512, SK, SK | 1 << 16, SK | 1 << 17, SK | 1 << 18, SK | 1 << 19, SK | 15 << 16, 0]
SK = 63 | 512
NAMES = {0: "full", 1: "-components", 2: "-regen", 4: "-map observations", 8: "-flat vectors -masks", 16: "-rewards",
         31 | 512: "record in, decode, occupancy map, record out only", 63 | 512: "record in, decode, occupancy map only",
         32: "-record store", 64: "-flat stage A", 128: "-flat cda", 256: "-flat tax", 512: "-masks",
         1024: "-planner copy-out", 2048: "-build", 4096: "-cda", 8192: "-gather", 16384: "-tax",
         32768: "full map rewrite instead of in-place", 24: "-flat -rewards (wave 0 tail)",
         2 | 4 | 512: "-regen -map obs -masks (wave 1 tail)",
         SK | 1 << 16: "skeleton - draw window", SK | 1 << 17: "skeleton - occupancy map", SK | 1 << 18: "skeleton - action decode",
         SK | 1 << 19: "skeleton - generator rows to registers", SK | 15 << 16: "skeleton - all four"}
LAUNCHES = 30
E = 4096


def main():
    import json

    agents = [int(x) for x in sys.argv[1:]] or [4, 10]
    times = {}
    for n in agents:
        env = make_env(dict(bench.C2_CFG, n_agents=n), n_envs=E, device="cuda:0")
        env.seed(1)
        env.reset()
        be = env.backend
        be.lib.aie_dev_set_skip_mask.argtypes = [ctypes.c_void_p, ctypes.c_int]
        print("n_agents", n, "step kernel instance", be.lib.aie_step_kernel_instance(be.handle), flush=True)
        for _ in range(300):
            a, p = be.sample_random_actions(1234)
            be.step(a, p)
        torch.cuda.synchronize()
        snap = be.arena.clone()
        a, p = be.sample_random_actions(1234)
        base = None
        for m in MASKS:
            be.arena.copy_(snap)
            be.lib.aie_dev_set_skip_mask(be.handle, m)
            for _ in range(5):
                be.step(a, p)
            be.arena.copy_(snap)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(LAUNCHES):
                be.step(a, p)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1000.0 / LAUNCHES
            if base is None:
                base = us
            print("  mask %6d %-40s %7.2f us per launch (%+6.2f)" % (m, NAMES.get(m, "?"), us, us - base), flush=True)
            times.setdefault(str(n), []).append({"mask": m, "phase_off": NAMES.get(m, "?"), "us_per_launch": us})
        be.lib.aie_dev_set_skip_mask(be.handle, 0)
        del env, be
    if os.environ.get("ABLATE_JSON"):  # launch times per switched-off phase (tools/spec_ablate_report.py adds the counters)
        json.dump(times, open(os.environ["ABLATE_JSON"], "w"), indent=1)


if __name__ == "__main__":
    main()
