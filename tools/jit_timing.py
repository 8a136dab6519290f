#!/usr/bin/env python
"""Development tool: step-launch time of configurations WITHOUT a compile-time instance on the generic kernel and on
run-time specialised kernels (aie_specialize), next to BASELINE configs[1]'s compile-time instance.  GPU only."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

HOST_US = 0.0

import bench  # noqa: E402
from helpers import make_env  # noqa: E402

CASES = {
    "C2 (compile-time instance)": dict(bench.C2_CFG),
    "phase-2: planner_gets_spatial_info=False": dict(bench.C2_CFG, planner_gets_spatial_info=False),
    "uniform 65-clump layout file": dict(bench.C2_CFG, env_layout_file="uniform_25x25_25each_65clump.txt"),
    "C3-like, 8 agents": dict(bench.C2_CFG, n_agents=8),
}


def per_launch(be, n=300):
    cur = [be.sample_random_actions(5, 0, slot=0), 0]

    def step():
        cur[0] = be.step_sample_next(cur[0][0], cur[0][1], 5, 0, next_slot=cur[1] ^ 1)
        cur[1] ^= 1

    for _ in range(100):
        step()
    best = 1e9
    global HOST_US
    HOST_US = 1e9
    for _ in range(4):  # (the first block after another environment was torn down can be several times slower)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        ev0.record()
        h0 = time.perf_counter()
        for _ in range(n):
            step()
        HOST_US = min(HOST_US, (time.perf_counter() - h0) / n * 1e6)
        ev1.record()
        torch.cuda.synchronize()
        best = min(best, ev0.elapsed_time(ev1) / n * 1e3)
    return best


for name, cfg in CASES.items():
    env = make_env(cfg, n_envs=4096, device="cuda:0")
    env.seed(1)
    env.reset()
    be = env.backend
    inst = be.lib.aie_step_kernel_instance(be.handle)
    t_first = per_launch(be)
    line = "%-44s instance %5d: %6.2f us per launch (host issue %.1f us)" % (name, inst, t_first, HOST_US)
    if inst < 0:
        t0 = time.perf_counter()
        ok = env.specialize()
        dt = time.perf_counter() - t0
        if ok:
            line += " | specialised (%.1f s): %6.2f us" % (dt, per_launch(be))
        else:
            line += " | specialisation unavailable: %s" % be.lib.aie_last_error(be.handle).decode()
    else:
        be.lib.aie_select_step_kernel(be.handle, 1)
        line += " | generic kernel: %6.2f us" % per_launch(be)
        os.environ["AIE_JIT_FORCE"] = "1"  # the same configuration compiled at run time: A/B against the build's instance
        if env.specialize():
            be.lib.aie_select_step_kernel(be.handle, 0)  # (release the pin on the generic kernel)
            assert be.lib.aie_step_kernel_instance(be.handle) == 1000
            line += " | run-time instance: %6.2f us (host issue %.1f us)" % (per_launch(be), HOST_US)
        del os.environ["AIE_JIT_FORCE"]
    print(line, flush=True)
    del env, be
