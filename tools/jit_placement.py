#!/usr/bin/env python
"""Development: does the launch time of the run-time instance follow where its code lies?  BASELINE configs[1]'s family
compiled at run time (AIE_JIT_FORCE) with N no-ops ahead of the kernels (AIE_JIT_PAD_NOPS), one fresh environment per N,
next to the build's instance.  GPU only.   python tools/jit_placement.py [N ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ["AIE_JIT_AUTO"] = "0"
import torch  # noqa: E402

import bench  # noqa: E402
from helpers import make_env  # noqa: E402


def per_launch(be, n=400):
    cur = [be.sample_random_actions(5, 0, slot=0), 0]

    def step():
        cur[0] = be.step_sample_next(cur[0][0], cur[0][1], 5, 0, next_slot=cur[1] ^ 1)
        cur[1] ^= 1

    for _ in range(150):
        step()
    best = 1e9
    for _ in range(3):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(n):
            step()
        ev1.record()
        torch.cuda.synchronize()
        best = min(best, ev0.elapsed_time(ev1) / n * 1e3)
    return best


pads = [int(x) for x in sys.argv[1:]] or [0, 16, 48, 64, 128, 256, 512, 1024, 2048, 4096, 8192]
env = make_env(dict(bench.C2_CFG), n_envs=4096, device="cuda:0")
env.seed(1)
env.reset()
print("build's instance %d: %.2f us per launch" % (env.backend.lib.aie_step_kernel_instance(env.backend.handle), per_launch(env.backend)), flush=True)
del env
for n in pads:
    os.environ["AIE_JIT_FORCE"] = "1"
    if n:
        os.environ["AIE_JIT_PAD_NOPS"] = str(n)
    else:
        os.environ.pop("AIE_JIT_PAD_NOPS", None)
    env = make_env(dict(bench.C2_CFG), n_envs=4096, device="cuda:0")
    env.seed(1)
    env.reset()
    be = env.backend
    ok = env.specialize()
    inst = be.lib.aie_step_kernel_instance(be.handle)
    print("run-time instance, %5d no-ops (%6d B) ahead: %s" % (n, 4 * n, "%.2f us per launch" % per_launch(be) if ok and inst == 1000 else "unavailable (%d)" % inst), flush=True)
    del env, be
