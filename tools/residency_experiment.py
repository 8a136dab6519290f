#!/usr/bin/env python
"""Experiment: step time against workgroups per CU (extra dynamic LDS lowers the residency): is one full round plus
a tail better or worse than balanced rounds?  usage: residency_experiment.py N_AGENTS PAD_BYTES..."""
import os

os.environ["AIE_DEV_LIB"] = "1"  # the aie_dev_* hooks live in libaie_hip_dev.so (-DAIE_DEV) only
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
from helpers import make_env  # noqa: E402

E = 4096
N_AGENTS = int(sys.argv[1])
for pad in [int(x) for x in sys.argv[2:]]:
    env = make_env(dict(bench.C2_CFG, n_agents=N_AGENTS), n_envs=E, device="cuda:0")
    env.seed(1)
    env.reset()
    be = env.backend
    be.lib.aie_dev_set_lds_pad(be.handle, pad)
    cur, slot = be.sample_random_actions(1234, 0, slot=0), 0
    for _ in range(300):
        cur, slot = be.step_sample_next(cur[0], cur[1], 1234, 0, next_slot=slot ^ 1), slot ^ 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 1500
    for _ in range(K):
        cur, slot = be.step_sample_next(cur[0], cur[1], 1234, 0, next_slot=slot ^ 1), slot ^ 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("pad=%5d  %.1f us/step  %.1f M agent-steps/s" % (pad, dt / K * 1e6, E * N_AGENTS * K / dt / 1e6))
    del env, be
