#!/usr/bin/env python
"""Experiment: does splitting the 4096 replicas into S independent shards stepped on S
HIP streams (so that one shard's compute phase overlaps another's store phase) help?"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
from helpers import make_env  # noqa: E402

E = 4096
N_AGENTS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
CFG = dict(bench.C2_CFG, n_agents=N_AGENTS)
for S in (1, 2, 4):
    envs, streams = [], []
    for s in range(S):
        env = make_env(CFG, n_envs=(E // S if S != 3 else [1366, 1365, 1365][s]), device="cuda:0", env_offset=s * 1366)
        env.seed(1)
        env.reset()
        envs.append(env)
        streams.append(torch.cuda.Stream())
    torch.cuda.synchronize()
    cur = []
    for env, st in zip(envs, streams):
        with torch.cuda.stream(st):
            cur.append([env.backend.sample_random_actions(1234, env.env_offset, slot=0), 0])

    def step_all():  # one launch per shard and step (aie_step_sample_next), as in bench.py
        for k, (env, st) in enumerate(zip(envs, streams)):
            with torch.cuda.stream(st):
                (a, p), slot = cur[k]
                cur[k] = [env.backend.step_sample_next(a, p, 1234, env.env_offset, next_slot=slot ^ 1), slot ^ 1]

    for _ in range(200):
        step_all()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 1500
    for _ in range(K):
        step_all()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("shards=%d  %.1f us/step  %.1f M agent-steps/s" % (S, dt / K * 1e6, E * N_AGENTS * K / dt / 1e6))
    del envs
