#!/usr/bin/env python
"""Timing of BASELINE configs[0]'s scenario (uniform 15x15: every reset draws a new source layout on the device):
step launch, full reset, masked reset of 1/50 of the replicas.  GPU only."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from helpers import make_env  # noqa: E402

E = 4096
cfg = dict(scenario_name="uniform/simple_wood_and_stone", n_agents=4, world_size=[15, 15], episode_length=1000,
           components=[["Build", {}], ["Gather", {}]], starting_agent_coin=10, starting_stone_coverage=0.10,
           starting_wood_coverage=0.10)
env = make_env(cfg, n_envs=E, device="cuda:0")
env.seed(1)
env.reset()
be = env.backend


def timed(fn, reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


cur = [be.sample_random_actions(5, 0, slot=0), 0]


def step():
    cur[0] = be.step_sample_next(cur[0][0], cur[0][1], 5, 0, next_slot=cur[1] ^ 1)
    cur[1] ^= 1


timed(step, 100)
print("step launch        %8.1f us" % timed(step, 400))
full = torch.ones(E, dtype=torch.uint8, device="cuda")
part = (torch.arange(E, device="cuda") % 50 == 0).to(torch.uint8)
one = torch.zeros(E, dtype=torch.uint8, device="cuda"); one[7] = 1
print("reset, all replicas %8.1f us" % timed(lambda: be.reset(full), 5))
print("reset, 1 in 50      %8.1f us" % timed(lambda: be.reset(part), 10))
print("reset, 1 replica    %8.1f us" % timed(lambda: be.reset(one), 10))
