"""Secondary measurement (BASELINE configs[4]): one-step-economy, 100 agents + SimpleLabor + planner
tax, 65536 replicas on one MI355X, uniform random policy.  Prints one JSON line.

    python tools/bench_ose.py [--envs 65536] [--steps 200]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=200)
    args = ap.parse_args()
    import numpy as np
    import torch

    from helpers import make_env

    n = 100
    rs = np.random.RandomState(4)
    cfg = dict(scenario_name="one-step-economy", world_size=[1, 1], n_agents=n, episode_length=2,
               components=[["SimpleLabor", {"skills": [float(x) for x in np.sort(1 + rs.rand(n) * 2)]}],
                           ["PeriodicBracketTax", {"bracket_spacing": "us-federal", "period": 1,
                                                   "tax_model": "model_wrapper"}]])
    env = make_env(cfg, n_envs=args.envs, device="cuda:0")
    env.seed(1)
    env.reset()
    be = env.backend

    def run(k):
        for i in range(k):
            a, p = be.sample_random_actions(7)
            be.step(a, p)
            if i % 2 == 1:
                be.reset(be.tensors["done"])

    run(20)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    run(args.steps)
    ev1.record()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    obs = sum(t[0].numel() * t.element_size() for k, t in be.tensors.items() if k.startswith("obs_"))
    rec = be.descs["inv_coin"][2][0]
    b = dict(obs=obs, state_rw=2 * rec, act=(n + 7) * 4, rew_done=(n + 1) * 4 + 1)
    b["total"] = sum(b.values())
    print(json.dumps({
        "metric": "agent-steps/sec, one_step_economy 100 agents + SimpleLabor + planner tax",
        "value": args.envs * n * args.steps / el, "unit": "agent-steps/s", "n_gpus": 1, "steps": args.steps,
        "ms_per_step": el / args.steps * 1e3, "note": "every second step is followed by the episode reset launch",
        "config": {"workload": "C5: one-step-economy, episode_length 2, uniform random policy", "envs_per_gpu": args.envs,
                   "n_agents": n},
        "roofline": {"bound": "hbm", "algorithmic_bytes_per_env_step": b,
                     "achieved_GBps_incl_resets": b["total"] * args.envs * args.steps / (ev0.elapsed_time(ev1) * 1e-3) / 1e9,
                     "peak": 8000.0}}))


if __name__ == "__main__":
    main()
