#!/usr/bin/env python
"""Test / measurement infrastructure: one process = one reference environment (the UNMODIFIED reference
`env.step`, base_env.py:929-1032, imported through oracle/ref_harness.py), pinned to one core, stepped with
uniform random actions for a fixed wall-clock window.  bench.py's `cpu_baseline` leg (kind "reference") starts P
of these concurrently and adds up their rates.

    python oracle/ref_worker.py --cfg-json '{...}' --core 3 --start <unix time> --seconds 10

Prints one JSON line: {"steps": K, "elapsed": seconds, "n_agents": n, "late": bool, "resets": R} (R = env.reset()
calls inside the timed window: the window is free-running, an episode end costs what the reference's reset costs).
"""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg-json", required=True)
    ap.add_argument("--core", type=int, default=-1)
    ap.add_argument("--start", type=float, default=0.0)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    if args.core >= 0:
        try:
            os.sched_setaffinity(0, {args.core})
        except OSError:
            pass
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[k] = "1"
    import numpy as np

    from ref_harness import load_reference_foundation

    foundation = load_reference_foundation()
    kw = json.loads(args.cfg_json)
    scenario = kw.pop("scenario_name")
    kw["components"] = [tuple(c) for c in kw["components"]]
    env = foundation.make_env_instance(scenario, **kw)
    env.seed(args.seed)
    env.reset()
    n = env.n_agents
    A = env.world.agents[0].action_spaces
    pl = env.world.planner
    p_dims = pl.action_spaces if pl.multi_action_mode else None
    rng = np.random.RandomState(1234 + args.seed)

    resets = [0]

    def one_step():
        acts = {str(i): int(a) for i, a in enumerate(rng.randint(0, A, size=n))}
        if p_dims is not None and len(np.atleast_1d(p_dims)):
            acts["p"] = [int(rng.randint(0, d)) for d in np.atleast_1d(p_dims)]
        _, _, done, _ = env.step(acts)
        if done["__all__"]:
            env.reset()
            resets[0] += 1

    for _ in range(20):
        one_step()
    late = time.time() > args.start
    while time.time() < args.start:
        time.sleep(0.001)
    t0 = time.time()
    resets[0] = 0
    end = max(t0, args.start) + args.seconds
    steps = 0
    while time.time() < end:
        for _ in range(10):
            one_step()
        steps += 10
    print(json.dumps({"steps": steps, "elapsed": time.time() - t0, "n_agents": n, "late": bool(late),
                      "resets": resets[0]}))


if __name__ == "__main__":
    main()
