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
    ap.add_argument("--episodes", type=int, default=0,
                    help="SURVEY 8(d)'s protocol instead of the free-running window: one warm-up episode from a fresh reset, "
                         "then this many timed WHOLE episodes including their resets (--seconds is ignored)")
    ap.add_argument("--policy", choices=["unmasked", "masked"], default="unmasked",
                    help="unmasked: i.i.d. uniform over the full action range (SURVEY 8(d)); masked: uniform over the entries "
                         "the current `action_mask` observation allows (what bench.py's mask-respecting GPU legs draw)")
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
    obs = [env.reset()]
    n = env.n_agents
    A = env.world.agents[0].action_spaces
    pl = env.world.planner
    # the planner always acts: one index per subspace in multi-action mode, one flat index otherwise (round 4 left the
    # single-action planner -- COVID's -- without an action)
    p_dims = np.atleast_1d(pl.action_spaces) if pl.multi_action_mode else None
    p_flat = None if pl.multi_action_mode else int(pl.action_spaces)
    rng = np.random.RandomState(1234 + args.seed)
    masked = args.policy == "masked"

    resets = [0]

    def pick(mask):
        ok = np.flatnonzero(np.asarray(mask).ravel() > 0.5)
        return int(ok[rng.randint(len(ok))]) if len(ok) else 0

    def agent_mask(ob, i):
        if "a" in ob:  # collated observations (COVID run config): [entries, n]
            return np.asarray(ob["a"]["action_mask"])[:, i]
        return ob[str(i)]["action_mask"]

    def one_step():
        ob = obs[0]
        if masked:
            acts = {str(i): pick(agent_mask(ob, i)) for i in range(n)}
        else:
            acts = {str(i): int(a) for i, a in enumerate(rng.randint(0, A, size=n))}
        if p_dims is not None and len(p_dims):
            if masked:
                m = np.asarray(ob["p"]["action_mask"]).ravel()
                acts["p"], lo = [], 0
                for d in p_dims:
                    acts["p"].append(pick(m[lo: lo + 1 + int(d)]))
                    lo += 1 + int(d)
            else:
                acts["p"] = [int(rng.randint(0, d)) for d in p_dims]
        elif p_flat:
            acts["p"] = pick(ob["p"]["action_mask"]) if masked else int(rng.randint(0, p_flat))
        ob2, _, done, _ = env.step(acts)
        obs[0] = ob2
        if done["__all__"]:
            obs[0] = env.reset()
            resets[0] += 1

    if args.episodes > 0:
        # 1 warm-up episode + N timed whole episodes incl. resets; every worker starts on the common clock
        obs[0] = env.reset()
        resets[0] = 0
        while resets[0] < 1:
            one_step()
        late = time.time() > args.start
        while time.time() < args.start:
            time.sleep(0.001)
        t0 = time.time()
        resets[0] = 0
        steps = 0
        while resets[0] < args.episodes:
            one_step()
            steps += 1
        print(json.dumps({"steps": steps, "elapsed": time.time() - t0, "n_agents": n, "late": bool(late),
                          "resets": resets[0], "policy": args.policy, "episodes": args.episodes}))
        return
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
                      "resets": resets[0], "policy": args.policy}))


if __name__ == "__main__":
    main()
