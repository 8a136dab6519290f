"""Secondary measurement (BASELINE configs[3]): CovidAndEconomySimulation, 51 states + planner,
8192 replicas on one MI355X, uniform random policy.  Prints one JSON line shaped like
bench.py's (the driver's bench line stays the C2 workload of bench.py).

    python tools/bench_covid.py [--envs 8192] [--steps 540] [--cpu-seconds 10]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=8192)
    ap.add_argument("--steps", type=int, default=540)
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--dev-skip", type=int, default=0, help="development: phases of the step kernel to skip")
    args = ap.parse_args()

    import numpy as np
    import torch

    from helpers import load_covid_golden
    from test_covid_golden import hip_env, make_oracle

    cfg = load_covid_golden("c4_covid_51ag")["cfg"]  # the shipped run config
    E, n = args.envs, 51
    env = hip_env(cfg, n_envs=E)
    env.reset()
    be = env.backend
    T = env.episode_length
    if args.dev_skip:
        import ctypes

        be.lib.aie_dev_set_skip_mask.argtypes = [ctypes.c_void_p, ctypes.c_int]
        be.lib.aie_dev_set_skip_mask(be.handle, args.dev_skip)

    def episode_steps(k):
        t = int(be.tensors["timestep"][0].item())
        for _ in range(k):
            a, p = be.sample_random_actions(99, 0)
            be.step(a, p)
            t += 1
            if t == T:
                be.reset(be.tensors["done"])
                t = 0

    episode_steps(50)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    episode_steps(args.steps)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0

    # per-launch kernel time (HIP events on the launch stream), mid-episode
    a, p = be.sample_random_actions(99, 0)
    be.reset()
    nk = min(300, T - 1)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nk)]
    torch.cuda.synchronize()
    for s, e in ev:
        s.record()
        be.step(a, p)
        e.record()
    torch.cuda.synchronize()
    durs = sorted(s.elapsed_time(e) for s, e in ev)
    avg_ms = sum(durs) / len(durs)
    m = env.model
    L = int(m["filter_len"])
    obs = sum(t[0].numel() * t.element_size() for k, t in be.tensors.items() if k.startswith("obs_a_")) \
        + (4 + 1 + 20) * 4
    state_rw = 2 * (8 + 1) * n * 4 + n  # float32 rows + cooldown, read+write; today's level byte
    hist = (L + 1) * n                  # each state's 601-day stringency window, one byte a day
    b = dict(history_window=hist, state_rw=state_rw, obs=obs, act=(n + 1) * 4, rew_done=(n + 1) * 4 + 1)
    b["total"] = sum(b.values())
    achieved = b["total"] * E / (avg_ms * 1e-3) / 1e9
    flops = 2.0 * L * int(m["num_filters"]) * n * E  # filter-bank FMAs (float64)
    out = {
        "metric": "agent-steps/sec, covid19_env 51 US-state agents + planner", "value": E * n * args.steps / el,
        "unit": "agent-steps/s", "n_gpus": 1, "steps": args.steps, "ms_per_step": el / args.steps * 1e3,
        "higher_is_better": True, "dtype": "f32 state, f64 filter bank (f32 observations)", "data": "synthetic",
        "config": {"workload": "C4: CovidAndEconomySimulation, run config covid_and_economy_environment.yaml, "
                               "uniform random policy", "envs_per_gpu": E, "n_agents": n},
        "roofline": {"bound": "hbm", "kernel": "aie_covid_step_kernel<5>", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_launch": b["total"] * E, "bytes_breakdown_per_env_step": b,
                     "avg_launch_ms": avg_ms, "median_launch_ms": durs[len(durs) // 2],
                     "f64_filter_tflops": flops / (avg_ms * 1e-3) / 1e12},
    }
    if args.cpu_seconds > 0:
        Ec = 64
        o = make_oracle(cfg, n_envs=Ec)
        o.reset()
        rng = np.random.RandomState(0)
        k, t1 = 0, time.perf_counter()
        while time.perf_counter() - t1 < args.cpu_seconds and k < T:
            o.step(rng.randint(0, 11, size=(Ec, n)), rng.randint(0, 21, size=Ec))
            k += 1
        dt = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": Ec * n * k / dt, "unit": "agent-steps/s", "cores": 1, "kind": "port",
                               "sample": "%d replicas x %d steps, batched NumPy restatement (oracle/covid_oracle.py), %.1f s"
                                         % (Ec, k, dt)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
