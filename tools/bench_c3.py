"""Secondary measurement (BASELINE configs[2], one GPU's share): gather-trade-build 25x25, 10 agents +
ContinuousDoubleAuction + PeriodicBracketTax, 4096 replicas (32768 over 8 GPUs = 4096 each), uniform
random policy.  Same timed loop as bench.py with the C3 workload; prints one JSON line.

    python tools/bench_c3.py [--steps 1000]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

bench.WORKLOAD = dict(bench.WORKLOAD, n_agents=10)
bench.WORKLOAD_NAME = "C3 (one GPU's share of 32768 replicas)"

if __name__ == "__main__":
    if "--no-cpu-baseline" not in sys.argv:
        sys.argv.append("--no-cpu-baseline")
    if "--steps" not in sys.argv:
        sys.argv += ["--steps", "1000"]
    bench.main()
