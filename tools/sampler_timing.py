#!/usr/bin/env python
"""Development: the policy sampler's launch time with parts of the kernel switched off (dev library,
AIE_SAMPLER_DEV_SKIP bits: 1 no entry loads, 2 no draw index, 4 no arithmetic, 8 empty kernel), one process per setting.
   python tools/sampler_timing.py            (GPU only)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    os.environ["AIE_DEV_LIB"] = "1"
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import torch

    import bench
    from helpers import make_env

    E = 4096
    env = make_env(dict(bench.C2_CFG), n_envs=E, device="cuda:0")
    env.seed(1)
    env.reset()
    be = env.backend
    MA, MP = be.tensors["obs_a_action_mask"].shape[-1], be.tensors["obs_p_action_mask"].shape[-1]
    la = torch.randn(E, 4, MA, device="cuda")
    lp = torch.randn(E, MP, device="cuda")
    big = torch.zeros(64 << 20, device="cuda")  # 256 MB: written between launches to push the inputs out of the caches
    flush = "flush" in sys.argv
    for _ in range(20):
        be.sample_policy_actions(la, lp, seed=5, env_offset=0)
    torch.cuda.synchronize()
    N = 100
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(N)]
    for a, b in evs:
        if flush:
            big.add_(1.0)
        a.record()
        be.sample_policy_actions(la, lp, seed=5, env_offset=0)
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    print("skip %s wpr_log2 %s %s: median %.1f us, p10 %.1f, p90 %.1f (event to event, one launch)" % (
        os.environ.get("AIE_SAMPLER_DEV_SKIP", "0"), os.environ.get("AIE_SAMPLER_WAVES_LOG2", "1"), "flushed" if flush else "warm",
        ts[N // 2], ts[N // 10], ts[9 * N // 10]))
else:
    import csv
    import glob
    import shutil

    out = "/tmp/sampler_timing_prof"
    for mode in ([], ["flush"]):
        for wl in os.environ.get("WAVES_LOG2", "1").split(","):
            for sk in os.environ.get("SKIPS", "0,1,2,3,4,5,7,8").split(","):
                env = dict(os.environ, AIE_SAMPLER_DEV_SKIP=sk, AIE_SAMPLER_WAVES_LOG2=wl, TMPDIR="/tmp")
                shutil.rmtree(out, ignore_errors=True)
                subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", out, "-o", "s", "--",
                                sys.executable, os.path.abspath(__file__), "child"] + mode, env=env, timeout=300, cwd="/tmp",
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                for f in glob.glob(out + "/**/*kernel_stats.csv", recursive=True):
                    for r in csv.DictReader(open(f)):
                        if "sample_policy" in r["Name"]:
                            print("skip %s waves_log2 %s %s: kernel average %.2f us (min %.2f) over %s launches" % (
                                sk, wl, "flushed" if mode else "warm", float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, r["Calls"]), flush=True)
