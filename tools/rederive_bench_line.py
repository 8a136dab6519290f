#!/usr/bin/env python
"""The bench line of a profiling call is printed before that call's counter summaries exist, so its derived roofline
fields (traffic, hbm_traffic_frac, issue_frac, valu_frac, bound) were computed from the PREVIOUS round's summaries.
This recomputes them from the summaries of the same call -- pure arithmetic on the line's own launch time, exactly
bench.py's formulas (tests/test_bench_accounting.py holds the committed lines to them):

   python tools/rederive_bench_line.py profiles/r04_c2_bench.json [...]     (the summaries must already be in profiles/)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

for path in sys.argv[1:]:
    line = json.loads(open(path).read().strip().splitlines()[-1])
    wl, rnd = bench.workload_of_profile(path)
    r = line["roofline"]
    t = r["avg_launch_ms"] * 1e-3
    traffic, tsrc = bench.measured_traffic(wl, line["config"]["envs_per_gpu"], max_round=rnd)
    insts, isrc, valu = bench.issue_counters(wl, max_round=rnd)
    r["traffic"] = traffic
    r["hbm_traffic_frac"] = traffic / t / 1e9 / bench.HBM_PEAK_GBS if traffic else None
    r["issue_frac"] = insts / (bench.N_SIMDS * bench.SM_CLOCK_HZ * t) if insts else None
    r["valu_frac"] = 4.0 * valu / (bench.N_SIMDS * bench.SM_CLOCK_HZ * t) if valu else None
    for k, v in (("traffic_source", tsrc), ("issue_source", isrc), ("wave_instructions_per_launch", insts),
                 ("valu_instructions_per_launch", valu)):
        if k in r:
            r[k] = v
    if "bound" in r:
        store = r.get("traffic_frac_of_store_roof")
        if r.get("store_roof_GBps_this_box") and traffic:
            store = r["traffic_frac_of_store_roof"] = (traffic / t / 1e9) / r["store_roof_GBps_this_box"]
        r["bound"] = ("hbm" if (store or 0) >= 0.7 or (r["hbm_traffic_frac"] or 0) >= 0.7 else
                      "valu" if (r["valu_frac"] or 0) >= 0.5 else "issue" if (r["issue_frac"] or 0) >= 0.6 else "latency")
    line["rederived"] = "roofline.traffic / *_frac / bound recomputed from the counter summaries of the same profiling call (tools/rederive_bench_line.py)"
    open(path, "w").write(json.dumps(line) + "\n")
    print(path, {k: r.get(k) for k in ("avg_launch_ms", "frac", "hbm_traffic_frac", "issue_frac", "valu_frac", "bound")})
