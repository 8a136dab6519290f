"""bench.py's roofline accounting reads committed rocprofv3 summaries (profiles/): the lookups must hold for every
workload the default run times, with and without a summary, and the derived fractions must follow from the numbers in
the files.  CPU only (no kernel is launched)."""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


@pytest.mark.parametrize("wl", sorted(bench.WORKLOADS))
def test_counter_lookups_return_their_triples_for_every_workload(wl):
    insts, src, valu = bench.issue_counters(wl)  # (None, None, None) for a workload without a summary, e.g. C2p
    assert (insts is None) == (src is None)
    if insts is not None:
        assert insts > 0 and os.path.exists(os.path.join(ROOT, src))
        assert valu is None or 0 < valu < insts
    traffic, tsrc = bench.measured_traffic(wl, bench.WORKLOADS[wl]["envs"])
    assert (traffic is None) == (tsrc is None)
    assert bench.measured_traffic(wl, bench.WORKLOADS[wl]["envs"] + 1) == (None, None)  # only valid at the profiled size


def test_side_workloads_are_known_workloads():
    for name, steps, warm in bench.SIDE_WORKLOADS:
        assert name in bench.WORKLOADS and steps > 0 and warm > 0


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(ROOT, "profiles", "r03_c*_bench.json"))))
def test_committed_round3_lines_agree_with_their_counter_summaries(path):
    """profiles/r03_<c>_bench.json: issue_frac / valu_frac / hbm_traffic_frac = the committed summaries of the same
    profiling call over the line's own launch time; rocprofv3's average launch duration agrees with the HIP events."""
    line = json.loads(open(path).read().strip().splitlines()[-1])
    r = line["roofline"]
    wl = os.path.basename(path).split("_")[1].upper().replace("C4X", "C4x")
    t = r["avg_launch_ms"] * 1e-3
    insts, _, valu = bench.issue_counters(wl)
    assert r["issue_frac"] == pytest.approx(insts / (bench.N_SIMDS * bench.SM_CLOCK_HZ * t), rel=1e-6)
    assert r["valu_frac"] == pytest.approx(4.0 * valu / (bench.N_SIMDS * bench.SM_CLOCK_HZ * t), rel=1e-6)
    traffic, _ = bench.measured_traffic(wl, line["config"]["envs_per_gpu"])
    assert r["hbm_traffic_frac"] == pytest.approx(traffic / t / 1e9 / bench.HBM_PEAK_GBS, rel=1e-6)
    assert r["frac"] == pytest.approx(r["achieved"] / bench.HBM_PEAK_GBS, rel=1e-9)
    stats = open(path.replace("_bench.json", "_kernel_stats.csv")).read().splitlines()
    step_rows = [row for row in stats[1:] if "step_kernel" in row and "reset" not in row]
    avg_ns = float(step_rows[0].split('",')[1].split(",")[2])
    assert avg_ns * 1e-6 == pytest.approx(r["avg_launch_ms"], rel=0.06)  # HIP events vs rocprofv3, same run
