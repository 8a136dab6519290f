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
    insts, src, valu = bench.issue_counters(wl)  # (None, None, None) for a workload without a summary, e.g. P2
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


@pytest.mark.parametrize("path", sorted(p for p in glob.glob(os.path.join(ROOT, "profiles", "r04_c*_bench.json"))
                                        + glob.glob(os.path.join(ROOT, "profiles", "r05_c*_bench.json"))
                                        if bench.workload_of_profile(p)[0] is not None))  # (not c2pi: its own shape)
def test_committed_round4_lines_agree_with_their_counter_summaries(path):
    """profiles/r04_<c>_bench.json: issue_frac / valu_frac / hbm_traffic_frac = the committed summaries of the same
    profiling call over the line's own launch time; rocprofv3's average launch duration agrees with the HIP events
    (C4x: a step is two launches, the step kernel and the window kernel)."""
    line = json.loads(open(path).read().strip().splitlines()[-1])
    r = line["roofline"]
    wl, rnd = bench.workload_of_profile(path)  # the summaries of that round (or the newest older ones) stand behind the line
    t = r["avg_launch_ms"] * 1e-3
    insts, _, valu = bench.issue_counters(wl, max_round=rnd)
    if insts is None:  # (a workload without SQ passes: traffic and launch time only)
        assert r.get("issue_frac") is None and r.get("valu_frac") is None
        insts = valu = 0.0
        r = dict(r, issue_frac=0.0, valu_frac=0.0)
    two_launches = wl in ("C4x", "C4xu")
    assert r["issue_frac"] == pytest.approx(insts / (bench.N_SIMDS * bench.SM_CLOCK_HZ * t), rel=1e-6)
    assert r["valu_frac"] == pytest.approx(4.0 * valu / (bench.N_SIMDS * bench.SM_CLOCK_HZ * t), rel=1e-6)
    traffic, _ = bench.measured_traffic(wl, line["config"]["envs_per_gpu"], max_round=rnd)
    assert r["hbm_traffic_frac"] == pytest.approx(traffic / t / 1e9 / bench.HBM_PEAK_GBS, rel=1e-6)
    assert r["frac"] == pytest.approx(r["achieved"] / bench.HBM_PEAK_GBS, rel=1e-4)  # (the committed line is the compact one: 5 significant digits)
    stats = open(path.replace("_bench.json", "_kernel_stats.csv")).read().splitlines()
    step_rows = [row for row in stats[1:] if ("step_kernel" in row or "window_kernel" in row) and "reset" not in row]
    avg_ns = sum(float(row.split('",')[1].split(",")[2]) for row in step_rows[:2 if two_launches else 1])
    assert avg_ns * 1e-6 == pytest.approx(r["avg_launch_ms"], rel=0.06)  # HIP events vs rocprofv3, same run


def _synthetic_full_result(n_gpus=1, long_strings=400):
    """A full bench result with EVERY optional part present and prose strings far longer than the real ones."""
    prose = "x" * long_strings
    roof = dict(bound="latency", roof="hbm", kernel="aie_step_kernel_spec<0>" + "k" * 40, achieved=7225.123456789,
                peak=8000.0, unit="GB/s", frac=0.903140432, traffic=83412345.678, traffic_source=prose,
                hbm_traffic_frac=0.4187654321, frac_moved=0.4187654321, traffic_round=6, traffic_stale=False, counters_round=4,
                counters_stale=True, issue_frac=0.2987654321, wave_instructions_per_launch=123456789,
                issue_source=prose, valu_frac=0.6087654321, valu_instructions_per_launch=98765432, valu_roof=prose,
                issue_roof=prose, algorithmic_bytes_per_launch=179961856.0, algorithmic_bytes_per_unit=10984.0,
                unit_of_work=prose, achieved_final_layout=8281.123, frac_final_layout=1.035,
                final_layout_bytes_per_launch=1.0e8, final_layout_bytes_per_env_step={k: 1234567 for k in "abcdefgh"},
                avg_launch_ms=0.0249071234, launches_timed=1999, reset_launches_in_region=99, reset_ms_in_region=1.234,
                note=prose * 3, store_roof_GBps_this_box=6950.1234, traffic_frac_of_store_roof=0.62, store_roof_note=prose)
    cfg = dict(workload=prose * 2, workload_short="C2 " + "w" * 140, parallelism_short="p" * 70, envs_per_gpu=4096,
               global_envs=4096 * n_gpus, n_agents=4, rng=prose, policy=prose, phasing=prose, parallelism=prose,
               dev_switches=[], kernel_specialisation="run time (aie_specialize)")
    cpu = dict(value=134418.123456, unit="agent-steps/s", cores=256, kind="reference", per_core=525.1, steps=1234567,
               resets=1234, seconds=10.0, sample_short="s" * 150, sample=prose * 2)
    out = dict(metric="agent-steps/sec, " + "m" * 60, value=6.22123456e8, unit="agent-steps/s", n_gpus=n_gpus, steps=2000,
               warmup=200, ms_per_step=0.0263123456, higher_is_better=True, scaling="weak", vs_baseline=None,
               dtype="u8/i32 state + f64 coin/utility (f32 observations)", data="synthetic", config=cfg,
               per_rank_seconds=[0.0526123456] * n_gpus, per_rank_avg_launch_ms=[0.0248123456] * n_gpus, host_issue_seconds=0.05, gpu_region_seconds=0.0521234,
               exchange_ok=True, roofline=roof, cpu_baseline=cpu,
               cpu_port=dict(value=3.8e6, unit="agent-steps/s", cores=256, kind="port", sample=prose),
               gather=dict(collectives=32, bytes_per_collective=6291456, wait_seconds=0.00123456))
    out["workloads"] = {name: dict(metric=prose, value=3.9e9, unit="agent-steps/s", steps=200, warmup=20,
                                   ms_per_step=1.6725123, dtype=prose, config=cfg, roofline=dict(roof),
                                   gpu_region_seconds=0.3345, cpu_baseline=dict(cpu))
                        for name, _, _ in bench.SIDE_WORKLOADS}
    out["workloads"]["C9"] = {"error": "RuntimeError(" + prose + ")"}
    return out


@pytest.mark.parametrize("n_gpus", [1, 8])
def test_driver_line_stays_under_4_kb_and_keeps_the_contract(n_gpus):
    """Round 3's driver record was unparsed because the last stdout line had grown past the driver's 8 KB tail.  The
    line built from a result with every optional part present (and absurdly long prose) stays under 4 KB, is valid
    JSON, and still carries the contract fields, `roofline` and `cpu_baseline`."""
    full = _synthetic_full_result(n_gpus)
    assert len(json.dumps(full)) > 20000  # the synthetic result is at least as large as round 3's real one
    text = bench.compact_line(full)
    assert len(text) < 4096 and "\n" not in text
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["config"]["global_envs"] == n_gpus * line["config"]["envs_per_gpu"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"]
    assert list(line["roofline"])[0] == "bound"  # what limits the kernel comes first: a `frac` above 1 is read against it
    # provenance of the counter-derived fields: which round's committed summaries, and whether they were
    # collected on the kernels of this tree
    assert line["roofline"]["frac_moved"] == pytest.approx(full["roofline"]["hbm_traffic_frac"], rel=1e-4)
    assert line["roofline"]["counters_round"] == 4 and line["roofline"]["counters_stale"] is True
    assert line["roofline"]["traffic_round"] == 6 and line["roofline"]["traffic_stale"] is False
    for name, _, _ in bench.SIDE_WORKLOADS:
        assert line["workloads"][name]["stale"] is True and list(line["workloads"][name])[0] == "bound"
    assert line["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-4)
    assert line["value"] == pytest.approx(full["value"], rel=1e-4)
    assert set(line["cpu_baseline"]) == {"value", "unit", "cores", "kind", "sample"}
    for name, _, _ in bench.SIDE_WORKLOADS:
        assert line["workloads"][name]["value"] == pytest.approx(3.9e9, rel=1e-4)
        assert line["workloads"][name]["cpu_ref"] == pytest.approx(134418.12, rel=1e-4)
    if n_gpus > 1:
        assert len(line["per_rank_seconds"]) == n_gpus and line["gather"]["collectives"] == 32
        assert len(line["per_rank_avg_launch_ms"]) == n_gpus


def test_emit_prints_the_compact_line_last(tmp_path, capsys):
    full = _synthetic_full_result()
    bench.emit(full, str(tmp_path / "detail.json"))
    lines = capsys.readouterr().out.strip().splitlines()
    assert len(lines) == 2 and len(lines[-1]) < 4096
    assert json.loads(lines[0]) == json.loads(open(tmp_path / "detail.json").read())
    assert json.loads(lines[-1])["detail"] == "bench_detail.json"


def test_summary_provenance_flags_summaries_of_other_kernels(tmp_path, monkeypatch):
    """bench.summary_provenance: the round comes from the file name; a summary without a source hash (rounds 1 - 5) or with
    another tree's hash is stale, one that carries this tree's hash is not."""
    import ai_economist_amd  # noqa: F401
    from ai_economist_amd import _build

    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (tmp_path / "profiles").mkdir()
    now = _build._source_hash()
    for name, payload in (("r04_c2_sq_counters.json", {}), ("r06_c2_pmc.json", {"source_hash": now}),
                          ("r06_c3_pmc.json", {"source_hash": "0" * 64})):
        (tmp_path / "profiles" / name).write_text(json.dumps(payload))
    assert bench.summary_provenance("profiles/r04_c2_sq_counters.json") == (4, True)
    assert bench.summary_provenance("profiles/r06_c2_pmc.json") == (6, False)
    assert bench.summary_provenance("profiles/r06_c3_pmc.json") == (6, True)
    assert bench.summary_provenance(None) == (None, None)
