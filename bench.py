#!/usr/bin/env python
"""bench.py -- random-policy rollout throughput of the batched Foundation env.step().

    python bench.py --gpus N --steps K --warmup W [--workload C1|C2|C2v|P2|C3|C4|C4x|C5] [--no-workloads]

Workloads = BASELINE.json configs[0..4].  The default run (1 GPU, C2) is the headline line and
ALSO times every other BASELINE configuration in a short window of the same invocation (`"workloads": {C1, C3, C4, C4x,
C5: {value, ms_per_step, roofline}}`), so that all of them are measured under the caller's clock.  C2 is the
configuration the metric is quoted on: layout_from_file/simple_wood_and_stone, 25x25 quadrant layout, 4 mobile agents + planner, Build +
ContinuousDoubleAuction(max_num_orders 5) + Gather + PeriodicBracketTax (model_wrapper, us-federal, period 100),
starting_agent_coin 10, episode_length 1000, 4096 replicas PER GPU (weak scaling), uniform random actions from a
counter RNG keyed (seed, global replica, t, slot).

One "step" = one env.step() of every replica of the rank (+ the masked reset launch of the replicas whose episode
ended, + with N > 1 the (reward, done) gather to rank 0, the path's only exchange).  Inputs are resident in HBM
before the timed region.  The replicas are DE-PHASED before the timed region: blocks of replicas are reset at
staggered points of a prologue, so that any window of the rollout -- including a 20-step one -- contains tax days,
order expiries, mid-episode order books and episode ends in their long-run proportions.

`--gpus N` with N > 1 started as a plain process re-executes itself under torch.distributed.run (one rank per GPU);
`--launcher torchrun` takes that path with --gpus 1 too (the N > 1 launch code on a one-GPU box).  `--generic-kernel`
(development) times the generic step kernel instead of the configuration's compile-time instance.

Prints (rank 0) the full result as one JSON line, writes it to bench_detail.json, and then -- LAST line of stdout, the
line the driver parses, <= 4 KB (compact_line) -- the contract fields plus the numbers of
  roofline      dominant kernel: algorithmic bytes per launch (SURVEY.md 8(d) figure, and the figure recomputed from
                the final layouts) / average launch duration measured live with HIP events on the launch stream;
                measured HBM traffic and issued wave-instructions from the committed rocprofv3 PMC summaries of the
                same command (-> hbm_traffic_frac, issue_frac, valu_frac = the share of all VALU pipe time the
                kernel's vector instructions fill, and `bound` derived from them); C5: the box's own pure-store roof
                measured live
  cpu_baseline  the UNMODIFIED reference env.step (oracle/_ref, kind "reference") on this host's cores, P pinned
                processes timed concurrently on a bounded window; `cpu_port` = the C restatement (oracle/) beside it
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

os.environ.setdefault("OMP_WAIT_POLICY", "passive")
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s achievable copy rate
INFINITY_CACHE_BYTES = 256 << 20  # MI355X memory-side cache (MI355X_MICROARCH.md): an arena that fits is re-read from it,
#                                   and FETCH_SIZE / WRITE_SIZE (L2 <-> fabric) then are not DRAM traffic
ACTION_SEED = 1234
ENV_SEED = 1
STAGGER_STRIDE = 20  # steps between the prologue's block resets = steps between reset launches in the rollout

GTB = [["Build", {}], ["ContinuousDoubleAuction", {"max_num_orders": 5}], ["Gather", {}], ["PeriodicBracketTax", {}]]
C2_CFG = dict(scenario_name="layout_from_file/simple_wood_and_stone", n_agents=4, world_size=[25, 25],
              episode_length=1000, components=GTB, starting_agent_coin=10,
              env_layout_file="quadrant_25x25_20each_30clump.txt")


def _c5_cfg():
    import numpy as np

    rs = np.random.RandomState(4)  # SimpleLabor's skills are a construction-time Monte-Carlo draw in the reference
    return dict(scenario_name="one-step-economy", n_agents=100, world_size=[1, 1], episode_length=2,
                components=[["SimpleLabor", {"skills": [float(x) for x in np.sort(1 + rs.rand(100) * 2)]}],
                            ["PeriodicBracketTax", {"bracket_spacing": "us-federal", "period": 1,
                                                    "tax_model": "model_wrapper"}]])


def _phase2_cfg():
    from ai_economist_amd import _specs

    kw = dict(_specs.PHASE2, scenario_name="layout_from_file/simple_wood_and_stone", dense_log_frequency=20)
    kw["components"] = [[name, dict(c)] for name, c in kw["components"]]
    return kw


def _c4_cfg():
    # the env block of the reference's training/run_configs/covid_and_economy_environment.yaml:10-39
    return dict(scenario_name="CovidAndEconomySimulation", collate_agent_step_and_reset_data=True,
                components=[["ControlUSStateOpenCloseStatus", {"action_cooldown_period": 28}],
                            ["FederalGovernmentSubsidy", {"num_subsidy_levels": 20, "subsidy_interval": 90,
                                                          "max_annual_subsidy_per_person": 20000}],
                            ["VaccinationCampaign", {"daily_vaccines_per_million_people": 3000, "delivery_interval": 1,
                                                     "vaccine_delivery_start_date": "2021-01-12"}]],
                economic_reward_crra_eta=2, episode_length=540, flatten_masks=True, flatten_observations=False,
                health_priority_scaling_agents=0.3, health_priority_scaling_planner=0.45,
                infection_too_sick_to_work_rate=0.1, multi_action_mode_agents=False, multi_action_mode_planner=False,
                n_agents=51, path_to_data_and_fitted_params="", pop_between_age_18_65=0.6,
                risk_free_interest_rate=0.03, world_size=[1, 1], start_date="2020-03-22", use_real_world_data=False,
                use_real_world_policies=False,
                # opt-in extension (not the default): the unemployment filter bank as an O(1) recurrence instead of the
                # reference's 600-tap window sum; `unemployed` within 1.5e-6 relative of the window sums.  Workload
                # "C4x" runs the default (window sums).
                filter_recurrence=True)


# name -> (description, cfg builder, replicas per GPU, SURVEY 8(d) B_alg per unit, units per replica-step,
#          kernel, counted agents)
C1_CFG = dict(scenario_name="uniform/simple_wood_and_stone", n_agents=4, world_size=[15, 15], episode_length=1000,
              components=[["Build", {}], ["Gather", {}]], starting_agent_coin=10, starting_stone_coverage=0.10,
              starting_wood_coverage=0.10)
WORKLOADS = {
    "C1": dict(short="BASELINE configs[0] scenario batched: uniform/simple_wood_and_stone 15x15, 4 agents, Build+Gather",
               desc="BASELINE configs[0]'s scenario batched: uniform/simple_wood_and_stone 15x15, 4 agents + planner, "
                    "Build+Gather (no auction, no taxes), episode_length 1000; every reset draws a new source layout on "
                    "the device",
               cfg=lambda: dict(C1_CFG), envs=4096, survey_bytes=5303.0, kernel="aie_step_kernel"),
    "C2": dict(short="BASELINE configs[1]: gather-trade-build 25x25, 4 agents + planner, Build+CDA+Gather+PeriodicBracketTax, 4096 replicas/GPU",
               desc="BASELINE configs[1]: gather-trade-build 25x25 quadrant layout, 4 agents + planner, Build+"
                    "ContinuousDoubleAuction(max_num_orders=5)+Gather+PeriodicBracketTax, episode_length 1000",
               cfg=lambda: dict(C2_CFG), envs=4096, survey_bytes=10984.0, kernel="aie_step_kernel"),
    "C2v": dict(short="configs[1] with starting_agent_coin=15, isoelastic_eta=0.5, episode_length=500 (same instance family)",
                desc="BASELINE configs[1] with other scalars (starting_agent_coin 15, isoelastic_eta 0.5, episode_length 500): "
                     "the compile-time instance of C2's family runs it, no call by the user",
                cfg=lambda: dict(C2_CFG, starting_agent_coin=15, isoelastic_eta=0.5, episode_length=500), envs=4096,
                survey_bytes=10984.0, kernel="aie_step_kernel"),
    "P2": dict(short="the reference's phase-2 training YAML env block (incl. dense_log_frequency 20; no episode being logged)",
               desc="SURVEY 8(d) C2'': the env block of the reference's tutorials/rllib/phase2/config.yaml as it stands "
                    "(planner without maps, fixed four skills, annealed tax cap, dense_log_frequency 20): runs on the "
                    "build's P2 instance; the dense-log replica records events in every 20th episode only "
                    "(aie_set_dense_log_active), the window times one of the other 19",
               cfg=lambda: _phase2_cfg(), envs=4096, survey_bytes=6601.0, kernel="aie_step_kernel", dense_log_off=True),
    "C3": dict(short="BASELINE configs[2], one GPU share: as C2 with 10 agents, 4096 replicas/GPU",
               desc="BASELINE configs[2], one GPU's share (32768 replicas over 8 GPUs = 4096 each): as C2 with 10 agents",
               cfg=lambda: dict(C2_CFG, n_agents=10), envs=4096, survey_bytes=7666.0, kernel="aie_step_kernel"),
    "C2f": dict(short="configs[1] with rng_mode='fast' (counter-based Philox2x32-10 stream per replica; NOT NumPy's stream)",
                desc="BASELINE configs[1] in the throughput mode rng_mode='fast' (include/aie.h AIE_RNG_FAST): every "
                     "np.random.* draw of the replicas comes from a counter-based stream (Philox2x32-10 keyed by seed + "
                     "global replica) instead of NumPy's MT19937 -- 16 B of generator state per replica instead of 2 496, "
                     "the regeneration computes only the words of its source cells; checked bit for bit against the "
                     "oracle's restatement of the same generator (tests/test_rng_fast.py); never the headline",
                cfg=lambda: dict(C2_CFG, rng_mode="fast"), envs=4096, survey_bytes=10984.0, kernel="aie_step_kernel",
                rng="fast"),
    "C1f": dict(short="configs[0] scenario batched with rng_mode='fast': layouts from a stream of their own, drawn ahead of the resets",
                desc="C1 in the throughput mode rng_mode='fast' (see C2f): a replica's k-th reset draws its source layout from "
                     "a counter stream keyed by the replica and k (csrc/aie_layout.h: aie_layout_stream), so the library draws "
                     "layouts ahead of their resets -- one refill launch per quarter of the replicas instead of one 0.3 ms "
                     "chain inside every masked reset",
                cfg=lambda: dict(C1_CFG, rng_mode="fast"), envs=4096, survey_bytes=5303.0, kernel="aie_step_kernel", rng="fast"),
    "C3f": dict(short="configs[2] one GPU share with rng_mode='fast' (counter-based stream; NOT NumPy's stream)",
                desc="BASELINE configs[2], one GPU's share, in the throughput mode rng_mode='fast' (see C2f)",
                cfg=lambda: dict(C2_CFG, n_agents=10, rng_mode="fast"), envs=4096, survey_bytes=7666.0,
                kernel="aie_step_kernel", rng="fast"),
    "C4": dict(short="BASELINE configs[3] COVID 51 states + planner, opt-in O(1) filter recurrence",
               desc="BASELINE configs[3]: CovidAndEconomySimulation, 51 US-state agents + planner, run config "
                    "covid_and_economy_environment.yaml, episode_length 540; filter_recurrence=True (opt-in O(1) update of "
                    "the unemployment filter bank, `unemployed` within 1.5e-6 relative of the default window sums)",
               cfg=_c4_cfg, envs=8192, survey_bytes=1580.0, kernel="aie_covid_step_kernel"),
    "C4x": dict(short="BASELINE configs[3] COVID 51 states + planner, reference-exact window sums",
               desc="BASELINE configs[3] with the default unemployment filter bank (the reference's 600-tap window sums; "
                     "C4 runs the opt-in O(1) recurrence)",
                cfg=lambda: dict(_c4_cfg(), filter_recurrence=False), envs=8192, survey_bytes=1580.0,
                kernel="aie_covid_step_kernel"),
    "C4xu": dict(short="BASELINE configs[3] COVID, window sums, SURVEY 8(d)'s UNMASKED uniform policy",
                 desc="BASELINE configs[3] with the reference's window sums under the policy SURVEY.md 8(d) prescribes: "
                      "agents U{0..10}, planner U{0..20} regardless of the masks (the reference's components do not re-check "
                      "them, covid19_components.py:180-199): stringency levels change on most days",
                 cfg=lambda: dict(_c4_cfg(), filter_recurrence=False), envs=8192, survey_bytes=1580.0,
                 kernel="aie_covid_step_kernel", policy="unmasked"),
    "C2@16384": dict(short="configs[1] at 16384 replicas/GPU (arena 0.72 GB: past the 256 MiB Infinity Cache)",
                     desc="BASELINE configs[1] with 16384 replicas on the GPU: the arena (0.72 GB) no longer fits the 256 MiB "
                          "Infinity Cache", cfg=lambda: dict(C2_CFG), envs=16384, survey_bytes=10984.0,
                     kernel="aie_step_kernel", profile_tag="c2_e16384"),
    "C2@65536": dict(short="configs[1] at 65536 replicas/GPU (arena 2.9 GB)",
                     desc="BASELINE configs[1] with 65536 replicas on the GPU (arena 2.9 GB)", cfg=lambda: dict(C2_CFG),
                     envs=65536, survey_bytes=10984.0, kernel="aie_step_kernel", profile_tag="c2_e65536"),
    "C5": dict(short="BASELINE configs[4]: one-step-economy 100 agents + SimpleLabor + tax",
               desc="BASELINE configs[4]: one-step-economy, 100 agents + SimpleLabor + PeriodicBracketTax(period 1), "
                    "episode_length 2",
               cfg=_c5_cfg, envs=65536, survey_bytes=987.0, kernel="aie_ose_step_kernel"),
}


def make_env(cfg, n_envs, **extra):
    from ai_economist_amd import foundation

    kw = dict(cfg)
    scenario = kw.pop("scenario_name")
    kw["components"] = [tuple(c) for c in kw["components"]]
    kw.update(extra)
    return foundation.make_env_instance(scenario, n_envs=n_envs, **kw)


def layout_bytes_per_env_step(be, wl):
    """B_alg of SURVEY.md 8(d) recomputed from the FINAL layouts: observation tensors in the reference's own
    format + state record read+write (+ C4: the 601-day stringency window each state streams) + actions +
    rewards/done."""
    obs = sum(t[0].numel() * t.element_size() for k, t in be.tensors.items() if k.startswith("obs_"))
    n = be.n
    if wl.startswith("C4"):
        L, F = int(be.cfg.covid.filter_len), int(be.cfg.covid.num_filters)
        state_rw = 2 * (8 + 1) * n * 4 + n
        if be.cfg.covid.filter_recurrence:  # O(1) filter update: F float64 sums per state r+w, 5 recent levels + today's per state
            b = dict(filter_sums_rw=2 * F * n * 8, history_bytes=6 * n, state_rw=state_rw, obs=obs, act=(n + 1) * 4,
                     rew_done=(n + 1) * 4 + 1)
        else:
            # window sums over the change events (round 4): F float64 sums per state r + w (formed one step ahead), the
            # states' event lists (4 B per live event, the batch's mean list length at the end of the window, read once
            # per step; + head / tail r + w), today's and yesterday's level rows -- the 601-day window is streamed only by
            # replicas whose list overflowed
            ht = be.tensors["stringency_change_head_tail"]
            live = float(((ht >> 16) - (ht & 0xffff)).float().mean().item())
            b = dict(window_sums_rw=2 * F * n * 8, event_lists=live * 4 * n + 2 * 4 * n, history_bytes=6 * n, state_rw=state_rw,
                     obs=obs, act=(n + 1) * 4, rew_done=(n + 1) * 4 + 1, mean_live_events_per_state=live)
            b = {k: v for k, v in b.items()}
    else:
        key = "cells" if "cells" in be.descs else "inv_coin"
        rec = be.descs[key][2][0]  # per-replica record bytes = stride of any record field along the replica axis
        b = dict(obs=obs, state_rw=2 * rec, act=n * 4 + be._act_p_width() * 4, rew_done=(n + 1) * 4 + 1)
    b["total"] = sum(v for k, v in b.items() if k != "mean_live_events_per_state")
    return b


def profile_tag(wl):
    """profiles/rNN_<tag>_{pmc,sq_counters,kernel_stats}.*: the committed rocprofv3 summaries of a workload."""
    return WORKLOADS[wl].get("profile_tag", wl.lower())


def workload_of_profile(path):
    """(workload, round) of a committed profiles/rNN_<tag>_<kind>.{json,csv} file."""
    import re

    m = re.match(r"^r(\d+)_(.+)_(bench\.json|pmc\.json|sq_counters\.json|kernel_stats\.csv)$", os.path.basename(path))
    if not m:
        return None, None
    for wl in WORKLOADS:
        if profile_tag(wl) == m.group(2):
            return wl, int(m.group(1))
    return None, int(m.group(1))


def measured_traffic(wl, envs_per_gpu, max_round=None):
    """HBM bytes per launch of the dominant kernel from the newest committed rocprofv3 PMC summary for this workload
    (profiles/*<wl>*pmc.json, made by tools/pmc_summary.py from separate FETCH_SIZE / WRITE_SIZE passes of this same
    command).  Counters cannot be collected from inside the run; only valid for the default batch size."""
    import glob
    import re

    def version(path):
        return [int(x) for x in re.findall(r"\d+", os.path.basename(path))]

    if wl not in WORKLOADS:
        return None, None
    tag = profile_tag(wl)
    files = [f for f in glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json"))
             if re.match(r"^r\d+_%s_pmc\.json$" % re.escape(tag), os.path.basename(f))]
    if wl == "C2":
        files += [f for f in glob.glob(os.path.join(ROOT, "profiles", "r01_v*_pmc.json"))]
    files = sorted(files, key=version)
    if max_round is not None:  # (tests: the summaries a committed line of that round was derived from)
        files = [f for f in files if version(f)[0] <= max_round]
    if not files or envs_per_gpu != WORKLOADS[wl]["envs"]:
        return None, None
    d = json.load(open(files[-1]))
    return d.get("hbm_bytes_per_launch"), os.path.relpath(files[-1], ROOT)


def summary_provenance(path):
    """(round, stale) of a committed counter summary: the round from its file name (profiles/rNN_...), stale = it was
    collected on kernels other than the ones this library was built from -- its `source_hash` (tools/pmc_summary.py,
    tools/sq_passes.sh; absent before round 6) is not the hash of the sources in this tree."""
    import re

    if not path:
        return None, None
    m = re.match(r"^r(\d+)_", os.path.basename(path))
    rnd = int(m.group(1)) if m else None
    try:
        have = json.load(open(os.path.join(ROOT, path))).get("source_hash")
    except (OSError, ValueError):
        have = None
    try:
        import ai_economist_amd  # noqa: F401
        from ai_economist_amd import _build

        now = _build._source_hash()
    except Exception:
        now = None
    return rnd, (have is None or now is None or have != now)


def usable_cores():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:  # cgroup v2 CPU quota
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def cpu_reference_baseline(cfg, seconds=10.0, max_procs=64, policy="unmasked", episodes=0):
    """The reference's own env.step (oracle/_ref or the live tree, through oracle/ref_harness.py): P = usable cores
    processes, one environment each, pinned, stepped concurrently with uniform random actions."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_harness

    if not ref_harness.reference_available():
        return None
    cfg = json.loads(json.dumps(cfg))
    for comp in cfg["components"]:  # `skills=` is an extension of the host mirror (the reference estimates them itself)
        comp[1].pop("skills", None)
    cfg.pop("filter_recurrence", None)  # (host-mirror extension as well)
    ncores = usable_cores()
    P = max(1, min(ncores, max_procs))
    try:
        cores = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cores = list(range(P))
    # episodes > 0: SURVEY 8(d)'s protocol (1 warm-up episode + `episodes` timed whole episodes incl. resets) instead of the
    # free-running window; the warm-up episode runs ahead of the common start
    warm = 6.0 + 0.05 * P
    if episodes:
        warm += 1.3 * float(cfg.get("episode_length", 1000)) / 1500.0 + 5.0  # (a reference episode: ~0.5 ms per step and worker)
    start = time.time() + warm
    procs = []
    for k in range(P):
        cmd = [sys.executable, os.path.join(ROOT, "oracle", "ref_worker.py"), "--cfg-json", json.dumps(cfg),
               "--core", str(cores[k % len(cores)]), "--start", repr(start), "--seconds", repr(seconds),
               "--seed", str(1 + k), "--policy", policy, "--episodes", str(int(episodes))]
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                                      env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")))
    rate, steps, late, n_agents, ok, resets = 0.0, 0, 0, None, 0, 0
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=seconds + 120)
            d = json.loads(out.strip().splitlines()[-1])
        except Exception:
            pr.kill()
            continue
        ok += 1
        n_agents = d["n_agents"]
        rate += d["steps"] * d["n_agents"] / d["elapsed"]
        steps += d["steps"]
        late += int(d["late"])
        resets += int(d.get("resets", 0))
    if not ok:
        return None
    if episodes:
        return dict(value=rate, unit="agent-steps/s", cores=ok, kind="reference", per_core=rate / ok, steps=steps, resets=resets,
                    policy=policy, episodes=episodes,
                    sample_short="%d pinned procs x 1 env, unmodified reference env.step, %s uniform policy, SURVEY 8(d): 1 warm-up + "
                                 "%d timed whole episodes incl. resets (%d steps)" % (ok, policy, episodes, steps),
                    sample="%d pinned processes x one environment each, the unmodified reference env.step (base_env.py:929-1032) "
                           "with uniform random actions (%s): one warm-up episode, then %d timed whole episodes including "
                           "their env.reset() per process, started on a common clock (SURVEY.md 8(d)'s protocol): %d steps in "
                           "total, %d agents each; host reports %d usable cores%s"
                           % (ok, policy, episodes, steps, n_agents, ncores, ", %d workers started late" % late if late else ""))
    return dict(value=rate, unit="agent-steps/s", cores=ok, kind="reference", per_core=rate / ok,
                steps=steps, resets=resets, seconds=seconds, policy=policy,
                sample_short="%d pinned procs x 1 env, unmodified reference env.step, %s uniform policy, free-running %.0f s "
                             "window: %d steps, %d env.reset() inside it" % (ok, policy, seconds, steps, resets),
                sample="%d pinned processes x one environment each, the unmodified reference env.step "
                       "(base_env.py:929-1032, %s) stepped concurrently for %.0f s with uniform random actions (%s): "
                       "%d steps in total (%d agents each), %d episode ends (env.reset()) inside the window -- a "
                       "free-running wall-clock window, not SURVEY 8(d)'s 3 whole episodes; host reports %d usable cores%s"
                       % (ok, "byte-compiled into oracle/_ref" if not ref_harness.reference_is_live_tree()
                          else "live tree", seconds, policy, steps, n_agents, resets, ncores,
                          ", %d workers started late" % late if late else ""))


def cpu_port_baseline(cfg, seconds_target=4.0):
    """The C restatement (oracle/aie_oracle.c, OpenMP over replicas) on a bounded sample: a second, labelled figure."""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from oracle_lib import OracleEnv

    E = 1024
    env = make_env(cfg, n_envs=E)
    o = OracleEnv(env.build_config(), env.layout_planes())
    o.seed(ENV_SEED)
    o.reset()
    rng = np.random.RandomState(ACTION_SEED)
    n = env.n_agents
    acts = rng.randint(0, 50, size=(20, E, n)).astype(np.int32)
    acts_p = rng.randint(0, 22, size=(20, E, 7)).astype(np.int32)
    th = usable_cores()
    for t in range(2):
        o.step(acts[t], acts_p[t], nthreads=th)
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds_target:
        for t in range(20):
            o.step(acts[t], acts_p[t], nthreads=th)
        steps += 20
    dt = time.perf_counter() - t0
    return dict(value=E * n * steps / dt, unit="agent-steps/s", cores=th, kind="port",
                sample="%d replicas x %d steps (%.1f s), C restatement of the reference step (oracle/aie_oracle.c), "
                       "OpenMP over replicas" % (E, steps, dt))


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) as a plain process: re-execute under torch.distributed.run."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(max(1, args.gpus)),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


class Rollout:
    """The timed loop of one rank: one launch per step, de-phased replicas, masked resets on a host-known
    schedule (block g of the replicas, g = e mod G, ends its episode at steps = offset_g mod episode_length)."""

    def __init__(self, wl, env, rank_offset, stagger=True, auto_reset=True, unmasked_covid=False):
        import torch

        self.torch = torch
        self.wl, self.env, self.be = wl, env, env.backend
        self.off = rank_offset
        self.T = int(env.episode_length)
        self.E = self.be.E
        self.t = 0  # rollout steps since the common reset
        self.fused = True  # every scenario draws the next step's random actions inside the step launch
        # C5 (2-step episodes): auto-reset -- the replicas restart inside the step launch that ends their episode
        # (aie_set_auto_reset), as a vectorised trainer runs it; no separate reset launches
        self.auto_reset = wl == "C5" and auto_reset
        if self.auto_reset:
            self.be.set_auto_reset(True)
        stride = STAGGER_STRIDE if (stagger and self.T > 2 * STAGGER_STRIDE) else 0
        self.G = (self.T // stride) if stride else 1
        self.stride = stride
        e = torch.arange(self.E, device=self.be.device)
        self.group_masks = [((e % self.G) == g).to(torch.uint8) for g in range(self.G)] if self.G > 1 else None
        self.reset_events = []
        # COVID: the random policy respects the action masks (a state's stringency levels only outside its cool-down,
        # subsidy levels only on interval starts) -- the policy a trainer that applies `action_mask` to its logits
        # starts from, and the one under which the reference's cool-down component means anything; drawn inside the
        # step launch like the uniform one (aie_step_sample_next_masked)
        self.masked = wl.startswith("C4") and not unmasked_covid
        self.cur = (self.be.sample_masked_actions if self.masked else self.be.sample_random_actions)(ACTION_SEED, self.off, slot=0)
        self.slot = 0

    def prologue(self):
        """De-phasing (outside the timed region): one episode length of steps; block g is reset after g * stride of
        them, so afterwards block g sits at timestep T - g * stride ... and the blocks' episodes end `stride` apart."""
        if self.G == 1:
            return 0
        for s in range(self.T):
            if s % self.stride == 0 and 0 < s // self.stride < self.G:
                self.be.reset(self.group_masks[s // self.stride])
            self._launch()
        self.t = 0
        # block 0 was never re-reset: its episode (T steps) ended exactly now
        self.be.reset(self.group_masks[0])
        return self.T

    def _launch(self):
        be = self.be
        if self.fused:
            self.cur = be.step_sample_next(self.cur[0], self.cur[1], ACTION_SEED, self.off, next_slot=self.slot ^ 1,
                                           masked=self.masked)
            self.slot ^= 1
        else:
            a, p = be.sample_random_actions(ACTION_SEED, self.off)
            be.step(a, p)

    def step(self, timed=False):
        self._launch()
        self.t += 1
        if self.G == 1:
            if self.t % self.T == 0 and not self.auto_reset:
                self._reset(self.be.tensors["done"], timed)
        elif self.t % self.stride == 0:
            # block g sits at timestep T - g * stride (mod T) at t = 0, so its episode ends at t = g * stride (mod T)
            g = (self.t % self.T) // self.stride
            if g < self.G:
                self._reset(self.group_masks[g], timed)

    def warm_reset_path(self):
        """One masked reset with an all-zero mask through the timed path (outside the timed region): the kernel finds
        nothing to do, the host-side first-use costs are paid here."""
        zero = self.torch.zeros(self.E, dtype=self.torch.uint8, device=self.be.device)
        self._reset(zero, True)
        self.reset_events.clear()

    def _reset(self, mask, timed):
        if timed:
            ev = (self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True))
            ev[0].record()
            self.be.reset(mask)
            ev[1].record()
            self.reset_events.append(ev)
        else:
            self.be.reset(mask)


SM_CLOCK_HZ = 2.4e9   # MI355X peak engine clock (MI355X_MICROARCH.md): the issue roof is one wave-instruction per SIMD and clock
N_SIMDS = 256 * 4


def issue_counters(wl, max_round=None):
    """Wave-instructions the dominant kernel issues per launch, from the newest committed SQ-counter summary of this
    workload (profiles/*_<wl>_sq_counters.json: rocprofv3 --pmc SQ_INSTS_* passes of this same command, tools/sq_passes.sh)."""
    import glob
    import re

    files = sorted([f for f in glob.glob(os.path.join(ROOT, "profiles", "*_sq_counters.json"))
                    if re.match(r"^r\d+_%s_sq_counters\.json$" % re.escape(profile_tag(wl)), os.path.basename(f))],
                   key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))])
    if max_round is not None:
        files = [f for f in files if int(re.findall(r"\d+", os.path.basename(f))[0]) <= max_round]
    if not files:
        return None, None, None
    d = json.load(open(files[-1]))
    if wl in ("C4x", "C4xu") and any("window_kernel" in k for k in d):
        # window sums: a step is two launches (the step kernel, then the change-event lists' upkeep): their sum
        rows = [v for k, v in d.items() if isinstance(v, dict) and ("covid_step_kernel" in k or "window_kernel" in k)]
        return (sum(v["wave_instructions_per_launch"] for v in rows), os.path.relpath(files[-1], ROOT),
                sum((v.get("counters") or {}).get("SQ_INSTS_VALU", 0.0) for v in rows))
    best = None
    for name, v in d.items():
        if isinstance(v, dict) and "reset" not in name and v.get("wave_instructions_per_launch"):
            if best is None or v["wave_instructions_per_launch"] > best[1]:
                best = (name, v["wave_instructions_per_launch"], (v.get("counters") or {}).get("SQ_INSTS_VALU"))
    return (best[1] if best else None), os.path.relpath(files[-1], ROOT), (best[2] if best else None)


def store_roof_gbs(device, nbytes=4 << 30):
    """Pure-store roof of THIS box, measured live: a device fill of 4 GiB (launch times of the store-bound
    one-step-economy kernel move by +-6 % between boxes, so the roof it is held against has to come from the same box)."""
    import torch

    buf = torch.empty(nbytes // 4, dtype=torch.int32, device=device)
    best = 0.0
    for _ in range(4):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        buf.fill_(7)
        ev1.record()
        torch.cuda.synchronize()
        best = max(best, nbytes / (ev0.elapsed_time(ev1) * 1e-3) / 1e9)
    del buf
    return best


def run_workload(wl, args, steps, warmup, rank, local_rank, world, device):
    """One workload's rollout on this rank: environment, de-phasing prologue, warm-up, the timed window.  Returns the
    JSON fields of the workload (rank 0) or None."""
    import gc

    import torch

    from ai_economist_amd.sharding import RewardLogGather

    W = WORKLOADS[wl]
    cfg = W["cfg"]()
    E = args.envs_per_gpu or W["envs"]
    env_offset = rank * E
    env = make_env(cfg, n_envs=E, device=device, env_offset=env_offset)
    env.seed(ENV_SEED)
    env.reset()
    be = env.backend
    n = env.n_agents
    if W.get("dense_log_off"):
        be.set_dense_log_active(False)
    if args.generic_kernel:  # development: what a configuration without an instance runs
        be.lib.aie_select_step_kernel(be.handle, 1)
    roll = Rollout(wl, env, env_offset, stagger=not args.no_stagger, auto_reset=not args.no_auto_reset,
                   unmasked_covid=args.covid_unmasked_policy or W.get("policy") == "unmasked")
    gc.collect()
    gc.disable()  # no collector pause between here and the end of the timed window (it may be as short as 20 launches);
    #               collected now, while the GPU has nothing queued: a pause later would let it run dry before the window
    prologue_steps = roll.prologue()

    # N > 1: (reward, done) of every step travel to the learner rank, 64 steps per collective, straight from the
    # log the step kernel fills (aie_set_reward_log) -- no per-step launches or collectives beside the step.
    # A failure here fails the run: the scaling line must not silently drop the exchange.
    gather = None
    if world > 1 or args.force_gather:
        gather = RewardLogGather(be, steps_per_gather=64, force_collective=args.force_gather)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    # warm-up runs the very code path of the timed region (event-bracketed resets included), so that first-use costs
    # (event creation, cold Python paths) are not charged to a short timed window
    warm1 = torch.cuda.Event(enable_timing=True)
    # one pair of HIP events on the launch stream around the whole timed region (K back-to-back step launches)
    # plus one pair around each of the (rare) reset launches inside it: average step-kernel launch period =
    # (region - resets) / K.  Bracketing every step launch would add ~4 us of queue packets per step.
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if not roll.auto_reset:
        roll.warm_reset_path()
    ev0.record()  # (a torch event creates its HIP event at the first record(): not inside the window)
    ev1.record()
    close_flag = close_one = None
    if os.environ.get("AIE_BENCH_CLOSE", "flag") == "flag":
        close_flag = torch.zeros(1, dtype=torch.int32).pin_memory()
        close_one = torch.ones(1, dtype=torch.int32, device=device)
        close_flag.copy_(close_one, non_blocking=True)  # (the copy path warmed up outside the window)
        torch.cuda.synchronize()
        close_flag.zero_()
    for _ in range(warmup):
        roll.step(timed=True)
        if gather is not None:
            gather.after_step()
    if gather is not None:
        gather.finish()  # the warm-up's (reward, done) rows leave before the window; the window ships its own
    warm1.record()
    # The opening bracket (barrier + synchronize) leaves the GPU idle; everything the host has to do before the first
    # timed launch is done BEFORE it (events exist, the reset-event list is cleared inside the window's bookkeeping),
    # so that the idle gap -- during which the clocks start to drop -- is as short as the bracket itself.
    n_warm_resets = len(roll.reset_events)
    # drain the GPU by polling: a blocking wait puts the host core to sleep for the ~30 ms the prologue still needs,
    # and the first calls after the wake-up (the window's first launches) run several times slower
    while not warm1.query():
        pass
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # The first timed launch goes out before anything else; the event pair that measures the launch period on the GPU
    # brackets launches 2..K (K > 1): an event record ahead of the first launch would only delay it.
    first_outside = steps > 1
    if first_outside:
        roll.step(timed=True)
        if gather is not None:
            gather.after_step()
        n_warm_resets = len(roll.reset_events)  # a reset issued with the first step lies outside the event pair
    ev0.record()
    for _ in range(steps - (1 if first_outside else 0)):
        roll.step(timed=True)
        if gather is not None:
            gather.after_step()
    ev1.record()
    if close_flag is not None:  # (behind the last launch, in stream order: a 4-byte copy into pinned host memory)
        close_flag.copy_(close_one, non_blocking=True)
    t_issue = time.perf_counter() - t0
    if gather is not None:
        gather.finish()
    # the blocking wait inside synchronize() wakes up ~0.1-0.2 ms after the GPU is done (interrupt path): poll first, so
    # that a 20-step window is not dominated by the wake-up latency of its closing bracket -- a word of pinned host memory
    # the stream writes behind its last launch (a plain memory read per poll; AIE_BENCH_CLOSE=event: hipEventQuery polls,
    # round 4's way)
    if close_flag is not None:
        while int(close_flag[0]) == 0:
            pass
    else:
        while not ev1.query():
            pass
    torch.cuda.synchronize()
    barrier()
    elapsed_local = time.perf_counter() - t0
    gc.enable()
    elapsed = elapsed_local
    per_rank = [elapsed_local]
    if world > 1:
        tt = torch.tensor([elapsed_local], dtype=torch.float64, device="cpu" if args.oversubscribe_one_gpu else device)
        allt = [torch.zeros_like(tt) for _ in range(world)]
        torch.distributed.all_gather(allt, tt)
        per_rank = [float(x.item()) for x in allt]
        elapsed = max(per_rank)

    out = None
    region_ms = ev0.elapsed_time(ev1)
    timed_resets = roll.reset_events[n_warm_resets:]
    reset_ms = sum(a.elapsed_time(b) for a, b in timed_resets)
    launches_in_region = steps - (1 if first_outside else 0)
    avg_ms = (region_ms - reset_ms) / launches_in_region
    per_rank_launch_ms = [avg_ms]
    if world > 1:  # every rank's own kernel time (HIP events on its launch stream): separates it from the gather's wait
        tt = torch.tensor([avg_ms], dtype=torch.float64, device="cpu" if args.oversubscribe_one_gpu else device)
        allt = [torch.zeros_like(tt) for _ in range(world)]
        torch.distributed.all_gather(allt, tt)
        per_rank_launch_ms = [float(x.item()) for x in allt]
    if rank == 0:
        lay = layout_bytes_per_env_step(be, wl)
        units = (n + 1) if wl.startswith("C4") else n  # C4's per-unit figure counts the planner
        survey_per_launch = W["survey_bytes"] * units * E
        layout_per_launch = lay["total"] * E
        achieved = survey_per_launch / (avg_ms * 1e-3) / 1e9
        achieved_layout = layout_per_launch / (avg_ms * 1e-3) / 1e9
        traffic, traffic_src = measured_traffic(wl, E)
        inst = int(be.lib.aie_step_kernel_instance(be.handle))
        kernel_name = W["kernel"]  # the name rocprofv3 lists: compile-time instances are template instantiations
        if inst == 1000:
            kernel_name = "aie_jit_step"  # run-time specialisation (aie_specialize)
        elif inst >= 0 and not wl.startswith("C4"):
            kernel_name = "%s_spec<%d>" % (W["kernel"], inst)
        elif wl.startswith("C4"):
            import numpy as np

            # window sums: two launches per step (the step, then the change-event lists' upkeep and the next step's sums)
            taps = np.asarray(env.model["unemp_conv_filters"], np.float64)
            tap_t = "float" if np.array_equal(taps.astype(np.float32).astype(np.float64), taps) else "double"
            nf = env.model["num_filters"]
            kernel_name = ("aie_covid_step_kernel<%d, false> + aie_covid_window_kernel<%d, %s>" % (nf, nf, tap_t)
                           if env.exact_filter_sums else "aie_covid_step_kernel<%d, true>" % nf)
        traffic_frac = (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None
        insts, insts_src, valu_insts = issue_counters(wl)
        issue_frac = (insts / (N_SIMDS * SM_CLOCK_HZ * avg_ms * 1e-3)) if (insts and E == W["envs"]) else None
        # a wave64 vector instruction keeps its SIMD's 16-lane pipe for 4 clocks: the share of all VALU pipe time the
        # kernel's vector instructions fill
        valu_frac = (4.0 * valu_insts / (N_SIMDS * SM_CLOCK_HZ * avg_ms * 1e-3)) if (valu_insts and E == W["envs"]) else None
        store_roof = store_roof_gbs(device) if wl == "C5" else None
        store_frac = None
        if store_roof and traffic:
            store_frac = (traffic / (avg_ms * 1e-3) / 1e9) / store_roof
        # what the counters say binds the launch: the memory system (measured traffic against the box's own store roof
        # / the HBM peak), instruction issue, or neither (dependent chains of too few resident waves)
        if (store_frac or 0) >= 0.7 or (traffic_frac or 0) >= 0.7:
            bound = "hbm"
        elif (valu_frac or 0) >= 0.5:
            bound = "valu"
        elif (issue_frac or 0) >= 0.6:
            bound = "issue"
        else:
            bound = "latency"
        arena = be.arena_info()
        roof = dict(
            arena_bytes=arena["bytes"], fits_infinity_cache=arena["bytes"] <= INFINITY_CACHE_BYTES,
            arena_allocator=arena["allocator"], arena_piece_mib=arena["piece_mib"],
            bound=bound, roof="hbm", kernel=kernel_name, achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
            frac=achieved / HBM_PEAK_GBS, traffic=traffic, traffic_source=traffic_src, hbm_traffic_frac=traffic_frac,
            frac_moved=traffic_frac,  # = hbm_traffic_frac: MEASURED bytes / time / peak (what `frac` > 1 has to be read against)
            traffic_round=summary_provenance(traffic_src)[0], traffic_stale=summary_provenance(traffic_src)[1],
            counters_round=summary_provenance(insts_src)[0], counters_stale=summary_provenance(insts_src)[1],
            issue_frac=issue_frac, wave_instructions_per_launch=insts, issue_source=insts_src,
            valu_frac=valu_frac, valu_instructions_per_launch=valu_insts,
            valu_roof="a wave64 VALU instruction occupies its SIMD for 4 clocks: %d SIMDs x %.1f GHz / 4" % (N_SIMDS, SM_CLOCK_HZ / 1e9),
            issue_roof="1 wave-instruction per SIMD and clock: %d SIMDs x %.1f GHz" % (N_SIMDS, SM_CLOCK_HZ / 1e9),
            algorithmic_bytes_per_launch=survey_per_launch, algorithmic_bytes_per_unit=W["survey_bytes"],
            unit_of_work="agent-step incl. planner" if wl.startswith("C4") else "agent-step",
            achieved_final_layout=achieved_layout, frac_final_layout=achieved_layout / HBM_PEAK_GBS,
            final_layout_bytes_per_launch=layout_per_launch, final_layout_bytes_per_env_step=lay,
            avg_launch_ms=avg_ms, launches_timed=launches_in_region, reset_launches_in_region=len(timed_resets),
            reset_ms_in_region=reset_ms,
            note="achieved/frac price SURVEY.md 8(d)'s algorithmic bytes per launch against the HBM peak; "
                 "*_final_layout does the same with the bytes of the layouts actually used; hbm_traffic_frac = measured "
                 "HBM bytes (committed rocprofv3 FETCH_SIZE/WRITE_SIZE summary of this command) / live launch time / "
                 "peak; issue_frac = wave-instructions per launch (committed SQ_INSTS_* summary) / (SIMDs x clock x live "
                 "launch time); valu_frac = 4 clocks x vector instructions per launch (same summary) / (SIMDs x clock x live "
                 "launch time); `bound` is derived from those fractions, not asserted")
        if store_roof:
            roof["store_roof_GBps_this_box"] = store_roof
            roof["traffic_frac_of_store_roof"] = store_frac
            roof["store_roof_note"] = ("pure-store roof measured live on this box (4 GiB device fill); "
                                       "profiles/r03_store_roof.json has the per-pattern roofs (tools/store_roof.hip)")
        agent_steps = world * E * n * steps
        out = {
            "metric": "agent-steps/sec, %s" % {"C1": "simple_wood_and_stone 15x15 4-agent Gather+Build batched envs",
                                                "C2": "gather-trade-build 25x25 4-agent batched envs",
                                                "C2v": "gather-trade-build 25x25 4-agent batched envs, other scalars",
                                                "C2f": "gather-trade-build 25x25 4-agent batched envs, counter-based RNG (not NumPy's stream)",
                                                "C3f": "gather-trade-build 25x25 10-agent batched envs, counter-based RNG (not NumPy's stream)",
                                                "C1f": "simple_wood_and_stone 15x15 4-agent Gather+Build batched envs, counter-based RNG (not NumPy's stream)",
                                                "P2": "gather-trade-build 25x25 4-agent batched envs, phase-2 YAML",
                                                "C3": "gather-trade-build 25x25 10-agent batched envs",
                                                "C4": "covid19_env 51 US-state agents + planner",
                                                "C4x": "covid19_env 51 US-state agents + planner",
                                                "C4xu": "covid19_env 51 US-state agents + planner, unmasked policy",
                                                "C2@16384": "gather-trade-build 25x25 4-agent batched envs, 16384 replicas",
                                                "C2@65536": "gather-trade-build 25x25 4-agent batched envs, 65536 replicas",
                                                "C5": "one_step_economy 100 agents + SimpleLabor + planner tax"}[wl],
            "value": agent_steps / elapsed, "unit": "agent-steps/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32 state, f64 filter bank (f32 observations)" if wl.startswith("C4") else
                      "u8/i32 state + f64 coin/utility (f32 observations)"),
            "data": "synthetic",
            "config": {
                "workload": "%s: %s; uniform random policy; mobile agents counted (planner excluded)" % (wl, W["desc"]),
                "workload_short": "%s %s, %s random policy, planner not counted" % (wl, W["short"], "mask-respecting" if roll.masked else "uniform"),
                "parallelism_short": (("replica sharding x%d on ONE GPU, gloo host-staged (reward,done) gather to rank 0"
                                       if args.oversubscribe_one_gpu else
                                       "replica sharding x%d, RCCL (reward,done) gather to rank 0") % world) if world > 1
                else "single GPU",
                "oversubscribed_one_gpu": bool(args.oversubscribe_one_gpu) or None,
                "envs_per_gpu": E, "global_envs": world * E, "n_agents": n,
                "policy_short": "masked" if roll.masked else "unmasked",
                "rng": ("per-replica counter-based stream, Philox2x32-10 (rng_mode='fast': NOT NumPy's stream; bit-exact "
                        "against the oracle's restatement of the same generator)" if W.get("rng") == "fast" else
                        "per-replica NumPy-legacy MT19937 (parity mode)"),
                "policy": ("uniform random over the actions the masks allow (counter RNG)" if roll.masked else
                           "uniform random (counter RNG)") + ("; the draw for step t+1 happens inside the launch of "
                                                             "step t (aie_step_sample_next), one launch per step"
                                                             if roll.fused else ""),
                "phasing": ("replicas de-phased in %d blocks, episode ends %d steps apart (prologue of %d untimed "
                            "steps); one masked reset launch per block end inside the timed region"
                            % (roll.G, roll.stride, prologue_steps)) if roll.G > 1 else
                           ("lock-step replicas; auto-reset: a replica restarts inside the step launch that ends its "
                            "episode, the terminal observations are replaced by the next episode's first ones"
                            if roll.auto_reset else "lock-step replicas, one reset launch per episode end"),
                "parallelism": ("replica sharding, %d ranks; (reward, done) of every step gathered to rank 0 over "
                                "RCCL, 64 steps per collective, overlapped with the steps" % world)
                if world > 1 else "single GPU",
                "dev_switches": [],
                "kernel_specialisation": ("run time (aie_specialize)" if inst == 1000 else
                                          "compile time" if inst >= 0 else "none (generic kernel)"),
            },
            "per_rank_seconds": per_rank, "per_rank_avg_launch_ms": per_rank_launch_ms,
            "host_issue_seconds": t_issue, "gpu_region_seconds": region_ms * 1e-3,
            "exchange_ok": (gather is not None) if (world > 1 or args.force_gather) else None,
            "roofline": roof,
        }
        if gather is not None:
            out["gather"] = {"collectives": gather.n_collectives, "bytes_per_collective": gather.bytes_per_collective,
                             "wait_seconds": gather.wait_seconds}
        out["_cfg"] = cfg
    del roll, gather, be, env
    gc.collect()
    torch.cuda.empty_cache()
    return out


# ---- the line the driver parses -----------------------------------------------------------------------------------
# The driver keeps an 8 KB tail of stdout: the LAST line has to be short.  Everything else (prose notes, per-workload
# configs, the layout byte breakdown, per-workload CPU baselines) goes to bench_detail.json and to an earlier line.
LINE_LIMIT = 4096
# frac = SURVEY 8(d)'s algorithmic bytes / launch time / peak; frac_final_layout = the same with the bytes of the layouts
# actually used (COVID: SURVEY's figure assumes the 601-day window is streamed every step, which neither kernel does any
# more -- the sums over the window are the same float64, the bytes are not)
ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_final_layout", "traffic", "frac_moved",
             "valu_frac", "avg_launch_ms", "algorithmic_bytes_per_launch", "arena_bytes", "fits_infinity_cache",
             "traffic_round", "traffic_stale", "counters_round", "counters_stale")
# (side entries: no kernel names -- the detail file has them -- so that thirteen of them fit the 4 KB line)
SIDE_KEYS = ("bound", "value", "avg_launch_ms", "frac", "frac_moved", "valu_frac")  # (ms_per_step, frac_final_layout: detail file)


def _sig(x, digits=5):
    """Numbers rounded to `digits` significant figures (the detail file keeps full precision)."""
    if isinstance(x, float):
        if x.is_integer() and abs(x) < 2 ** 53:  # byte counts stay exact
            return int(x)
        return float("%.*g" % (digits, x))
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def compact_line(out):
    """The <= 4 KB driver line from the full result: the contract fields, the short config, the roofline numbers,
    cpu_baseline {value, unit, cores, kind, sample}, one flat object per side workload.  No prose."""
    cfg = out["config"]
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {"workload": cfg["workload_short"], "envs_per_gpu": cfg["envs_per_gpu"],
                      "global_envs": cfg["global_envs"], "n_agents": cfg["n_agents"],
                      "kernel_specialisation": cfg["kernel_specialisation"], "parallelism": cfg["parallelism_short"]}
    if cfg.get("oversubscribed_one_gpu"):
        line["config"]["oversubscribed_one_gpu"] = True  # (N ranks on one device: the N > 1 path, not a scaling figure)
    line["roofline"] = {k: out["roofline"].get(k) for k in ROOF_KEYS}
    line["gpu_region_seconds"] = out["gpu_region_seconds"]
    if out["n_gpus"] > 1 or out.get("gather"):
        line["per_rank_seconds"] = out["per_rank_seconds"]
        # kernel time per rank beside the wall time per rank: what separates the step launches from the gather's wait
        line["per_rank_avg_launch_ms"] = out.get("per_rank_avg_launch_ms")
        line["gather"] = out.get("gather")
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind")}
        line["cpu_baseline"]["sample"] = cb.get("sample_short") or cb.get("sample", "")[:160]
    if out.get("cpu_port"):
        line["cpu_port"] = {k: out["cpu_port"].get(k) for k in ("value", "cores", "kind")}
    if out.get("workloads"):
        sides = {}
        for name, r in out["workloads"].items():
            if "error" in r:
                sides[name] = {"error": r["error"][:80]}
                continue
            if r.get("compact"):  # workloads with their own shape (C2pi: a policy in the loop)
                sides[name] = dict(r["compact"])
                continue
            flat = dict(r.get("roofline") or {}, **{k: r.get(k) for k in ("value", "ms_per_step", "gpu_region_seconds")})
            sides[name] = {k: flat.get(k) for k in SIDE_KEYS if flat.get(k) is not None}
            if flat.get("counters_stale") or flat.get("traffic_stale"):
                # a counter-derived field of this entry (valu_frac / bound, frac_moved) comes from a committed summary that
                # was collected on other kernels than this tree's; which round's: the detail file
                sides[name]["stale"] = True
            if name.startswith("C4"):
                sides[name]["policy"] = (r.get("config") or {}).get("policy_short")
            if not flat.get("fits_infinity_cache", True):
                sides[name]["arena_bytes"] = flat.get("arena_bytes")
            if name == "C5":
                sides[name].update({k: flat.get(k) for k in ("arena_allocator", "arena_piece_mib")})
                if r.get("fresh_process"):
                    sides[name]["fresh_ms"] = r["fresh_process"].get("avg_launch_ms")
            if r.get("cpu_baseline"):
                sides[name]["cpu_ref"] = r["cpu_baseline"]["value"]
                if name.startswith("C4"):
                    sides[name]["cpu_policy"] = r["cpu_baseline"].get("policy")
        line["workloads"] = sides
    line["detail"] = "bench_detail.json"
    line = _sig(line)
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_LIMIT:  # cannot happen with the key sets above (tests/test_bench_accounting.py); never truncate JSON
        for name in list(line.get("workloads", {})):
            line["workloads"][name] = {k: line["workloads"][name].get(k) for k in ("value", "avg_launch_ms", "frac")}
        text = json.dumps(line, separators=(",", ":"))
    assert len(text) <= LINE_LIMIT, len(text)
    return text


# the other BASELINE configurations a default run also times, in short windows: (workload, steps, warm-up)
# (C5: the first ~10 launches over its 7 GB arena run 15 % slower than the steady state -- 1.68 ms per launch measured
# with 6 warm-up launches, 1.447 ms with 10, 30 or 60 on the same box)
SIDE_CPU_BASELINES = ("C1", "C3", "C4x", "C4xu", "C5")  # (C2v, P2 and C4 step the same reference code as C2 and C4x)
SIDE_CPU_SECONDS = 3.0


def emit(out, detail_file):
    """Full result -> bench_detail.json + an earlier stdout line; the compact line LAST."""
    detail = json.dumps(out)
    try:
        with open(detail_file, "w") as f:
            f.write(detail + "\n")
    except OSError:
        pass
    print(detail, flush=True)
    print(compact_line(out), flush=True)


SIDE_WORKLOADS = [("C1", 200, 20), ("C1f", 200, 20), ("C2v", 200, 20), ("C2f", 200, 20), ("P2", 200, 20), ("C3", 200, 20), ("C3f", 200, 20),
                  ("C4", 200, 20), ("C4x", 100, 10),
                  ("C4xu", 100, 10), ("C2@16384", 100, 20), ("C2@65536", 60, 10), ("C5", 60, 20)]


def run_policy_workload(args, steps, warmup, device):
    """C2pi: BASELINE configs[1] with a POLICY in the loop: a small torch MLP on `obs_a_flat` /
    `obs_p_flat`, `action_mask` applied to its logits, Gumbel-max sampling, actions written on the device, aie_step --
    issued call by call from Python (eager) and as one captured hipGraph replay per step
    (ai_economist_amd/rollout.py).  Everything else in this file fuses the random draw into the step kernel; this is
    what a trainer's rollout worker sees."""
    import gc

    import torch

    from ai_economist_amd.rollout import GraphedStep, MaskedMLPPolicy

    E = 4096
    env = make_env(dict(C2_CFG), n_envs=E, device=device)
    env.seed(ENV_SEED)
    env.reset()
    be = env.backend
    n = env.n_agents
    pol = MaskedMLPPolicy(be, hidden=128)
    gs = GraphedStep(env, pol, auto_reset=True, warmup=3)
    # the policy alone, captured the same way: what share of a replayed step is the policy's
    g_pol = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_pol):
        pol(be.tensors, gs.actions_a, gs.actions_p)

    def timed(fn):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        gc.collect()
        gc.disable()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(steps):
            fn()
        ev1.record()
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        gc.enable()
        return dt, t_issue, ev0.elapsed_time(ev1) * 1e-3

    dt_e, iss_e, gpu_e = timed(lambda: gs.eager(1))
    dt_g, iss_g, gpu_g = timed(lambda: gs.replay(1))
    dt_p, _, gpu_p = timed(g_pol.replay)
    assert int(be.tensors["error_flags"].abs().sum()) == 0  # the policy honoured the masks
    out = {
        "metric": "agent-steps/sec, gather-trade-build 25x25 4-agent batched envs, torch MLP policy in the loop",
        "value": E * n * steps / dt_g, "unit": "agent-steps/s", "steps": steps, "warmup": warmup,
        "ms_per_step": dt_g / steps * 1e3, "host_issue_seconds": iss_g, "gpu_region_seconds": gpu_g,
        "eager": {"value": E * n * steps / dt_e, "ms_per_step": dt_e / steps * 1e3, "host_issue_seconds": iss_e,
                  "gpu_region_seconds": gpu_e},
        "policy_only_ms_per_step": gpu_p / steps * 1e3,
        "config": {"workload": "C2pi: BASELINE configs[1], 4096 replicas, lock-step with auto-reset; policy = 3-layer MLP "
                               "(hidden 128, fp32, random init) on obs_a_flat [E,4,%d] and obs_p_flat [E,%d], action_mask "
                               "applied to the logits, Gumbel-max sampling on the device; one hipGraph replay per step "
                               "(policy + aie_step + auto-reset launch) vs the same calls issued from Python"
                               % (be.tensors["obs_a_flat"].shape[-1], be.tensors["obs_p_flat"].shape[-1]),
                   "envs_per_gpu": E, "n_agents": n},
    }
    out["compact"] = _sig({"value": out["value"], "ms_per_step": out["ms_per_step"], "host_issue_s": iss_g,
                           "eager_value": out["eager"]["value"], "eager_ms_per_step": out["eager"]["ms_per_step"],
                           "eager_host_issue_s": iss_e, "policy_ms": out["policy_only_ms_per_step"], "steps": steps})
    del gs, g_pol, pol, be, env
    gc.collect()
    torch.cuda.empty_cache()
    return out


def c5_fresh_process(trials=3, piece_mb=None):
    """C5's launch time in FRESH processes (its 7 GB arena lands on the memory channels differently from process to
    process and box to box: the figure of the long-lived bench process alone says little): `trials` sub-processes of 60
    timed launches each; min / median of their avg_launch_ms."""
    res = []
    env = dict(os.environ)
    if piece_mb is not None:
        env["AIE_ARENA_PIECE_MB"] = str(piece_mb)
    for _ in range(trials):
        try:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", "C5", "--steps", "60", "--warmup",
                                  "20", "--no-cpu-baseline", "--no-workloads", "--detail-file", os.devnull],
                                 capture_output=True, text=True, timeout=300, env=env).stdout
            lines = out.strip().splitlines()
            d = json.loads(lines[-2])  # (the full result; the compact driver line follows it)
            res.append((d["roofline"]["avg_launch_ms"], d["roofline"].get("arena_piece_mib")))
        except Exception:
            pass
    if not res:
        return None
    res.sort()
    # piece_mib: what each process's arena was mapped with -- forced (piece_mb) or what the library's store probe chose
    # at aie_create (csrc/aie_capi.hip: aie_arena_alloc)
    return {"avg_launch_ms": res[len(res) // 2][0], "min": res[0][0], "max": res[-1][0], "trials": len(res),
            "piece_mib": [r[1] for r in res], "piece_forced": piece_mb is not None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", choices=sorted(WORKLOADS) + ["C2pi"], default="C2",
                    help="C2pi: configs[1] with a torch MLP policy in the loop, alone (one JSON line; no CPU legs)")
    ap.add_argument("--envs-per-gpu", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-episodes", type=int, default=0,
                    help="time the headline's cpu_baseline on SURVEY 8(d)'s protocol -- 1 warm-up + N timed WHOLE episodes incl. "
                         "resets per process -- instead of the default free-running 10 s window (3 = the survey's figure; takes "
                         "~4 x episode_length x 0.5 ms per process)")
    ap.add_argument("--no-workloads", action="store_true",
                    help="headline workload only (default: a 1-GPU C2 run also times C1, C3, C4, C4x, C5 in short windows)")
    ap.add_argument("--generic-kernel", action="store_true",
                    help="development: time the generic step kernel instead of the configuration's compile-time instance")
    ap.add_argument("--no-stagger", action="store_true", help="keep all replicas in lock-step (round-1 behaviour)")
    ap.add_argument("--no-auto-reset", action="store_true",
                    help="C5: separate reset launches instead of restarting replicas inside the step launch")
    ap.add_argument("--covid-unmasked-policy", action="store_true",
                    help="C4 / C4x: the uniform policy that ignores the action masks (round 3's; stringency levels then change "
                         "on most days, the change-event lists overflow and the replicas stream their whole window)")
    ap.add_argument("--c5-piece-probe", action="store_true",
                    help="C5: also time fresh processes with forced 16 / 64 / 128 MiB arena pieces (development)")
    ap.add_argument("--force-gather", action="store_true",
                    help="run the N > 1 reward-log gather in a 1-rank group (exercises the RCCL path on one GPU)")
    ap.add_argument("--detail-file", default=os.path.join(ROOT, "bench_detail.json"),
                    help="where the full (uncompacted) result goes; the last stdout line is the <= 4 KB driver line")
    ap.add_argument("--oversubscribe-one-gpu", action="store_true",
                    help="development / tests: all N ranks share HIP device 0 (a gloo group and the host-staged reward-log "
                         "gather: RCCL refuses two ranks on one device) -- the N > 1 path with real kernels on a 1-GPU box; "
                         "the line says so (config.oversubscribed_one_gpu) and is no scaling measurement")
    ap.add_argument("--launcher", choices=["auto", "torchrun"], default="auto",
                    help="torchrun: re-execute under torch.distributed.run even with --gpus 1 (the N > 1 launch path on one GPU)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.launcher == "torchrun"):
        if args.gpus > 1:  # a clear error instead of N ranks fighting over fewer devices
            import torch

            have = torch.cuda.device_count()
            if args.gpus > have and not args.oversubscribe_one_gpu:
                sys.exit("bench.py: --gpus %d, but only %d HIP device(s) are visible on this node" % (args.gpus, have))
        sys.exit(self_launch(args))

    import torch

    from ai_economist_amd.sharding import dist_info

    dev_env = sorted(k for k in os.environ if k.startswith("AIE_DEV_"))
    if dev_env:
        sys.exit("bench.py refuses to run with development switches set: %s" % dev_env)

    rank, local_rank, world = dist_info()
    if args.oversubscribe_one_gpu:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        sys.exit("bench.py: rank %d wants HIP device %d, but only %d are visible" % (rank, local_rank, torch.cuda.device_count()))
    if args.gpus > 1 or world > 1 or args.force_gather or args.launcher == "torchrun":
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29513")
        if args.oversubscribe_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        assert world == max(1, args.gpus), "world size %d != --gpus %d" % (world, args.gpus)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    if args.workload == "C2pi":  # the policy-in-the-loop side workload on its own (profiling)
        r = run_policy_workload(args, min(args.steps, 400), min(args.warmup, 40), device)
        r.pop("compact", None)
        print(json.dumps(r), flush=True)
        return
    out = run_workload(args.workload, args, args.steps, args.warmup, rank, local_rank, world, device)
    if world > 1:
        # every rank empties its native stdio buffers (RCCL's version banner) before rank 0 prints the JSON line, so
        # that nothing follows it on the job's stdout
        try:
            import ctypes

            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        torch.distributed.barrier()

    if rank == 0:
        cfg = out.pop("_cfg")
        wl = args.workload
        side_cfgs = {}
        if world == 1 and wl == "C2" and not args.no_workloads and not args.envs_per_gpu:
            # every BASELINE configuration under the same clock, in the same invocation (short windows)
            sides = {}
            for swl, ssteps, swarm in SIDE_WORKLOADS:
                try:
                    r = run_workload(swl, args, ssteps, swarm, rank, local_rank, world, device)
                    side_cfgs[swl] = r.pop("_cfg", None)
                    sides[swl] = {k: r[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype",
                                                    "config", "roofline", "gpu_region_seconds")}
                except Exception as exc:  # a side line must not take the headline down
                    sides[swl] = {"error": repr(exc)}
            try:
                sides["C2pi"] = run_policy_workload(args, 200, 20, device)
            except Exception as exc:
                sides["C2pi"] = {"error": repr(exc)}
            if "error" not in sides.get("C5", {"error": 1}):
                # the same C5 window in two fresh processes (where a process's 7 GB arena lands differs from process to
                # process and box to box; the arena's piece size is NOT what separates fast from slow boxes: round 5,
                # tools/c5_piece_experiment.sh -- `--c5-piece-probe` repeats that experiment here)
                fresh = c5_fresh_process(2)
                if fresh:
                    sides["C5"]["fresh_process"] = fresh
                    if args.c5_piece_probe:
                        alts = {mb: c5_fresh_process(1, piece_mb=mb) for mb in (16, 64, 128)}
                        sides["C5"]["fresh_process_other_piece_sizes"] = alts
            out["workloads"] = sides
        if not args.no_cpu_baseline and world == 1:
            ref = None
            try:
                ref = cpu_reference_baseline(cfg, episodes=args.cpu_baseline_episodes)
            except Exception as exc:  # the reference leg must not take the GPU line down
                out["cpu_baseline_error"] = repr(exc)
            if wl in ("C1", "C2", "C3"):
                port = cpu_port_baseline(cfg)
                if ref is not None:
                    out["cpu_baseline"], out["cpu_port"] = ref, port
                else:
                    out["cpu_baseline"] = port
            elif ref is not None:
                out["cpu_baseline"] = ref
            # the reference's CPU step beside EVERY configuration of the run: a short window each
            for swl in SIDE_CPU_BASELINES:
                if swl in side_cfgs and "error" not in out["workloads"][swl]:
                    try:
                        r = cpu_reference_baseline(side_cfgs[swl], seconds=SIDE_CPU_SECONDS,
                                                   policy=out["workloads"][swl]["config"].get("policy_short", "unmasked")
                                                   if swl.startswith("C4") else "unmasked")
                        if r is not None:
                            out["workloads"][swl]["cpu_baseline"] = r
                    except Exception as exc:
                        out["workloads"][swl]["cpu_baseline_error"] = repr(exc)
        try:  # whatever native libraries (RCCL's version banner) left in the C stdio buffer goes out first, so that
            import ctypes  # the JSON line is the last line on stdout

            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        emit(out, args.detail_file)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
