"""Configurations that get a COMPILE-TIME instance of the step kernel (csrc/aie_spec_generated.h).

For each entry the host mirror builds the `aie_config`, the build-time tool csrc/aie_specgen.c derives the parameter
block from it (the same aie_build_params the library runs at aie_create) and normalises away what depends on the
batch; the image is baked into `aie_step_kernel_spec<K>`.  At aie_create an environment whose normalised block
equals an image byte for byte runs on that instance -- any replica count, any seed, and any value of the scalars the
normalisation blanks (an image stands for a family, csrc/aie_layout.h: aie_spec_normalize) -- everything else runs on
the generic kernel until its run-time specialisation is ready (aie_specialize).  The list is the BASELINE component tuple (Build, ContinuousDoubleAuction, Gather,
PeriodicBracketTax on the 25x25 quadrant layout) at the agent counts of BASELINE configs[1] and configs[2], plus the
phase-2 setting of the reference's tutorials (planner without spatial observations).
"""
import ctypes
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
OUT = os.path.join(CSRC, "aie_spec_generated.h")

GTB = [("Build", {}), ("ContinuousDoubleAuction", {"max_num_orders": 5}), ("Gather", {}), ("PeriodicBracketTax", {})]
_BASE = dict(world_size=[25, 25], episode_length=1000, components=GTB, starting_agent_coin=10,
             env_layout_file="quadrant_25x25_20each_30clump.txt")
_OSE = dict(n_agents=100, world_size=[1, 1], episode_length=2,
            components=[("SimpleLabor", {}), ("PeriodicBracketTax", {"bracket_spacing": "us-federal", "period": 1,
                                                                     "tax_model": "model_wrapper"})])
# the env blocks of the reference's two training YAMLs (tutorials/rllib/phase1|phase2/config.yaml): phase 1 = free
# market (taxes disabled, energy warm-up "auto"), phase 2 = the planner sets the tax rates (no maps for the planner,
# annealed rate cap); both with the four fixed skills / start locations.  Their scalars are the family's run-time part.
def _phase(disable_taxes, **extra):
    tax = dict(bracket_spacing="us-federal", disable_taxes=disable_taxes, period=100, tax_annealing_schedule=[-100, 0.001],
               usd_scaling=1000)
    if not disable_taxes:
        tax.update(rate_disc=0.05, tax_model="model_wrapper")
    return dict(n_agents=4, world_size=[25, 25], episode_length=1000, env_layout_file="quadrant_25x25_20each_30clump.txt",
                components=[("Build", dict(build_labor=10, payment=10, payment_max_skill_multiplier=3, skill_dist="pareto")),
                            ("ContinuousDoubleAuction", dict(max_bid_ask=10, max_num_orders=5, order_duration=50, order_labor=0.25)),
                            ("Gather", dict(collect_labor=1, move_labor=1, skill_dist="pareto")),
                            ("PeriodicBracketTax", tax)],
                energy_cost=0.21, fixed_four_skill_and_loc=True, flatten_masks=True, flatten_observations=True,
                isoelastic_eta=0.23, multi_action_mode_agents=False, multi_action_mode_planner=True,
                planner_gets_spatial_info=False, planner_reward_type="coin_eq_times_productivity", starting_agent_coin=0,
                **extra)


PHASE1 = _phase(True, energy_warmup_constant=10000, energy_warmup_method="auto")
PHASE2 = _phase(False, energy_warmup_constant=0)

# (name, scenario, kwargs, kernel family, waves per SIMD): "gtb" -> aie_step_kernel_spec<K>, "ose" ->
# aie_ose_step_kernel_spec<K>.  Waves per SIMD = what the instance's LDS footprint lets a CU hold (two waves per
# workgroup, four SIMDs per CU, 160 KB of LDS in 1280-byte granules, at most 16 workgroups): C2's 6 736 B and C3's
# 9 696 B both allow 16 workgroups -> 8 waves per SIMD, 64 VGPRs (tests/test_gpu_parity.py checks the figure against
# the LDS size the library reports); the one-step-economy instance (one wave per workgroup): 9.9 KB of LDS -> 16
# workgroups per CU -> 4 waves per SIMD, 128 VGPRs.
SPECS = [
    ("C2: gather-trade-build 25x25, 4 agents (BASELINE configs[1])", "layout_from_file/simple_wood_and_stone",
     dict(_BASE, n_agents=4), "gtb", 8),
    ("C3: gather-trade-build 25x25, 10 agents (BASELINE configs[2])", "layout_from_file/simple_wood_and_stone",
     dict(_BASE, n_agents=10), "gtb", 8),
    # SimpleLabor's skills (a construction-time Monte-Carlo draw in the reference) are run-time data, not part of the image
    ("C5: one-step-economy, 100 agents (BASELINE configs[4])", "one-step-economy", _OSE, "ose", 4),
    ("C1: uniform 15x15, 4 agents, Build + Gather (BASELINE configs[0]'s scenario)", "uniform/simple_wood_and_stone",
     dict(n_agents=4, world_size=[15, 15], episode_length=1000, components=[("Build", {}), ("Gather", {})],
          starting_agent_coin=10, starting_stone_coverage=0.10, starting_wood_coverage=0.10), "gtb", 8),
    ("P2: the reference's phase-2 training YAML (tutorials/rllib/phase2/config.yaml)", "layout_from_file/simple_wood_and_stone",
     PHASE2, "gtb", 8),
    ("P1: the reference's phase-1 training YAML (tutorials/rllib/phase1/config.yaml)", "layout_from_file/simple_wood_and_stone",
     PHASE1, "gtb", 8),
    # the throughput mode (rng_mode="fast": a counter-based stream per replica instead of NumPy's MT19937, include/aie.h
    # AIE_RNG_FAST) of the two headline configurations -- never the headline itself
    ("C2f: C2 with rng_mode='fast'", "layout_from_file/simple_wood_and_stone", dict(_BASE, n_agents=4, rng_mode="fast"), "gtb", 8),
    ("C3f: C3 with rng_mode='fast'", "layout_from_file/simple_wood_and_stone", dict(_BASE, n_agents=10, rng_mode="fast"), "gtb", 8),
    # C1's scenario in the throughput mode: layouts from a stream of their own, drawn ahead of their resets
    ("C1f: C1 with rng_mode='fast'", "uniform/simple_wood_and_stone",
     dict(n_agents=4, world_size=[15, 15], episode_length=1000, components=[("Build", {}), ("Gather", {})],
          starting_agent_coin=10, starting_stone_coverage=0.10, starting_wood_coverage=0.10, rng_mode="fast"), "gtb", 8),
]


def _tool():
    exe = os.path.join(tempfile.gettempdir(), "aie_specgen_%d" % os.getuid())
    src = os.path.join(CSRC, "aie_specgen.c")
    deps = [src, os.path.join(CSRC, "aie_layout.h"), os.path.join(ROOT, "include", "aie.h")]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        tmp = "%s.%d" % (exe, os.getpid())  # never execute a half-written tool (several ranks may build at once)
        subprocess.run(["gcc", "-O1", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, src, "-o", tmp, "-lm"], check=True)
        os.replace(tmp, exe)
    return exe


def images():
    from . import foundation

    exe = _tool()
    out = []
    import numpy as np

    for name, scenario, kw, family, waves in SPECS:
        state = np.random.get_state()  # SimpleLabor draws its (irrelevant here) skills from the global stream
        env = foundation.make_env_instance(scenario, n_envs=1, **dict(kw, components=[tuple(c) for c in kw["components"]]))
        np.random.set_state(state)
        cfg = env.build_config()
        raw = bytes(ctypes.string_at(ctypes.addressof(cfg), ctypes.sizeof(cfg)))
        img = subprocess.run([exe], input=raw, check=True, capture_output=True).stdout
        out.append((name, img, family, waves))
    return out


def header_text():
    imgs = images()
    lines = ["/* GENERATED by ai-economist_amd/_specs.py (run by _build.build()): normalised aie_params images of the",
             "   configurations that get a compile-time instance of the step kernel.  Do not edit. */", "#pragma once",
             "#define AIE_N_SPECS %d" % len(imgs), "template <int K> struct aie_spec_image;"]
    for k, (name, img, _family, waves) in enumerate(imgs):
        lines.append("/* %d: %s */" % (k, name))
        lines.append("alignas(16) static constexpr unsigned char aie_spec_bytes_%d[%d] = {" % (k, len(img)))
        for off in range(0, len(img), 32):
            lines.append("  " + ",".join(str(b) for b in img[off:off + 32]) + ",")
        lines.append("};")
        lines.append("template <> struct aie_spec_image<%d> { static constexpr const unsigned char* bytes = aie_spec_bytes_%d; "
                     "static constexpr int waves = %d; };" % (k, k, waves))
    for fam in ("gtb", "ose"):
        lines.append("#define AIE_SPEC_LIST_%s(X) " % fam.upper() +
                     " ".join("X(%d)" % k for k, im in enumerate(imgs) if im[2] == fam))
    lines.append("static const unsigned char* const aie_spec_table[AIE_N_SPECS + 1] = {" +
                 ", ".join("aie_spec_bytes_%d" % k for k in range(len(imgs))) + (", " if imgs else "") + "nullptr};")
    lines.append("static const char* const aie_spec_names[AIE_N_SPECS + 1] = {" +
                 ", ".join('"%s"' % im[0] for im in imgs) + (", " if imgs else "") + "nullptr};")
    return "\n".join(lines) + "\n"


def write_header():
    """(Re)writes csrc/aie_spec_generated.h if its content changed; returns True if it did."""
    text = header_text()
    old = open(OUT).read() if os.path.exists(OUT) else None
    if old == text:
        return False
    tmp = "%s.%d" % (OUT, os.getpid())
    with open(tmp, "w") as f:
        f.write(text)
    os.replace(tmp, OUT)
    return True
