#!/usr/bin/env python
"""Generates tests/golden/*.npz by running the UNMODIFIED reference Foundation
(/root/reference, imported through oracle/ref_harness.py) under a fixed seed.

Run (in the build container only; the GPU box has no /root/reference):
    python oracle/gen_golden.py

Each fixture holds, for one env replica:
  cfg_json             the make_env_instance kwargs (scenario_name + kwargs)
  pre_reset_mt/pos     NumPy legacy MT19937 state injected before reset()
  s0_<field>           full state right after reset() (field names = include/aie.h)
  actions_a [T,n], actions_p [T,NB]
  st_<field> [T,...]   state after each step
  rew [T,n+1]          reference rewards (f64)
  obs_steps [K], ob_<tensor> [K,...]   observations at selected steps (0 = reset obs)
"""
import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_extract import extract_obs, extract_state, rewards_array  # noqa: E402
from ref_harness import load_reference_foundation  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def sample_actions(env, rng, t_steps, p_move=0.45, p_build=0.12, p_trade=0.33):
    """Biased random policy so that gathers / builds / trades actually happen."""
    n = env.n_agents
    ag = env.world.agents[0]
    names = ag._action_names
    # single-action index ranges per subspace
    ranges = {}
    base = 1
    for nm in names:
        d = ag.action_dim[nm]
        ranges[nm] = (base, base + d)
        base += d
    A = base
    acts = np.zeros((t_steps, n), np.int32)
    trade_names = [nm for nm in names if nm.startswith("ContinuousDoubleAuction")]
    if names == ["SimpleLabor"]:
        acts = rng.randint(0, A, size=(t_steps, n)).astype(np.int32)
    for t in range(t_steps if names != ["SimpleLabor"] else 0):
        for i in range(n):
            u = rng.rand()
            if u < p_move and "Gather" in ranges:
                lo, hi = ranges["Gather"]
                acts[t, i] = rng.randint(lo, hi)
            elif u < p_move + p_build and "Build" in ranges:
                acts[t, i] = ranges["Build"][0]
            elif u < p_move + p_build + p_trade and trade_names:
                nm = trade_names[rng.randint(0, len(trade_names))]
                lo, hi = ranges[nm]
                # low asks / high bids trade more often
                acts[t, i] = rng.randint(lo, hi)
            else:
                acts[t, i] = 0
    pl = env.world.planner
    nb = len(pl._action_names)
    if nb and pl._action_names[0] != "PassiveAgentPlaceholder":
        dims = [pl.action_dim[nm] for nm in pl._action_names]
        acts_p = np.stack([rng.randint(0, d, size=t_steps) for d in dims], axis=1).astype(np.int32)
    else:
        acts_p = np.zeros((t_steps, 0), np.int32)
    return acts, acts_p, A


def run_case(name, cfg, seed, t_steps, obs_steps, action_seed=123, action_kw=None,
             n_episodes=1):
    foundation = load_reference_foundation()
    kwargs = dict(cfg)
    scenario = kwargs.pop("scenario_name")
    kwargs["components"] = [tuple(c) for c in kwargs["components"]]
    out_seed = seed + 1000
    np.random.seed(out_seed)  # SimpleLabor / SplitLayout draw skills from the global stream at construction
    env = foundation.make_env_instance(scenario, **kwargs)
    np.random.seed(seed)
    st = np.random.get_state()
    out = {"cfg_json": np.array(json.dumps(cfg))}
    out["construction_seed"] = np.array(out_seed, np.int64)
    out["pre_reset_mt"] = np.array(st[1], np.uint32)
    out["pre_reset_pos"] = np.array(st[2], np.int32)
    obs = env.reset()
    s0 = extract_state(env)
    for k, v in s0.items():
        out["s0_" + k] = v
    rng = np.random.RandomState(action_seed)
    acts, acts_p, A = sample_actions(env, rng, t_steps, **(action_kw or {}))
    out["actions_a"] = acts
    out["actions_p"] = acts_p
    states = []
    rews = []
    dones = []
    obs_rec = {}
    kept = []

    def keep(t, o):
        kept.append(t)
        for k, v in extract_obs(env, o).items():
            obs_rec.setdefault(k, []).append(v)

    if 0 in obs_steps:
        keep(0, obs)
    reset_states = []
    for t in range(t_steps):
        ad = {str(i): int(acts[t, i]) for i in range(env.n_agents)}
        if acts_p.shape[1]:
            ad["p"] = [int(x) for x in acts_p[t]]
        obs, rew, done, _ = env.step(ad)
        states.append(extract_state(env))
        rews.append(rewards_array(env, rew))
        dones.append(done["__all__"])
        if (t + 1) in obs_steps:
            keep(t + 1, obs)
        if done["__all__"] and t + 1 < t_steps:
            # multi-episode fixtures: reset continues the same MT stream
            obs = env.reset()
            reset_states.append((t + 1, extract_state(env)))
    for k in states[0].keys():
        if k == "mt":
            # 2.5 KB of incompressible words per step: keep a CRC per step + the final state
            out["st_mt_crc"] = np.array(
                [zlib.crc32(s["mt"].tobytes()) for s in states], np.uint32)
            out["final_mt"] = states[-1]["mt"]
            continue
        out["st_" + k] = np.stack([s[k] for s in states])
    out["rew"] = np.stack(rews)
    out["done"] = np.array(dones, np.uint8)
    out["obs_steps"] = np.array(kept, np.int32)
    for k, v in obs_rec.items():
        out["ob_" + k] = np.stack(v)
    if reset_states:
        out["reset_at"] = np.array([t for t, _ in reset_states], np.int32)
        for k in reset_states[0][1].keys():
            if k == "mt":
                continue
            out["rs_" + k] = np.stack([s[k] for _, s in reset_states])
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    if scenario == "one-step-economy":
        print("%-28s %7.1f KB  A=%d" % (name, os.path.getsize(path) / 1024.0, A))
        return
    print("%-28s %7.1f KB  A=%d  gathers=%d builds=%d trades=%d" % (
        name, os.path.getsize(path) / 1024.0, A,
        sum(len(g) for g in env.get_component("Gather").gathers) if _has(env, "Gather") else -1,
        sum(len(b) for b in env.get_component("Build").builds) if _has(env, "Build") else -1,
        sum(len(x) for x in env.get_component("ContinuousDoubleAuction").executed_trades)
        if _has(env, "ContinuousDoubleAuction") else -1))


def _has(env, name):
    return name in [c.name for c in env.components]


GTB = [
    ["Build", {}],
    ["ContinuousDoubleAuction", {"max_num_orders": 5}],
    ["Gather", {}],
    ["PeriodicBracketTax", {}],
]

CASES = {
    # BASELINE configs[0] (C1): uniform 15x15, 4 agents, Build+Gather (mirrors the reference's
    # tests/test_env.py minus the auction); 2 short episodes => 2 random layouts
    "c1_uniform15_4ag": dict(
        cfg=dict(scenario_name="uniform/simple_wood_and_stone", n_agents=4, world_size=[15, 15],
                 episode_length=90, components=[["Build", {}], ["Gather", {}]], starting_agent_coin=10,
                 starting_stone_coverage=0.10, starting_wood_coverage=0.10),
        seed=17, t_steps=180, obs_steps=[0, 1, 45, 90, 91, 180],
        action_kw=dict(p_move=0.6, p_build=0.3, p_trade=0.0)),
    # BASELINE configs[4] (C5) at 1 replica: one-step-economy, 100 agents, 2-step episodes
    "c5_one_step_economy_100ag": dict(
        cfg=dict(scenario_name="one-step-economy", n_agents=100, world_size=[1, 1], episode_length=2,
                 components=[["SimpleLabor", {}],
                             ["PeriodicBracketTax", {"bracket_spacing": "us-federal", "period": 1,
                                                     "tax_model": "model_wrapper"}]]),
        seed=13, t_steps=8, obs_steps=[0, 1, 2, 3, 4, 8]),
    # BASELINE configs[1] (C2) at 1 replica: quadrant layout, 4 agents, Build+CDA+Gather+Tax
    "c2_quadrant_4ag": dict(
        cfg=dict(scenario_name="layout_from_file/simple_wood_and_stone", n_agents=4,
                 world_size=[25, 25], episode_length=1000, components=GTB,
                 starting_agent_coin=10,
                 env_layout_file="quadrant_25x25_20each_30clump.txt"),
        seed=1, t_steps=320, obs_steps=[0, 1, 2, 57, 100, 101, 200, 201, 320]),
    # denser layout => many gathers/builds/trades
    "c2_uniform65_4ag": dict(
        cfg=dict(scenario_name="layout_from_file/simple_wood_and_stone", n_agents=4,
                 world_size=[25, 25], episode_length=1000, components=GTB,
                 starting_agent_coin=10, resource_regen_prob=0.05,
                 env_layout_file="uniform_25x25_25each_65clump.txt"),
        seed=7, t_steps=320, obs_steps=[0, 1, 99, 100, 101, 250]),
    # BASELINE configs[2] (C3) at 1 replica: 10 agents
    "c3_quadrant_10ag": dict(
        cfg=dict(scenario_name="layout_from_file/simple_wood_and_stone", n_agents=10,
                 world_size=[25, 25], episode_length=1000, components=GTB,
                 starting_agent_coin=10,
                 env_layout_file="quadrant_25x25_20each_30clump.txt"),
        seed=3, t_steps=220, obs_steps=[0, 1, 100, 101, 220]),
    # BASELINE configs[0]-like on a fixed layout: Build+Gather only, 15x15
    "c1_puremixed15_4ag": dict(
        cfg=dict(scenario_name="layout_from_file/simple_wood_and_stone", n_agents=4,
                 world_size=[15, 15], episode_length=1000,
                 components=[["Build", {}], ["Gather", {}]], starting_agent_coin=10,
                 env_layout_file="env-pure_and_mixed-15x15.txt"),
        seed=11, t_steps=200, obs_steps=[0, 1, 50, 200],
        action_kw=dict(p_move=0.6, p_build=0.3, p_trade=0.0)),
    # skills: pareto build skill, lognormal gather bonus (rand() per pickup matters),
    # short episodes => done / reset / completions bookkeeping, planner obs off
    "skills_short_episodes": dict(
        cfg=dict(scenario_name="layout_from_file/simple_wood_and_stone", n_agents=4,
                 world_size=[25, 25], episode_length=60,
                 components=[["Build", {"skill_dist": "pareto",
                                        "payment_max_skill_multiplier": 3}],
                             ["ContinuousDoubleAuction", {"max_num_orders": 5,
                                                          "order_duration": 20}],
                             ["Gather", {"skill_dist": "lognormal"}],
                             ["PeriodicBracketTax", {"period": 25}]],
                 starting_agent_coin=20, planner_gets_spatial_info=False,
                 resource_regen_prob=0.1,
                 env_layout_file="uniform_25x25_25each_65clump.txt"),
        seed=21, t_steps=150, obs_steps=[0, 1, 25, 26, 60, 61, 150]),
    # the paper / phase-2 tutorial setting: fixed_four_skill_and_loc + pareto skills
    "fixed_four_skill": dict(
        cfg=dict(scenario_name="layout_from_file/simple_wood_and_stone", n_agents=4,
                 world_size=[25, 25], episode_length=1000,
                 components=[["Build", {"skill_dist": "pareto",
                                        "payment_max_skill_multiplier": 3}],
                             ["ContinuousDoubleAuction", {"max_num_orders": 5}],
                             ["Gather", {}],
                             ["PeriodicBracketTax", {"period": 100,
                                                     "bracket_spacing": "us-federal"}]],
                 starting_agent_coin=0, fixed_four_skill_and_loc=True,
                 planner_gets_spatial_info=False, isoelastic_eta=0.23,
                 energy_cost=0.21, energy_warmup_constant=10000,
                 energy_warmup_method="auto",
                 env_layout_file="quadrant_25x25_20each_30clump.txt"),
        seed=5, t_steps=1000, obs_steps=[0, 1, 100, 101, 150, 500, 1000]),  # a whole episode: auto_warmup over 1000 steps
    # dynamic-layout variants: a water cross with openings / randomly drawn resource zones,
    # 2 short episodes each => 2 generated layouts (np.random.shuffle / rand / randn draws)
    "quadrant_dyn15_4ag": dict(
        cfg=dict(scenario_name="quadrant/simple_wood_and_stone", n_agents=4, world_size=[15, 15],
                 episode_length=70, components=GTB, starting_agent_coin=10,
                 starting_stone_coverage=0.10, starting_wood_coverage=0.10),
        seed=23, t_steps=140, obs_steps=[0, 1, 35, 70, 71, 140]),
    "split_layout25_5ag": dict(
        cfg=dict(scenario_name="split_layout/simple_wood_and_stone", n_agents=5, world_size=[25, 25],
                 episode_length=50,
                 components=[["Build", {"skill_dist": "pareto", "payment_max_skill_multiplier": 3}],
                             ["ContinuousDoubleAuction", {"max_num_orders": 5}], ["Gather", {}],
                             ["PeriodicBracketTax", {"period": 20}]],
                 starting_agent_coin=10, water_row=11, skill_rank_of_top_agents=[0, 3],
                 env_layout_file="uniform_25x25_25each_65clump.txt"),
        seed=31, t_steps=100, obs_steps=[0, 1, 50, 51, 100]),
    # neighbourhood regeneration (regen_halfwidth 2 / 1: scipy convolve2d over the source blocks) and the
    # WealthRedistribution component in front of the tax component
    "quadrant_halfwidth_wealth_4ag": dict(
        cfg=dict(scenario_name="quadrant/simple_wood_and_stone", n_agents=4, world_size=[16, 16],
                 episode_length=60,
                 components=GTB[:3] + [["WealthRedistribution", {}], ["PeriodicBracketTax", {"period": 20}]],
                 starting_agent_coin=10, starting_stone_coverage=0.10, starting_wood_coverage=0.10,
                 wood_regen_halfwidth=2, wood_regen_weight=0.6, stone_regen_halfwidth=1, stone_regen_weight=0.4),
        seed=37, t_steps=120, obs_steps=[0, 1, 20, 60, 61, 120]),
    # tax_model "saez" while the sample buffer is short of its 500 samples: np.random.uniform bracket rates at
    # every period start (drawn between the step's other draws), (income, marginal rate) pairs collected
    "saez_random_rate_phase_5ag": dict(
        cfg=dict(scenario_name="layout_from_file/simple_wood_and_stone", n_agents=5, world_size=[25, 25],
                 episode_length=50,
                 components=GTB[:3] + [["PeriodicBracketTax", {"period": 5, "tax_model": "saez", "rate_max": 0.9}]],
                 starting_agent_coin=15, resource_regen_prob=0.08,
                 env_layout_file="uniform_25x25_25each_65clump.txt"),
        seed=41, t_steps=100, obs_steps=[0, 1, 5, 6, 50, 51, 100]),
    "multizone16_4ag": dict(
        cfg=dict(scenario_name="multi_zone/simple_wood_and_stone", n_agents=4, world_size=[16, 16],
                 episode_length=60, components=[["Build", {}], ["Gather", {}]], starting_agent_coin=10,
                 num_partitions_row=4, num_partitions_col=4, num_wood_zones=3, num_stone_zones=3,
                 num_wood_and_stone_zones=2, starting_stone_coverage=0.08, starting_wood_coverage=0.08),
        seed=29, t_steps=120, obs_steps=[0, 1, 60, 61, 120],
        action_kw=dict(p_move=0.6, p_build=0.3, p_trade=0.0)),
}


def main():
    only = sys.argv[1:]
    for name, kw in CASES.items():
        if only and name not in only:
            continue
        run_case(name, **kw)


if __name__ == "__main__":
    main()
