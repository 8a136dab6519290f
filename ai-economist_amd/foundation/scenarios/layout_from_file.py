"""`layout_from_file/simple_wood_and_stone` (reference:
F/scenarios/simple_wood_and_stone/layout_from_file.py:17-267): fixed Wood / Stone /
Water layout, stochastic regeneration, egocentric crop observations, isoelastic
utility rewards.  Dynamics -> scenario_step_regen / write_observations /
compute_rewards in csrc/aie_kernels.hip.
"""
import os

import numpy as np

from ... import _cabi
from ..base_env import BaseEnvironment, scenario_registry

_LAYOUTS = None


def _builtin_layout(name):
    global _LAYOUTS
    if _LAYOUTS is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "layouts.npz")
        with np.load(path) as z:
            _LAYOUTS = {k: z[k] for k in z.files}
    return _LAYOUTS.get(name)


def parse_layout_string(text):
    """';'-separated rows of symbols: W wood, S stone, @ water (layout_from_file.py:99-112)."""
    rows = text.split(";")
    h, w = len(rows), max(len(r) for r in rows)
    grid = np.zeros((h, w), np.uint8)
    code = {"W": 1, "S": 2, "@": 3}
    for r, row in enumerate(rows):
        for c, sym in enumerate(row):
            grid[r, c] = code.get(sym, 0)
    return grid


@scenario_registry.add
class LayoutFromFile(BaseEnvironment):
    supports_batched_components = True  # (and every scenario derived from it: uniform/, quadrant/, multi_zone/, split_layout/)
    name = "layout_from_file/simple_wood_and_stone"
    agent_subclasses = ["BasicMobileAgent", "BasicPlanner"]
    required_entities = ["Wood", "Stone", "Water"]

    def __init__(self, *base_env_args, planner_gets_spatial_info=True, full_observability=False,
                 mobile_agent_observation_range=5,
                 env_layout_file="quadrant_25x25_20each_30clump.txt", resource_regen_prob=0.01,
                 fixed_four_skill_and_loc=False, starting_agent_coin=0, isoelastic_eta=0.23,
                 energy_cost=0.21, energy_warmup_constant=0, energy_warmup_method="decay",
                 planner_reward_type="coin_eq_times_productivity",
                 mixing_weight_gini_vs_coin=0.0, **base_env_kwargs):
        super().__init__(*base_env_args, **base_env_kwargs)
        self._planner_gets_spatial_info = bool(planner_gets_spatial_info)
        self._full_observability = bool(full_observability)
        self._mobile_agent_observation_range = int(mobile_agent_observation_range)

        # layout: a built-in name, a path to a ';'-separated text file, or the text itself
        grid = _builtin_layout(env_layout_file)
        if grid is None:
            if os.path.exists(env_layout_file):
                with open(env_layout_file, "r") as f:
                    grid = parse_layout_string(f.read())
            elif ";" in env_layout_file:
                grid = parse_layout_string(env_layout_file)[: self.world_size[0], : self.world_size[1]]
            else:
                raise FileNotFoundError("unknown env_layout_file {!r}".format(env_layout_file))
        H, W = self.world_size
        full = np.zeros((max(H, grid.shape[0]), max(W, grid.shape[1])), np.uint8)
        full[: grid.shape[0], : grid.shape[1]] = grid
        if full[H:, :].any() or full[:, W:].any():
            raise IndexError("layout does not fit world_size {}".format(self.world_size))
        self.env_layout = full[:H, :W]
        self._source_maps = {
            "Wood": (self.env_layout == 1).astype(np.uint8),
            "Stone": (self.env_layout == 2).astype(np.uint8),
            "Water": (self.env_layout == 3).astype(np.uint8),
        }

        self.layout_specs = dict(
            Wood={"regen_weight": float(resource_regen_prob), "regen_halfwidth": 0, "max_health": 1},
            Stone={"regen_weight": float(resource_regen_prob), "regen_halfwidth": 0, "max_health": 1},
        )
        assert 0 <= self.layout_specs["Wood"]["regen_weight"] <= 1
        assert 0 <= self.layout_specs["Stone"]["regen_weight"] <= 1
        self.starting_agent_coin = float(starting_agent_coin)
        assert self.starting_agent_coin >= 0.0
        self.isoelastic_eta = float(isoelastic_eta)
        assert 0.0 <= self.isoelastic_eta <= 1.0
        self.energy_cost = float(energy_cost)
        assert self.energy_cost >= 0
        self.energy_warmup_method = energy_warmup_method.lower()
        assert self.energy_warmup_method in ["decay", "auto"]
        self.energy_warmup_constant = float(energy_warmup_constant)
        assert self.energy_warmup_constant >= 0
        self.planner_reward_type = str(planner_reward_type).lower()
        if self.planner_reward_type not in _cabi.PLANNER_REWARD:
            raise NotImplementedError("No valid planner reward selected!")
        self.mixing_weight_gini_vs_coin = float(mixing_weight_gini_vs_coin)
        assert 0 <= self.mixing_weight_gini_vs_coin <= 1.0

        self.fixed_four_skill_and_loc = bool(fixed_four_skill_and_loc)
        self._ranked_locs = []
        self._avg_ranked_skill = None
        if self.fixed_four_skill_and_loc:
            bm = self.get_component("Build")
            assert bm.skill_dist == "pareto"
            pmsm = bm.payment_max_skill_multiplier
            # expected skill of the i-th ranked agent: a fixed-seed Monte-Carlo estimate
            # (layout_from_file.py:181-194); a private RandomState leaves no global trace.
            rs = np.random.RandomState(1)
            pareto_samples = rs.pareto(4, size=(100000, self.n_agents))
            clipped = np.minimum(pmsm, (pmsm - 1) * pareto_samples + 1)
            self._avg_ranked_skill = np.sort(clipped, axis=1).mean(axis=0) * bm.payment
            starts = [(0, W - 1), (H - 1, 0), (0, 0), (W - 1, W - 1)]  # :197-206 (sic: W-1)
            groups = np.floor(np.arange(self.n_agents) * (4 / self.n_agents)).astype(int)
            n_in_group = np.zeros(4, dtype=int)
            for g in groups:
                pos = n_in_group[g]
                r0, c0 = starts[g]
                dr, dc = pos // 4, pos % 4
                r = r0 + dr if g in (0, 2) else r0 - dr
                c = c0 - dc if g in (0, 3) else c0 + dc
                self._ranked_locs.append((int(r), int(c)))
                n_in_group[g] += 1
            for (r, c) in self._ranked_locs:
                if not (0 <= r < H and 0 <= c < W) or self._source_maps["Water"][r, c]:
                    raise ValueError("fixed_four_skill_and_loc start {} is not accessible".format((r, c)))

    def layout_planes(self):
        return (self._source_maps["Stone"], self._source_maps["Wood"], self._source_maps["Water"])

    def world_flat_keys(self):
        """The scenario's scalar observations (layout_from_file.py:474-517), see foundation/obs_keys.py."""
        inv = [("world-inventory-%s" % r, 1, True) for r in ("Coin", "Stone", "Wood")]
        loc = [("world-loc-col", 1, True), ("world-loc-row", 1, True)]
        a = inv + ([] if self._full_observability else loc)
        pa = [] if self._full_observability else inv + (loc if self._planner_gets_spatial_info else [])
        return a, list(inv), pa

    def scenario_metrics(self, tensors):
        from .. import metrics

        return metrics.gtb_scenario_metrics(self, tensors)

    def fill_scenario_config(self, cfg):
        cfg.has_water = 1
        cfg.shared_layout = 1
        cfg.planner_gets_spatial_info = int(self._planner_gets_spatial_info)
        cfg.full_observability = int(self._full_observability)
        cfg.obs_range = self._mobile_agent_observation_range
        cfg.fixed_four_skill_and_loc = int(self.fixed_four_skill_and_loc)
        cfg.energy_warmup_method = _cabi.WARMUP[self.energy_warmup_method]
        cfg.planner_reward_type = _cabi.PLANNER_REWARD[self.planner_reward_type]
        for i, r in enumerate(["Stone", "Wood"]):
            cfg.regen_halfwidth[i] = self.layout_specs[r]["regen_halfwidth"]
            cfg.max_health[i] = self.layout_specs[r]["max_health"]
            cfg.regen_weight[i] = self.layout_specs[r]["regen_weight"]
        cfg.starting_agent_coin = self.starting_agent_coin
        cfg.isoelastic_eta = self.isoelastic_eta
        cfg.energy_cost = self.energy_cost
        cfg.energy_warmup_constant = self.energy_warmup_constant
        cfg.mixing_weight_gini_vs_coin = self.mixing_weight_gini_vs_coin
        if self.fixed_four_skill_and_loc:
            for i, (r, c) in enumerate(self._ranked_locs):
                cfg.ranked_locs[i][0] = r
                cfg.ranked_locs[i][1] = c
                cfg.avg_ranked_skill[i] = float(self._avg_ranked_skill[i])


@scenario_registry.add
class SplitLayout(LayoutFromFile):
    """`split_layout/simple_wood_and_stone` (layout_from_file.py:653-800): the file layout plus
    a row of water midway; ranked pareto build skills handed out in a random order at reset;
    the listed skill ranks start above the water, everybody else below."""
    name = "split_layout/simple_wood_and_stone"

    def __init__(self, *args, water_row=None, skill_rank_of_top_agents=None, **kwargs):
        super().__init__(*args, **kwargs)
        if self.fixed_four_skill_and_loc:
            raise ValueError("The split layout scenario does not support fixed_four_skill_and_loc. "
                             "Set this to False.")
        if water_row is None:
            self._water_line = self.world_size[0] // 2
        else:
            self._water_line = int(water_row)
            assert 0 < self._water_line < self.world_size[0] - 1
        for landmark, landmark_map in self._source_maps.items():
            landmark_map[self._water_line, :] = 1 if landmark == "Water" else 0
        if skill_rank_of_top_agents is None:
            skill_rank_of_top_agents = [0]
        if isinstance(skill_rank_of_top_agents, (int, float)):
            self.skill_rank_of_top_agents = [int(skill_rank_of_top_agents)]
        elif isinstance(skill_rank_of_top_agents, (tuple, list)):
            self.skill_rank_of_top_agents = list(set(skill_rank_of_top_agents))
        else:
            raise TypeError("skill_rank_of_top_agents must be a scalar index, or a list of scalar indices.")
        for rank in self.skill_rank_of_top_agents:
            assert 0 <= rank < self.n_agents
        assert 0 < len(self.skill_rank_of_top_agents) < self.n_agents
        bm = self.get_component("Build")
        assert bm.skill_dist == "pareto"
        pmsm = bm.payment_max_skill_multiplier
        # Like the reference (:750-757) the ranked skills are a Monte-Carlo estimate drawn from
        # the GLOBAL NumPy stream at construction time (no fixed seed here, unlike
        # fixed_four_skill_and_loc).  Index 0 = highest skill.
        pareto_samples = np.random.pareto(4, size=(100000, self.n_agents))
        clipped = np.minimum(pmsm, (pmsm - 1) * pareto_samples + 1)
        self._avg_ranked_skill = (np.sort(clipped, axis=1).mean(axis=0) * bm.payment)[::-1]

    def fill_scenario_config(self, cfg):
        super().fill_scenario_config(cfg)
        cfg.split_water_line = self._water_line
        for rank in self.skill_rank_of_top_agents:
            cfg.split_top_ranks[rank >> 5] |= 1 << (rank & 31)
        for i in range(self.n_agents):
            cfg.avg_ranked_skill[i] = float(self._avg_ranked_skill[i])
