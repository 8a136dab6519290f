"""`uniform/simple_wood_and_stone` (reference:
F/scenarios/simple_wood_and_stone/dynamic_layout.py:16-702).  Same step / observation /
reward kernels as layout_from_file; what differs is the reset: a fresh random source
layout per episode (no Water), and agents placed in a random order.

The layout generation (dynamic_layout.py:313-392: uniform draws, a shrinking threshold,
then growth by convolving with random 7x7 kernels; MultiZone's per-reset zone shuffle,
Quadrant's empty water lines) runs INSIDE the reset kernel, from each replica's own
legacy-NumPy stream (csrc/aie_kernels.hip: layout_generate; restated for the checker in
oracle/aie_oracle.c and pinned there against the live reference): a reset has no host
round trip.  This module supplies the static source-probability maps and the kwargs.
`generate_layout*` below is the same procedure on the host (NumPy / SciPy), kept for worlds
larger than the device path's 4096 cells and as documentation of the reference's steps.
"""
import numpy as np

from ... import _cabi
from ..base_env import scenario_registry
from .layout_from_file import LayoutFromFile


def _empty_like_reference(planes):
    """world.maps.empty (world.py:307-312): no resource and no landmark on the tile."""
    return np.sum(np.stack(planes), axis=0) == 0


@scenario_registry.add
class Uniform(LayoutFromFile):
    name = "uniform/simple_wood_and_stone"
    agent_subclasses = ["BasicMobileAgent", "BasicPlanner"]
    required_entities = ["Wood", "Stone"]

    def __init__(self, *base_env_args, planner_gets_spatial_info=True, full_observability=False,
                 mobile_agent_observation_range=5, starting_wood_coverage=0.025,
                 wood_regen_halfwidth=0, wood_regen_weight=0.01, wood_max_health=1,
                 starting_stone_coverage=0.025, stone_regen_halfwidth=0, stone_regen_weight=0.01,
                 stone_max_health=1, wood_clumpiness=0.35, stone_clumpiness=0.5, gradient_steepness=8,
                 checker_source_blocks=False, starting_agent_coin=0, isoelastic_eta=0.23,
                 energy_cost=0.21, energy_warmup_constant=0, energy_warmup_method="decay",
                 planner_reward_type="coin_eq_times_productivity", mixing_weight_gini_vs_coin=0.0,
                 **base_env_kwargs):
        # the shared part (observation / reward kwargs) is validated by the parent with an
        # empty layout; the layout itself is generated at every reset
        super().__init__(
            *base_env_args, planner_gets_spatial_info=planner_gets_spatial_info,
            full_observability=full_observability,
            mobile_agent_observation_range=mobile_agent_observation_range, env_layout_file=";",
            resource_regen_prob=0.0, fixed_four_skill_and_loc=False,
            starting_agent_coin=starting_agent_coin, isoelastic_eta=isoelastic_eta,
            energy_cost=energy_cost, energy_warmup_constant=energy_warmup_constant,
            energy_warmup_method=energy_warmup_method, planner_reward_type=planner_reward_type,
            mixing_weight_gini_vs_coin=mixing_weight_gini_vs_coin, **base_env_kwargs)
        H, W = self.world_size
        self.layout_specs = dict(Wood={}, Stone={})
        if starting_wood_coverage >= 1:
            starting_wood_coverage /= np.prod(self.world_size)
        if starting_stone_coverage >= 1:
            starting_stone_coverage /= np.prod(self.world_size)
        assert (starting_stone_coverage + starting_wood_coverage) < 0.5
        self._checker_source_blocks = bool(checker_source_blocks)
        cc, rr = np.meshgrid(np.arange(W) % 2, np.arange(H) % 2)
        self._checker_mask = (rr + cc) == 1
        m = 2 if self._checker_source_blocks else 1
        self.layout_specs["Wood"]["starting_coverage"] = float(starting_wood_coverage) * m
        self.layout_specs["Stone"]["starting_coverage"] = float(starting_stone_coverage) * m
        assert 0 < self.layout_specs["Wood"]["starting_coverage"] < 1
        assert 0 < self.layout_specs["Stone"]["starting_coverage"] < 1
        self.layout_specs["Wood"]["regen_halfwidth"] = int(wood_regen_halfwidth)
        self.layout_specs["Stone"]["regen_halfwidth"] = int(stone_regen_halfwidth)
        assert 0 <= self.layout_specs["Wood"]["regen_halfwidth"] <= 3
        assert 0 <= self.layout_specs["Stone"]["regen_halfwidth"] <= 3
        self.layout_specs["Wood"]["regen_weight"] = float(wood_regen_weight)
        self.layout_specs["Stone"]["regen_weight"] = float(stone_regen_weight)
        assert 0 <= self.layout_specs["Wood"]["regen_weight"] <= 1
        assert 0 <= self.layout_specs["Stone"]["regen_weight"] <= 1
        self.layout_specs["Wood"]["max_health"] = int(wood_max_health)
        self.layout_specs["Stone"]["max_health"] = int(stone_max_health)
        assert self.layout_specs["Wood"]["max_health"] > 0
        assert self.layout_specs["Stone"]["max_health"] > 0
        self.clumpiness = {"Wood": float(wood_clumpiness), "Stone": float(stone_clumpiness)}
        assert all(0 <= v <= 1 for v in self.clumpiness.values())
        self.gradient_steepness = float(gradient_steepness)
        assert self.gradient_steepness >= 1.0
        self.source_prob_maps = self.make_source_prob_maps()

    def make_source_prob_maps(self):
        """Row gradient ** steepness, normalised; Stone is Wood's map flipped -- and, as in
        the reference (dynamic_layout.py:306-307), scaled by WOOD's coverage."""
        H, W = self.world_size
        grad = np.arange(H)[:, None].repeat(W, axis=1) ** self.gradient_steepness
        grad = grad / np.mean(grad)
        cov = self.layout_specs["Wood"]["starting_coverage"]
        return {"Wood": grad * cov, "Stone": grad[-1::-1] * cov}

    def generate_layout(self, rs):
        """One reset_starting_layout() (dynamic_layout.py:313-392) driven by the legacy
        RandomState `rs`.  Returns (stone_src, wood_src) uint8 planes."""
        from scipy import signal

        source_maps = {}
        happy, tries = False, 0
        while tries < 100 and not happy:
            source_maps = {}
            placed = []  # planes that make tiles non-empty: resource + source block
            for resource in ["Wood", "Stone"]:
                cov = self.layout_specs[resource]["starting_coverage"]
                clump = 1 - np.clip(self.clumpiness[resource], 0.0, 0.99)
                source_prob = self.source_prob_maps[resource] * 0.1 * clump
                empty = (_empty_like_reference(placed) if placed
                         else np.ones(source_prob.shape, dtype=bool))
                tmp = rs.rand(*source_prob.shape)
                maybe = (tmp < source_prob) * empty
                n_tries = 0
                while np.mean(maybe) < cov * clump:
                    tmp *= 0.9
                    maybe = (tmp < source_prob) * empty
                    n_tries += 1
                    if n_tries > 200:
                        break
                while np.mean(maybe) < cov:
                    kernel = rs.randn(7, 7) > 0
                    grown = signal.convolve2d(
                        maybe + (0.2 * rs.randn(*maybe.shape)) - 0.25, kernel.astype(np.float32), "same")
                    maybe = np.maximum(grown > 0, maybe) * empty
                source_maps[resource] = maybe
                placed += [np.asarray(maybe, np.float64), np.asarray(maybe, np.float64)]
            happy = True
            for resource in ["Wood", "Stone"]:
                q = np.mean(source_maps[resource]) / self.layout_specs[resource]["starting_coverage"]
                if not (1 / 1.4) <= q <= 1.4:
                    happy = False
            tries += 1
        if self._checker_source_blocks:
            source_maps = {k: v * self._checker_mask for k, v in source_maps.items()}
        return ((np.asarray(source_maps["Stone"]) > 0).astype(np.uint8),
                (np.asarray(source_maps["Wood"]) > 0).astype(np.uint8))

    def generate_layout_flags(self, rs):
        """The packed static cell flags (1 water | 2 stone source | 4 wood source) of one
        freshly generated layout; subclasses add water / re-draw their probability maps."""
        stone, wood = self.generate_layout(rs)
        return (2 * stone + 4 * wood).astype(np.uint8)

    def layout_planes(self):
        z = np.zeros([self.n_envs] + list(self.world_size), np.uint8)
        return (z, z, z)

    def scenario_metrics(self, tensors):
        from .. import metrics

        return metrics.gtb_scenario_metrics(self, tensors)

    layout_gen = _cabi.LAYOUT_UNIFORM
    DEVICE_LAYOUT_MAX_CELLS = 4096  # csrc/aie_layout.h: the generator's planes live in LDS (<= 160 KB per workgroup)

    @property
    def layouts_on_device(self):
        return int(np.prod(self.world_size)) <= self.DEVICE_LAYOUT_MAX_CELLS

    def fill_scenario_config(self, cfg):
        super().fill_scenario_config(cfg)
        cfg.has_water = 0
        cfg.shared_layout = 0
        cfg.reset_random_order = 1
        for i, r in enumerate(["Stone", "Wood"]):
            cfg.regen_halfwidth[i] = self.layout_specs[r]["regen_halfwidth"]
            cfg.max_health[i] = self.layout_specs[r]["max_health"]
            cfg.regen_weight[i] = self.layout_specs[r]["regen_weight"]
            cfg.layout_coverage[i] = float(self.layout_specs[r]["starting_coverage"])
            cfg.layout_clump[i] = float(1 - np.clip(self.clumpiness[r], 0.0, 0.99))
        cfg.layout_gen = self.layout_gen if self.layouts_on_device else _cabi.LAYOUT_FIXED
        if not self.layouts_on_device and self.rng_mode != "numpy":
            raise NotImplementedError("rng_mode='fast': worlds above %d cells draw their source layouts on the host from "
                                      "the replica's NumPy stream, which the counter-based generator does not have; use "
                                      "rng_mode='numpy' or a smaller world" % self.DEVICE_LAYOUT_MAX_CELLS)
        cfg.layout_checker = int(self._checker_source_blocks)
        if cfg.layout_gen != _cabi.LAYOUT_FIXED:
            # constant tensors that go with this configuration (the device backend uploads them once; the CPU checker
            # fills its arena from the same attribute)
            cfg._model_tensors = {"layout_source_prob": np.stack([
                np.asarray(self.source_prob_maps["Stone"], np.float64),
                np.asarray(self.source_prob_maps["Wood"], np.float64)]).reshape(1, 2, -1)}

    def upload_model_constants(self, backend):
        for name, arr in getattr(backend.cfg, "_model_tensors", {}).items():
            backend.upload(name, arr)

    def host_pre_reset(self, env_mask):
        """Generates a fresh source layout for every replica about to be reset, continuing
        that replica's own MT19937 stream."""
        if self.layouts_on_device:
            return  # the reset kernel draws the layout itself
        be = self.backend
        torch = __import__("torch")
        torch.cuda.synchronize(be.device)
        if env_mask is None:
            which = np.arange(self.n_envs)
        else:
            which = np.nonzero(torch.as_tensor(env_mask).to("cpu").numpy().reshape(-1))[0]
        if len(which) == 0:
            return
        idx = torch.as_tensor(which, device=be.device)
        t = be.tensors
        keys = t["mt"][idx].cpu().numpy().view(np.uint32)
        pos = t["mt_pos"][idx].cpu().numpy()
        hasg = t["mt_has_gauss"][idx].cpu().numpy()
        gauss = t["mt_gauss"][idx].cpu().numpy()
        flags = np.zeros((len(which),) + tuple(self.world_size), np.uint8)
        rs = np.random.RandomState()
        for k in range(len(which)):
            rs.set_state(("MT19937", keys[k], int(pos[k]), int(hasg[k]), float(gauss[k])))
            flags[k] = self.generate_layout_flags(rs)
            st = rs.get_state()
            keys[k], pos[k], hasg[k], gauss[k] = st[1], st[2], st[3], st[4]
        t["cell_flags"][idx] = torch.as_tensor(flags, device=be.device)
        t["mt"][idx] = torch.as_tensor(keys.view(np.int32), device=be.device)
        t["mt_pos"][idx] = torch.as_tensor(pos, device=be.device)
        t["mt_has_gauss"][idx] = torch.as_tensor(hasg, device=be.device)
        t["mt_gauss"][idx] = torch.as_tensor(gauss, device=be.device)


@scenario_registry.add
class MultiZone(Uniform):
    """`multi_zone/simple_wood_and_stone` (dynamic_layout.py:705-873): the world is cut into
    num_partitions_row x num_partitions_col regions; a random subset of them are wood, stone
    or mixed zones, re-drawn (np.random.shuffle) at every reset."""
    name = "multi_zone/simple_wood_and_stone"

    def __init__(self, *args, num_partitions_row=8, num_partitions_col=8, num_wood_zones=6,
                 num_stone_zones=6, num_wood_and_stone_zones=4, **kwargs):
        self.num_partitions_row = num_partitions_row
        self.num_partitions_col = num_partitions_col
        self.zone_specs = {"Wood": (0, num_wood_zones), "Stone": (1, num_stone_zones),
                           "WoodStone": (2, num_wood_and_stone_zones)}
        super().__init__(*args, **kwargs)

    def make_source_prob_maps(self, rs=None):
        """dynamic_layout.py:778-864.  `rs`: the replica's stream at reset time; at construction
        the reference shuffles with the global NumPy stream, and so does this."""
        shuffle = np.random.shuffle if rs is None else rs.shuffle
        zone_names = list(self.zone_specs.keys())
        zone_indices = [v[0] for _, v in self.zone_specs.items()]
        num_zones_per_type = [self.zone_specs[k][1] for k in zone_names]
        num_zones = sum(num_zones_per_type)
        npr, npc = self.num_partitions_row, self.num_partitions_col
        num_regions = npr * npc
        assert num_regions >= num_zones
        size_r = int(np.ceil(self.world_size[0] / npr))
        size_c = int(np.ceil(self.world_size[1] / npc))
        grid = np.concatenate([np.repeat(zone_indices, num_zones_per_type),
                               np.array([-1] * (num_regions - num_zones))])
        shuffle(grid)
        grid = grid.reshape((npr, npc))
        out = {}
        for res in ("Wood", "Stone"):
            prob = np.where(np.logical_or(np.equal(grid, self.zone_specs[res][0]),
                                          np.equal(grid, self.zone_specs["WoodStone"][0])),
                            np.ones_like(grid), np.zeros_like(grid))
            prob = np.kron(prob, np.ones((size_r, size_c)))
            prob = prob[: self.world_size[0], : self.world_size[1]]
            prob = prob / np.mean(prob)
            assert prob.shape[0] == self.world_size[0] and prob.shape[1] == self.world_size[1]
            # (the reference scales both maps by WOOD's coverage, :860-863)
            out[res] = prob * self.layout_specs["Wood"]["starting_coverage"]
        return out

    layout_gen = _cabi.LAYOUT_MULTI_ZONE

    def fill_scenario_config(self, cfg):
        super().fill_scenario_config(cfg)
        cfg.mz_rows, cfg.mz_cols = int(self.num_partitions_row), int(self.num_partitions_col)
        for k, name in enumerate(("Wood", "Stone", "WoodStone")):
            cfg.mz_zones[k] = int(self.zone_specs[name][1])

    def generate_layout_flags(self, rs):
        self.source_prob_maps = self.make_source_prob_maps(rs)  # reset_starting_layout :866-872
        return super().generate_layout_flags(rs)


@scenario_registry.add
class Quadrant(Uniform):
    """`quadrant/simple_wood_and_stone` (dynamic_layout.py:875-1024): a cross of water with
    openings splits the map into four quadrants; wood is likelier towards one corner, stone
    towards another."""
    name = "quadrant/simple_wood_and_stone"
    required_entities = Uniform.required_entities + ["Water"]

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        height, width = self.world_size
        o0, o1 = 0.2, 0.35
        rN = (0.5 + np.arange(height)) / height
        cN = (0.5 + np.arange(width)) / width
        rSeg = ((rN < o0) + (rN > o1)) * ((rN < 1 - o1) + (rN > 1 - o0))
        cSeg = ((cN < o0) + (cN > o1)) * ((cN < 1 - o1) + (cN > 1 - o0))
        water = np.zeros((height, width))
        water[:, height // 2] = rSeg  # (the reference swaps height/width here; square maps only)
        water[width // 2, :] = cSeg
        self._water = water
        for k, v in self.source_prob_maps.items():
            v = v * (1 - self._water)
            self.source_prob_maps[k] = v / np.sum(v)

    def make_source_prob_maps(self):  # :960-990
        height, width = self.world_size
        g = np.arange(height)[:, None].repeat(width, axis=1) ** (self.gradient_steepness / 2)
        w_grad = g[::-1]
        g = np.arange(width)[None].repeat(height, axis=0) ** (self.gradient_steepness / 2)
        s_grad = g[:, ::-1]
        prob_sum = s_grad + w_grad
        s_grad = prob_sum * s_grad
        w_grad = prob_sum * w_grad
        return {"Stone": s_grad / np.sum(s_grad), "Wood": w_grad / np.sum(w_grad)}

    def generate_layout_flags(self, rs):  # reset_starting_layout :992-1024
        stone, wood = self.generate_layout(rs)
        height, width = self.world_size
        for plane in (stone, wood):  # nothing on the water lines, openings included
            plane[:, height // 2] = 0
            plane[width // 2, :] = 0
        return ((self._water > 0) * 1 + 2 * stone + 4 * wood).astype(np.uint8)

    layout_gen = _cabi.LAYOUT_QUADRANT

    def layout_planes(self):
        z = np.zeros([self.n_envs] + list(self.world_size), np.uint8)
        water = np.broadcast_to((self._water > 0).astype(np.uint8), z.shape).copy()
        return (z, z, water)

    def fill_scenario_config(self, cfg):
        super().fill_scenario_config(cfg)
        cfg.has_water = 1
