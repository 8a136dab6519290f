"""`Gather` (reference: F/components/move.py:16-91; dynamics -> gather_component_step)."""
from ... import _cabi
from .base import BaseComponent, component_registry


@component_registry.add
class Gather(BaseComponent):
    name = "Gather"
    required_entities = ["Coin", "House", "Labor"]
    agent_subclasses = ["BasicMobileAgent"]
    comp_id = _cabi.COMP_GATHER

    def __init__(self, *base_args, move_labor=1.0, collect_labor=1.0, skill_dist="none",
                 **base_kwargs):
        super().__init__(*base_args, **base_kwargs)
        self.move_labor = float(move_labor)
        assert self.move_labor >= 0
        self.collect_labor = float(collect_labor)
        assert self.collect_labor >= 0
        self.skill_dist = skill_dist.lower()
        assert self.skill_dist in ["none", "pareto", "lognormal"]

    def get_n_actions(self, agent_cls_name):
        return 4 if agent_cls_name == "BasicMobileAgent" else None

    def agent_state_fields(self):
        return {"bonus_gather_prob": "bonus_gather_prob"}

    def fill_config(self, cfg):
        cfg.move_labor = self.move_labor
        cfg.collect_labor = self.collect_labor
        cfg.gather_skill_dist = _cabi.SKILL[self.skill_dist]
