"""`Build` (reference: F/components/build.py:15-110; dynamics -> build_component_step
in csrc/aie_kernels.hip)."""
from ... import _cabi
from .base import BaseComponent, component_registry


@component_registry.add
class Build(BaseComponent):
    name = "Build"
    component_type = "Build"
    required_entities = ["Wood", "Stone", "Coin", "House", "Labor"]
    agent_subclasses = ["BasicMobileAgent"]
    comp_id = _cabi.COMP_BUILD

    def __init__(self, *base_args, payment=10, payment_max_skill_multiplier=1,
                 skill_dist="none", build_labor=10.0, **base_kwargs):
        super().__init__(*base_args, **base_kwargs)
        self.payment = int(payment)
        assert self.payment >= 0
        self.payment_max_skill_multiplier = int(payment_max_skill_multiplier)
        assert self.payment_max_skill_multiplier >= 1
        self.resource_cost = {"Wood": 1, "Stone": 1}
        self.build_labor = float(build_labor)
        assert self.build_labor >= 0
        self.skill_dist = skill_dist.lower()
        assert self.skill_dist in ["none", "pareto", "lognormal"]

    def get_n_actions(self, agent_cls_name):
        return 1 if agent_cls_name == "BasicMobileAgent" else None

    def agent_state_fields(self):
        return {"build_payment": "build_payment", "build_skill": "build_skill"}

    def fill_config(self, cfg):
        cfg.build_payment = self.payment
        cfg.build_payment_max_skill_multiplier = self.payment_max_skill_multiplier
        cfg.build_skill_dist = _cabi.SKILL[self.skill_dist]
        cfg.build_labor = self.build_labor
