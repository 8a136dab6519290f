"""`SimpleLabor` (reference: F/components/simple_labor.py:15-134; dynamics ->
labor_component_step in csrc/aie_kernels_ose.hip)."""
import numpy as np

from ... import _cabi
from .base import BaseComponent, component_registry


@component_registry.add
class SimpleLabor(BaseComponent):
    name = "SimpleLabor"
    required_entities = ["Coin"]
    agent_subclasses = ["BasicMobileAgent"]
    comp_id = _cabi.COMP_SIMPLE_LABOR

    def __init__(self, *base_args, mask_first_step=True, payment_max_skill_multiplier=3,
                 pareto_param=4.0, skills=None, **base_kwargs):
        super().__init__(*base_args, **base_kwargs)
        self.num_labor_hours = 100
        assert isinstance(mask_first_step, bool)
        self.mask_first_step = mask_first_step
        self.pareto_param = float(pareto_param)
        assert self.pareto_param > 0
        self.payment_max_skill_multiplier = float(payment_max_skill_multiplier)
        pmsm = self.payment_max_skill_multiplier
        if skills is not None:
            self.skills = np.asarray(skills, np.float64)
            assert self.skills.shape == (self.n_agents,)
        else:
            # Like the reference (simple_labor.py:66-74) the expected ranked skills are a
            # Monte-Carlo estimate drawn from the GLOBAL NumPy stream at construction time,
            # so `np.random.seed(s)` before make_env_instance reproduces its values.
            pareto_samples = np.random.pareto(4, size=(1000, self.n_agents))
            clipped = np.minimum(pmsm, (pmsm - 1) * pareto_samples + 1)
            self.skills = np.sort(clipped, axis=1).mean(axis=0)

    def get_n_actions(self, agent_cls_name):
        return self.num_labor_hours if agent_cls_name == "BasicMobileAgent" else None

    def agent_state_fields(self):
        return {"skill": "skill", "production": "production"}

    def fill_config(self, cfg):
        if self.n_agents > _cabi.MAX_AGENTS_WIDE:
            raise ValueError("SimpleLabor supports at most {} agents".format(_cabi.MAX_AGENTS_WIDE))
        cfg.labor_mask_first_step = int(self.mask_first_step)
        cfg.labor_num_hours = self.num_labor_hours
        cfg.labor_pmsm = self.payment_max_skill_multiplier
        for i, v in enumerate(self.skills):
            cfg.labor_skills[i] = float(v)
