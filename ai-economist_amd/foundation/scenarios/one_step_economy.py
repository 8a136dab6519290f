"""`one-step-economy` (reference: F/scenarios/one_step_economy/one_step_economy.py:15-336):
no map; the planner sets taxes, the agents choose labor.  Dynamics / observations /
rewards -> csrc/aie_kernels_ose.hip."""
import numpy as np

from ... import _cabi
from ..base_env import BaseEnvironment, scenario_registry


@scenario_registry.add
class OneStepEconomy(BaseEnvironment):
    name = "one-step-economy"
    agent_subclasses = ["BasicMobileAgent", "BasicPlanner"]
    required_entities = ["Coin"]

    def __init__(self, *base_env_args, agent_reward_type="coin_minus_labor_cost", isoelastic_eta=0.23,
                 labor_exponent=2.0, labor_cost=1.0, planner_reward_type="inv_income_weighted_utility",
                 mixing_weight_gini_vs_coin=0, **base_env_kwargs):
        super().__init__(*base_env_args, **base_env_kwargs)
        self.labor_cost = labor_cost
        self.agent_reward_type = agent_reward_type
        if agent_reward_type not in _cabi.AGENT_REWARD:
            raise NotImplementedError("unknown agent_reward_type {!r}".format(agent_reward_type))
        self.isoelastic_eta = isoelastic_eta
        self.labor_exponent = labor_exponent
        self.planner_reward_type = planner_reward_type
        # one_step_economy.py:316-335 knows exactly these two social welfare functions
        if planner_reward_type not in ("coin_eq_times_productivity", "inv_income_weighted_utility"):
            raise NotImplementedError("No valid planner reward selected!")
        self.mixing_weight_gini_vs_coin = mixing_weight_gini_vs_coin
        self.planner_starting_coin = 0
        for c in self.components:
            if c.name not in ("SimpleLabor", "PeriodicBracketTax", "WealthRedistribution"):
                raise NotImplementedError(
                    "one-step-economy is implemented for SimpleLabor, PeriodicBracketTax, WealthRedistribution")

    def layout_planes(self):
        z = np.zeros(self.world_size, np.uint8)
        return (z, z, z)

    def world_flat_keys(self):
        """one_step_economy.py:118-183: only the planner has scenario observations."""
        return [], [("world-equality", 1, True), ("world-normalized_per_capita_productivity", 1, True)], []

    def scenario_metrics(self, tensors):
        from .. import metrics

        return metrics.ose_scenario_metrics(self, tensors)

    def fill_scenario_config(self, cfg):
        cfg.scenario = _cabi.SCN_ONE_STEP_ECONOMY
        cfg.shared_layout = 1
        cfg.ose_agent_reward_type = _cabi.AGENT_REWARD[self.agent_reward_type]
        cfg.isoelastic_eta = float(self.isoelastic_eta)
        cfg.ose_labor_exponent = float(self.labor_exponent)
        cfg.ose_labor_cost = float(self.labor_cost)
        cfg.planner_reward_type = _cabi.PLANNER_REWARD[self.planner_reward_type]
        cfg.mixing_weight_gini_vs_coin = float(self.mixing_weight_gini_vs_coin)
