#!/usr/bin/env python
"""Golden fixtures for USER-REGISTERED components (tests/golden/custom_*.npz): the unmodified reference Foundation with a
toy component -- defined HERE against the reference's own BaseComponent and registered through its open component
registry (F/base/base_component.py:378, F/base/registrar.py:48-66) -- listed among the built-in ones.  The same component
written as an ai_economist_amd.foundation.BatchedComponent (tests/test_batched_component.py) has to reproduce the
fixture: state after every step, rewards, the flat observation vectors with the component's keys at their sorted
positions.

Run in the build container only (needs /root/reference):   python oracle/gen_golden_custom.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import GTB, run_case  # noqa: E402
from ref_harness import load_reference_foundation  # noqa: E402


def register_reference_toys():
    foundation = load_reference_foundation()
    from ai_economist.foundation.base.base_component import BaseComponent, component_registry

    if component_registry.has("CoinSubsidy"):
        return foundation

    @component_registry.add
    class CoinSubsidy(BaseComponent):
        """Every `every`-th timestep agent i receives amount * (i + 1) coin; observes when the next payment is due."""
        name = "CoinSubsidy"
        required_entities = ["Coin"]
        agent_subclasses = ["BasicMobileAgent", "BasicPlanner"]

        def __init__(self, *args, amount=0.5, every=3, **kwargs):
            super().__init__(*args, **kwargs)
            self.amount = float(amount)
            self.every = int(every)

        def get_n_actions(self, agent_cls_name):
            return None

        def get_additional_state_fields(self, agent_cls_name):
            return {}

        def component_step(self):
            if self.world.timestep % self.every == 0:
                for agent in self.world.agents:
                    agent.state["inventory"]["Coin"] += self.amount * (agent.idx + 1)

        def generate_observations(self):
            nxt = float(self.every - self.world.timestep % self.every)
            obs = {str(agent.idx): {"next_in": nxt, "share": [self.amount * (agent.idx + 1), self.amount]}
                   for agent in self.world.agents}
            obs[self.world.planner.idx] = {"next_in": nxt}
            return obs

        def generate_masks(self, completions=0):
            return {}

    @component_registry.add
    class LaborRelief(BaseComponent):
        """Halves every agent's accumulated labor at the reset and caps it at `cap` after every step (a component that
        edits an endogenous quantity and takes part in reset)."""
        name = "LaborRelief"
        required_entities = ["Labor"]
        agent_subclasses = ["BasicMobileAgent"]

        def __init__(self, *args, cap=3.0, **kwargs):
            super().__init__(*args, **kwargs)
            self.cap = float(cap)

        def get_n_actions(self, agent_cls_name):
            return None

        def get_additional_state_fields(self, agent_cls_name):
            return {}

        def additional_reset_steps(self):
            for agent in self.world.agents:
                agent.state["inventory"]["Coin"] += 2.0  # a reset-time edit that shows in the first observations

        def component_step(self):
            for agent in self.world.agents:
                agent.state["endogenous"]["Labor"] = min(agent.state["endogenous"]["Labor"], self.cap)

        def generate_observations(self):
            return {}

        def generate_masks(self, completions=0):
            return {}

    return foundation


CASES = {
    # the subsidy BETWEEN the built-ins (after the auction, before Gather and the taxes): three launches per step
    "custom_subsidy_mid_4ag": dict(
        cfg=dict(scenario_name="layout_from_file/simple_wood_and_stone", n_agents=4, world_size=[25, 25], episode_length=60,
                 components=[GTB[0], GTB[1], ["CoinSubsidy", {"amount": 0.75, "every": 4}], GTB[2], GTB[3]],
                 starting_agent_coin=10, env_layout_file="quadrant_25x25_20each_30clump.txt"),
        seed=21, t_steps=75, obs_steps=[0, 1, 3, 4, 5, 60, 61, 75]),
    # a host component first and one last, on a generated layout, across an episode end
    "custom_first_and_last_5ag": dict(
        cfg=dict(scenario_name="uniform/simple_wood_and_stone", n_agents=5, world_size=[15, 15], episode_length=20,
                 components=[["LaborRelief", {"cap": 2.5}], ["Build", {}], ["Gather", {}],
                             ["CoinSubsidy", {"amount": 0.25, "every": 2}]],
                 starting_agent_coin=4, starting_stone_coverage=0.12, starting_wood_coverage=0.12),
        seed=5, t_steps=45, obs_steps=[0, 1, 2, 20, 21, 45], action_kw=dict(p_move=0.6, p_build=0.3, p_trade=0.0)),
    # no built-in between two host components, 10 agents, flatten_observations off is the test's side (same fixture)
    "custom_adjacent_10ag": dict(
        cfg=dict(scenario_name="layout_from_file/simple_wood_and_stone", n_agents=10, world_size=[25, 25], episode_length=40,
                 components=[GTB[0], ["CoinSubsidy", {"amount": 1.0, "every": 5}], ["LaborRelief", {"cap": 4.0}], GTB[1], GTB[2],
                             GTB[3]],
                 starting_agent_coin=10, env_layout_file="quadrant_25x25_20each_30clump.txt"),
        seed=9, t_steps=40, obs_steps=[0, 1, 5, 6, 40]),
}


if __name__ == "__main__":
    register_reference_toys()
    for name, kw in CASES.items():
        run_case(name, **kw)
