"""Base class for component specs (reference: F/base/base_component.py:16-375)."""
from ..registrar import Registry


class BaseComponent:
    name = ""
    component_type = None
    agent_subclasses = None
    required_entities = None
    comp_id = 0  # AIE_COMP_* in include/aie.h

    def __init__(self, n_agents, episode_length, inventory_scale=1):
        assert self.name
        assert isinstance(self.agent_subclasses, (tuple, list)) and self.agent_subclasses
        assert isinstance(self.required_entities, (tuple, list))
        assert isinstance(episode_length, int) and episode_length > 0
        self.n_agents = int(n_agents)
        self._episode_length = episode_length
        self._inventory_scale = float(inventory_scale)

    @property
    def episode_length(self):
        return int(self._episode_length)

    @property
    def inv_scale(self):
        return self._inventory_scale

    @property
    def shorthand(self):
        return self.name if self.component_type is None else self.component_type

    def get_n_actions(self, agent_cls_name):
        raise NotImplementedError

    def agent_state_fields(self):
        """{agent.state field the component adds (get_additional_state_fields): state tensor};
        used when the reference-shaped per-agent state is rebuilt (dense logs)."""
        return {}

    def fill_config(self, cfg):
        """Writes this component's kwargs into an AieConfig (ctypes)."""
        raise NotImplementedError


component_registry = Registry(BaseComponent)
