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


class BatchedComponent(BaseComponent):
    """A user's component whose dynamics are HOST code over the batch (round 6).

    The reference's component registry is open (F/base/base_component.py:378, F/base/registrar.py:48-66): anybody can
    subclass BaseComponent, register the class and list it in `components`.  The built-in components' dynamics are device
    kernels here, which a Python class cannot join -- but it can run BETWEEN launches.  A registered subclass of this class
    is placed in the component list like any other; the environment then steps through `aie_step_range` (include/aie.h):
    the built-in components ahead of it in one launch, then its `component_step(tensors)` as torch code on the zero-copy
    state tensors of all replicas at once, then the next stretch of built-ins, ... and the end of the step (regeneration,
    observations, masks, rewards) in the last launch.  Same order of effects as the reference's
    `for component in self._components: component.component_step()` (base_env.py:985-987).

        @foundation.components.add
        class CoinSubsidy(foundation.BatchedComponent):
            name = "CoinSubsidy"
            required_entities = ["Coin"]
            agent_subclasses = ["BasicMobileAgent"]

            def __init__(self, *args, amount=1.0, **kwargs):
                super().__init__(*args, **kwargs)
                self.amount = float(amount)

            def component_step(self, t):                    # t: {name: tensor [n_envs, ...]}, the arena's own memory
                t["inv_coin"] += self.amount

            def generate_observations(self, t):             # optional: {"a": {key: [n_envs, n_agents(, k)]}, "p": {key: [n_envs(, k)]}}
                return {"a": {"amount": t["inv_coin"].new_full(t["inv_coin"].shape, self.amount)}, "p": {}}

    What it can do: read and write every state tensor (`env.tensors`: inv_coin, inv_res, labor, loc_r / loc_c, stone /
    wood / house_owner, ... the names of include/aie.h's tensor table), add observations (they enter the flat vectors at
    their sorted-key position "<name>-<key>" exactly as the reference packs them, base_env.py:561-612, or appear under
    that key with flatten_observations=False), keep its own torch state, take part in reset (`additional_reset_steps`).
    What it cannot (yet): own an ACTION subspace (`get_n_actions` must return None: the action layout and the masks are
    the kernels'), draw from a replica's NumPy stream, run in the COVID / one-step-economy scenarios, be captured in a
    hipGraph with data-dependent Python control flow.  Cost: one extra launch per stretch, the full-featured kernel
    instead of the configuration's instance, and whatever the hook's torch code costs -- an extension point, not the hot
    path.  tests/test_batched_component.py holds a toy component against the same component added to the live
    reference."""
    comp_id = 0  # no device kernel
    is_batched_host_component = True

    def get_n_actions(self, agent_cls_name):
        return None

    def fill_config(self, cfg):
        return None

    def component_step(self, tensors):
        raise NotImplementedError

    def generate_observations(self, tensors):
        return {"a": {}, "p": {}}

    def additional_reset_steps(self, tensors, env_mask=None):
        """Called after the reset kernel; `env_mask`: the uint8 [n_envs] mask of the replicas that were reset (None:
        all).  Return True when state tensors were edited (the observations are then rewritten)."""
        return False


component_registry = Registry(BaseComponent)
