"""Host-side mirror of `ai_economist.foundation` for the MI355X batched backend.

Same entry points as the reference package (F/__init__.py:7-18):

    from ai_economist_amd import foundation
    env = foundation.make_env_instance("layout_from_file/simple_wood_and_stone",
                                       n_envs=4096, **same_kwargs_as_the_reference)
    obs = env.reset(); obs, rew, done, info = env.step(actions)

`foundation.scenarios / components / agents / resources / landmarks / endogenous`
are the registries; register your own Scenario/Component spec classes with
`@foundation.scenarios.add` / `@foundation.components.add`.
"""
from .components import BaseComponent, BatchedComponent  # noqa: F401  (base classes of user-registered components)
from .components import component_registry as components
from .entities import agent_registry as agents
from .entities import endogenous_registry as endogenous
from .entities import landmark_registry as landmarks
from .entities import resource_registry as resources
from . import utils  # save_episode_log / load_episode_log (F/utils.py)
from .scenarios import scenario_registry as scenarios


def make_env_instance(scenario_name, reference_format=False, **kwargs):
    """Looks the scenario up by name and constructs it (F/__init__.py:16-18).
    reference_format=True: a one-replica environment behind the reference's single-environment surface (per-actor
    dictionaries, env.world.agents, ...; foundation/reference_view.py) instead of the batched one."""
    scenario_class = scenarios.get(scenario_name)
    if reference_format:
        from .reference_view import ReferenceFormatEnv

        # the reference fills previous_episode_metrics at every episode end (base_env.py:763-765); a one-replica
        # environment can afford the device->host read that costs
        kwargs.setdefault("track_episode_metrics", True)
        return ReferenceFormatEnv(scenario_class(**dict(kwargs, n_envs=1)))
    return scenario_class(**kwargs)
