"""One replica of a batched environment behind the reference's own single-environment surface
(F/base/base_env.py): per-actor observation / reward / info dictionaries keyed by the agents' `idx` ("0" ... "p"),
`done["__all__"]`, `env.all_agents`, `env.world.agents`, `env.world.planner`, `env.get_agent(idx)`.

    env = foundation.make_env_instance(reference_format=True, **the_reference_config)   # one replica
    obs = env.reset()                          # {"0": {...}, ..., "p": {...}}
    obs, rew, done, info = env.step({"0": 3, "p": [0, 1, ...]})       # or {} for all NO-OP

Presentation only: every call goes to the batched environment (`env.batched`), whose tensors stay available for
anything that wants them without the host copies this view makes."""
import numpy as np


class AgentView:
    """What user code reads off the reference's agent objects (F/base/base_agent.py): idx, the registered action
    subspaces and the action mode; `state` is a host snapshot of the replica's record for this actor."""

    def __init__(self, env, idx, name, names, multi_action_mode):
        self._env = env
        self.idx = idx
        self.name = name
        self.multi_action_mode = bool(multi_action_mode)
        self._names = list(names)  # [(subspace name, n actions)]

    @property
    def action_spaces(self):
        """base_agent.py:173-186: an int (single-action mode: all actions + 1 NO-OP) or an array of sizes."""
        if self.multi_action_mode:
            return np.array([d + 1 for _, d in self._names], dtype=np.int64)
        return 1 + sum(d for _, d in self._names)

    @property
    def action_dim(self):
        return {nm: d for nm, d in self._names}

    @property
    def state(self):
        """The actor's `state` dictionary as the dense log would record it now (a host read of the replica's record).
        Scenarios whose state dictionaries are accumulated step by step (COVID) have one only while the episode is
        being dense-logged."""
        if self._env is None:
            raise AttributeError("this agent view describes action spaces only (no environment attached)")
        b = self._env.batched
        if b._dense_log_this_episode and b._dense_logger is not None:
            return b._dense_logger.states_snapshot()[str(self.idx)]
        from .dense_log import DenseLogger

        logger = b.make_dense_logger()
        if type(logger) is not DenseLogger:
            raise NotImplementedError("agent.state of this scenario is available while the episode is dense-logged "
                                      "(reset(force_dense_logging=True))")
        logger.begin_episode()
        return logger.states_snapshot()[str(self.idx)]


class _World:
    def __init__(self, agents, planner):
        self.agents = agents
        self.planner = planner
        self.n_agents = len(agents)


class ReferenceFormatEnv:
    def __init__(self, batched):
        assert batched.n_envs == 1, "the reference's single-environment surface shows one replica (n_envs=1)"
        self.batched = batched
        names_a, names_p = batched.action_subspace_names()
        agents = [AgentView(self, i, "BasicMobileAgent", names_a, batched.multi_action_mode_agents)
                  for i in range(batched.n_agents)]
        planner = AgentView(self, "p", "BasicPlanner", names_p, batched.multi_action_mode_planner)
        self.world = _World(agents, planner)

    def __getattr__(self, name):  # n_agents, components, get_component, metrics of the batch, dense logs, ...
        if name == "batched":  # (not set yet: do not recurse)
            raise AttributeError(name)
        return getattr(self.batched, name)

    @property
    def all_agents(self):
        return self.world.agents + [self.world.planner]

    def get_agent(self, agent_idx):
        """base_env.py:521-535."""
        if str(agent_idx) == "p":
            return self.world.planner
        return self.world.agents[int(agent_idx)]

    @property
    def metrics(self):
        return self.batched.metrics_of(0)

    @property
    def previous_episode_metrics(self):
        m = self.batched.previous_episode_metrics
        return None if m is None else {k: v[0].item() for k, v in m.items()}

    def seed(self, seed):
        self.batched.seed(seed)

    def _obs(self):
        return self.batched.as_reference_dicts(0)

    def reset(self, seed_state=None, force_dense_logging=False):
        self.batched.reset(seed_state=seed_state, force_dense_logging=force_dense_logging)
        return self._obs()

    def step(self, actions=None, seed_state=None):
        b = self.batched
        b.step(actions if actions else None, seed_state=seed_state)
        t = b.backend.tensors
        ra = t["rewards_a"][0].cpu().numpy()
        rew = {str(i): float(ra[i]) for i in range(b.n_agents)}
        rew["p"] = float(t["rewards_p"][0].item())
        done = {"__all__": bool(t["done"][0].item())}
        info = {k: {} for k in rew}
        return self._obs(), rew, done, info
