"""`FoundationEnvWrapper` with the surface of the reference's device seam (F/env_wrapper.py:96-420: the class its
WarpDrive trainer drives): `reset_all_envs()`, `step_all_envs()`, `reset_only_done_envs()`, per-actor
`observation_space` / `action_space` dictionaries on the environment.  Here the "device side" is the batched
environment itself; there is no data manager to push to -- the tensors ARE the device data
(`wrapper.env.tensors`, `include/aie.h: aie_get_tensor`).

    wrapper = FoundationEnvWrapper(env_name="CovidAndEconomySimulation", env_config=cfg, num_envs=8192)
    wrapper.reset_all_envs()
    for _ in range(T):
        wrapper.step_all_envs({"a": actions_a, "p": actions_p})    # device tensors in, nothing copied out
        wrapper.reset_only_done_envs()
"""
import numpy as np

try:  # the reference builds gym spaces; without gym the same information in two tiny stand-ins
    from gym.spaces import Box, Discrete, MultiDiscrete
except Exception:  # pragma: no cover - depends on the image
    class Discrete:
        def __init__(self, n):
            self.n, self.dtype, self.shape = int(n), np.int32, ()

        def __repr__(self):
            return "Discrete(%d)" % self.n

    class MultiDiscrete:
        def __init__(self, nvec):
            self.nvec, self.dtype = np.asarray(nvec, np.int64), np.int32
            self.shape = self.nvec.shape

        def __repr__(self):
            return "MultiDiscrete(%s)" % (self.nvec.tolist(),)

    class Box:
        def __init__(self, low, high, shape, dtype):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

        def __repr__(self):
            return "Box(%s, %s)" % (self.shape, np.dtype(self.dtype).name)

BIG_NUMBER = 1e20


class FoundationEnvWrapper:
    def __init__(self, env_obj=None, env_name=None, env_config=None, num_envs=1, use_cuda=True, env_registrar=None,
                 event_messenger=None, process_id=0, device=None):
        from . import make_env_instance
        from .reference_view import AgentView

        if env_obj is not None:
            self.env = env_obj
        else:
            assert env_name is not None and env_config is not None
            kw = dict(env_config)
            kw.pop("scenario_name", None)
            self.env = make_env_instance(env_name, n_envs=int(num_envs), device=device, **kw)
        env = self.env
        self.n_envs = env.n_envs
        self.n_agents = env.num_agents  # mobile agents + the planner, as in the reference
        self.episode_length = env.episode_length
        self.name = env.name
        self.use_cuda = True  # the batched environment always steps on the device
        self.reset_on_host = True
        names_a, names_p = env.action_subspace_names()
        views = [AgentView(None, i, "BasicMobileAgent", names_a, env.multi_action_mode_agents)
                 for i in range(env.n_agents)]
        views.append(AgentView(None, "p", "BasicPlanner", names_p, env.multi_action_mode_planner))
        env.action_space = {}
        for v in views:
            sp = MultiDiscrete(v.action_spaces) if v.multi_action_mode else Discrete(v.action_spaces)
            sp.dtype = np.int32
            env.action_space[str(v.idx)] = sp
        obs = self.obs_at_reset()
        env.observation_space = {
            k: {kk: Box(-BIG_NUMBER, BIG_NUMBER, tuple(t.shape[1:]), np.float32) for kk, t in d.items()}
            for k, d in obs.items()}
        assert set(env.observation_space.keys()) == set(env.action_space.keys())

    # ---- F/env_wrapper.py:267-420 ----
    def obs_at_reset(self):
        return self._reformat_obs(self.env.reset())

    def _reformat_obs(self, obs):
        """Per-actor keys "0" ... "p" over the batched tensors: obs["3"][key] is the [E, ...] slice of agent 3 (a
        view, nothing is copied)."""
        out = {}
        n = self.env.n_agents
        # gather-trade-build / one-step-economy tensors carry the agent axis right behind the replica axis; the COVID
        # scenario keeps the reference's collated layout, agent axis last (F/env_wrapper.py:387-396)
        last = getattr(self.env, "supports_unflattened_observations", False)
        for i in range(n):
            d = {}
            for k, v in obs["a"].items():
                if not hasattr(v, "shape") or v.dim() < 2:
                    continue
                if last and v.shape[-1] == n:
                    d[k] = v[..., i]
                elif not last and v.shape[1] == n:
                    d[k] = v[:, i]
            out[str(i)] = d
        out["p"] = {k: v for k, v in obs["p"].items() if hasattr(v, "shape")}
        return out

    def _reformat_rew(self, rew):
        out = {str(i): rew["a"][:, i] for i in range(self.env.n_agents)}
        out["p"] = rew["p"]
        return out

    def reset_all_envs(self):
        obs = self.obs_at_reset()
        self.reset_on_host = False
        return obs

    def reset_only_done_envs(self):
        """Resets the replicas whose `done` flag is set (F/env_wrapper.py:341-353), on the device."""
        self.env.reset(self.env.tensors["done"])
        return {}

    def step_all_envs(self, actions=None):
        """One step of every replica; like the reference's device path it returns nothing -- observations, rewards
        and done flags are the environment's tensors."""
        self.env.step(actions)
        return None

    def reset(self):
        return self.reset_all_envs()

    def step(self, actions=None):
        obs, rew, done, info = self.env.step(actions)
        return self._reformat_obs(obs), self._reformat_rew(rew), done, info
