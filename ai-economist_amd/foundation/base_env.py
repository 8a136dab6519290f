"""BaseEnvironment: the Gym-style reset()/step() surface, batched over E replicas.

Mirrors the constructor contract, properties and call order of the reference's
`BaseEnvironment` (F/base/base_env.py:24-1120) but every replica-level computation is
a HIP kernel behind the C ABI (include/aie.h); this class only validates kwargs,
builds the `aie_config`, and hands out batched torch views.

Differences that come with batching (documented in INTEGRATION.md):
  * obs / rew / done are batched tensors keyed "a" (all mobile agents, leading dims
    [E, n_agents]) and "p" (planner, leading dim [E]) -- the layout the reference uses
    with collate_agent_step_and_reset_data=True, with the agent axis FIRST;
    `as_reference_dicts(e)` rebuilds the reference's exact per-replica dict.
  * randomness is one legacy-NumPy MT19937 stream PER replica (replica e of
    `seed(s)` == the reference after `env.seed(s + e)`), not the process-global stream.
"""
import ctypes

import numpy as np

from .. import _cabi
from .components import component_registry
from .entities import endogenous_registry, landmark_registry, resource_registry
from .registrar import Registry


class BaseEnvironment:
    name = ""
    agent_subclasses = []
    required_entities = None
    # scenarios whose observations are already per-key arrays (COVID) accept
    # flatten_observations=False, which is how the reference's run config sets it
    supports_unflattened_observations = False
    # host components (foundation.BatchedComponent) run between launches of aie_step_range: gather-trade-build scenarios
    supports_batched_components = False

    def __init__(self, components=None, n_agents=None, world_size=None, episode_length=1000,
                 multi_action_mode_agents=False, multi_action_mode_planner=True,
                 flatten_observations=True, flatten_masks=True,
                 allow_observation_scaling=True, dense_log_frequency=None,
                 world_dense_log_frequency=50, collate_agent_step_and_reset_data=False,
                 seed=None, n_envs=1, device=None, env_offset=0, track_episode_metrics=False, rng_mode="numpy"):
        assert self.name
        assert isinstance(self.agent_subclasses, (tuple, list)) and len(self.agent_subclasses) > 0
        assert isinstance(self.required_entities, (tuple, list))
        assert isinstance(world_size, (tuple, list)) and len(world_size) == 2
        self.world_size = list(world_size)
        assert isinstance(n_agents, int) and n_agents >= 2
        self.n_agents = n_agents
        self.num_agents = n_agents + 1  # + planner, as in base_env.py:227-230
        assert isinstance(components, (tuple, list))

        def spec_is_valid(spec):
            if isinstance(spec, (tuple, list)):
                return len(spec) == 2 and isinstance(spec[0], str) and isinstance(spec[1], dict)
            if isinstance(spec, dict):
                return (len(spec) == 1 and isinstance(list(spec.keys())[0], str)
                        and isinstance(list(spec.values())[0], dict))
            return False

        assert all(spec_is_valid(c) for c in components)
        self._episode_length = int(episode_length)
        assert self._episode_length >= 1
        self.multi_action_mode_agents = bool(multi_action_mode_agents)
        self.multi_action_mode_planner = bool(multi_action_mode_planner)
        self._allow_observation_scaling = bool(allow_observation_scaling)
        # flatten_masks=False (base_env.py:706-756): the kernels still write the flattened masks; the observation dicts
        # then carry {"<Component>[.<sub-action>]": view} instead of the vector (foundation/obs_keys.py: mask_keys)
        self._flatten_masks = bool(flatten_masks)
        self._mask_key_views = None
        # flatten_observations=False (base_env.py:591-612): the kernels still write the packed `flat` vectors; the
        # observation dicts then hand out every key as a zero-copy slice of them (foundation/obs_keys.py)
        self._flatten_observations = bool(flatten_observations)
        self._flat_key_views = None
        # dense logs (base_env.py:148-163, 273-283): replica `dense_log_replica` of the batch is
        # the one whose episodes are logged; the device records its component events
        self._create_dense_log_every = None
        if dense_log_frequency is not None:
            self._create_dense_log_every = int(dense_log_frequency)
            assert self._create_dense_log_every >= 1
        self.world_dense_log_frequency = int(world_dense_log_frequency)
        assert self.world_dense_log_frequency >= 1
        self._dense_log_this_episode = False
        self._dense_logger = None
        self._dense_log = {"world": [], "states": [], "actions": [], "rewards": []}
        self._last_ep_dense_log = dict(self._dense_log)
        # replay log of the logged replica (base_env.py:359-360, 445-471): the RNG state at reset and before every
        # step plus the actions -- recorded for the episodes that are dense-logged (a per-step device read)
        self._replay_log = {"reset": dict(seed_state=None), "step": []}
        self._last_ep_replay_log = dict(self._replay_log)
        self.collate_agent_step_and_reset_data = True
        self.n_envs = int(n_envs)
        assert self.n_envs >= 1
        self.env_offset = int(env_offset)  # global index of replica 0 (multi-GPU sharding)
        self._device = device
        # which generator stands behind the replicas' np.random.* draws (include/aie.h: AIE_RNG_*): "numpy" = NumPy's legacy
        # MT19937 stream per replica, bit for bit with the reference (the default); "fast" = a counter-based stream
        # (Philox2x32-10) -- a throughput mode the reference does not have, NOT stream-compatible with NumPy
        if rng_mode not in ("numpy", "fast"):
            raise ValueError("rng_mode must be 'numpy' (parity with the reference) or 'fast' (counter-based stream), got %r" % (rng_mode,))
        self.rng_mode = rng_mode

        self._entities = {"resources": ["Coin"], "landmarks": [], "endogenous": ["Labor"]}
        self._register_entities(self.required_entities)
        self._components = []
        self._components_dict = {}
        self._shorthand_lookup = {}
        for spec in components:
            if isinstance(spec, (tuple, list)):
                cname, ckw = spec
            else:
                cname, ckw = list(spec.keys())[0], list(spec.values())[0]
            ccls = component_registry.get(cname)
            if getattr(ccls, "is_batched_host_component", False):
                if not self.supports_batched_components:
                    raise NotImplementedError("host components (foundation.BatchedComponent) run in the gather-trade-build "
                                              "scenarios only; {!r} is listed in a {} environment".format(cname, self.name))
                if dense_log_frequency is not None:
                    raise NotImplementedError("host components and dense logs do not combine yet (the logged replica's "
                                              "event rows are per launch)")
            elif not int(getattr(ccls, "comp_id", 0)):
                # The registry is open like the reference's (base_component.py:378, registrar.py:48-66), but a component's
                # dynamics are a device kernel here: Python component_step / generate_observations code cannot run
                # inside a batched launch.  Said at construction, not as "unknown component id 0" at the first reset.
                raise NotImplementedError(
                    "component {!r} ({}) has no device kernel: this backend runs the reference's built-in components only "
                    "(Build, ContinuousDoubleAuction, Gather, PeriodicBracketTax, WealthRedistribution, SimpleLabor, "
                    "ControlUSStateOpenCloseStatus, FederalGovernmentSubsidy, VaccinationCampaign) on the device; a "
                    "component of your own subclasses foundation.BatchedComponent (its component_step runs as torch code "
                    "on the state tensors between two launches)".format(cname, ccls.__name__))
            self._register_entities(ccls.required_entities)
            obj = ccls(self.n_agents, self._episode_length, inventory_scale=self.inv_scale, **ckw)
            if obj.name in self._components_dict:
                raise ValueError("component {} listed twice".format(obj.name))
            obj._env = self  # components whose state lives on the device reach it through their environment
            self._components.append(obj)
            self._components_dict[obj.name] = obj
            self._shorthand_lookup[obj.shorthand] = obj
        # host components: (built-in components listed before it, component), in list order
        self._host_components = []
        n_builtin = 0
        for obj in self._components:
            if getattr(obj, "is_batched_host_component", False):
                if obj.get_n_actions("BasicMobileAgent") or obj.get_n_actions("BasicPlanner"):
                    raise NotImplementedError("host component {!r}: action subspaces are the kernels' (get_n_actions must "
                                              "return None)".format(obj.name))
                self._host_components.append((n_builtin, obj))
            else:
                n_builtin += 1
        self._n_builtin_components = n_builtin
        self._host_obs_tables = None

        self._completions = 0
        self._last_ep_metrics = None
        # previous_episode_metrics (base_env.py:763-765) costs a device->host copy of every non-observation state
        # tensor at each full reset that follows an episode: opt in
        self._track_episode_metrics = bool(track_episode_metrics)
        self._backend = None
        self._pending_seed = None if seed is None else int(seed)
        if seed is not None:
            assert isinstance(seed, (int, float)) and int(seed) > 0

    # ---- entity bookkeeping (base_env.py:368-408) ----
    def _register_entities(self, entities):
        for entity in entities:
            if resource_registry.has(entity):
                if entity not in self._entities["resources"]:
                    self._entities["resources"].append(entity)
            elif landmark_registry.has(entity):
                if entity not in self._entities["landmarks"]:
                    self._entities["landmarks"].append(entity)
            elif endogenous_registry.has(entity):
                if entity not in self._entities["endogenous"]:
                    self._entities["endogenous"].append(entity)
            else:
                raise KeyError("Unknown entity: {}".format(entity))

    @property
    def episode_length(self):
        return int(self._episode_length)

    @property
    def inv_scale(self):
        return 0.01 if self._allow_observation_scaling else 1

    @property
    def resources(self):
        return sorted(self._entities["resources"])

    @property
    def landmarks(self):
        return sorted(self._entities["landmarks"])

    @property
    def endogenous(self):
        return sorted(self._entities["endogenous"])

    @property
    def components(self):
        return self._components

    def get_component(self, component_name):
        if component_name not in self._components_dict:
            if component_name not in self._shorthand_lookup:
                raise KeyError(
                    "No component with name or shorthand name {} found; registered "
                    "components are:\n\t".format(component_name)
                    + "\n\t".join(self._components_dict.keys()))
            return self._shorthand_lookup[component_name]
        return self._components_dict[component_name]

    # ---- C-ABI config ----
    def fill_scenario_config(self, cfg):
        raise NotImplementedError

    def layout_planes(self):
        """Returns (stone_src, wood_src, water) uint8 [H, W] (or [E, H, W]) planes."""
        raise NotImplementedError

    def upload_model_constants(self, backend):
        """Hook: named constant tensors a scenario pushes once after the device env exists
        (the counterpart of the reference's get_data_dictionary()).  Default: none."""
        return None

    def build_config(self):
        cfg = _cabi.AieConfig()
        ctypes.memset(ctypes.byref(cfg), 0, ctypes.sizeof(cfg))
        cfg.abi_version = _cabi.ABI_VERSION
        cfg.n_envs = self.n_envs
        cfg.n_agents = self.n_agents
        cfg.world_h, cfg.world_w = int(self.world_size[0]), int(self.world_size[1])
        cfg.episode_length = self._episode_length
        cfg.multi_action_mode_agents = int(self.multi_action_mode_agents)
        cfg.multi_action_mode_planner = int(self.multi_action_mode_planner)
        cfg.allow_observation_scaling = int(self._allow_observation_scaling)
        cfg.dense_log_replicas = 1 if self._create_dense_log_every is not None else 0
        cfg.rng_mode = _cabi.RNG_FAST if self.rng_mode == "fast" else _cabi.RNG_NUMPY
        builtin = [c for c in self._components if not getattr(c, "is_batched_host_component", False)]
        if len(builtin) > _cabi.MAX_COMPONENTS:
            raise ValueError("too many components")
        cfg.n_components = len(builtin)
        for i, c in enumerate(builtin):
            cfg.components[i] = c.comp_id
            c.fill_config(cfg)
        self.fill_scenario_config(cfg)
        return cfg

    # ---- device backend ----
    @property
    def backend(self):
        if self._backend is None:
            from ..env import DeviceBackend  # needs torch + the HIP library; fails loudly

            self._backend = DeviceBackend(self.build_config(), self.layout_planes(),
                                          device=self._device)
            self.upload_model_constants(self._backend)
            if self._pending_seed is not None:
                self._backend.seed(self._pending_seed + self.env_offset)
        return self._backend

    def specialize(self, required=False):
        """Kernels compiled for THIS configuration at run time (aie_specialize: hiprtc, cached on disk), the way the
        build specialises the BASELINE configurations.  True when the environment now runs on them; results are
        bit-identical either way."""
        return self.backend.specialize(required=required)

    def seed(self, seed):
        """Replica e gets the NumPy legacy stream of `np.random.seed(seed + env_offset + e)`
        (reference: BaseEnvironment.seed, base_env.py:481-494)."""
        assert isinstance(seed, (int, float))
        seed = int(seed)
        assert seed > 0
        self._pending_seed = seed
        if self._backend is not None:
            self._backend.seed(seed + self.env_offset)

    def set_rng_state(self, keys, pos):
        """Injects raw MT19937 states (reference: reset/step(seed_state=...))."""
        self.backend.set_rng_state(keys, pos)

    def tensor(self, name):
        return self.backend.tensors[name]

    @property
    def tensors(self):
        return self.backend.tensors

    # axis of the agents' mask tensor that runs over the action entries: the last one ([E, n, A]); the collated COVID
    # observations are [E, 1 + levels, n] (covid19_env.py: masks stacked along axis 0)
    mask_axis_agents = -1

    def _obs(self):
        obs = self._obs_raw()
        host = None
        if self._host_components:
            merged = self._host_observations(obs)
            host = {who: merged[who].pop("_host_keys", {}) for who in ("a", "p")}
            if self._flatten_observations or self.supports_unflattened_observations:
                obs = merged
        if not self._flatten_observations and not self.supports_unflattened_observations:
            obs = self._unflatten(obs)
            if host:  # the unflattened form: every key on its own
                for who in ("a", "p"):
                    obs[who].update(host[who])
        if not self._flatten_masks:
            obs = self._unflatten_masks(obs)
        return obs

    def _unflatten_masks(self, obs):
        """The reference's `flatten_masks=False` form (base_env.py:749-756): obs[...]["action_mask"] is a dictionary
        {"<Component>" or "<Component>.<sub-action>": mask of that subspace}, here zero-copy views of the flattened
        mask tensors (float32 0 / 1 where the reference holds uint8 lists), NO-OP entries left out as in the reference."""
        if self._mask_key_views is None:
            from .obs_keys import mask_keys

            self._mask_key_views = mask_keys(self)
            t = self.backend.tensors
            ax = self.mask_axis_agents
            assert self._mask_key_views["sizes"]["a"] == t["obs_a_action_mask"].shape[ax], "agent mask table vs tensor"
            assert self._mask_key_views["sizes"]["p"] == t["obs_p_action_mask"].shape[-1], "planner mask table vs tensor"
        tab = self._mask_key_views
        out = {who: dict(d) for who, d in obs.items()}
        ma, mp = obs["a"]["action_mask"], obs["p"]["action_mask"]
        out["a"]["action_mask"] = {key: ma.narrow(self.mask_axis_agents, off, size) for key, off, size in tab["a"]}
        out["p"]["action_mask"] = {key: mp.narrow(-1, off, size) for key, off, size in tab["p"]}
        return out

    def _unflatten(self, obs):
        """The reference's `flatten_observations=False` form: every scalar / vector observation under its own key
        (views of the packed vectors, no copies); `time` stays, the 2-D+ arrays and `action_mask` are untouched.
        The planner's per-agent fragments become obs["p"]["agents"][key] = [E, n(, size)]."""
        if self._flat_key_views is None:
            from .obs_keys import flat_keys

            self._flat_key_views = flat_keys(self)
            sizes = self._flat_key_views["sizes"]
            t = self.backend.tensors
            assert sizes["a"] == t["obs_a_flat"].shape[-1] and sizes["p"] == t["obs_p_flat"].shape[-1], \
                "key table does not match the packed vectors"
            if "obs_p_agents" in t:
                assert sizes["pa"] == t["obs_p_agents"].shape[-1]
        tab = self._flat_key_views
        out = {"a": {}, "p": {}}
        for who, flat_name in (("a", "flat"), ("p", "flat")):
            for k, v in obs[who].items():
                if k not in ("flat", "agents"):
                    out[who][k] = v
            flat = obs[who][flat_name]
            for key, off, size, scalar in tab[who]:
                if key == "time":
                    continue
                out[who][key] = flat[..., off] if scalar else flat[..., off:off + size]
        if "agents" in obs["p"]:
            pa = obs["p"]["agents"]
            out["p"]["agents"] = {key: (pa[..., off] if scalar else pa[..., off:off + size])
                                  for key, off, size, scalar in tab["pa"]}
        return out

    # ---- RNG state of one replica in np.random.get_state() form (reset / step(seed_state=...), replay logs) ----
    def rng_state(self, e=0):
        t = self.backend.tensors
        if "mt" not in t:  # a scenario without random draws
            return None
        # (rng_mode "fast": the same 5-tuple with the counter stream's four state words under the name "PHILOX2X32")
        return ("MT19937" if self.rng_mode == "numpy" else "PHILOX2X32", t["mt"][e].cpu().numpy().view(np.uint32).copy(), int(t["mt_pos"][e].item()),
                int(t["mt_has_gauss"][e].item()), float(t["mt_gauss"][e].item()))

    def set_replica_rng_state(self, seed_state, e=0):
        """base_env.py:871-881 / 968-978 for replica e."""
        import torch

        assert isinstance(seed_state, (tuple, list))
        assert len(seed_state) == 5
        t = self.backend.tensors
        if "mt" not in t:
            return
        key = np.array(seed_state[1], dtype=np.uint32)
        if self.rng_mode == "numpy":
            assert key.shape == (624,) and str(seed_state[0]) == "MT19937"
        else:
            assert key.shape == (_cabi.RNG_FAST_STATE_WORDS,) and str(seed_state[0]) == "PHILOX2X32"
        t["mt"][e].copy_(torch.from_numpy(key.view(np.int32).copy()))
        t["mt_pos"][e] = int(seed_state[2])
        t["mt_has_gauss"][e] = int(seed_state[3])
        t["mt_gauss"][e] = float(seed_state[4])

    @property
    def replay_log(self):
        """The (possibly still growing) replay log of the logged replica's current episode."""
        return self._replay_log

    @property
    def previous_episode_replay_log(self):
        """Replay log of the logged replica's most recent completed, logged episode: feeding
        `reset(force_dense_logging=True, **log["reset"])` and `step(**s) for s in log["step"]` to a one-replica
        environment reproduces the episode (base_env.py:455-471)."""
        return self._last_ep_replay_log

    def reset(self, env_mask=None, force_dense_logging=False, seed_state=None):
        """Resets all replicas (or those selected by the uint8/bool device tensor
        `env_mask`, e.g. the `done` tensor) and returns batched observations.
        force_dense_logging: log the coming episode of replica 0 even if it is not one of the
        every-`dense_log_frequency`-th episodes (base_env.py:883-891); needs an environment
        built with dense_log_frequency set (the device event buffer exists only then)."""
        if self._backend is None and self._pending_seed is None:
            # the reference falls back on whatever the global NumPy stream holds;
            # here an unseeded env is seeded from the OS once.
            self._pending_seed = int(np.random.SeedSequence().generate_state(1)[0] % (2 ** 31 - 1)) + 1
        if force_dense_logging and self._create_dense_log_every is None:
            raise ValueError("force_dense_logging needs an environment created with dense_log_frequency")
        if (self._track_episode_metrics and self._backend is not None and env_mask is None
                and bool(self._backend.tensors["done"].all().item())):
            # the episode that just ended, before its state is replaced (the reference stores the metrics when the
            # last step of an episode finishes, base_env.py:763-765; a full reset is the batch's episode boundary)
            self._last_ep_metrics = {k: np.array(v, copy=True) for k, v in self.metrics.items()}
        log_replica_resets = self._create_dense_log_every is not None and (
            env_mask is None or bool(env_mask[0].item()))
        if seed_state is not None:  # the logged replica's stream, before the reset draws from it
            self.set_replica_rng_state(seed_state, 0)
        if log_replica_resets:
            # completed episodes of the logged replica, before this reset (base_env.py:885-891)
            done_eps = int(self.backend.tensors["completions"][0].item()) if self._backend is not None else 0
            self._dense_log_this_episode = bool(force_dense_logging) or done_eps % self._create_dense_log_every == 0
            # the logged replica records event rows (full-featured kernel) only in the episodes that are logged; in the
            # others it steps with the rest of the batch on the fast kernel (include/aie.h: aie_set_dense_log_active)
            self.backend.set_dense_log_active(self._dense_log_this_episode)
        if log_replica_resets:
            self._replay_log = {"reset": dict(seed_state=self.rng_state(0) if self._dense_log_this_episode else None),
                                "step": []}
        self.host_pre_reset(env_mask)
        if self._host_components and env_mask is not None:
            env_mask = env_mask.clone()  # (often the live `done` tensor, which the reset clears; the hooks need it afterwards)
        self.backend.reset(env_mask)
        if self._host_components:
            # The reference resets the components in list order, then the scenario (base_env.py:905-911).  The built-in
            # components' resets all ran inside the reset kernel; the one of them that looks at what a host component may
            # have edited -- PeriodicBracketTax snapshots the agents' coin (redistribution.py:1106-1110) -- takes its
            # snapshot again between the hooks listed ahead of it and those behind it; the utilities the first rewards are
            # measured from and the observations follow at the end (the scenario's own reset steps come last there too).
            tax_at = next((k for k, c in enumerate(b for b in self._components if not getattr(b, "is_batched_host_component", False))
                           if c.name == "PeriodicBracketTax"), None)
            edited_ahead = edited_behind = retaken = False
            for n_before, comp in self._host_components:
                ahead = tax_at is None or n_before <= tax_at  # listed ahead of the tax component (or there is none)
                if not ahead and edited_ahead and not retaken:
                    self.backend.step_range(None, None, 0, 0, _cabi.STEP_OBSERVE | _cabi.STEP_RETAX)
                    retaken = True
                e = bool(comp.additional_reset_steps(self.backend.tensors, env_mask))
                edited_ahead, edited_behind = edited_ahead or (e and ahead), edited_behind or (e and not ahead)
            if edited_ahead or edited_behind:  # the reset kernel's observations no longer show the state: rewrite them
                retax = _cabi.STEP_RETAX if (tax_at is not None and edited_ahead and not retaken) else 0
                self.backend.step_range(None, None, 0, 0, _cabi.STEP_OBSERVE | _cabi.STEP_REBASE | retax)
        if log_replica_resets:
            self._dense_log = {"world": [], "states": [], "actions": [], "rewards": []}
            if self._dense_log_this_episode:
                if self._dense_logger is None:
                    self._dense_logger = self.make_dense_logger()
                self._dense_logger.begin_episode()
                self._dense_log = self._dense_logger.log
        return self._obs()

    def make_dense_logger(self):
        """The object that assembles replica 0's dense log (scenarios with their own state dictionaries override)."""
        from .dense_log import DenseLogger

        return DenseLogger(self, 0)

    @property
    def dense_log(self):
        """The (possibly still growing) dense log of the logged replica's current episode."""
        return self._dense_log

    @property
    def previous_episode_dense_log(self):
        """Dense log of the logged replica's most recent completed, logged episode."""
        return self._last_ep_dense_log

    def action_subspace_names(self):
        """([(name, n_actions)] of the mobile agents, [...] of the planner), in action-index
        order (base_agent.py:116-171: "<Component>" or "<Component>.<sub-action>")."""
        out = []
        for cls in ("BasicMobileAgent", "BasicPlanner"):
            names = []
            for comp in self._components:
                if cls not in comp.agent_subclasses:
                    continue
                n = comp.get_n_actions(cls)
                if n is None or n == 0:
                    continue
                if isinstance(n, int):
                    names.append((comp.name, n))
                else:
                    names.extend(("%s.%s" % (comp.name, sub), int(k)) for sub, k in n)
            out.append(names)
        return out[0], out[1]

    def host_pre_reset(self, env_mask):
        """Hook for scenarios whose reset has a host-side part (e.g. uniform/...: a fresh
        random layout per episode).  Default: nothing."""
        return None

    def _from_reference_actions(self, actions):
        """A reference-style {"0": action or [sub-actions], ..., "p": ...} dictionary (base_env.py:929-945) as the
        batched tensors of a ONE-replica environment; missing actors do nothing."""
        import torch

        be = self.backend
        wa, wp = be.act_a_numel // self.n_agents, max(1, be.act_p_numel)  # (one replica)
        a = np.zeros((1, self.n_agents, wa), np.int32)
        p = np.zeros((1, wp), np.int32)
        for k, v in actions.items():
            v = np.asarray(v, np.int32).reshape(-1)
            if str(k) == "p":
                p[0, :v.size] = v
            else:
                a[0, int(k), :v.size] = v
        return torch.from_numpy(a).to(be.device), torch.from_numpy(p).to(be.device)

    def step(self, actions=None, seed_state=None):
        """actions: None (all NO-OP), or {"a": int32 [E, n_agents(, n_subspaces)],
        "p": int32 [E, n_planner_subspaces]} device tensors; a one-replica environment also takes the reference's
        per-actor dictionary.  seed_state: RNG state to give replica 0 before the step (base_env.py:968-978).
        Returns the batched (obs, rew, done, info) of base_env.py:929-1032."""
        a = p = None
        if seed_state is not None:
            self.set_replica_rng_state(seed_state, 0)
        if actions is not None:
            assert isinstance(actions, dict)
            if self.n_envs == 1 and actions and all(str(k) == "p" or str(k).isdigit() for k in actions) and (
                    any(str(k).isdigit() for k in actions) or not hasattr(actions.get("p"), "data_ptr")):
                a, p = self._from_reference_actions(actions)
                actions = {"a": a, "p": p}
            unknown = [k for k in actions if k not in ("a", "p")]
            if unknown:
                # a reference-style {"0": 3, "1": 0, ..., "p": [...]} dict would silently turn into NO-OPs
                raise ValueError("batched actions are {'a': int32 [n_envs, n_agents(, n_subspaces)], 'p': int32 "
                                 "[n_envs, planner subspaces]}; unexpected keys %r" % (unknown,))
            a = actions.get("a")
            p = actions.get("p")
        logging = self._dense_log_this_episode and self._dense_logger is not None
        if logging:
            self._dense_logger.before_step(a, p)
            self._replay_log["step"].append(dict(actions=self._dense_logger.reference_actions(a, p),
                                                 seed_state=self.rng_state(0)))
        if self._host_components:
            self._step_with_host_components(a, p)
        else:
            self.backend.step(a, p)
        t = self.backend.tensors
        if logging:
            self._dense_logger.after_step()
            if bool(t["done"][0].item()):  # _finalize_logs, base_env.py:763-814
                self._last_ep_dense_log = self._dense_logger.finalize()
                self._last_ep_replay_log = self._replay_log
                self._dense_log_this_episode = False
        rew = {"a": t["rewards_a"], "p": t["rewards_p"]}
        done = {"__all__": t["done"]}
        info = {"a": {}, "p": {}}
        return self._obs(), rew, done, info

    def _step_with_host_components(self, a, p):
        """One step with foundation.BatchedComponent hooks: the built-in components in stretches (aie_step_range), the
        hooks between them in list order (base_env.py:985-987), the end of the step in the last launch."""
        be = self.backend
        if a is None or p is None:  # every launch of the step decodes the same buffers
            za, zp = be._action_buffers(0)
            a = za.zero_() if a is None else a
            p = zp.zero_() if p is None else p
        lo, first = 0, True
        for hi, comp in self._host_components:
            if first or hi > lo:
                be.step_range(a, p, lo, hi, _cabi.STEP_HEAD if first else 0)  # (0: a plain middle stretch)
            first = False
            comp.component_step(be.tensors)
            lo = hi
        be.step_range(a, p, lo, self._n_builtin_components, _cabi.STEP_TAIL | (_cabi.STEP_HEAD if first else 0))

    def _host_observations(self, obs):
        """Adds the host components' observations to the raw observation dict: under "<Component>-<key>" and, with
        flatten_observations, merged into the flat vectors at their sorted-key position (base_env.py:561-612, 644-673:
        scalars and vectors of an actor are concatenated in sorted key order)."""
        import torch

        be = self.backend
        extra = {"a": {}, "p": {}}
        for _, comp in self._host_components:
            o = comp.generate_observations(be.tensors) or {}
            for who in ("a", "p"):
                for k, v in (o.get(who) or {}).items():
                    extra[who]["%s-%s" % (comp.name, k)] = v
        if not extra["a"] and not extra["p"]:
            return obs
        if self._host_obs_tables is None:
            from .obs_keys import flat_keys

            base = flat_keys(self)
            tables = {}
            for who, lead in (("a", 2), ("p", 1)):  # leading dims: [E, n] / [E]
                items = [(key, size, ("k", off)) for key, off, size, _ in base[who]]
                col = base["sizes"][who]
                for key in sorted(extra[who]):
                    v = extra[who][key]
                    size = 1 if v.dim() == lead else int(v.shape[-1])
                    items.append((key, size, ("x", col)))
                    col += size
                perm = []
                for key, size, (_src, off) in sorted(items, key=lambda it: it[0]):
                    perm.extend(range(off, off + size))
                tables[who] = torch.as_tensor(perm, dtype=torch.int64, device=be.device)
            self._host_obs_tables = tables
        out = {who: dict(d) for who, d in obs.items()}
        for who, lead in (("a", 2), ("p", 1)):
            if not extra[who]:
                continue
            parts = [obs[who]["flat"]]
            for key in sorted(extra[who]):
                v = extra[who][key].to(torch.float32)
                parts.append(v.unsqueeze(-1) if v.dim() == lead else v)
            out[who]["flat"] = torch.cat(parts, dim=-1).index_select(-1, self._host_obs_tables[who])
            out[who]["_host_keys"] = {k: extra[who][k] for k in extra[who]}
        return out

    def check_errors(self):
        """Raises what the reference raises from inside step() / reset() for conditions a batched launch can only
        record (tensor `error_flags`, include/aie.h AIE_ERR_*): an action index outside an action space
        (ValueError, F/components/move.py:133-134, build.py:158-159) or a reset that found no free tile for an agent
        (TimeoutError, layout_from_file.py:366-368).  One device->host read; call it where a check is affordable."""
        t = self.backend.tensors
        if "error_flags" not in t:
            return
        flags = t["error_flags"]
        if not bool(flags.any().item()):
            return
        f = flags.cpu().numpy()
        bad = np.nonzero(f)[0]
        e = int(bad[0])
        if f[e] & 4:
            raise TimeoutError("replica %d (and %d more): reset found no free tile for an agent in 200 tries"
                               % (e, bad.size - 1))
        who = "agent" if f[e] & 1 else "planner"
        raise ValueError("replica %d (and %d more): %s action index outside its action space (treated as NO-OP)"
                         % (e, bad.size - 1, who))

    # ---- metrics (base_env.py:420-432) ----
    def scenario_metrics(self, tensors):
        """{metric key: ndarray [E]} from host copies of the state tensors; None if the
        scenario reports nothing (reference: BaseEnvironment.scenario_metrics)."""
        return None

    @property
    def metrics(self):
        """The combined scenario + component metrics, every value an array over the E replicas
        (the reference returns one scalar per key for its single env)."""
        from . import metrics as _metrics

        t = {k: v.cpu().numpy() for k, v in self.backend.tensors.items()
             if not k.startswith("obs_") and not k.startswith("model_") and not k.startswith("saez_global")
             and k not in ("mt", "cells")}
        return _metrics.env_metrics(self, t)

    @property
    def previous_episode_metrics(self):
        """env.metrics as they stood at the end of the last completed episode (captured by the full reset that
        follows it; None before that, for environments that are only ever reset through a mask, and unless the
        environment was created with track_episode_metrics=True)."""
        return self._last_ep_metrics

    def metrics_of(self, e):
        """The reference-shaped metrics dict (scalars) of replica e."""
        return {k: v[e].item() for k, v in self.metrics.items()}

    def as_reference_dicts(self, e):
        """Rebuilds the reference's per-replica observation dict
        ({"0": {...}, ..., "p": {..., "p0": ...}}) for replica e as NumPy arrays -- with
        `flatten_observations=False` in the reference's unflattened form (scalars as floats), with
        `flatten_masks=False` every actor's "action_mask" as the reference's dictionary {"<Component>" or
        "<Component>.<sub-action>": list of uint8} without the NO-OP entries (base_env.py:749-756)."""
        out = self._reference_dicts_flat_masks(e)
        if not self._flatten_masks:
            from .obs_keys import mask_keys

            if self._mask_key_views is None:
                self._mask_key_views = mask_keys(self)
            tab = self._mask_key_views
            t = self.backend.tensors
            ma = t["obs_a_action_mask"][e].cpu().numpy()  # [n, entries]; the collated COVID masks are [entries, n]
            if self.mask_axis_agents != -1:
                ma = ma.T
            mp = t["obs_p_action_mask"][e].cpu().numpy()
            for i in range(self.n_agents):
                out[str(i)]["action_mask"] = {key: ma[i, off:off + size].astype(np.uint8).tolist() for key, off, size in tab["a"]}
            out["p"]["action_mask"] = {key: mp[off:off + size].astype(np.uint8).tolist() for key, off, size in tab["p"]}
        return out

    def _reference_dicts_flat_masks(self, e):
        t = self.backend.tensors
        out = {}
        unflat = not self._flatten_observations and not self.supports_unflattened_observations
        if unflat:
            obs = self._unflatten(self._obs_raw())

            def val(x):
                x = x.cpu().numpy()
                return float(x) if x.ndim == 0 else x

            for i in range(self.n_agents):
                out[str(i)] = {k: val(v[e, i]) for k, v in obs["a"].items()}
            d = {k: val(v[e]) for k, v in obs["p"].items() if k != "agents"}
            for i in range(self.n_agents):
                d["p%d" % i] = {k: val(v[e, i]) for k, v in obs["p"].get("agents", {}).items()}
            out["p"] = d
            return out
        for i in range(self.n_agents):
            d = {}
            for k, v in t.items():
                if k.startswith("obs_a_"):
                    d[k[6:]] = v[e, i].cpu().numpy()
            out[str(i)] = d
        d = {}
        for k, v in t.items():
            if k == "obs_p_agents":
                arr = v[e].cpu().numpy()
                for i in range(self.n_agents):
                    d["p%d" % i] = arr[i]
            elif k.startswith("obs_p_"):
                d[k[6:]] = v[e].cpu().numpy()
        out["p"] = d
        return out

    def _obs_raw(self):
        t = self.backend.tensors
        obs = {"a": {}, "p": {}}
        for k, v in t.items():
            if k.startswith("obs_a_"):
                obs["a"][k[6:]] = v
            elif k.startswith("obs_p_"):
                obs["p"][k[6:]] = v
        return obs


scenario_registry = Registry(BaseEnvironment)
