"""Device backend: owns the arena (a torch uint8 CUDA/HIP tensor), the `aie_env` handle
and zero-copy torch views of every exported tensor.  torch is plumbing here (memory,
streams, distributed); all compute happens in the HIP kernels behind the C ABI.
"""
import ctypes as C

import numpy as np

from . import _cabi, _native

_TORCH_DTYPES = None


def _torch():
    import torch

    global _TORCH_DTYPES
    if _TORCH_DTYPES is None:
        _TORCH_DTYPES = [torch.uint8, torch.int8, torch.int16, torch.int32, torch.int32,
                         torch.float32, torch.float64]  # uint32 exposed as int32 bits
    return torch


class AieError(RuntimeError):
    pass


class DeviceBackend:
    def __init__(self, cfg, layout_planes, device=None):
        torch = _torch()
        if not torch.cuda.is_available():
            raise AieError("no HIP device visible: the batched env has no CPU fallback")
        # AIE_DEV_LIB=1: the -DAIE_DEV build with the development hooks (tools/, a few tests); never the default
        import os

        self.lib = _native.lib(dev=os.environ.get("AIE_DEV_LIB") == "1")
        self.cfg = cfg
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        if self.device.index is None:  # "cuda": the CURRENT device, which is where torch allocates the arena
            self.device = torch.device("cuda", torch.cuda.current_device())
        nbytes = self.lib.aie_arena_bytes(C.byref(cfg))
        if nbytes < 0:
            raise self._err(None, nbytes)
        # Small arenas are torch allocations (the caching allocator hands them out without a device call).  Large ones
        # (AIE_ARENA_VMM_MIN_MB, default 1 GiB, and up) are allocated by the library as a virtual range backed by 64 MiB
        # physical pieces: the store-bound one-step-economy launch over its 7 GB arena runs 14 % faster that way than on
        # one allocation of torch's or hipMalloc's (csrc/aie_capi.hip: aie_arena_alloc, profiles/r04_c5_alloc.json);
        # the library owns that memory until aie_destroy, `self.arena` is a view of it.
        lib_owned = int(nbytes) >= (int(os.environ.get("AIE_ARENA_VMM_MIN_MB", "1024")) << 20) and \
            os.environ.get("AIE_ARENA_PIECE_MB", "64") != "0"
        self.arena = None if lib_owned else torch.zeros(int(nbytes), dtype=torch.uint8, device=self.device)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = self.lib.aie_create(C.byref(cfg), self.device.index, self.arena.data_ptr() if self.arena is not None else None,
                                     int(nbytes), C.byref(h))
        if rc != 0:
            raise self._err(None, rc)
        self.handle = h
        try:
            self._finish_init(cfg, layout_planes, lib_owned, nbytes)
        except BaseException:
            # nothing else knows the handle yet: destroy it (and a library-owned arena with it) instead of leaking it.
            # Exactly once: an owner object, once it exists, destroys the handle in its __del__ -- which may
            # run right here or later, when the last view of the arena goes -- so it is disarmed first and the one
            # aie_destroy call is this one.
            owner = getattr(self, "_arena_owner", None)
            if owner is not None:
                owner.handle = None
            self._arena_owner = None
            self.tensors = {}
            self.arena = None
            try:
                self.lib.aie_destroy(self.handle)
            finally:
                self.handle = None
            raise

    def _finish_init(self, cfg, layout_planes, lib_owned, nbytes):
        torch = _torch()
        if lib_owned:
            d0 = _cabi.AieTensorDesc()
            self.lib.aie_tensor_at(self.handle, 0, C.byref(d0))
            base = int(d0.data) - int(d0.arena_offset)

            class _Arena:
                """The library's allocation as a CUDA array (torch wraps it without copying and keeps this object alive
                for as long as any view of the arena lives): it owns the environment handle, so the memory is released
                -- aie_destroy -- when the last tensor that points into it is gone, not when the backend is closed."""
                __cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (base, False), "version": 2}

                def __init__(self, lib, handle):
                    self.lib, self.handle = lib, handle

                def __del__(self):
                    try:
                        if self.handle:
                            self.lib.aie_destroy(self.handle)
                    except Exception:
                        pass

            self._arena_owner = _Arena(self.lib, self.handle)
            self.arena = torch.as_tensor(self._arena_owner, device=self.device)
        self.E = cfg.n_envs
        self.n = cfg.n_agents
        self.descs = {}
        self.tensors = {}
        d = _cabi.AieTensorDesc()
        for i in range(self.lib.aie_num_tensors(self.handle)):
            self.lib.aie_tensor_at(self.handle, i, C.byref(d))
            name = d.name.decode()
            self.descs[name] = (d.dtype, tuple(d.shape[: d.ndim]), tuple(d.stride[: d.ndim]),
                                d.arena_offset)
            self.tensors[name] = self._view(*self.descs[name])
        stone_src, wood_src, water = [np.ascontiguousarray(p, np.uint8) for p in layout_planes]
        self._check(self.lib.aie_set_layout(self.handle, stone_src.ctypes.data,
                                            wood_src.ctypes.data, water.ctypes.data))
        self._rand_a = None
        self._rand_p = None
        # element counts the kernels index (aie_kernels.hip: decode_actions reads aa[(e*n + i) * act_a_width + s] and
        # ap[e * act_p_width + b]): checked on every call, a wrongly shaped buffer would be read out of bounds
        wa = 1 if not cfg.multi_action_mode_agents else max(1, self._n_sub_a())
        self.act_a_numel = self.E * self.n * wa
        self.act_p_numel = self.E * self._act_p_width_for(cfg)

    # ---- plumbing ----
    def _err(self, handle, rc):
        msg = self.lib.aie_last_error(handle)
        msg = msg.decode() if msg else ""
        if rc == _cabi.E_NOTFOUND:
            return KeyError(msg)
        if rc == _cabi.E_INVALID:
            return ValueError(msg)
        if rc == _cabi.E_UNSUPPORTED:
            return NotImplementedError(msg)
        return AieError("aie error %d: %s" % (rc, msg))

    def _check(self, rc):
        if rc != 0:
            raise self._err(self.handle, rc)

    def _view(self, dtype, shape, stride, offset):
        torch = _torch()
        tdt = _TORCH_DTYPES[dtype]
        es = torch.empty(0, dtype=tdt).element_size()
        base = self.arena[offset:]
        usable = (base.numel() // es) * es
        typed = base[:usable].view(tdt)
        assert all(s % es == 0 for s in stride)
        return torch.as_strided(typed, shape, tuple(s // es for s in stride))

    def _stream(self):
        return C.c_void_p(_torch().cuda.current_stream(self.device).cuda_stream)

    def _ptr(self, t, dtype, what, numel=None):
        torch = _torch()
        if t is None:
            return None
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(np.asarray(t), device=self.device)
        if numel is not None and t.numel() != numel:
            raise ValueError("%s has %d elements (shape %s), the environment needs %d: actions_a is int32 "
                             "[n_envs, n_agents(, n_subspaces)], actions_p int32 [n_envs, planner subspaces], env_mask "
                             "uint8 [n_envs]" % (what, t.numel(), tuple(t.shape), numel))
        if t.device != self.device:
            t = t.to(self.device)
        if t.dtype != dtype:
            t = t.to(dtype)
        t = t.contiguous()
        self._keep = getattr(self, "_keep", [])
        self._keep = self._keep[-8:] + [t]  # keep alive until the stream has consumed it
        return C.c_void_p(t.data_ptr())

    # ---- API ----
    def seed(self, base_seed):
        """Replica e gets the stream of seed base_seed + e: np.random.seed(base_seed + e) (rng_mode "numpy"), or the
        counter stream keyed by base_seed + e (rng_mode "fast": aie_seed_fast, 48 bits of the seed are used)."""
        if self.cfg.rng_mode == _cabi.RNG_FAST:
            self._check(self.lib.aie_seed_fast(self.handle, C.c_uint64(base_seed & 0xFFFFFFFFFFFFFFFF), C.c_int64(0), self._stream()))
        else:
            self._check(self.lib.aie_seed(self.handle, C.c_uint32(base_seed & 0xFFFFFFFF), self._stream()))

    def set_rng_state(self, keys, pos):
        words = _cabi.RNG_FAST_STATE_WORDS if self.cfg.rng_mode == _cabi.RNG_FAST else _cabi.MT_N
        keys = np.ascontiguousarray(keys, np.uint32).reshape(self.E, words)
        pos = np.ascontiguousarray(pos, np.int32).reshape(self.E)
        self._check(self.lib.aie_set_rng_state(self.handle, keys.ctypes.data, pos.ctypes.data))

    def reset(self, env_mask=None):
        torch = _torch()
        m = self._ptr(env_mask, torch.uint8, "env_mask", self.E)
        self._check(self.lib.aie_reset(self.handle, m, self._stream()))

    def step(self, actions_a=None, actions_p=None):
        torch = _torch()
        a = self._ptr(actions_a, torch.int32, "actions_a", self.act_a_numel)
        p = self._ptr(actions_p, torch.int32, "actions_p", self.act_p_numel)
        self._check(self.lib.aie_step(self.handle, a, p, self._stream()))

    def step_range(self, actions_a, actions_p, comp_lo, comp_hi, phases):
        """aie_step_range: the built-in components [comp_lo, comp_hi) and the named parts of a step (_cabi.STEP_HEAD /
        STEP_TAIL / STEP_OBSERVE) -- what foundation.BatchedComponent hooks run between."""
        torch = _torch()
        a = self._ptr(actions_a, torch.int32, "actions_a", self.act_a_numel)
        p = self._ptr(actions_p, torch.int32, "actions_p", self.act_p_numel)
        self._check(self.lib.aie_step_range(self.handle, a, p, int(comp_lo), int(comp_hi), int(phases), self._stream()))

    def set_reward_log(self, n_slots):
        """Allocates a reward log of `n_slots` step slots, f32 [n_slots, E, n_agents + 2] = (agent rewards,
        planner reward, done), and makes every following step fill the next slot (slot 0 first, wrapping).
        n_slots = 0 switches it off.  Returns the log tensor (or None)."""
        torch = _torch()
        if not n_slots:
            self._check(self.lib.aie_set_reward_log(self.handle, None, 0))
            self.reward_log = None
            return None
        self.reward_log = torch.zeros((int(n_slots), self.E, self.n + 2), dtype=torch.float32, device=self.device)
        self._check(self.lib.aie_set_reward_log(self.handle, C.c_void_p(self.reward_log.data_ptr()), int(n_slots)))
        return self.reward_log

    def rewind_reward_log(self):
        """The next step fills slot 0 of the current reward log again."""
        if self.reward_log is not None:
            self._check(self.lib.aie_set_reward_log(self.handle, C.c_void_p(self.reward_log.data_ptr()),
                                                    int(self.reward_log.shape[0])))

    def set_auto_reset(self, on=True):
        """Replicas restart inside / right behind the step that ends their episode (include/aie.h:
        aie_set_auto_reset): `done` and the rewards are the terminal step's, state and observations the new episode's."""
        self._check(self.lib.aie_set_auto_reset(self.handle, 1 if on else 0))

    def set_dense_log_active(self, on=True):
        """Dense-log replicas record event rows (and run the full-featured kernel) only while an episode is being
        logged (include/aie.h: aie_set_dense_log_active); otherwise they step with the rest of the batch."""
        self._check(self.lib.aie_set_dense_log_active(self.handle, 1 if on else 0))

    def step_sample_next(self, actions_a, actions_p, seed, env_offset=0, next_slot=1, masked=False):
        """One launch: step with (actions_a, actions_p) and fill the action buffers of `next_slot`
        with the uniform random policy's next draw (same values as sample_random_actions; masked=True, COVID only:
        as sample_masked_actions, from the masks this step writes)."""
        torch = _torch()
        a = self._ptr(actions_a, torch.int32, "actions_a", self.act_a_numel)
        p = self._ptr(actions_p, torch.int32, "actions_p", self.act_p_numel)
        na, np_ = self._action_buffers(next_slot)
        fn = self.lib.aie_step_sample_next_masked if masked else self.lib.aie_step_sample_next
        self._check(fn(
            self.handle, a, p, C.c_uint64(seed), C.c_int64(env_offset),
            C.c_void_p(na.data_ptr()), C.c_void_p(np_.data_ptr()), self._stream()))
        return na, np_

    def arena_info(self):
        """{"bytes", "allocator": "caller" (a torch tensor) | "hipMalloc" | "vmm", "piece_mib"} -- aie_arena_info."""
        b, a, pc = C.c_int64(), C.c_int32(), C.c_int64()
        self._check(self.lib.aie_arena_info(self.handle, C.byref(b), C.byref(a), C.byref(pc)))
        name = _cabi.ARENA_ALLOCATORS[a.value]
        return {"bytes": int(b.value), "allocator": "torch" if name == "caller" else name, "piece_mib": int(pc.value) >> 20}

    def specialize(self, required=False):
        """aie_specialize: kernels compiled for this configuration at run time (hiprtc, cached).  Returns True when the
        environment now runs on specialised kernels; False (or, with required=True, an exception) when that is not
        possible here -- the generic kernel keeps running, with identical results."""
        rc = self.lib.aie_specialize(self.handle)
        if rc != 0 and required:
            raise self._err(self.handle, rc)
        return rc == 0

    def sample_masked_actions(self, seed, env_offset=0, slot=0):
        """Like sample_random_actions, but every sub-action is drawn uniformly among the
        entries the current `action_mask` observations allow."""
        self.sample_random_actions  # noqa: B018  (buffers are shared)
        a, p = self._action_buffers(slot)
        self._check(self.lib.aie_sample_masked_actions(
            self.handle, C.c_uint64(seed), C.c_int64(env_offset),
            C.c_void_p(a.data_ptr()), C.c_void_p(p.data_ptr()), self._stream()))
        return a, p

    def sample_policy_actions(self, logits_a, logits_p, seed, env_offset=0, slot=0, out=None):
        """Categorical sampling from the caller's policy logits under the current action masks (aie_sample_policy_actions:
        Gumbel-max in one launch, replayable from a captured graph).  logits_a: float32 [E, n, MA] in the layout of the
        agents' flattened action mask (COVID: [E, n, 1 + levels]), logits_p: float32 [E, MP]; either may be None.
        Returns the action buffers (int32 [E, n, width], [E, width_p]) -- `out=(a, p)` to fill the caller's own."""
        torch = _torch()
        a, p = out if out is not None else self._action_buffers(slot)
        la = lp = None
        if logits_a is not None:
            per_agent = self.tensors["obs_a_action_mask"].numel() // (self.E * self.n)
            la = self._ptr(logits_a, torch.float32, "logits_a", self.E * self.n * per_agent)
        if logits_p is not None:
            lp = self._ptr(logits_p, torch.float32, "logits_p", self.tensors["obs_p_action_mask"].numel())
        self._check(self.lib.aie_sample_policy_actions(
            self.handle, la, lp, C.c_uint64(seed), C.c_int64(env_offset),
            C.c_void_p(a.data_ptr()) if logits_a is not None else None,
            C.c_void_p(p.data_ptr()) if logits_p is not None else None, self._stream()))
        return a, p

    def _action_buffers(self, slot):
        torch = _torch()
        if self._rand_a is None:
            wa = 1 if not self.cfg.multi_action_mode_agents else max(1, self._n_sub_a())
            wp = self._act_p_width()
            self._rand_a = [torch.zeros((self.E, self.n, wa), dtype=torch.int32, device=self.device) for _ in range(2)]
            self._rand_p = [torch.zeros((self.E, wp), dtype=torch.int32, device=self.device) for _ in range(2)]
        return self._rand_a[slot], self._rand_p[slot]

    def sample_random_actions(self, seed, env_offset=0, slot=0):
        """Fills (and returns) caller-owned action buffers with the benchmark's uniform random
        policy.  `slot` selects one of two buffer pairs so that the actions of step t+1 can be
        sampled (on another stream) while step t still reads its own."""
        a, p = self._action_buffers(slot)
        self._check(self.lib.aie_sample_random_actions(
            self.handle, C.c_uint64(seed), C.c_int64(env_offset),
            C.c_void_p(a.data_ptr()), C.c_void_p(p.data_ptr()), self._stream()))
        return a, p

    def _n_sub_a(self):
        comps = list(self.cfg.components)[: self.cfg.n_components]
        return sum({_cabi.COMP_BUILD: 1, _cabi.COMP_CDA: 4, _cabi.COMP_GATHER: 1,
                    _cabi.COMP_SIMPLE_LABOR: 1}.get(c, 0) for c in comps)

    def _act_p_width(self):
        return self._act_p_width_for(self.cfg)

    @staticmethod
    def _act_p_width_for(cfg):
        has_planner_actions = (
            _cabi.COMP_TAX in list(cfg.components)[: cfg.n_components]
            and cfg.tax_model == 0 and not cfg.tax_disable)
        if cfg.multi_action_mode_planner and has_planner_actions:
            return cfg.tax_n_brackets
        return 1

    def upload(self, name, array):
        dtype, shape, _, _ = self.descs[name]
        arr = np.ascontiguousarray(np.broadcast_to(np.asarray(array), shape), np.dtype(_cabi.DTYPES[dtype]))
        self._check(self.lib.aie_upload(self.handle, name.encode(), arr.ctypes.data, arr.nbytes))

    def download(self, name):
        dtype, shape, _, _ = self.descs[name]
        arr = np.empty(shape, np.dtype(_cabi.DTYPES[dtype]))
        self._check(self.lib.aie_download(self.handle, name.encode(), arr.ctypes.data, arr.nbytes))
        return arr

    def invalidate_observations(self, e=None):
        """Call after editing state tensors from outside the kernels: the next step rewrites the
        map observations in full instead of updating them in place (`obs_valid`)."""
        if "obs_valid" in self.tensors:
            if e is None:
                self.tensors["obs_valid"].zero_()
            else:
                self.tensors["obs_valid"][e] = 0

    def load_state(self, state, e=None):
        """Injects a host state ({field: array without env dim}) into replica e (or all)."""
        torch = _torch()
        self.invalidate_observations(e)
        for k, v in state.items():
            if k in ("stone_src", "wood_src", "water"):
                continue
            if k not in self.tensors:
                continue
            t = self.tensors[k]
            arr = np.asarray(v)
            if k == "mt":
                arr = arr.astype(np.uint32).view(np.int32)
            src = torch.as_tensor(arr, device=self.device).to(t.dtype)
            if e is None:
                t[...] = src
            else:
                t[e] = src
        if "stone_src" in state:
            fl = (np.asarray(state["water"], np.uint8) + 2 * np.asarray(state["stone_src"], np.uint8)
                  + 4 * np.asarray(state["wood_src"], np.uint8)).astype(np.uint8)
            src = torch.as_tensor(fl, device=self.device)
            if self.cfg.shared_layout and self.cfg.layout_gen == _cabi.LAYOUT_FIXED and self.cfg.rng_mode == _cabi.RNG_FAST:
                # one source layout for the whole batch (the regeneration's source list is derived from it once,
                # aie_set_layout): a state that carries another layout cannot be injected into some replicas only
                if not bool((self.tensors["cell_flags"][0 if e is None else e] == src).all()):
                    raise ValueError("load_state: the state's source / water planes differ from the environment's shared "
                                     "layout (shared_layout=1): create the environment with that layout")
            if e is None:
                self.tensors["cell_flags"][...] = src
            else:
                self.tensors["cell_flags"][e] = src
            if "regen_src_list" in self.tensors:
                # the regeneration's source doubles (record fields the reset kernel derives from the flags,
                # csrc/aie_layout.h: o_src_list): Wood cells ascending, then Stone cells
                hw = int(np.asarray(state["wood_src"]).size)
                d = np.concatenate([np.flatnonzero(np.asarray(state["wood_src"]).reshape(-1)),
                                    hw + np.flatnonzero(np.asarray(state["stone_src"]).reshape(-1))])
                cap = int(self.tensors["regen_src_list"].shape[-1])
                lst = np.zeros(cap, np.int16)
                lst[: min(cap, d.size)] = d[:cap].astype(np.uint16).view(np.int16)
                lst_t = torch.as_tensor(lst, device=self.device)
                if e is None:
                    self.tensors["regen_src_list"][...] = lst_t
                    self.tensors["regen_src_n"][...] = int(d.size)
                else:
                    self.tensors["regen_src_list"][e] = lst_t
                    self.tensors["regen_src_n"][e] = int(d.size)
            if "regen_source_count" in self.tensors:
                # bookkeeping the reset kernel derives from the layout (source blocks per regeneration window,
                # csrc/aie_layout.h: regen_conv): a layout injected here needs the same counts
                cnt = self._source_window_counts(np.asarray(state["stone_src"]), np.asarray(state["wood_src"]))
                src = torch.as_tensor(cnt, device=self.device)
                if e is None:
                    self.tensors["regen_source_count"][...] = src
                else:
                    self.tensors["regen_source_count"][e] = src

    def _source_window_counts(self, stone_src, wood_src):
        out = []
        for plane, hw in ((stone_src, int(self.cfg.regen_halfwidth[0])), (wood_src, int(self.cfg.regen_halfwidth[1]))):
            p = np.pad((np.asarray(plane) > 0).astype(np.int32), hw)
            H, W = np.asarray(plane).shape
            acc = np.zeros((H, W), np.int32)
            for dr in range(2 * hw + 1):
                for dc in range(2 * hw + 1):
                    acc += p[dr: dr + H, dc: dc + W]
            out.append(acc)
        return np.stack(out).astype(np.uint8)

    def close(self):
        """Releases the environment.  A library-owned arena (1 GiB and up) is freed when the LAST tensor view of it is
        gone -- action buffers and tensors the caller still holds keep it alive; `free_now()` does not wait for them."""
        if getattr(self, "handle", None):
            if getattr(self, "_arena_owner", None) is not None:
                # the library owns the arena: the handle is destroyed when the last view of the arena is gone
                self._arena_owner = None
                self.arena = None
                self.tensors = {}
                self._rand_a = self._rand_p = None
            else:
                self.lib.aie_destroy(self.handle)
            self.handle = None

    def free_now(self):
        """close() + the memory back NOW (large environments created back to back: a stale view of a library-owned
        7 GB arena would otherwise keep it until the garbage collector finds it).  Views the caller still holds dangle
        afterwards -- use only when none are."""
        import gc

        owner = getattr(self, "_arena_owner", None)
        self.close()
        if owner is not None:
            try:
                _torch().cuda.synchronize(self.device)
            except Exception:
                pass
            h, owner.handle = owner.handle, None  # (the owner's __del__ then has nothing left to destroy)
            if h:
                self.lib.aie_destroy(h)
        gc.collect()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
