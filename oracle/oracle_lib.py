"""Test infrastructure ONLY: ctypes wrapper of oracle/aie_oracle.c (the CPU
restatement).  Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg -- never by the product package.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from ai_economist_amd import _cabi  # noqa: E402

_LIB = None
NP_DTYPES = [np.uint8, np.int8, np.int16, np.int32, np.uint32, np.float32, np.float64]


def build():
    subprocess.run(["make", "-s", "-C", HERE], check=True)
    return os.path.join(HERE, "_build", "libaie_oracle.so")


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, "_build", "libaie_oracle.so")
        src = os.path.join(HERE, "aie_oracle.c")
        hdr = os.path.join(ROOT, "ai-economist_amd", "csrc", "aie_layout.h")
        if (not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src)
                or os.path.getmtime(path) < os.path.getmtime(hdr)):
            path = build()
        L = C.CDLL(path)
        vp = C.c_void_p
        L.aie_oracle_params.restype = C.c_int
        L.aie_oracle_params.argtypes = [C.POINTER(_cabi.AieConfig), vp, vp, C.c_char_p, C.c_int]
        L.aie_oracle_sizeof_params.restype = C.c_int
        L.aie_oracle_sizeof_table.restype = C.c_int
        L.aie_oracle_step.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int]
        L.aie_oracle_step.restype = None
        L.aie_oracle_step_mt.argtypes = [vp, vp, vp, vp, C.c_int]
        L.aie_oracle_step_mt.restype = None
        L.aie_oracle_saez_period_start.argtypes = [vp, vp]
        L.aie_oracle_saez_period_start.restype = None
        L.aie_oracle_reset.argtypes = [vp, vp, vp, C.c_int, C.c_int]
        L.aie_oracle_reset.restype = None
        L.aie_oracle_seed.argtypes = [vp, vp, C.c_uint32]
        L.aie_oracle_seed.restype = None
        L.aie_oracle_seed64.argtypes = [vp, vp, C.c_uint64]
        L.aie_oracle_seed64.restype = None
        L.aie_oracle_philox2x32_10.argtypes = [vp, C.c_uint32, vp]
        L.aie_oracle_philox2x32_10.restype = None
        L.aie_oracle_layout_stream.argtypes = [vp, vp]
        L.aie_oracle_layout_stream.restype = None
        L.aie_oracle_sample_policy_actions.argtypes = [vp, vp, vp, vp, C.c_uint64, C.c_int64, vp, vp]
        L.aie_oracle_sample_policy_actions.restype = None
        L.aie_oracle_sampler_expf.argtypes = [C.c_float]
        L.aie_oracle_sampler_expf.restype = C.c_float
        L.aie_oracle_sampler_uniform.argtypes = [C.c_uint32]
        L.aie_oracle_sampler_uniform.restype = C.c_float
        L.aie_oracle_sample_row.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint32]
        L.aie_oracle_sample_row.restype = C.c_int32
        L.aie_oracle_sampler_entry_rng.argtypes = [C.c_uint32, C.c_uint32]
        L.aie_oracle_sampler_entry_rng.restype = C.c_uint32
        _LIB = L
    return _LIB


class TensorTable(C.Structure):
    _fields_ = [("n", C.c_int32), ("_pad", C.c_int32), ("t", _cabi.AieTensorDesc * _cabi.MAX_TENSORS)]


class OracleEnv:
    """E replicas stepped on the CPU by the C restatement, in a NumPy arena with the
    device's exact byte layout."""

    def __init__(self, cfg, layout_planes=None):
        L = lib()
        self.cfg = cfg
        assert L.aie_oracle_sizeof_table() == C.sizeof(TensorTable), (
            L.aie_oracle_sizeof_table(), C.sizeof(TensorTable))
        self._params = C.create_string_buffer(L.aie_oracle_sizeof_params())
        self._table = TensorTable()
        err = C.create_string_buffer(256)
        rc = L.aie_oracle_params(C.byref(cfg), self._params, C.byref(self._table), err, 256)
        if rc != 0:
            raise ValueError("aie_build_params failed (%d): %s" % (rc, err.value.decode()))
        self.E = cfg.n_envs
        self.n = cfg.n_agents
        self.descs = {}
        arena_bytes = 0
        for i in range(self._table.n):
            d = self._table.t[i]
            self.descs[d.name.decode()] = d
        L.aie_oracle_arena_bytes.restype = C.c_int64
        arena_bytes = int(L.aie_oracle_arena_bytes(self._params))
        self.arena = np.zeros(arena_bytes, np.uint8)
        self.t = {name: self._view(d) for name, d in self.descs.items()}
        if "house_owner" in self.t:
            self.t["house_owner"][...] = -1
        if "saez_elas" in self.t:  # elas_t = elas_tm1 = 0.5 at construction (redistribution.py:263-266)
            self.t["saez_elas"][:, 0:2] = 0.5
        if layout_planes is not None and "cell_flags" in self.t:
            self.set_layout(*layout_planes)
        for name, arr in getattr(cfg, "_model_tensors", {}).items():  # constants that travel with the configuration
            self.t[name][...] = np.asarray(arr).reshape(self.t[name].shape)

    def set_global_saez_buffer(self, pairs):
        """PeriodicBracketTax.set_global_saez_buffer (redistribution.py:530-533) for every replica of this arena."""
        pairs = np.asarray(pairs, np.float64).reshape(-1, 2)
        self.t["saez_global_buffer"][0, : len(pairs)] = pairs
        self.t["saez_global_len"][...] = len(pairs)

    def saez_period_start(self):
        """Runs PeriodicBracketTax's period-start rate update (tax_model "saez") on every replica."""
        lib().aie_oracle_saez_period_start(self._params, self.arena.ctypes.data_as(C.c_void_p))

    def _view(self, d):
        dt = np.dtype(NP_DTYPES[d.dtype])
        shape = tuple(d.shape[i] for i in range(d.ndim))
        strides = tuple(d.stride[i] for i in range(d.ndim))
        return np.ndarray(shape=shape, dtype=dt, buffer=self.arena.data,
                          offset=d.arena_offset, strides=strides)

    def set_layout(self, stone_src, wood_src, water):
        fl = (np.asarray(water, np.uint8) * 1 + np.asarray(stone_src, np.uint8) * 2
              + np.asarray(wood_src, np.uint8) * 4).astype(np.uint8)
        self.t["cell_flags"][...] = fl  # broadcasts [H,W] -> [E,H,W]

    def load_state(self, state, e=None):
        """state: {field: array without env dim}; e=None broadcasts to all replicas."""
        for k, v in state.items():
            if k in ("stone_src", "wood_src", "water"):
                continue
            if k not in self.t:
                continue
            if e is None:
                self.t[k][...] = v
            else:
                self.t[k][e] = v
        if "stone_src" in state:
            fl = (np.asarray(state["water"], np.uint8) + 2 * np.asarray(state["stone_src"], np.uint8)
                  + 4 * np.asarray(state["wood_src"], np.uint8)).astype(np.uint8)
            if e is None:
                self.t["cell_flags"][...] = fl
            else:
                self.t["cell_flags"][e] = fl

    def seed(self, base_seed):
        lib().aie_oracle_seed64(self._params, self.arena.ctypes.data, base_seed)  # (rng_mode "numpy": the low 32 bits)

    def sample_policy_actions(self, logits_a, logits_p, seed, env_offset=0, width_a=1, width_p=1):
        """aie_sample_policy_actions on this arena's masks: (actions_a int32 [E, n, width_a], actions_p int32 [E, width_p];
        the widths are the action buffers' -- sub-actions per agent / planner)."""
        la = np.ascontiguousarray(logits_a, np.float32)
        lp = np.ascontiguousarray(logits_p, np.float32)
        a = np.zeros((self.E, self.n, int(width_a)), np.int32)
        p = np.zeros((self.E, int(width_p)), np.int32)
        lib().aie_oracle_sample_policy_actions(self._params, self.arena.ctypes.data, la.ctypes.data, lp.ctypes.data, seed,
                                               env_offset, a.ctypes.data, p.ctypes.data)
        return a, p

    def reset(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8).ctypes.data
        lib().aie_oracle_reset(self._params, self.arena.ctypes.data, m, 0, self.E)

    def step(self, actions_a=None, actions_p=None, nthreads=1):
        a = None if actions_a is None else np.ascontiguousarray(actions_a, np.int32)
        p = None if actions_p is None else np.ascontiguousarray(actions_p, np.int32)
        pa = None if a is None else a.ctypes.data
        pp = None if p is None or p.size == 0 else p.ctypes.data
        if nthreads > 1:
            lib().aie_oracle_step_mt(self._params, self.arena.ctypes.data, pa, pp, nthreads)
        else:
            lib().aie_oracle_step(self._params, self.arena.ctypes.data, pa, pp, 0, self.E)
