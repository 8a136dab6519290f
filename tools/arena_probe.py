#!/usr/bin/env python
"""Development probe: the one-step-economy kernel runs at two speeds (1.45 vs 1.67 ms per launch) with the same binary
on the same box in the same process, depending on where its 7 GB arena lies.
   python tools/arena_probe.py fresh [count]     a fresh arena from torch's allocator per environment
   python tools/arena_probe.py offsets [bytes..] the arena at chosen offsets inside ONE 8 GB + 1 GiB allocation
   python tools/arena_probe.py contiguous [count]  arenas from hipExtMallocWithFlags(hipDeviceMallocContiguous)
   python tools/arena_probe.py hipmalloc|vmm|vmm2m [count]  arenas from tools/arena_alloc.hip (hipMalloc; hipMemCreate +
                                                 hipMemMap in one piece / in 128 MiB pieces of 2 MiB granules)
   python tools/arena_probe.py procs [N] [out.json]  VERDICT r3 #7: every allocator in N FRESH processes each (what
                                                 matters is the placement a process gets at its first allocation), the
                                                 launch times per allocator -> JSON
GPU only."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
from helpers import make_env  # noqa: E402

WL = os.environ.get("PROBE_WL", "C5")
SLACK = 1 << 30
keep = []
big = None
real_zeros = torch.zeros


class Blob:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def run(mode, off=0, launches=40, warm=15):
    def zeros(n, *a, dtype=None, device=None, **kw):
        global big
        if dtype == torch.uint8 and isinstance(n, int) and n > (1 << 26):
            if mode == "offsets":
                if big is None:
                    big = real_zeros(n + SLACK, dtype=torch.uint8, device=device)
                v = big[off: off + n]
                v.zero_()
                return v
            if mode in ("hipmalloc", "vmm", "vmm2m"):
                lib = ctypes.CDLL(os.path.join(ROOT, "tools", "bin", "libarena_alloc.so"))
                lib.arena_alloc.restype = ctypes.c_void_p
                lib.arena_alloc.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t)]
                gran = ctypes.c_size_t(0)
                ptr = lib.arena_alloc({"hipmalloc": 0, "vmm": 2, "vmm2m": 3}[mode], n, 0, ctypes.byref(gran))
                if ptr:
                    t = torch.as_tensor(Blob(ptr, n), device=device)
                    t.zero_()
                    return t
                print("   arena_alloc(%s) failed" % mode, flush=True)
            if mode == "contiguous":
                hip = ctypes.CDLL("libamdhip64.so")
                hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
                ptr = ctypes.c_void_p()
                rc = hip.hipExtMallocWithFlags(ctypes.byref(ptr), n, 0x4)  # hipDeviceMallocContiguous
                if rc == 0:
                    t = torch.as_tensor(Blob(ptr.value, n), device=device)
                    t.zero_()
                    return t
                print("   hipExtMallocWithFlags(contiguous) failed: rc", rc, flush=True)
        return real_zeros(n, *a, dtype=dtype, device=device, **kw)

    torch.zeros = zeros
    try:
        W = bench.WORKLOADS[WL]
        env = make_env(W["cfg"](), n_envs=W["envs"], device="cuda:0")
        env.seed(1)
        env.reset()  # (the backend, and with it the arena, is created here)
    finally:
        torch.zeros = real_zeros
    be = env.backend
    if WL == "C5":
        be.lib.aie_set_auto_reset(be.handle, 1)
    for _ in range(warm):
        a, p = be.sample_random_actions(1234)
        be.step(a, p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a, p = be.sample_random_actions(1234)
    e0.record()
    for _ in range(launches):
        be.step(a, p)
    e1.record()
    torch.cuda.synchronize()
    print("%s %-10s arena at %#x (offset %d)  %.4f ms per launch" % (WL, mode, be.arena.data_ptr(), off,
                                                                      e0.elapsed_time(e1) / launches), flush=True)
    if mode == "fresh":
        keep.append(env)  # (holding the previous arenas makes the allocator hand out new addresses)


mode = sys.argv[1] if len(sys.argv) > 1 else "fresh"
if mode == "pieces":  # the VMM arena in pieces of several sizes, N fresh processes each
    import json
    import re
    import subprocess

    n_procs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    res = {}
    for mb in (2, 16, 64, 128, 256, 1024):
        times = []
        for k in range(n_procs):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "vmm2m", "1"], capture_output=True, text=True, timeout=600,
                               env=dict(os.environ, ARENA_PIECE_MB=str(mb)))
            m = re.search(r"([0-9.]+) ms per launch", r.stdout)
            times.append(float(m.group(1)) if m else None)
        res["%d MiB pieces" % mb] = times
        print(mb, times, flush=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r04_c5_alloc_pieces.json"), "w"), indent=1)
    sys.exit(0)
if mode == "procs":
    import json
    import re
    import subprocess

    n_procs = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    out_path = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", "r04_c5_alloc.json")
    res = {}
    for alloc in ("fresh", "hipmalloc", "contiguous", "vmm", "vmm2m"):
        times = []
        for k in range(n_procs):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), alloc, "1"], capture_output=True, text=True, timeout=600)
            m = re.search(r"([0-9.]+) ms per launch", r.stdout)
            times.append(float(m.group(1)) if m else None)
            print(alloc, k, times[-1], (r.stderr.strip().splitlines() or [""])[-1][:120] if not m else "", flush=True)
        ok = [t for t in times if t is not None]
        res[{"fresh": "torch caching allocator"}.get(alloc, alloc)] = dict(
            ms_per_launch=times, best=min(ok) if ok else None, worst=max(ok) if ok else None,
            at_or_below_1p50=sum(1 for t in ok if t <= 1.50), processes=n_procs)
    res["workload"] = "%s: one fresh process per sample, 15 warm-up + 40 timed launches each (tools/arena_probe.py procs)" % WL
    json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps(res, indent=1))
    sys.exit(0)
if mode == "offsets":
    for off in [int(x, 0) for x in sys.argv[2:]] or [0, 256, 4096, 1 << 16, 1 << 20, 1 << 21, 3 << 21, 1 << 24, 1 << 27, 1 << 29, 0]:
        run(mode, off)
else:
    for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
        run(mode)
