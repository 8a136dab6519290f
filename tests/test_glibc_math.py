"""csrc/aie_glibc_math.h (the utilities' pow / exp) must equal the host's libm bit for bit: the reference computes
`coin ** (1 - eta)` and `exp(-v / c)` with libm (F/scenarios/utils/rewards.py:40, layout_from_file.py:255-267) and an
integer state field depends on the sign of the result's last bits (layout_from_file.py:553-557).

CPU: the header compiled as plain C against libm on millions of inputs.  GPU: the device build of the same header."""
import ctypes
import os
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ai-economist_amd", "csrc")

SHIM = r"""
#include "aie_glibc_math.h"
#include <math.h>
#include <string.h>
long check_pow(const double* x, const double* y, long n, double* first_bad) {
  long bad = 0;
  for (long i = 0; i < n; ++i) {
    const double a = pow(x[i], y[i]), b = aie_pow_glibc(x[i], y[i]);
    if (memcmp(&a, &b, 8)) { if (!bad) { first_bad[0] = x[i]; first_bad[1] = y[i]; } ++bad; }
  }
  return bad;
}
long check_log(const double* x, long n, double* first_bad) {
  long bad = 0;
  for (long i = 0; i < n; ++i) {
    const double a = log(x[i]), b = aie_log_glibc(x[i]);
    if (memcmp(&a, &b, 8)) { if (!bad) first_bad[0] = x[i]; ++bad; }
  }
  return bad;
}
long check_exp(const double* x, long n, double* first_bad) {
  long bad = 0;
  for (long i = 0; i < n; ++i) {
    const double a = exp(x[i]), b = aie_exp_glibc(x[i]);
    if (memcmp(&a, &b, 8)) { if (!bad) first_bad[0] = x[i]; ++bad; }
  }
  return bad;
}
"""


def _inputs(n, seed):
    rng = np.random.RandomState(seed)
    q = n // 4
    x = np.concatenate([
        np.exp((rng.rand(q) - 0.5) * 80.0),                      # wide log-uniform coin
        rng.rand(q) * 2000.0,                                     # realistic coin
        np.floor(rng.rand(q) * 500.0) + np.floor(rng.rand(q) * 8) / 8.0,  # exact binary fractions
        np.exp((rng.rand(n - 3 * q) - 0.5) * 1400.0),             # the whole double range
    ])
    y = np.concatenate([
        rng.rand(q), np.full(q, 0.77), 1.0 - np.floor(rng.rand(q) * 100) / 100.0, rng.rand(n - 3 * q)])
    y = np.where(y <= 0, 0.5, y)
    special = np.array([0.0, 1.0, 2.0, 10.0, 5e-324, 2.0 ** -1060, 1e308, 0.1, 1.0 - 2.0 ** -53])
    x = np.concatenate([x, special, special])
    y = np.concatenate([y, np.full(len(special), 0.77), np.full(len(special), 1.0)])
    xe = np.concatenate([(rng.rand(n // 3) - 0.5) * 1500.0, -rng.rand(n // 3) * 40.0,
                         (rng.rand(n // 3) - 0.5) * 2.0 ** (-(rng.rand(n // 3) * 70).astype(int)),
                         np.array([0.0, -0.0, -745.2, -708.4, 709.7, -1e-300, 1e-300])])
    return np.ascontiguousarray(x), np.ascontiguousarray(y), np.ascontiguousarray(xe)


def _log_inputs(n, seed):
    rng = np.random.RandomState(seed)
    q = n // 4
    return np.ascontiguousarray(np.concatenate([
        rng.rand(q), 1.0 - rng.rand(q), 0.9 + 0.2 * rng.rand(q), np.exp((rng.rand(n - 3 * q) - 0.5) * 1480.0),
        np.array([1.0, 5e-324, 2.0 ** -1040, 1e308, 0.9375, 1.064697265625, 2.0, 10.0])]))


def test_header_equals_host_libm_bit_for_bit():
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "shim.c")
        open(src, "w").write(SHIM)
        so = os.path.join(d, "shim.so")
        subprocess.run(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-shared", "-fPIC", "-I" + CSRC, src, "-o", so, "-lm"],
                       check=True)
        lib = ctypes.CDLL(so)
        dp = ctypes.POINTER(ctypes.c_double)
        lib.check_pow.restype = ctypes.c_long
        lib.check_pow.argtypes = [dp, dp, ctypes.c_long, dp]
        lib.check_exp.restype = ctypes.c_long
        lib.check_exp.argtypes = [dp, ctypes.c_long, dp]
        x, y, xe = _inputs(4_000_000, 7)
        bad = np.zeros(2)
        nb = lib.check_pow(x.ctypes.data_as(dp), y.ctypes.data_as(dp), len(x), bad.ctypes.data_as(dp))
        assert nb == 0, "pow(%r, %r) differs from libm (%d of %d)" % (bad[0], bad[1], nb, len(x))
        nb = lib.check_exp(xe.ctypes.data_as(dp), len(xe), bad.ctypes.data_as(dp))
        assert nb == 0, "exp(%r) differs from libm (%d of %d)" % (bad[0], nb, len(xe))
        xl = _log_inputs(4_000_000, 5)
        lib.check_log.restype = ctypes.c_long
        lib.check_log.argtypes = [dp, ctypes.c_long, dp]
        nb = lib.check_log(xl.ctypes.data_as(dp), len(xl), bad.ctypes.data_as(dp))
        assert nb == 0, "log(%r) differs from libm (%d of %d)" % (bad[0], nb, len(xl))


@pytest.mark.gpu
def test_device_pow_exp_equal_host_libm_bit_for_bit():
    import math

    import torch

    from ai_economist_amd import _native

    lib = _native.lib(dev=True)  # the hook lives in the -DAIE_DEV build only
    vp = ctypes.c_void_p
    lib.aie_test_glibc_math.restype = ctypes.c_int
    lib.aie_test_glibc_math.argtypes = [ctypes.c_int, vp, vp, vp, ctypes.c_int64, vp]
    x, y, xe = _inputs(1_000_000, 11)
    dev = torch.device("cuda:0")
    tx, ty = torch.as_tensor(x, device=dev), torch.as_tensor(y, device=dev)
    out = torch.empty_like(tx)
    assert lib.aie_test_glibc_math(0, tx.data_ptr(), ty.data_ptr(), out.data_ptr(), len(x), None) == 0
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    want = np.array([math.pow(a, b) for a, b in zip(x.tolist(), y.tolist())])  # Python float ** -> libm pow
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), \
        "%d of %d device pow results differ from libm" % ((got.view(np.uint64) != want.view(np.uint64)).sum(), len(x))
    te = torch.as_tensor(xe, device=dev)
    oute = torch.empty_like(te)
    assert lib.aie_test_glibc_math(1, te.data_ptr(), None, oute.data_ptr(), len(xe), None) == 0
    torch.cuda.synchronize()
    got = oute.cpu().numpy()

    def host_exp(a):  # math.exp -> libm exp (NumPy's own AVX-512 exp is a different algorithm)
        try:
            return math.exp(a)
        except OverflowError:
            return np.inf

    want = np.array([host_exp(a) for a in xe.tolist()])
    bad = np.nonzero(got.view(np.uint64) != want.view(np.uint64))[0]
    assert bad.size == 0, "%d of %d device exp results differ from libm, e.g. exp(%r): %r vs %r" % (
        bad.size, len(xe), xe[bad[0]], got[bad[0]], want[bad[0]])
    xl = _log_inputs(1_000_000, 13)
    tl = torch.as_tensor(xl, device=dev)
    outl = torch.empty_like(tl)
    assert lib.aie_test_glibc_math(2, tl.data_ptr(), None, outl.data_ptr(), len(xl), None) == 0
    torch.cuda.synchronize()
    got = outl.cpu().numpy()
    want = np.array([math.log(a) for a in xl.tolist()])  # math.log -> libm log
    bad = np.nonzero(got.view(np.uint64) != want.view(np.uint64))[0]
    assert bad.size == 0, "%d of %d device log results differ from libm, e.g. log(%r): %r vs %r" % (
        bad.size, len(xl), xl[bad[0]], got[bad[0]], want[bad[0]])
