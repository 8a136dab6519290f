/* Bit-exact restatement of glibc 2.35's double-precision pow() and exp() as the x86-64 FMA build executes them
 * (sysdeps/ieee754/dbl-64/e_pow.c, e_exp.c; multiarch variants __ieee754_pow_fma / __ieee754_exp_fma, which every
 * x86-64 CPU with FMA3 — all EPYC hosts of MI355X nodes — selects).
 *
 * Why: the reference computes utilities with libm (`coin ** (1 - eta)`, F/scenarios/utils/rewards.py:40;
 * `1 - exp(-v / c)`, layout_from_file.py:255-267), and `_auto_warmup_integrator += (mean agent reward > 0)`
 * (layout_from_file.py:553-557) is an INTEGER state field decided by the sign of a ~1e-16 float. The device's own
 * libm (ocml) is 1-2 ulp off glibc, which let that counter drift (round-1 VERDICT, weak #1). These functions follow
 * glibc's published algorithm (ARM optimized-routines) operation by operation, including which multiply-adds the
 * distribution's build FUSED (gcc -mfma with fp-contract=fast): the operation sequence below was read off the
 * disassembly of libm-2.35.a:e_pow-fma.o / e_exp-fma.o. Every multiply-add is therefore written explicitly
 * (`__builtin_fma` where fused, contraction switched off otherwise). Tables: aie_glibc_tables.h (generated).
 *
 * Compiles as device code under hipcc and as plain C under gcc (-mfma -ffp-contract=off) — the CPU test
 * (tests/test_glibc_math.py) checks this very header against libm on millions of inputs; a GPU test does the same
 * for the device build.
 */
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define AIE_GLIBC_FN static __device__ __forceinline__
#define AIE_GLIBC_TABLE static __device__ const
#define AIE_GLIBC_NOCONTRACT _Pragma("clang fp contract(off)")
#else
#define AIE_GLIBC_FN static inline
#define AIE_GLIBC_TABLE static const
#define AIE_GLIBC_NOCONTRACT /* gcc: build with -ffp-contract=off */
#endif
#include "aie_glibc_tables.h"

AIE_GLIBC_FN double aie_gl_f64(uint64_t u) {
  union { uint64_t u; double d; } c;
  c.u = u;
  return c.d;
}
AIE_GLIBC_FN uint64_t aie_gl_u64(double d) {
  union { uint64_t u; double d; } c;
  c.d = d;
  return c.u;
}

/* exp's subnormal / overflow tail (e_exp.c: specialcase). */
AIE_GLIBC_FN double aie_gl_exp_special(double tmp, uint64_t sbits, uint64_t ki) {
  AIE_GLIBC_NOCONTRACT
  if ((ki & 0x80000000ull) == 0) { /* k > 0 */
    sbits -= 1009ull << 52;
    const double scale = aie_gl_f64(sbits);
    return 0x1p1009 * __builtin_fma(scale, tmp, scale);
  }
  sbits += 1022ull << 52;
  const double scale = aie_gl_f64(sbits);
  const double st = scale * tmp;
  double y = scale + st;
  if (y < 1.0) {
    double lo = scale - y;
    lo = lo + st;
    const double hi = 1.0 + y;
    double t = 1.0 - hi;
    t = t + y;
    t = t + lo;
    y = (t + hi) - 1.0;
    if (y == 0.0) y = 0.0;
  }
  return 0x1p-1022 * y;
}

/* 2^(k/N) * exp(r) core shared by exp() and pow(); `xtail` is pow's low word of y*log(x). */
AIE_GLIBC_FN double aie_gl_exp_core(double x, double xtail, int with_tail, uint32_t abstop) {
  AIE_GLIBC_NOCONTRACT
  const double InvLn2N = aie_gl_f64(aie_glibc_exp_head[0]), Shift = aie_gl_f64(aie_glibc_exp_head[1]);
  const double NegLn2hiN = aie_gl_f64(aie_glibc_exp_head[2]), NegLn2loN = aie_gl_f64(aie_glibc_exp_head[3]);
  const double C2 = aie_gl_f64(aie_glibc_exp_head[4]), C3 = aie_gl_f64(aie_glibc_exp_head[5]);
  const double C4 = aie_gl_f64(aie_glibc_exp_head[6]), C5 = aie_gl_f64(aie_glibc_exp_head[7]);
  double kd = __builtin_fma(x, InvLn2N, Shift);
  const uint64_t ki = aie_gl_u64(kd);
  kd = kd - Shift;
  double r = __builtin_fma(kd, NegLn2hiN, x);
  r = __builtin_fma(kd, NegLn2loN, r);
  if (with_tail) r = xtail + r;
  const uint32_t idx = 2u * (uint32_t)(ki & 127u);
  const uint64_t top = ki << 45;
  const double tail = aie_gl_f64(aie_glibc_exp_tab[idx]);
  const uint64_t sbits = aie_glibc_exp_tab[idx + 1] + top;
  const double q23 = __builtin_fma(r, C3, C2);
  const double s = r + tail;
  const double r2 = r * r;
  const double q45 = __builtin_fma(r, C5, C4);
  const double t = __builtin_fma(q23, r2, s);
  const double r4 = r2 * r2;
  const double tmp = __builtin_fma(r4, q45, t);
  if (abstop == 0) return aie_gl_exp_special(tmp, sbits, ki);
  const double scale = aie_gl_f64(sbits);
  return __builtin_fma(scale, tmp, scale);
}

/* exp(x) for finite x (e_exp.c: __exp). NaN / inf inputs are outside the callers' domain and return x + 1. */
AIE_GLIBC_FN double aie_exp_glibc(double x) {
  AIE_GLIBC_NOCONTRACT
  uint32_t abstop = (uint32_t)(aie_gl_u64(x) >> 52) & 0x7ffu;
  if (abstop - 0x3c9u >= 0x3fu) {
    if ((int32_t)(abstop - 0x3c9u) < 0) return 1.0 + x;      /* |x| < 2^-54 */
    if (abstop >= 0x409u) {                                   /* |x| >= 1024 */
      if (abstop == 0x7ffu) return 1.0 + x;
      return (aie_gl_u64(x) >> 63) ? 0.0 : aie_gl_f64(0x7ff0000000000000ull);
    }
    abstop = 0; /* 512 <= |x| < 1024: result may over/underflow, handled in the special tail */
  }
  return aie_gl_exp_core(x, 0.0, 0, abstop);
}

/* pow(x, y) for x >= 0 finite and y > 0 finite, 2^-65 <= y < 2^63 (the utilities' domain: coin >= 0, 0 < 1 - eta <= 1).
 * Anything else returns NaN so that a caller outside the domain is found at once. */
AIE_GLIBC_FN double aie_pow_glibc(double x, double y) {
  AIE_GLIBC_NOCONTRACT
  uint64_t ix = aie_gl_u64(x);
  const uint64_t iy = aie_gl_u64(y);
  const uint32_t topx = (uint32_t)(ix >> 52), topy = (uint32_t)(iy >> 52);
  if ((topy & 0x7ffu) - 0x3beu >= 0x80u || (iy >> 63)) return aie_gl_f64(0x7ff8000000000000ull);
  if (topx - 1u >= 0x7feu) {
    if (ix == 0) return 0.0;                                   /* pow(+0, y > 0) */
    if (topx != 0) return aie_gl_f64(0x7ff8000000000000ull);   /* negative, inf, nan: outside the domain */
    ix = aie_gl_u64(x * 0x1p52);                               /* subnormal x */
    ix &= 0x7fffffffffffffffull;
    ix -= 52ull << 52;
  }
  /* log_inline (e_pow.c:60-124) */
  const double Ln2hi = aie_gl_f64(aie_glibc_powlog_head[0]), Ln2lo = aie_gl_f64(aie_glibc_powlog_head[1]);
  const double A0 = aie_gl_f64(aie_glibc_powlog_head[2]), A1 = aie_gl_f64(aie_glibc_powlog_head[3]);
  const double A2 = aie_gl_f64(aie_glibc_powlog_head[4]), A3 = aie_gl_f64(aie_glibc_powlog_head[5]);
  const double A4 = aie_gl_f64(aie_glibc_powlog_head[6]), A5 = aie_gl_f64(aie_glibc_powlog_head[7]);
  const double A6 = aie_gl_f64(aie_glibc_powlog_head[8]);
  const uint64_t tmp = ix - 0x3fe6955500000000ull;
  const uint32_t i = (uint32_t)(tmp >> 45) & 127u;
  const int64_t k = (int64_t)tmp >> 52;
  const uint64_t iz = ix - (tmp & (0xfffull << 52));
  const double z = aie_gl_f64(iz), kd = (double)k;
  const double invc = aie_gl_f64(aie_glibc_powlog_tab[3 * i]);
  const double logc = aie_gl_f64(aie_glibc_powlog_tab[3 * i + 1]);
  const double logctail = aie_gl_f64(aie_glibc_powlog_tab[3 * i + 2]);
  const double t1 = __builtin_fma(kd, Ln2hi, logc);
  const double r = __builtin_fma(z, invc, -1.0);
  const double ar = r * A0;
  const double lo1 = __builtin_fma(kd, Ln2lo, logctail);
  const double q1 = __builtin_fma(r, A2, A1);
  const double q2 = __builtin_fma(r, A4, A3);
  const double t2 = r + t1;
  const double ar2 = r * ar;
  const double d12 = t1 - t2;
  const double ar3 = r * ar2;
  const double lo3 = __builtin_fma(ar, r, -ar2);
  const double lo2 = d12 + r;
  const double q3 = __builtin_fma(r, A6, A5);
  const double hi = t2 + ar2;
  const double d2h = t2 - hi;
  const double q4 = __builtin_fma(q3, ar2, q2);
  const double lo4 = d2h + ar2;
  const double q5 = __builtin_fma(ar2, q4, q1);
  double lo = lo1 + lo2;
  lo = lo + lo3;
  lo = lo + lo4;
  lo = __builtin_fma(ar3, q5, lo);
  const double lhi = hi + lo;
  double llo = hi - lhi;
  llo = llo + lo;
  /* pow (e_pow.c:345-363) */
  const double ehi = y * lhi;
  const double c = __builtin_fma(lhi, y, -ehi);
  const double elo = __builtin_fma(y, llo, c);
  uint32_t abstop = (uint32_t)(aie_gl_u64(ehi) >> 52) & 0x7ffu;
  if (abstop - 0x3c9u >= 0x3fu) {
    if ((int32_t)(abstop - 0x3c9u) < 0) return 1.0 + ehi;    /* |y log x| < 2^-54 (x == 1 lands here) */
    if (abstop >= 0x409u) return (aie_gl_u64(ehi) >> 63) ? 0.0 : aie_gl_f64(0x7ff0000000000000ull);
    abstop = 0;
  }
  return aie_gl_exp_core(ehi, elo, 1, abstop);
}

/* log(x) (e_log.c: __log; operation order read off libm-2.35.a:e_log-fma.o). Used by the legacy NumPy
 * distributions behind the skill draws (legacy_gauss, standard_exponential -> pareto, lognormal:
 * numpy/random/src/legacy/legacy-distributions.c), whose results become build payments, i.e. coin. */
AIE_GLIBC_FN double aie_log_glibc(double x) {
  AIE_GLIBC_NOCONTRACT
  uint64_t ix = aie_gl_u64(x);
  if (ix - 0x3fee000000000000ull <= 0x308ffffffffffull) { /* 1 - 2^-4 <= x < 1 + 0x1.09p-4 */
    if (ix == 0x3ff0000000000000ull) return 0.0;
    const double B0 = aie_gl_f64(aie_glibc_log_head[7]), B1 = aie_gl_f64(aie_glibc_log_head[8]);
    const double B2 = aie_gl_f64(aie_glibc_log_head[9]), B3 = aie_gl_f64(aie_glibc_log_head[10]);
    const double B4 = aie_gl_f64(aie_glibc_log_head[11]), B5 = aie_gl_f64(aie_glibc_log_head[12]);
    const double B6 = aie_gl_f64(aie_glibc_log_head[13]), B7 = aie_gl_f64(aie_glibc_log_head[14]);
    const double B8 = aie_gl_f64(aie_glibc_log_head[15]), B9 = aie_gl_f64(aie_glibc_log_head[16]);
    const double B10 = aie_gl_f64(aie_glibc_log_head[17]);
    const double r = x - 1.0;
    double p1 = __builtin_fma(r, B2, B1);
    double p4 = __builtin_fma(r, B5, B4);
    const double r2 = r * r;
    double p7 = __builtin_fma(r, B8, B7);
    p1 = __builtin_fma(r2, B3, p1);
    p4 = __builtin_fma(r2, B6, p4);
    const double r3 = r * r2;
    p7 = __builtin_fma(r2, B9, p7);
    p7 = __builtin_fma(r3, B10, p7);
    double P = __builtin_fma(p7, r3, p4);
    P = __builtin_fma(P, r3, p1);
    const double rw = __builtin_fma(r, 0x1p27, r);
    const double rhi = __builtin_fma(-0x1p27, r, rw);
    const double rhi2 = rhi * rhi;
    const double rlo = r - rhi;
    const double hi = __builtin_fma(rhi2, B0, r);
    const double d = r - hi;
    const double s = r + rhi;
    double lo = __builtin_fma(rhi2, B0, d);
    const double t = B0 * rlo;
    lo = __builtin_fma(t, s, lo);
    const double y = __builtin_fma(P, r3, lo);
    return hi + y;
  }
  const uint32_t top = (uint32_t)(ix >> 48);
  if (top - 0x0010u >= 0x7ff0u - 0x0010u) {
    if (ix * 2 == 0) return -aie_gl_f64(0x7ff0000000000000ull);              /* log(0) = -inf */
    if (ix == 0x7ff0000000000000ull) return x;                                 /* log(inf) */
    if ((top & 0x8000u) || (top & 0x7ff0u) == 0x7ff0u) return aie_gl_f64(0x7ff8000000000000ull);
    ix = aie_gl_u64(x * 0x1p52);                                               /* subnormal */
    ix -= 52ull << 52;
  }
  const double Ln2hi = aie_gl_f64(aie_glibc_log_head[0]), Ln2lo = aie_gl_f64(aie_glibc_log_head[1]);
  const double A0 = aie_gl_f64(aie_glibc_log_head[2]), A1 = aie_gl_f64(aie_glibc_log_head[3]);
  const double A2 = aie_gl_f64(aie_glibc_log_head[4]), A3 = aie_gl_f64(aie_glibc_log_head[5]);
  const double A4 = aie_gl_f64(aie_glibc_log_head[6]);
  const uint64_t tmp = ix - 0x3fe6000000000000ull;
  const uint32_t i = (uint32_t)(tmp >> 45) & 127u;
  const int64_t k = (int64_t)tmp >> 52;
  const uint64_t iz = ix - (tmp & (0xfffull << 52));
  const double invc = aie_gl_f64(aie_glibc_log_tab[2 * i]), logc = aie_gl_f64(aie_glibc_log_tab[2 * i + 1]);
  const double z = aie_gl_f64(iz), kd = (double)k;
  const double r = __builtin_fma(z, invc, -1.0);
  const double w = __builtin_fma(kd, Ln2hi, logc);
  const double q12 = __builtin_fma(r, A2, A1);
  const double hi = r + w;
  const double r2 = r * r;
  double lo = w - hi;
  lo = lo + r;
  lo = __builtin_fma(kd, Ln2lo, lo);
  const double r3 = r * r2;
  double q34 = __builtin_fma(r, A4, A3);
  lo = __builtin_fma(r2, A0, lo);
  q34 = __builtin_fma(q34, r2, q12);
  const double y = __builtin_fma(r3, q34, lo);
  return y + hi;
}
