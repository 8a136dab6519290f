// PeriodicBracketTax, tax_model == "saez": the Saez-formula rate update of a period start
// (reference: F/components/redistribution.py:437-513 compute_and_set_new_period_rates_from_saez_formula,
// :548-596 estimate_uniform_income_elasticity, :598-752 get_binned_saez_welfare_weight_and_pareto_params,
// :754-790 get_saez_marginal_rates, :792-823 bracketize_schedule).
//
// The formula reads nothing the coming step produces (sample buffer, elasticity estimates, running
// average all date from earlier tax days), so it runs in its own launch BEFORE the step kernel:
// one wavefront per replica; replicas whose tax_cycle_pos != 1 or whose buffer has not reached
// _buffer_size samples leave at once.  Result: `next_rates` in the replica's Saez block
// (aie_layout.h: a_saez), which the step kernel's tax component copies into the record at the period
// start (or draws np.random.uniform rates instead while the buffer is short); elasticity estimates and
// the running average are updated here.  f64 throughout; the 2x2 OLS normal equations are solved in
// closed form (the reference calls np.linalg.inv: agreement ~1e-12 relative, stated in DESIGN.md).
#pragma once
#include "aie_kernels.hip"

namespace aie {

__device__ __forceinline__ double saez_pareto(const aie_params& P, double z) {  // :636-643
  if (P.c.saez_pareto_weight_uniform) return 1.0;
  return 1.0 / (z > 1.0 ? z : 1.0);
}
__device__ __forceinline__ double saez_clip01(double x) { return x < 0 ? 0 : (x > 1 ? 1 : x); }
// sum over the wave, fixed order (lane 0 + lane 1 + ...): every lane gets the same value
__device__ __forceinline__ double wave_sum_ordered(double v) {
  double s = 0;
  for (int j = 0; j < AIE_NT; ++j) s += bcast(v, j);
  return s;
}

}  // namespace aie

#define AIE_SAEZ_T AIE_SAEZ_BINS

extern "C" __global__ void __launch_bounds__(AIE_NT)
aie_saez_kernel(const aie_params* __restrict__ params, uint8_t* __restrict__ arena) {
  using namespace aie;
  const aie_params& P = *params;
  const int e = (int)blockIdx.x, lane = (int)threadIdx.x, T = AIE_SAEZ_T, NB = P.NB;
  uint8_t* blk = arena + P.a_saez + (int64_t)e * P.saez_stride;
  const uint8_t* rec = arena + P.a_records + (int64_t)e * P.rec_bytes;
  int32_t* hdr = reinterpret_cast<int32_t*>(blk);
  if (*reinterpret_cast<const int32_t*>(rec + P.o_tax_cycle_pos) != 1) return;
  // the `saez_buffer` property :514-525: the trainer's global buffer (if one was set) followed by the samples this
  // replica added since the buffers were last reset, else the local buffer
  const int llen = hdr[0];
  const uint8_t* gblk = arena + P.a_saez_global;
  const int glen = P.saez_global_cap ? *reinterpret_cast<const int32_t*>(gblk) : 0;
  const int tail = glen > 0 ? (hdr[2] < llen ? hdr[2] : llen) : llen;  // local samples in use
  const int len = glen + tail;
  if (!hdr[1]) {  // :444-449
    if (len < P.c.saez_buffer_size) return;
    if (lane == 0) hdr[1] = 1;
  }
  double* el = reinterpret_cast<double*>(blk + AIE_SAEZ_OFF_ELAS);  // elas_t, elas_tm1, log_z0_t, log_z0_tm1
  double* avg = reinterpret_cast<double*>(blk + AIE_SAEZ_OFF_AVG);
  double* next = reinterpret_cast<double*>(blk + AIE_SAEZ_OFF_NEXT);
  const double* lbuf = reinterpret_cast<const double*>(blk + AIE_SAEZ_OFF_BUF) + 2 * (llen - tail);
  const double* gbuf = reinterpret_cast<const double*>(gblk + 16);
  auto smp = [&](int k, int j) { return k < glen ? gbuf[2 * k + j] : lbuf[2 * (k - glen) + j]; };
  const double* edges = P.saez_edges;

  __shared__ int s_counts[AIE_SAEZ_T];
  __shared__ double s_pz[AIE_SAEZ_T + 1], s_dens[AIE_SAEZ_T + 1], s_g[AIE_SAEZ_T + 1], s_az[AIE_SAEZ_T + 1],
      s_taus[AIE_SAEZ_T + 1], s_bt[AIE_SAEZ_T + 1];

  // ---- estimate_uniform_income_elasticity :548-596: samples with z > 0 and tau < 1 ----
  const double elas_tm1 = el[0], log_z0_tm1 = el[2];
  double elas_t = elas_tm1, log_z0_t = log_z0_tm1;
  {
    double cnt = 0, st = 0;
    for (int k = lane; k < len; k += AIE_NT) {
      const double z = smp(k, 0), tau = smp(k, 1);
      if (z > 0 && tau < 1) { cnt += 1; st += tau; }
    }
    const double m = wave_sum_ordered(cnt);
    if (m >= 10) {
      const double mean = wave_sum_ordered(st) / m;
      double sq = 0;
      for (int k = lane; k < len; k += AIE_NT) {
        const double z = smp(k, 0), tau = smp(k, 1);
        if (z > 0 && tau < 1) sq += (tau - mean) * (tau - mean);
      }
      const double sd = sqrt(wave_sum_ordered(sq) / m);
      if (!(sd < 1e-6)) {
        double sxx = 0, sx = 0, sxy = 0, sy = 0;
        for (int k = lane; k < len; k += AIE_NT) {
          const double z = smp(k, 0), tau = smp(k, 1);
          if (z > 0 && tau < 1) {
            double t1 = 1 - tau; if (t1 < 1e-9) t1 = 1e-9;
            double zz = z; if (zz < 1e-9) zz = 1e-9;
            const double x = log(t1), y = log(zz);
            sxx += x * x; sx += x; sxy += x * y; sy += y;
          }
        }
        sxx = wave_sum_ordered(sxx); sx = wave_sum_ordered(sx);
        sxy = wave_sum_ordered(sxy); sy = wave_sum_ordered(sy);
        const double det = sxx * m - sx * sx;
        const double i00 = m / det, i01 = -sx / det, i11 = sxx / det;
        const double elas = i00 * sxy + i01 * sy;
        log_z0_t = i01 * sxy + i11 * sy;
        elas_t = ((1 - 0.98) * (elas > 0.0 ? elas : 0.0)) + (0.98 * elas_tm1);
      }
    }
  }
  if (lane == 0) { el[1] = elas_tm1; el[3] = log_z0_tm1; el[0] = elas_t; el[2] = log_z0_t; }
  if (P.c.saez_fixed_elas_given) elas_t = P.c.saez_fixed_elas;

  // ---- np.histogram(incomes, bins=edges): [e_i, e_i+1), the last bin closed on the right ----
  for (int i = lane; i < T; i += AIE_NT) s_counts[i] = 0;
  __syncthreads();
  double n_below = 0, n_above = 0, w_above = 0, sum_above = 0;
  for (int k = lane; k < len; k += AIE_NT) {
    const double z = smp(k, 0);
    if (z < edges[0]) n_below += 1;
    else if (z > edges[T]) { n_above += 1; w_above += saez_pareto(P, z); sum_above += z; }
    else {
      int lo = 0, hi = T;  // largest i with edges[i] <= z
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (edges[mid] <= z) lo = mid; else hi = mid; }
      atomicAdd(&s_counts[lo], 1);
    }
  }
  n_below = wave_sum_ordered(n_below); n_above = wave_sum_ordered(n_above);
  w_above = wave_sum_ordered(w_above); sum_above = wave_sum_ordered(sum_above);
  __syncthreads();
  // pareto(max(z, 0)) == 1 for every income below the first edge (0), with either weight type
  const double w_below = n_below;
  double part = 0, cnt_part = 0;
  for (int i = lane; i < T; i += AIE_NT) {
    const double pb = (double)s_counts[i] * saez_pareto(P, 0.5 * (edges[i] + edges[i + 1]));
    s_dens[i] = pb;  // unnormalised for now
    part += pb; cnt_part += (double)s_counts[i];
  }
  const double norm = ((wave_sum_ordered(part) + w_below) + w_above) + 1e-9;
  const double n_total = wave_sum_ordered(cnt_part) + n_below + n_above;
  __syncthreads();
  for (int i = lane; i < T; i += AIE_NT) { s_dens[i] = s_dens[i] / norm; s_pz[i] = (double)s_counts[i] / n_total; }
  if (lane == 0) { s_dens[T] = w_above / norm; s_pz[T] = n_above / n_total; }
  __syncthreads();
  // ---- sequential scans (every lane runs them on the same LDS values: uniform) ----
  {  // weight / probability of incomes >= z, then the bin-centred g(z) :687-697
    double cd = 0, cp = 0;
    for (int i = T; i >= 0; --i) {
      cd = (i == T) ? s_dens[i] : cd + s_dens[i];
      cp = (i == T) ? s_pz[i] : cp + s_pz[i];
      if (lane == 0) s_g[i] = cd / (cp + 1e-9);
    }
  }
  {  // compute_binned_a_distribution :700-744
    double cum_pz = s_pz[0] + n_below / n_total;
    for (int i = 0; i < T; ++i) {
      if (i > 0) cum_pz = saez_clip01(cum_pz + s_pz[i]);
      if (lane == 0) {
        const double p_geq = 1 - cum_pz + (0.5 * s_pz[i]);
        double a = NAN;
        if (s_pz[i] != 0) {
          const double z = 0.5 * (edges[i] + edges[i + 1]);
          a = z * s_pz[i] / (saez_clip01(p_geq) + 1e-9) / (edges[i + 1] - edges[i]);
        }
        s_az[i] = a;
      }
    }
    if (lane == 0) {
      double a = 0.0;
      if (n_above > 0) { const double mean_above = sum_above / n_above; a = mean_above / (mean_above - edges[T] + 1e-9); }
      s_az[T] = a;
    }
  }
  __syncthreads();
  for (int i = lane; i <= T; i += AIE_NT) {  // get_saez_marginal_rates :754-757
    const double gz = i < T ? 0.5 * (s_g[i] + s_g[i + 1]) : s_g[T];
    s_taus[i] = (1.0 - gz) / (1.0 - gz + s_az[i] * elas_t + 1e-9);
  }
  __syncthreads();
  if (lane == 0) {  // gaps (bins without incomes): np.linspace between the neighbouring real rates :759-788
    double last_rate = 0.0;
    int last_idx = -1;
    for (int i = 0; i <= T; ++i) {
      const double tau = s_taus[i];
      if (tau != tau) continue;
      if (i - last_idx > 1) {
        const int gap = i - last_idx - 1;
        const double step = (tau - last_rate) / (double)(gap + 1);
        for (int j = 1; j <= gap; ++j) s_taus[last_idx + j] = (double)j * step + last_rate;
      }
      last_rate = tau; last_idx = i;
    }
  }
  __syncthreads();
  // ---- bracketize_schedule :792-823 + np.clip + running average :497-512 ----
  const double lo_rate = P.c.tax_rate_min;
  const double hi_rate = P.c.tax_annealing
                             ? aie_annealed_tax_limit(*reinterpret_cast<const int32_t*>(rec + P.o_tax_last_completions),
                                                      P.c.tax_annealing_warmup, P.c.tax_annealing_slope, P.c.tax_rate_max)
                             : P.c.tax_rate_max;
  double last_total = 0;
  for (int b = 0; b < NB; ++b) {
    double r;
    if (b + 1 < NB) {
      const double income = P.c.tax_bracket_cutoffs[b + 1];
      for (int i = lane; i <= T; i += AIE_NT) {
        double past = income - edges[i]; if (past < 0) past = 0;
        const double size = i < T ? edges[i + 1] - edges[i] : __builtin_huge_val();
        s_bt[i] = s_taus[i] * (size < past ? size : past);
      }
      __syncthreads();
      double due = np_sum_small(s_bt, T + 1);  // np.sum of 101 values, NumPy's pairwise order
      if (due < 0) due = 0;
      r = (due - last_total) / (P.c.tax_bracket_cutoffs[b + 1] - P.c.tax_bracket_cutoffs[b]);
      last_total = due;
      __syncthreads();
    } else {
      r = s_taus[T];
    }
    if (r < lo_rate) r = lo_rate;
    if (r > hi_rate) r = hi_rate;
    if (lane == 0) {
      next[b] = r;
      avg[b] = (avg[b] * 0.99) + (r * 0.01);
    }
  }
}
