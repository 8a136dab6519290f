// aie_kernels_covid.hip -- the COVID-19 scenario + its three components as ONE fused gfx950
// kernel per env.step() (the reference launches five CUDA kernels through WarpDrive:
// F/components/covid19_components_step.cu:10-262, F/scenarios/covid19/covid19_env_step.cu:274-619).
//
// Parity target is the reference's CPU path (`use_cuda=False`):
//   ControlUSStateOpenCloseStatus.component_step   F/components/covid19_components.py:180-221
//   FederalGovernmentSubsidy.component_step        :393-443
//   VaccinationCampaign.component_step             :615-627
//   CovidAndEconomyEnvironment.scenario_step       F/scenarios/covid19/covid19_env.py:744-792
//     sir_step :1477-1515, unemployment_step :1374-1441, economy_step :1444-1475
//   generate_observations :919-993, generate_masks (components), compute_reward :995-1173
//
// Mapping: one wavefront per replica, lane s = US state s.  Nothing is shared between lanes
// except the three 51-term planner sums, which go through LDS in NumPy's pairwise order.
// Every float32 / float64 conversion below sits where NumPy's promotion rules put it in the
// reference (float32 arrays; float32 x int32 -> float64; NumPy scalars are strongly typed),
// and the file is compiled with FP contraction off so that a*b+c stays two roundings.
// The only deliberate reassociation is the unemployment filter bank: the reference
// materialises delta[l]*w[s,f]*filt[f,l] and np.sum()s 3000 terms; here each lane streams its
// own 600-day window of stringency levels once and keeps F float64 accumulators (FMA),
// weighting them by w[s,f] at the end -- a float64 reordering, ~1e-16 relative.
#pragma clang fp contract(off)

// ablations (tools/covid_ablate.py, -DAIE_DEV build): bits of aie_dev_set_skip_mask -- 1 today's history byte store,
// 2 history byte loads (constants instead), 4 observation stores, 8 the per-state episode sums, 16 state row stores
#ifdef AIE_DEV
#define CV_SKIP(P, bit) (((P).dev_skip_mask & (bit)) != 0)
#else
#define CV_SKIP(P, bit) false
#endif

namespace aie {

__device__ __forceinline__ float np_sum_f32_lds(const float* a, int n) {
  // NumPy pairwise_sum for n <= 128 (numpy/core/src/umath/loops_utils.h.src): 8 strided
  // partial sums over the first n - n%8 elements, a fixed combination tree, then the tail.
  if (n < 8) {
    float r = 0.f;  // np.add.reduce starts from the first element; 0 + a0 is exact
    for (int i = 0; i < n; ++i) r = r + a[i];
    return r;
  }
  float r[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) r[k] = a[k];
  int i = 8;
  for (; i < n - (n % 8); i += 8) {
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = r[k] + a[i + k];
  }
  float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) res = res + a[i];
  return res;
}

typedef const double __attribute__((address_space(4))) * cv_tap_ptr;

__device__ __forceinline__ int wave_max_i32(int v) {  // wave-uniform maximum over the 64 lanes
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const int w = __shfl_xor(v, o, 64); v = w > v ? w : v; }
  return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ int wave_min_i32(int v) { return -wave_max_i32(-v); }

struct CvLane {
  float S, I, R, D, V, U, prod, subsidy;
  int level;     // stringency level in force at the current timestep
  int cooldown;  // ControlUSStateOpenCloseStatus.action_in_cooldown_until
};

__device__ __forceinline__ uint8_t* cv_hist_base(const aie_params& P, uint8_t* arena, int e) {
  return arena + P.a_cv_hist + (int64_t)e * P.cv_nch * P.cv_row;
}
__device__ __forceinline__ uint8_t* cv_hist_at(const aie_params& P, uint8_t* hist, int s, int tau) {
  return hist + (int64_t)(tau >> 4) * P.cv_row + s * 16 + (tau & 15);
}

// the recent-days ring (aie_layout.h: o_cv_ring): level of state s on history day tau (any of the last 32)
__device__ __forceinline__ uint8_t* cv_ring_at(const aie_params& P, uint8_t* rec, int s, int tau) {
  return rec + P.o_cv_ring + (tau & 31) * 64 + s;
}

// crra_nonlinearity (covid19_env.py:1056-1078), float32 like the reference's arrays
__device__ __forceinline__ float cv_crra(float x, float eta) {
  float ax = 365.0f * x;
  ax = fminf(fmaxf(ax, 0.1f), 3.0f);
  const float ome = 1.0f - eta;
  const float num = powf(ax, ome) - 1.0f;
  return (1.0f + num / ome) / 365.0f;
}
__device__ __forceinline__ float cv_minmax(float x, float lo, float hi) { return (x - lo) / (hi - lo + 1e-10f); }

// generate_observations + the three components' obs and masks, for timestep t.
__device__ __forceinline__ void cv_write_observations(const aie_params& P, uint8_t* __restrict__ arena, int e, int s,
                                                      int t, const CvLane& a, int subsidy_level,
                                                      int lag_level /* level on day t - beta_delay + 1 */) {
  const aie_covid_config& V = P.c.covid;
  const int n = P.n, NL = P.cv_NL, NS = P.cv_NS, T = P.c.episode_length;
  const bool on = s < n;
  const double* K = reinterpret_cast<const double*>(arena + P.a_cv_consts);
  float* oa = reinterpret_cast<float*>(arena + P.a_cv_obs_a + (int64_t)e * P.cv_nrow_obs * n * 4);
  float* op = reinterpret_cast<float*>(arena + P.a_cv_obs_p + (int64_t)e * (4 + P.MP) * 4);
  const int sl = on ? s : n - 1;
  const double pop = K[AIE_CV_K_POP * 64 + sl];
  // float32 state / int32 population -> float64, stored as float32 (:930-947)
  const float f6[6] = {a.S, a.I, a.R, a.D, a.V, a.U};
  const float time_f = (float)((double)t / (double)T);
  const int until = V.subsidy_interval - t % V.subsidy_interval;
  const float t_sub = (float)((double)until / (double)V.subsidy_interval);
  const float sub_lvl = (float)((double)subsidy_level / (double)NS);
  // VaccinationCampaign.generate_observations :629-653
  const int nt = t + 1, tf = P.cv_t_first_delivery, di = V.delivery_interval;
  double tv;
  if (nt <= tf) {
    tv = (double)(tf - nt) / (double)di;
    tv = tv < 1.0 ? tv : 1.0;
  } else {
    tv = (double)(di - nt % di);
  }
  const float t_vac = (float)(tv / (double)di);
  // lagged stringency level (:957-970): days before the episode divide in float64, the episode's own in float32
  const int tb = t - V.beta_delay + 1;
  const float lag = tb < 0 ? (float)((double)lag_level / (double)NL) : (float)lag_level / (float)NL;
  if (on) {
#pragma unroll
    for (int k = 0; k < 6; ++k) oa[(AIE_CV_OB_STATE + k) * n + s] = (float)((double)f6[k] / pop);
    oa[AIE_CV_OB_PROD * n + s] = a.prod / (float)K[AIE_CV_K_MAX_PROD * 64 + s];
    oa[AIE_CV_OB_LAG * n + s] = lag;
    oa[AIE_CV_OB_TIME * n + s] = time_f;
    oa[AIE_CV_OB_POLICY * n + s] = (float)a.level / (float)NL;
    oa[AIE_CV_OB_T_SUBSIDY * n + s] = t_sub;
    oa[AIE_CV_OB_SUBSIDY_LEVEL * n + s] = sub_lvl;
    oa[AIE_CV_OB_T_VACCINE * n + s] = t_vac;
    // generate_masks: NO-OP always allowed; levels only outside the cooldown (:97-108,223-241)
    const float open = (t >= a.cooldown || V.replay_policies) ? 1.0f : 0.0f;
    oa[AIE_CV_OB_MASK * n + s] = 1.0f;
    for (int k = 1; k <= NL; ++k) oa[(AIE_CV_OB_MASK + k) * n + s] = open;
  }
  if (s == 0) {
    op[0] = time_f;
    op[1] = t_sub;
    op[2] = sub_lvl;
    op[3] = t_vac;
  }
  // planner mask: subsidy levels selectable only on the first day of an interval (:445-469)
  const float pm = (t % V.subsidy_interval == 0 || V.replay_policies) ? 1.0f : 0.0f;
  for (int k = s; k < P.MP; k += AIE_NT) op[4 + k] = k == 0 ? 1.0f : pm;
}

}  // namespace aie

// One-time derivation of what every reset needs from the shared pre-episode table (run by aie_upload whenever
// "model_stringency_level_history_0" is written): the table in the per-replica history format (reset then copies
// 16-byte rows instead of assembling them byte by byte) and, with filter_recurrence, A_0 -- each filter's discounted
// sum of the pre-episode level changes (Horner, oldest first).  One wavefront, lane = state.
extern "C" __global__ void __launch_bounds__(AIE_NT)
    aie_covid_prepare_kernel(const aie_params* __restrict__ params, uint8_t* __restrict__ arena) {
  const aie_params& P = *params;
  const int s = (int)threadIdx.x, n = P.n, L = P.cv_L;
  const bool on = s < n;
  const int sl = on ? s : n - 1;
  const uint8_t* h0 = arena + P.a_cv_hist0;
  uint8_t* img = arena + P.a_cv_hist0c;
  for (int c = 0; c < P.cv_nch; ++c) {
    if (s * 16 < P.cv_row) {
      uint32_t w[4] = {0u, 0u, 0u, 0u};
      if (on) {
        for (int j = 0; j < 16; ++j) {
          const int tau = 16 * c + j;
          if (tau <= L) w[j >> 2] |= (uint32_t)h0[tau * n + s] << (8 * (j & 3));
        }
      }
      *reinterpret_cast<uint4*>(img + (int64_t)c * P.cv_row + s * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
  if (P.cv_ev_groups) {
    // the pre-episode days' level changes as every state's initial event list (what reset copies); a list that does not
    // fit makes every replica start "dense" (the window sums then stream the history as they always did)
    uint32_t* ev0 = reinterpret_cast<uint32_t*>(arena + P.a_cv_ev0);
    int32_t* ht0 = reinterpret_cast<int32_t*>(arena + P.a_cv_ev0 + (int64_t)P.cv_ev_groups * 1024);
    const int cap = 4 * P.cv_ev_groups;
    int cnt = 0;
    bool over = false;
    int prev = h0[sl];
    for (int tau = 1; tau <= L; ++tau) {
      const int lev = h0[tau * n + sl];
      if (lev != prev) {
        if (cnt < cap) ev0[((cnt >> 2) * 64 + s) * 4 + (cnt & 3)] = (uint32_t)tau | ((uint32_t)((lev - prev) & 0xff) << 16);
        else over = true;
        cnt += cnt < cap ? 1 : 0;
      }
      prev = lev;
    }
    ht0[s] = on ? (cnt << 16) : 0;
    const bool any_over = __ballot(on && over) != 0ull;
    if (s == 0) ht0[64] = any_over ? 1 : 0;
  }
  if (!P.c.covid.filter_recurrence) {
    // window sums: step 1's sums over its window WITHOUT its own day -- history days 2 .. L meet taps 0 .. L - 2 -- in
    // the order the reference adds the days (what the tail of every step leaves for the next one, here for the first)
    double* acc0 = reinterpret_cast<double*>(arena + P.a_cv_acc0);
    const double* G = reinterpret_cast<const double*>(arena + P.a_cv_filters) + (int64_t)AIE_CV_TAP_PAD_FRONT * P.cv_F;
    double A[AIE_COVID_MAX_FILTERS];
    for (int f = 0; f < AIE_COVID_MAX_FILTERS; ++f) A[f] = 0.0;
    int prev = h0[1 * n + sl];
    for (int tau = 2; tau <= L; ++tau) {
      const int lev = h0[tau * n + sl];
      const double d = (double)(lev - prev);
      prev = lev;
      for (int f = 0; f < P.cv_F; ++f) A[f] = __builtin_fma(d, G[(int64_t)(tau - 2) * P.cv_F + f], A[f]);
    }
    for (int f = 0; f < P.cv_F; ++f) acc0[f * 64 + s] = on ? A[f] : 0.0;
  }
  if (P.c.covid.filter_recurrence) {
    double* acc0 = reinterpret_cast<double*>(arena + P.a_cv_acc0);
    for (int f = 0; f < P.cv_F; ++f) {
      const double r = P.c.covid.filter_decay[f];
      double A = 0.0;
      int prev = h0[sl];
      for (int tau = 1; tau <= L; ++tau) {
        const int lev = h0[tau * n + sl];
        A = A * r + (double)(lev - prev);
        prev = lev;
      }
      acc0[f * 64 + s] = on ? A : 0.0;
    }
  }
}

// ---- reset: CovidAndEconomyEnvironment.reset_starting_layout / reset_agent_states /
// additional_reset_steps (covid19_env.py:1175-1293) + the components' additional_reset_steps.
extern "C" __global__ void __launch_bounds__(AIE_NT)
    aie_covid_reset_kernel(const aie_params* __restrict__ params, uint8_t* __restrict__ arena,
                           const uint8_t* __restrict__ env_mask, int keep_rewards) {
  using namespace aie;
  const aie_params& P = *params;
  const int e = replica_of_block((int)blockIdx.x, P.E);
  if (env_mask && !env_mask[e]) return;
  const int s = (int)threadIdx.x, n = P.n, L = P.cv_L, PT = P.cv_pitch;
  const bool on = s < n;
  uint8_t* rec = arena + P.a_records + (int64_t)e * P.rec_bytes;
  float* st = reinterpret_cast<float*>(rec + P.o_cv_state);
  const double* K = reinterpret_cast<const double*>(arena + P.a_cv_consts);
  uint8_t* hist = cv_hist_base(P, arena, e);
  // history: days -L..0 from the shared table (already in the history format: aie_covid_prepare_kernel), the
  // episode's own days zeroed
  const uint8_t* h0 = arena + P.a_cv_hist0;
  const uint8_t* img = arena + P.a_cv_hist0c;
  if (s * 16 < P.cv_row)
    for (int c = 0; c < P.cv_nch; ++c)
      *reinterpret_cast<uint4*>(hist + (int64_t)c * P.cv_row + s * 16) =
          *reinterpret_cast<const uint4*>(img + (int64_t)c * P.cv_row + s * 16);
  CvLane a;
  const int sl = on ? s : n - 1;
  for (int tau = L - 31; tau <= L; ++tau)  // the ring: the 32 days up to day 0 (days before the table begins: level 1)
    *cv_ring_at(P, rec, s, tau) = on ? (tau >= 0 ? h0[tau * n + s] : (uint8_t)1) : (uint8_t)0;
  if (P.cv_ev_groups) {  // the pre-episode level changes (aie_covid_prepare_kernel): every state's list, head/tail, dense flag
    const uint4* ev0 = reinterpret_cast<const uint4*>(arena + P.a_cv_ev0);
    const int32_t* ht0 = reinterpret_cast<const int32_t*>(arena + P.a_cv_ev0 + (int64_t)P.cv_ev_groups * 1024);
    uint4* evb = reinterpret_cast<uint4*>(arena + P.a_cv_events + (int64_t)e * P.cv_ev_groups * 1024);
    const int ht = ht0[s];
    const int groups = wave_max_i32(((ht >> 16) + 3) >> 2);
    for (int g = 0; g < groups; ++g) evb[g * 64 + s] = ev0[g * 64 + s];
    if (on) reinterpret_cast<int32_t*>(rec + P.o_cv_ev_ht)[s] = ht;
    if (s == 0) *reinterpret_cast<int32_t*>(rec + P.o_cv_dense) = ht0[64];
  } else if (s == 0) {
    *reinterpret_cast<int32_t*>(rec + P.o_cv_dense) = 0;
  }
  {  // recurrence: A_0 of every filter; window sums: step 1's sums before its own day (aie_covid_prepare_kernel)
    double* accs = reinterpret_cast<double*>(rec + P.o_cv_acc);
    const double* acc0 = reinterpret_cast<const double*>(arena + P.a_cv_acc0);
    if (on)
      for (int f = 0; f < P.cv_F; ++f) accs[f * PT + s] = acc0[f * 64 + s];
  }
  a.S = (float)K[AIE_CV_K_S0 * 64 + sl];
  a.I = (float)K[AIE_CV_K_I0 * 64 + sl];
  a.R = (float)K[AIE_CV_K_R0 * 64 + sl];
  a.D = (float)K[AIE_CV_K_D0 * 64 + sl];
  a.V = (float)K[AIE_CV_K_V0 * 64 + sl];
  a.U = (float)K[AIE_CV_K_U0 * 64 + sl];
  a.prod = 0.f;
  a.subsidy = 0.f;
  a.level = h0[L * n + sl];
  a.cooldown = 0;
  if (on) {  // the record's rows are packed: n lanes each (aie_layout.h: cv_pitch)
    st[AIE_CV_ST_S * PT + s] = a.S;
    st[AIE_CV_ST_I * PT + s] = a.I;
    st[AIE_CV_ST_R * PT + s] = a.R;
    st[AIE_CV_ST_D * PT + s] = a.D;
    st[AIE_CV_ST_V * PT + s] = a.V;
    st[AIE_CV_ST_U * PT + s] = a.U;
    st[AIE_CV_ST_PROD * PT + s] = 0.f;
    st[AIE_CV_ST_SUBSIDY * PT + s] = 0.f;
    st[AIE_CV_ST_HEALTH_INDEX * PT + s] = 0.f;
    st[AIE_CV_ST_ECONOMIC_INDEX * PT + s] = 0.f;
    for (int k = 0; k < AIE_CV_SUM_COUNT; ++k) reinterpret_cast<double*>(rec + P.o_cv_sums)[k * PT + s] = 0.0;
    reinterpret_cast<int32_t*>(rec + P.o_cv_cooldown)[s] = 0;
  }
  if (s == 0) {
    *reinterpret_cast<int32_t*>(rec + P.o_cv_subsidy_level) = 0;
    *reinterpret_cast<int32_t*>(rec + P.o_timestep) = 0;
    reinterpret_cast<float*>(rec + P.o_cv_p_index)[0] = 0.f;
    reinterpret_cast<float*>(rec + P.o_cv_p_index)[1] = 0.f;
    if (!keep_rewards) {
      arena[P.a_done + e] = 0;
      reinterpret_cast<float*>(arena + P.a_rew_p)[e] = 0.f;
    }
  }
  if (on && !keep_rewards) reinterpret_cast<float*>(arena + P.a_rew_a)[(int64_t)e * n + s] = 0.f;
  {  // the lagged observation at t = 0 (:957-970)
    const int tb = 1 - P.c.covid.beta_delay;
    const int lag_level = tb < 0 ? (arena + P.a_cv_lag_obs)[(tb + P.c.covid.beta_delay) * n + sl] : a.level;
    cv_write_observations(P, arena, e, s, 0, a, 0, lag_level);
  }
}

// ---- one env.step() (base_env.py:929-1032) ----
// What follows a window-sum step (filter_recurrence off), as a launch of its own (aie_covid_window_kernel).  Every filter's sum over the 600-day window is a sum over the
// window's NON-ZERO level changes -- a day without a change adds fma(0, tap, acc) == acc to the reference's sum, so the
// sum over the change events in the same order is the same float64, bit for bit -- and all of those but the step's own
// day are known one step ahead.  So a step ends by (1) recording its own change in the state's event list, (2) forming
// the NEXT step's sums over everything but that step's own day -- over the list (one burst of list loads, per-lane
// reads of the LDS tap table), or, once a list has overflowed, over the streamed history as the kernel always did --
// and leaving them in the record, where the next step finds them with its other loads and adds its own day's term.
// A launch of its own because the step kernel sits at the register file's limit (64 registers for 8 waves per SIMD,
// i.e. the whole batch resident): inlined ahead of the step this work cost ~20 spilled values and 15 us, inlined behind
// it still 8.  Also writes the long history ([chunk][state][16 days]).
template <int F, typename TapT>
__device__ __forceinline__ void cv_window_tail(const aie_params& P, uint8_t* __restrict__ arena, uint8_t* __restrict__ rec,
                                               uint8_t* __restrict__ hist, const int e, const int s, const int t,
                                               const int level, const int prev_level, int dense, const int ev_ht,
                                               const TapT* taps) {
  using namespace aie;
  const int n = P.n, L = P.cv_L, PT = P.cv_pitch;
  const bool on = s < n;
  const int sl = on ? s : n - 1;
  const int d_new_i = level - prev_level;
  const bool grow = on && d_new_i != 0;
  const int head = ev_ht & 0xffff, tail = ev_ht >> 16, cap = 4 * P.cv_ev_groups;
  bool flush_now = false;
  uint32_t flushw[4] = {0u, 0u, 0u, 0u};
  if (!dense && __ballot(grow && tail >= cap) != 0ull) {
    // a state's list is full: from here to the next reset the replica streams its whole window; the open chunk's
    // days so far come from the recent-days ring
    dense = 1;
    flush_now = true;
    if (s == 0) *reinterpret_cast<int32_t*>(rec + P.o_cv_dense) = 1;
#pragma unroll
    for (int j = 0; j < 15; ++j)
      if (j < ((L + t) & 15)) flushw[j >> 2] |= (uint32_t)*cv_ring_at(P, rec, sl, ((L + t) & ~15) + j) << (8 * (j & 3));
  }
  // the long history: a replica on event lists gets a state's 16 bytes when their chunk is complete (from the ring); a
  // streaming one every day (51 one-byte stores spread over 816 bytes); the step that switches writes the open chunk's
  // days so far in one piece
  if (!CV_SKIP(P, 1)) {
    if (dense && !flush_now) {
      if (on) *cv_hist_at(P, hist, s, L + t) = (uint8_t)level;
    } else if (flush_now) {
      uint32_t w[4] = {flushw[0], flushw[1], flushw[2], flushw[3]};
      w[((L + t) & 15) >> 2] |= (uint32_t)level << (8 * ((L + t) & 3));
      if (s * 16 < P.cv_row)
        *reinterpret_cast<uint4*>(hist + (int64_t)((L + t) >> 4) * P.cv_row + s * 16) = on ? make_uint4(w[0], w[1], w[2], w[3]) : make_uint4(0u, 0u, 0u, 0u);
    } else if (((L + t) & 15) == 15) {
      uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int j = 0; j < 15; ++j) w[j >> 2] |= (uint32_t)*cv_ring_at(P, rec, sl, L + t - 15 + j) << (8 * (j & 3));
      w[3] |= (uint32_t)level << 24;
      if (s * 16 < P.cv_row)
        *reinterpret_cast<uint4*>(hist + (int64_t)((L + t) >> 4) * P.cv_row + s * 16) = on ? make_uint4(w[0], w[1], w[2], w[3]) : make_uint4(0u, 0u, 0u, 0u);
    }
  }
  if (t >= P.c.episode_length) return;  // the episode is over: reset supplies the first step's sums
  const int t1 = t + 1;                 // the step the sums are for: window = history days t1 + 1 .. t1 + L, its own day = L + t1
  double acc[F];
#pragma unroll
  for (int f = 0; f < F; ++f) acc[f] = 0.0;
  if (!dense) {
    int32_t* htrow = reinterpret_cast<int32_t*>(rec + P.o_cv_ev_ht);
    uint32_t* evb = reinterpret_cast<uint32_t*>(arena + P.a_cv_events + (int64_t)e * P.cv_ev_groups * 1024);
    const int g_lo = wave_min_i32(on ? head >> 2 : 0x7fff), g_hi = wave_max_i32(on ? (tail + 3) >> 2 : 0);
    const uint4* evq = reinterpret_cast<const uint4*>(evb) + s;
    int expired = 0;
    // the list in memory holds the events before today's, oldest first; the groups are fetched CV_BURST at a time
    // (one memory round trip covers the 32 events of a two-week cool-down over the whole window)
#ifndef AIE_CV_BURST
#define AIE_CV_BURST 8
#endif
    constexpr int CV_BURST = AIE_CV_BURST;
    for (int g0 = g_lo; g0 < g_hi; g0 += CV_BURST) {
      uint4 qb[CV_BURST];
#pragma unroll
      for (int k = 0; k < CV_BURST; ++k)
        qb[k] = (on && 4 * (g0 + k) < tail) ? evq[(g0 + k) * 64] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
      for (int k = 0; k < CV_BURST; ++k) {
        if (g0 + k >= g_hi) break;  // (wave-uniform)
        const uint32_t w4[4] = {qb[k].x, qb[k].y, qb[k].z, qb[k].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int idx = 4 * (g0 + k) + j, tau = (int)(w4[j] & 0xffffu);
          const bool live = on && idx >= head && idx < tail;
          if (live && tau <= t1) expired += 1;  // leaves the window (at most one per step: the days are distinct)
          if (live && tau > t1) {
            const double d = (double)((int)(w4[j] << 8) >> 24);
            const TapT* tp = taps + (tau - t1) * F;  // tap l = tau - t1 - 1 lives in row l + 1
#pragma unroll
            for (int f = 0; f < F; ++f) acc[f] = __builtin_fma(d, (double)tp[f], acc[f]);
          }
        }
      }
    }
    if (grow) {  // today's change: still in registers (its tap tomorrow: L - 2), and onto the list
      const double d = (double)d_new_i;
      const TapT* tp = taps + (L - 1) * F;
#pragma unroll
      for (int f = 0; f < F; ++f) acc[f] = __builtin_fma(d, (double)tp[f], acc[f]);
      evb[((tail >> 2) * 64 + s) * 4 + (tail & 3)] = (uint32_t)(L + t) | ((uint32_t)(d_new_i & 0xff) << 16);
    }
    if (on) htrow[s] = (head + expired) | ((tail + (grow ? 1 : 0)) << 16);
  } else {
    // the whole window, day by day: deltas of the daily levels of history days t1 .. t1 + L; the delta between days
    // tau' - 1 and tau' meets tap l = tau' - t1 - 1.  The tap table is zero-padded (AIE_CV_TAP_PAD_FRONT rows before tap
    // 0, zeros after tap L - 1), so whole 16-day chunks are processed without any window test.  Chunks are fetched in
    // groups of AIE_CV_GROUP, one group ahead.  Today's level is still in registers and the next step's own day counts
    // as "no change" here (the step adds its term itself): both bytes are patched into the stream.
    const double* __restrict__ G = reinterpret_cast<const double*>(arena + P.a_cv_filters);  // [row][F]
    const int c0 = t1 >> 4;
    const int d_today = L + t, d_next = L + t1;
    const int ngroups = ((L >> 4) + 2 + AIE_CV_GROUP - 1) / AIE_CV_GROUP;
    const uint8_t* row = hist + sl * 16 + (int64_t)c0 * P.cv_row;
    int carry = 0;
    uint4 cur[AIE_CV_GROUP], nxt[AIE_CV_GROUP];
#pragma unroll
    for (int k = 0; k < AIE_CV_GROUP; ++k) cur[k] = *reinterpret_cast<const uint4*>(row + (int64_t)k * P.cv_row);
    for (int g = 0; g < ngroups; ++g) {
      if (g + 1 < ngroups) {
#pragma unroll
        for (int k = 0; k < AIE_CV_GROUP; ++k)
          nxt[k] = *reinterpret_cast<const uint4*>(row + (int64_t)((g + 1) * AIE_CV_GROUP + k) * P.cv_row);
      }
#pragma unroll
      for (int k = 0; k < AIE_CV_GROUP; ++k) {
        const int c = c0 + g * AIE_CV_GROUP + k;
        uint32_t ww[4] = {cur[k].x, cur[k].y, cur[k].z, cur[k].w};
        if (c == (d_today >> 4)) {
          if (flush_now) {  // the chunk's earlier days are not in the long history yet in the step that switches
#pragma unroll
            for (int q = 0; q < 4; ++q) ww[q] = flushw[q];
          }
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (q == ((d_today & 15) >> 2)) ww[q] = (ww[q] & ~(0xffu << (8 * (d_today & 3)))) | ((uint32_t)level << (8 * (d_today & 3)));
        }
        if (c == (d_next >> 4)) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (q == ((d_next & 15) >> 2)) ww[q] = (ww[q] & ~(0xffu << (8 * (d_next & 3)))) | ((uint32_t)level << (8 * (d_next & 3)));
        }
        // wave-uniform pointer into the read-only tap table -> s_load through the scalar cache
        cv_tap_ptr g0 = (cv_tap_ptr)(uintptr_t)(G + (int64_t)(16 * c - t1 - 1 + AIE_CV_TAP_PAD_FRONT) * F);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int lev = (int)((ww[j >> 2] >> (8 * (j & 3))) & 0xffu);
          const double d = (double)(lev - carry);
          carry = lev;
#pragma unroll
          for (int f = 0; f < F; ++f) acc[f] = __builtin_fma(d, g0[j * F + f], acc[f]);
          // 16*F doubles of taps per chunk do not fit the SGPR file: tie the pointer to the
          // accumulator every 4 days so that only 4*F taps are fetched ahead of their use
          if ((j & 3) == 3) {
#pragma unroll
            for (int f = 0; f < F; ++f) asm volatile("" : "+s"(g0), "+v"(acc[f]));
          }
        }
      }
#pragma unroll
      for (int k = 0; k < AIE_CV_GROUP; ++k) cur[k] = nxt[k];
    }
  }
  if (on) {
    double* accs = reinterpret_cast<double*>(rec + P.o_cv_acc);
#pragma unroll
    for (int f = 0; f < F; ++f) accs[f * PT + s] = acc[f];
  }
}

// `red`: this replica's three reduction rows in LDS; `taps` (window sums only): the workgroup's LDS copy of the filter
// taps, row 0 zero, row 1 + l = tap l ([filter_len + 1][F] doubles).
template <int F, bool RECUR>
__device__ __forceinline__ void cv_step_body(const aie_params* __restrict__ params, uint8_t* __restrict__ arena,
                                             const int32_t* __restrict__ act_a, const int32_t* __restrict__ act_p,
                                             const NextActions& next, const int e, const int s, float (*red)[64]) {
  using namespace aie;
  const aie_params& P = *params;
  const aie_covid_config& V = P.c.covid;
  const int n = P.n, L = P.cv_L, NL = P.cv_NL, NS = P.cv_NS, PT = P.cv_pitch;
  const bool on = s < n;
  const int sl = on ? s : n - 1;  // idle lanes shadow the last state (loads stay in bounds)
  uint8_t* rec = arena + P.a_records + (int64_t)e * P.rec_bytes;
  float* st = reinterpret_cast<float*>(rec + P.o_cv_state);
  int32_t* cool = reinterpret_cast<int32_t*>(rec + P.o_cv_cooldown);
  const double* K = reinterpret_cast<const double*>(arena + P.a_cv_consts);
  uint8_t* hist = cv_hist_base(P, arena, e);
  const int T = P.c.episode_length;
  const int t = uni(*reinterpret_cast<const int32_t*>(rec + P.o_timestep)) + 1;
  // the replica's call counters (beside the timestep in the record): the draw index of the synthetic random policy and
  // this step's slot of the reward log (aie_set_reward_log; nullptr = off).  Lane 0 stores them back at the end.
  const int sample_t = uni(*reinterpret_cast<const int32_t*>(rec + P.o_sample_t));
  float* __restrict__ rew_log = rew_log_claim(P, reinterpret_cast<int32_t*>(rec + P.o_rew_slot),
                                              reinterpret_cast<int32_t*>(rec + P.o_rew_epoch), P.E, n, s == 0);
  if (t > T) {
    // episode over: the caller has to reset (the reference would index past its arrays).  The call counters still move
    // with the batch's: a replica that sat a launch out writes the same reward-log slot as its peers, and
    // draws with the same index, once it is reset.
    if (s == 0 && (next.a || next.p)) *reinterpret_cast<int32_t*>(rec + P.o_sample_t) = sample_t + 1;
    return;
  }
  // Every load of the replica's record is issued here, before anything waits: a replica is one wavefront whose whole
  // step is a dependent chain (timestep -> history bytes -> state -> stores), and all 8192 of BASELINE configs[3] are
  // resident at once, so a launch lasts as long as that chain.  What does not depend on the timestep travels beside
  // it; the read-modify-write accumulators (episode sums, index sums) are read now and only written at the end.
  double* sums = reinterpret_cast<double*>(rec + P.o_cv_sums);
  float* pidx = reinterpret_cast<float*>(rec + P.o_cv_p_index);  // planner.state[...] += ... :1160-1161
  double sum_u0, sum_s0, sum_p0, sum_b0;
  float hidx0, eidx0, pidx0, pidx1, S1, I1, R1, V1, D1;
  int cool0;
#define CV_LOAD_STATE()                                                                                      \
  do {                                                                                                       \
    S1 = st[AIE_CV_ST_S * PT + sl]; I1 = st[AIE_CV_ST_I * PT + sl]; R1 = st[AIE_CV_ST_R * PT + sl];            \
    V1 = st[AIE_CV_ST_V * PT + sl]; D1 = st[AIE_CV_ST_D * PT + sl];                                           \
    cool0 = cool[sl];                                                                                        \
  } while (0)
#define CV_LOAD_ACCUMULATORS()                                                                                \
  do {                                                                                                       \
    sum_u0 = sums[AIE_CV_SUM_UNEMPLOYED * PT + sl]; sum_s0 = sums[AIE_CV_SUM_STRINGENCY * PT + sl];           \
    sum_p0 = sums[AIE_CV_SUM_PRODUCTIVITY * PT + sl]; sum_b0 = sums[AIE_CV_SUM_SUBSIDY * PT + sl];            \
    hidx0 = st[AIE_CV_ST_HEALTH_INDEX * PT + sl]; eidx0 = st[AIE_CV_ST_ECONOMIC_INDEX * PT + sl];             \
    pidx0 = pidx[0]; pidx1 = pidx[1];                                                                        \
  } while (0)
  CV_LOAD_STATE();
  // (the read-modify-write accumulators -- episode sums, index sums -- are read now and only written at the end)
  CV_LOAD_ACCUMULATORS();
  double acc[F];
  // the lagged stringency level of the new observation (:957-970), fetched with the other history bytes: a load issued
  // behind today's stores would wait for them (memory operations of a wave complete in order)
  const int tb = t - V.beta_delay + 1;
  int lag_level;
  if (tb < 0) lag_level = (arena + P.a_cv_lag_obs)[(tb + V.beta_delay) * n + sl];
  else if (V.beta_delay == 1) lag_level = -1;  // today's level, known further down
  else lag_level = V.beta_delay <= 32 ? *cv_ring_at(P, rec, sl, L + tb) : *cv_hist_at(P, hist, sl, L + tb);

  // ---- ControlUSStateOpenCloseStatus.component_step :180-221 ----
  int act = act_a ? act_a[(int64_t)e * n + sl] : 0;
  if (V.replay_policies) act = (arena + P.a_cv_replay_a)[(int64_t)(t - 1) * 64 + sl];  // :181-186: yesterday's recorded level
  const int prev_level = CV_SKIP(P, 2) ? 1 : *cv_ring_at(P, rec, sl, L + t - 1);
  if (act < 0 || act > NL) act = 0;
  CvLane a;
  a.level = act == 0 ? prev_level : act;
  if (lag_level < 0) lag_level = a.level;
  a.cooldown = cool0;
  if (t == a.cooldown + 1) a.cooldown += act == 0 ? 1 : V.action_cooldown_period;

  // ---- FederalGovernmentSubsidy.component_step :393-443 ----
  int sub_level = uni(*reinterpret_cast<const int32_t*>(rec + P.o_cv_subsidy_level));
  if (V.replay_policies) {  // :394-425: the level the recorded subsidies amount to on this day
    sub_level = uni(reinterpret_cast<const int32_t*>(arena + P.a_cv_replay_p)[t - 1]);
  } else if ((t - 1) % V.subsidy_interval == 0) {
    int ap = act_p ? uni(act_p[e]) : 0;
    if (ap < 0 || ap > NS) ap = 0;
    sub_level = ap;
  }
  a.subsidy = (float)(((double)sub_level / (double)NS) * K[AIE_CV_K_MAX_DAILY_SUBSIDY * 64 + sl]);

  // ---- VaccinationCampaign.component_step :615-627 (consumed by the scenario step below) ----
  const int vac = (t >= V.time_when_vaccine_delivery_begins && t % V.delivery_interval == 0)
                      ? (int)K[AIE_CV_K_VACCINES_PER_DELIVERY * 64 + sl]
                      : 0;

  // ---- sir_step :1477-1515 ----
  const double pop = K[AIE_CV_K_POP * 64 + sl];
  {
    const int beta_level = CV_SKIP(P, 2) ? 1 : (V.beta_delay <= 31 ? *cv_ring_at(P, rec, sl, L + t - V.beta_delay)
                                                                    : *cv_hist_at(P, hist, sl, L + t - V.beta_delay));  // days before the data: level 1
    const float beta = (float)(K[AIE_CV_K_BETA_INTERCEPT * 64 + sl] + K[AIE_CV_K_BETA_SLOPE * 64 + sl] * (double)beta_level);
    const float s_eps = S1 + 1e-10f;
    const double q = (double)vac / (double)s_eps;
    const float frac_vaccinated = (float)(q < 1.0 ? q : 1.0);
    const double vaccinated_t = (double)vac < (double)S1 ? (double)vac : (double)S1;
    const double si_over_n = ((double)S1 / pop) * (double)I1;
    const float one_minus = 1.0f - frac_vaccinated;
    const float dS = (float)(((double)(-beta) * si_over_n) * (double)one_minus - vaccinated_t);
    const float gI = (float)V.gamma * I1;
    const float dR = (float)((double)gI + vaccinated_t);
    const float dI = -dS - dR;
    const float dV = (float)vaccinated_t;
    a.S = fmaxf(S1 + dS, 0.f);
    a.I = fmaxf(I1 + dI, 0.f);
    a.R = fmaxf(R1 + dR, 0.f);
    a.V = fmaxf(V1 + dV, 0.f);
    a.D = (float)V.death_rate * (a.R - a.V);
  }
  // use_real_world_data :734-757: the recorded day (float64 table), clamped at zero; the global state keeps the
  // float32 cast, the economy step below works on the table values themselves
  const double* __restrict__ rws = reinterpret_cast<const double*>(arena + P.a_cv_replay_state);
  const int64_t rw_plane = (int64_t)(T + 1) * 64;
  double rw_I = 0.0, rw_D = 0.0;
  if (V.replay_data) {
    rw_I = fmax(rws[1 * rw_plane + (int64_t)t * 64 + sl], 0.0);
    rw_D = fmax(rws[4 * rw_plane + (int64_t)t * 64 + sl], 0.0);
    a.S = (float)fmax(rws[0 * rw_plane + (int64_t)t * 64 + sl], 0.0);
    a.I = (float)rw_I;
    a.R = (float)fmax(rws[2 * rw_plane + (int64_t)t * 64 + sl], 0.0);
    a.V = (float)fmax(rws[3 * rw_plane + (int64_t)t * 64 + sl], 0.0);
    a.D = (float)rw_D;
  }

  // ---- unemployment_step :1374-1441 ----
  // deltas of the 601 most recent daily levels (history index tau in [t, t+L]); the delta
  // between days tau'-1 and tau' meets filter tap l = tau' - t - 1.  The tap table is
  // zero-padded (AIE_CV_TAP_PAD_FRONT rows before tap 0, zeros after tap L-1), so whole
  // 16-day chunks are processed without any window test: days outside the window meet a zero
  // tap.  Chunks are fetched in groups of AIE_CV_GROUP, one group ahead, to keep ~128 B per
  // lane in flight (HBM latency >> the ~300 cycles of FMA work in one chunk).
  double unemployed;
  {
    if constexpr (RECUR) {
      // The taps are exp(-age / lambda_f): each filter's discounted delta sum over the window obeys
      //   A_t = r_f * (A_{t-1} - r_f^(L-1) * d_old) + d_new,
      // d_new = today's level change (tap L-1, weight 1), d_old = the change between history days t-1 and t, which
      // had tap 0 at step t-1 and now leaves the window.  O(1) per step and state instead of L * F multiply-adds
      // and a 601-byte history read; A_0 comes from the reset kernel (Horner over the pre-episode days).
      double* accs = reinterpret_cast<double*>(rec + P.o_cv_acc);
      // (history days <= L are the pre-episode days every replica shares: read from the shared table -- one
      // cache-resident 51-byte row per day -- instead of this replica's own copy while t <= L)
      int lev_t, lev_tm1;
      if (CV_SKIP(P, 2)) lev_t = lev_tm1 = 1;
      else if (t <= L) {
        const uint8_t* h0 = arena + P.a_cv_hist0;
        lev_t = h0[t * n + sl];
        lev_tm1 = h0[(t - 1) * n + sl];
      } else {
        lev_t = *cv_hist_at(P, hist, sl, t);
        lev_tm1 = *cv_hist_at(P, hist, sl, t - 1);
      }
      const double d_old = (double)(lev_t - lev_tm1), d_new = (double)(a.level - prev_level);
#pragma unroll
      for (int f = 0; f < F; ++f) {
        const double r = V.filter_decay[f];
        acc[f] = r * (accs[f * PT + sl] - V.filter_tail[f] * d_old) + d_new;
        if (on) accs[f * PT + s] = acc[f];
      }
    } else {
      // window sums: the sums over the window's earlier days are in the record (aie_covid_window_kernel left them there
      // behind the previous step; reset for the first); today's change (history day L + t, tap L - 1) is the window's
      // last day
      const double* accs = reinterpret_cast<const double*>(rec + P.o_cv_acc);
      const double d = (double)(a.level - prev_level);
      cv_tap_ptr tp = (cv_tap_ptr)(uintptr_t)(reinterpret_cast<const double*>(arena + P.a_cv_filters) +
                                              (int64_t)(AIE_CV_TAP_PAD_FRONT + L - 1) * F);  // (one row, the same for every lane)
#pragma unroll
      for (int f = 0; f < F; ++f) acc[f] = __builtin_fma(d, tp[f], accs[f * PT + sl]);
    }
    double x = 0.0;
#pragma unroll
    for (int f = 0; f < F; ++f) {
      x = x + K[(AIE_CV_K_CONV_W0 + f) * 64 + sl] * acc[f];
    }
    const double excess = x <= 20.0 ? log(1.0 + exp(x)) : x;  // softplus :1358-1372
    unemployed = ((excess + K[AIE_CV_K_UNEMP_BIAS * 64 + sl]) * pop) / 100.0;
    if (V.replay_data) unemployed = rws[5 * rw_plane + (int64_t)t * 64 + sl];  // :815-818 (not clamped)
    a.U = (float)unemployed;
  }


  // ---- economy_step :1444-1475 ----
  {
    const float incapacitated = (float)V.infection_too_sick_to_work_rate * a.I + a.D;
    const float p1865 = (float)V.population_between_age_18_65;
    double cant_work = (double)(incapacitated * p1865) + unemployed;
    if (V.replay_data)  // float32 scalars times float64 table values: NumPy computes in float64
      cant_work = ((double)(float)V.infection_too_sick_to_work_rate * rw_I + rw_D) * (double)p1865 + unemployed;
    const double workers = pop * (double)p1865;
    const double diff = workers - cant_work;
    a.prod = (float)((diff > 0.0 ? diff : 0.0) * (double)(float)V.daily_production_per_worker) + a.subsidy;
  }

  // ---- state write-back (rows of n lanes, packed: a replica's rows are one contiguous block) ----
  // today's level: one whole 64-byte row of the ring; the long history -- [chunk][state][16 days], the layout the window
  // sums stream -- gets today's byte every step only when those sums run (51 one-byte stores spread over 816 bytes:
  // seven partly written lines a step); the recurrence writes a state's 16 bytes once their chunk is complete
  *cv_ring_at(P, rec, s, L + t) = on ? (uint8_t)a.level : (uint8_t)0;
  if (RECUR && !CV_SKIP(P, 1) && ((L + t) & 15) == 15) {  // (window sums: the tail writes the long history)
    uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 15; ++j) w[j >> 2] |= (uint32_t)*cv_ring_at(P, rec, sl, L + t - 15 + j) << (8 * (j & 3));
    w[3] |= (uint32_t)a.level << 24;
    if (s * 16 < P.cv_row)
      *reinterpret_cast<uint4*>(hist + (int64_t)((L + t) >> 4) * P.cv_row + s * 16) = on ? make_uint4(w[0], w[1], w[2], w[3]) : make_uint4(0u, 0u, 0u, 0u);
  }
  if (on) {
    cool[s] = a.cooldown;
    if (!CV_SKIP(P, 16)) {
      st[AIE_CV_ST_S * PT + s] = a.S;
      st[AIE_CV_ST_I * PT + s] = a.I;
      st[AIE_CV_ST_R * PT + s] = a.R;
      st[AIE_CV_ST_D * PT + s] = a.D;
      st[AIE_CV_ST_V * PT + s] = a.V;
      st[AIE_CV_ST_U * PT + s] = a.U;
      st[AIE_CV_ST_PROD * PT + s] = a.prod;
      st[AIE_CV_ST_SUBSIDY * PT + s] = a.subsidy;
    }
    // per-state sums over the days of the episode, for scenario_metrics :1613-1687
    if (!CV_SKIP(P, 8)) {
      sums[AIE_CV_SUM_UNEMPLOYED * PT + s] = sum_u0 + (double)a.U;
      sums[AIE_CV_SUM_STRINGENCY * PT + s] = sum_s0 + (double)a.level;
      sums[AIE_CV_SUM_PRODUCTIVITY * PT + s] = sum_p0 + (double)a.prod;
      sums[AIE_CV_SUM_SUBSIDY * PT + s] = sum_b0 + (double)a.subsidy;
    }
  }
  if (s == 0) {
    *reinterpret_cast<int32_t*>(rec + P.o_timestep) = t;
    *reinterpret_cast<int32_t*>(rec + P.o_cv_subsidy_level) = sub_level;
  }

  // ---- compute_reward :995-1173 ----
  const float marginal_deaths = a.D - D1;
  red[0][s] = on ? marginal_deaths : 0.f;
  red[1][s] = on ? a.subsidy : 0.f;
  red[2][s] = on ? a.prod : 0.f;
  AIE_WSYNC();  // (a replica is one wavefront: its LDS writes are ordered, no workgroup barrier needed)
  const float eta = (float)V.economic_reward_crra_eta;
  const float rnf = (float)V.reward_normalization_factor;
  if (on) {
    const float hn = (float)K[AIE_CV_K_HEALTH_NORM * 64 + s];
    float h = (float)(((double)(-marginal_deaths) * V.value_of_life) / (double)hn);
    float ec = cv_crra(a.prod / (float)K[AIE_CV_K_ECON_NORM * 64 + s], eta);
    h = cv_minmax(h, (float)K[AIE_CV_K_MIN_HEALTH * 64 + s], (float)K[AIE_CV_K_MAX_HEALTH * 64 + s]);
    ec = cv_minmax(ec, (float)K[AIE_CV_K_MIN_ECON * 64 + s], (float)K[AIE_CV_K_MAX_ECON * 64 + s]);
    const float wh = (float)K[AIE_CV_K_W_HEALTH * 64 + s], we = (float)K[AIE_CV_K_W_ECON * 64 + s];
    const float ra = ((wh * h + we * ec) / (wh + we)) / rnf;
    reinterpret_cast<float*>(arena + P.a_rew_a)[(int64_t)e * n + s] = ra;
    if (rew_log) rew_log[(int64_t)e * (n + 2) + s] = ra;
    st[AIE_CV_ST_HEALTH_INDEX * PT + s] = hidx0 + h;  // agent.state["Health Index"] += ... :1123-1125 (float32)
    st[AIE_CV_ST_ECONOMIC_INDEX * PT + s] = eidx0 + ec;
  }
  if (s == 0) {
    const float sum_md = np_sum_f32_lds(red[0], n);
    const float sum_sub = np_sum_f32_lds(red[1], n);
    const float sum_pp = np_sum_f32_lds(red[2], n);
    double ph = ((double)(-sum_md) * V.value_of_life) / (double)(float)V.planner_health_norm;
    const float cost = (1.0f + (float)V.risk_free_interest_rate) * sum_sub;
    float pe = cv_crra((sum_pp - cost) / (float)V.planner_economic_norm, eta);
    const float lo_h = (float)V.min_marginal_planner_health_index, hi_h = (float)V.max_marginal_planner_health_index;
    ph = (ph - (double)lo_h) / (double)(hi_h - lo_h + 1e-10f);
    pe = cv_minmax(pe, (float)V.min_marginal_planner_economic_index, (float)V.max_marginal_planner_economic_index);
    const float wph = (float)V.weightage_on_marginal_planner_health_index;
    const float wpe = (float)V.weightage_on_marginal_planner_economic_index;
    const double rp = (((double)wph * ph + (double)(wpe * pe)) / (double)(wph + wpe)) / (double)rnf;
    reinterpret_cast<float*>(arena + P.a_rew_p)[e] = (float)rp;
    if (rew_log) {
      rew_log[(int64_t)e * (n + 2) + n] = (float)rp;
      rew_log[(int64_t)e * (n + 2) + n + 1] = t >= T ? 1.0f : 0.0f;
    }
    pidx[0] = (float)((double)pidx0 + ph);
    pidx[1] = pidx1 + pe;
    arena[P.a_done + e] = t >= T ? 1 : 0;
    if (t >= T) *reinterpret_cast<int32_t*>(rec + P.o_completions) += 1;
  }

  // ---- observations + masks for the new timestep ----
  if (!CV_SKIP(P, 4)) cv_write_observations(P, arena, e, s, t, a, sub_level, lag_level);
  if (next.a || next.p) {  // aie_step_sample_next: the uniform random policy's draw for the next step, one lane per slot
    const int per_env = P.n * P.act_a_width + P.act_p_width;
    if (next.masked) {
      // aie_step_sample_next_masked: uniform over what the masks just written allow -- a state's levels only outside its
      // cooldown, the planner's subsidy levels only on the first day of an interval; NO-OP always (the allowed set is
      // {0} or {0 .. N}, so the pick-th allowed entry of aie_sample_masked_actions is the pick itself)
      if (s <= n) {
        const uint32_t u = aie_counter_rng(next.seed, (uint64_t)(next.env_offset + e), (uint64_t)sample_t, (uint64_t)s);
        const bool open = s < n ? (t >= a.cooldown || V.replay_policies) : (t % V.subsidy_interval == 0 || V.replay_policies);
        const int count = open ? 1 + (s < n ? NL : NS) : 1;
        const int32_t pick = (int32_t)(((uint64_t)u * (uint64_t)count) >> 32);
        if (s < n) { if (next.a) next.a[(int64_t)e * n + s] = pick; }
        else if (next.p) next.p[e] = pick;
      }
    } else {
      for (int j = s; j < per_env; j += AIE_NT) sample_action_slot(P, next.seed, next.env_offset, (int64_t)sample_t, e, j, next.a, next.p);
    }
    if (s == 0) *reinterpret_cast<int32_t*>(rec + P.o_sample_t) = sample_t + 1;
  }
  // window sums: aie_covid_window_kernel, launched behind this one, records today's change and forms the next step's sums
  if constexpr (!RECUR)
    if (s == 0) *reinterpret_cast<int32_t*>(rec + P.o_cv_tail_pending) = 1;
}

#undef CV_LOAD_STATE
#undef CV_LOAD_ACCUMULATORS

// One replica per workgroup (one wavefront).
template <int F, bool RECUR>
__global__ void __launch_bounds__(AIE_NT) __attribute__((amdgpu_waves_per_eu(8, 8)))
    aie_covid_step_kernel(const aie_params* __restrict__ params, uint8_t* __restrict__ arena,
                          const int32_t* __restrict__ act_a, const int32_t* __restrict__ act_p, NextActions next) {
  __shared__ float red[3][64];
  cv_step_body<F, RECUR>(params, arena, act_a, act_p, next, aie::replica_of_block((int)blockIdx.x, params->E),
                         (int)threadIdx.x, red);
}

// Window sums (filter_recurrence off): the second launch of a step.  AIE_CV_WIN_WAVES replicas (one wavefront each) per
// workgroup share an LDS copy of the filter taps -- [1 + filter_len][F], row 0 zero -- which the sums over the change
// events index per lane.  TapT: double, or float when every uploaded tap is a float32 value (the reference's are:
// covid19_env.py:242-247 builds them in float32) -- the same numbers in half the LDS bytes and with odd-dword rows, i.e.
// per-lane reads of different rows spread over all 64 banks instead of colliding on 32 bank pairs.
template <int F, typename TapT>
__global__ void __launch_bounds__(AIE_CV_WIN_WAVES * AIE_NT) __attribute__((amdgpu_waves_per_eu(8, 8)))
    aie_covid_window_kernel(const aie_params* __restrict__ params, uint8_t* __restrict__ arena) {
  using namespace aie;
  extern __shared__ __attribute__((aligned(16))) uint8_t cv_lds[];
  const aie_params& P = *params;
  const int b = (int)blockIdx.x * AIE_CV_WIN_WAVES + (int)(threadIdx.x >> 6);
  const int e = replica_of_block(b < P.E ? b : 0, P.E), s = (int)(threadIdx.x & 63);
  uint8_t* rec = arena + P.a_records + (int64_t)e * P.rec_bytes;
  // this wave's loads go out ahead of the table copy: the step just taken (pending flag), the streaming flag, the
  // state's list head / tail, today's and yesterday's level (the recent-days ring)
  const int n = P.n, L = P.cv_L;
  const int sl = s < n ? s : n - 1;
  const int pending = b < P.E ? uni(*reinterpret_cast<const int32_t*>(rec + P.o_cv_tail_pending)) : 0;
  const int t = uni(*reinterpret_cast<const int32_t*>(rec + P.o_timestep));
  const int dense = uni(*reinterpret_cast<const int32_t*>(rec + P.o_cv_dense));
  const int ev_ht = reinterpret_cast<const int32_t*>(rec + P.o_cv_ev_ht)[sl];
  const int level = *cv_ring_at(P, rec, sl, L + t), prev_level = *cv_ring_at(P, rec, sl, L + t - 1);
  TapT* taps = reinterpret_cast<TapT*>(cv_lds);
  const int rows = L + 1;
  const double* G = reinterpret_cast<const double*>(arena + P.a_cv_filters) + (int64_t)(AIE_CV_TAP_PAD_FRONT - 1) * F;
  for (int q = (int)threadIdx.x; q < rows * F; q += AIE_CV_WIN_WAVES * AIE_NT) taps[q] = (TapT)G[q];  // (row PAD_FRONT - 1 is a zero row)
  __syncthreads();
  if (!pending) return;  // the step launch did nothing for this replica (episode over, or past the batch)
  if (s == 0) *reinterpret_cast<int32_t*>(rec + P.o_cv_tail_pending) = 0;
  cv_window_tail<F, TapT>(P, arena, rec, cv_hist_base(P, arena, e), e, s, t, level, prev_level, dense, ev_ht, taps);
}
// dynamic LDS of the window-sum kernel
__host__ __device__ inline size_t aie_covid_win_lds_bytes(const aie_params& P, size_t tap_bytes) {
  return ((size_t)(P.cv_L + 1) * P.cv_F * tap_bytes + 255) & ~(size_t)255;
}
