// aie_kernels_ose.hip -- "one-step-economy" + SimpleLabor + PeriodicBracketTax
// (BASELINE configs[4]): a map-less scenario with up to 128 agents per replica.
//
// One wavefront per replica; agents are strided over the 64 lanes (lane l owns agents
// l and l + 64).  State is a few f64 vectors in LDS; the step is dominated by writing the
// observations (each agent's flat vector repeats the shared tax fragment: ~88 KB per
// replica-step at n = 100), so the shared part is built once in LDS and streamed out.
//
// Reference: F/scenarios/one_step_economy/one_step_economy.py,
// F/components/simple_labor.py, F/components/redistribution.py.
#include "aie_kernels.hip"

// Threads per replica (a multiple of the wavefront size; every wave keeps its own copy of the generator rows
// and performs the same draws, the first wave writes them back).  Two waves per replica were measured at
// BASELINE configs[4] (100 agents, 65 536 replicas): 1.49 G agent-steps/s vs 1.54 G with one -- the kernel is not
// limited by one wave's store issue rate, so one wave (and 13 replicas per CU) it stays.
#define OSE_NT 64

// development (libaie_hip_dev.so only): per-workgroup clock stamps of the step's phases (tools/ose_trace.py)
#ifdef AIE_DEV
#define OSE_STAMP(c, k)                                                                                        \
  do {                                                                                                         \
    if ((c).R.dev_trace && (c).tid == 0) (c).R.dev_trace[12 * blockIdx.x + (k)] = wall_clock64();              \
  } while (0)
// ablations (tools/ose_ablate.py): bits of aie_dev_set_skip_mask -- 1 flat rows, 2 mask rows, 4 metrics atomics,
// 8 the small observation tensors, 16 record store, 32 record load (the LDS image is then garbage: timing only)
#define OSE_SKIP(c, bit) (((c).R.dev_skip_mask & (bit)) != 0)
#else
#define OSE_STAMP(c, k) do { } while (0)
#define OSE_SKIP(c, bit) false
#endif

namespace aie {

// Per-lane registers of a replica's wave: lane l owns agents l and l + 64 (n <= 128).  The decoded SimpleLabor
// actions and the two per-agent fields a step only reads -- skill and the escrow account (aie_layout.h keeps them
// behind the generator key, outside the LDS image) -- never touch LDS.
struct OseLane {  // scalars, not arrays: an array indexed by a loop variable ends up in scratch unless the loop unrolls
  int act0, act1;
  double skill0, skill1, esc0, esc1;
  // this replica's per-agent episode accumulators behind env.metrics (mo_tax_income / mo_tax_paid), fetched with the
  // record: a tax day adds to them and stores them back with plain stores -- the wave owns its replica's block, and
  // 300 no-return atomics per replica-step (20 M per launch at BASELINE configs[4]) cost 0.31 of a 1.93 ms launch
  double met_inc0, met_inc1, met_paid0, met_paid1;
  float skobs0, skobs1;  // the SimpleLabor-skill observation (float)(skill / pmsm): the doubles die with the labor step
  __device__ __forceinline__ float skobs(int k) const { return k ? skobs1 : skobs0; }
  __device__ __forceinline__ int act(int k) const { return k ? act1 : act0; }
  __device__ __forceinline__ double skill(int k) const { return k ? skill1 : skill0; }
  __device__ __forceinline__ double esc(int k) const { return k ? esc1 : esc0; }
};

// the (at most two) agents of the calling lane: k = 0, 1 <-> agent i = tid, tid + 64
#define OSE_MY_AGENTS(k, i, n) _Pragma("unroll") for (int k = 0, i = c.tid; k < 2; ++k, i += OSE_NT) if (i < (n))

struct OseScratch {
  double* sorted;    // [n] sorted incomes / sorted coin
  double* coin;      // [n]
  double* tmp;       // [n]
  double* part;      // [n + 1]
  float* tmpl_a;     // [FA] shared part of an agent's flat vector
  float* tmpl_p;     // [FP]
};

__host__ __device__ inline size_t ose_lds_bytes(const aie_params& P) {
  size_t b = (size_t)rec_lds_bytes(P);
  b += AIE_MAX_BRACKETS * 4;
  b = (b + 15) / 16 * 16;
  b += (size_t)(4 * P.n + 2) * 8;
  b += (size_t)(pad4(P.FA > P.MA ? P.FA : P.MA) + pad4(P.FP)) * 4;  // agent row template (flat vector, then mask) + planner's
  return (b + 15) / 16 * 16;
}

__device__ __forceinline__ Ctx ose_make_ctx(const aie_params& P, const aie_params& R, uint8_t* lds, int e, int tid,
                                             OseScratch& s, uint8_t* arena) {
  uint8_t* q = lds + rec_lds_bytes(P);
  int32_t* act_p = reinterpret_cast<int32_t*>(q);
  q += AIE_MAX_BRACKETS * 4;
  q = lds + ((q - lds) + 15) / 16 * 16;
  s.sorted = reinterpret_cast<double*>(q);
  s.coin = s.sorted + P.n;
  s.tmp = s.coin + P.n;
  s.part = s.tmp + P.n;
  q += (4 * P.n + 2) * 8;
  s.tmpl_a = reinterpret_cast<float*>(q);
  s.tmpl_p = s.tmpl_a + pad4(P.FA > P.MA ? P.FA : P.MA);
  uint8_t* met = arena + R.a_metrics + (int64_t)e * P.met_bytes;
  int32_t* ev = e < R.ev_replicas ? reinterpret_cast<int32_t*>(arena + R.a_events + (int64_t)e * R.ev_stride) : nullptr;
  return Ctx{P, R, lds, act_p, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, met, arena, ev, P.c.tax_model == AIE_TAX_SAEZ, true, tid, e,
             /*rtab=*/R.c.tax_disc_rates, /*mtab=*/P.mask_test, /*mtwin=*/nullptr, /*skipm=*/0};
}

// The draws of np.random.permutation(n) (World.get_random_order_agents, world.py:418-422) for a caller that never
// looks at the order: only the position of the stream afterwards matters.  Fisher-Yates index i = n-1 .. 1 consumes
// 32-bit words until one satisfies (word & mask(i)) <= i.  A block of up to 64 consecutive words of the generator
// window is tempered at once (lane l: word pos + l) and all indices that share a mask (i in [2^k, 2^(k+1))) are
// resolved together: word l of the block is accepted iff (w_l & mask) <= i0 - (accepted words before l) -- a fixed
// point over the lanes (A -> ballot(v_l <= i0 - popcount(A below l))); its solution is unique (induction over l) and
// equals the sequential loop's accept set, and the iteration reaches it because after k rounds the first k lanes are
// final.  ~40 instructions per (block, mask group) instead of ~12 per index: 10 us -> 3 us of a C5 wave's step.
__device__ __forceinline__ void rng_skip_permutation(MT& m, int lane, int n) {
  int i = n - 1;
  while (i >= 1) {
    if (m.pos >= AIE_MT_N) {
      if (m.fast) {  // the counter stream: the next block (its words are computed where they are read, no rows)
        m.fblk += 1u;
        m.twists += 1;
      } else {
        mt_twist(m, lane);
      }
      m.pos = 0;
    }
    const int pos = m.pos, cnt = min(64, AIE_MT_N - pos);
    uint32_t w;
    if (m.fast) {  // word pos + lane: element (pos + lane) & 1 of pair (pos + lane) >> 1
      uint32_t x0, x1;
      fast_pair(m.fkey, m.fblk, m.fsalt, (pos + lane) >> 1, x0, x1);
      w = ((pos + lane) & 1) ? x1 : x0;
    } else {
      // word pos + lane lives in row (pos + lane) >> 6, lane (pos + lane) & 63: at most two adjacent rows
      const int row_a = pos >> 6, src = (pos + lane) & 63;
      uint32_t ra = m.r[0], rb = m.r[1];
#pragma unroll
      for (int j = 1; j < 10; ++j) {
        ra = (row_a == j) ? m.r[j] : ra;
        rb = (row_a + 1 == j) ? m.r[j] : rb;
      }
      const uint32_t wa = lane_get(ra, src), wb = lane_get(rb, src);
      w = mt_temper((pos & 63) + lane < 64 ? wa : wb);
    }
    int start = 0;  // first word of the block not consumed yet
    while (i >= 1 && start < cnt) {
      uint32_t mask = (uint32_t)i;
      mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8;
      const int g_lo = (int)(mask >> 1) + 1;  // the smallest index with this mask
      const int need = i - g_lo + 1;          // accepted words that finish the group
      const bool valid = lane >= start && lane < cnt;
      const int v = (int)(w & mask);
      uint64_t A = __ballot(valid && v <= i);
      for (;;) {
        const int before = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(A >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)A, 0u));
        const uint64_t A2 = __ballot(valid && v <= i - before);
        if (A2 == A) break;
        A = A2;
      }
      const int total = __popcll(A);
      if (total >= need) {  // the group ends inside the block: behind its need-th accepted word
        const int before = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(A >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)A, 0u));
        const uint64_t last = __ballot(((A >> lane) & 1ull) && before == need - 1);
        start = __ffsll((unsigned long long)last);  // (index of that word) + 1
        i = g_lo - 1;
      } else {
        i -= total;
        start = cnt;
      }
    }
    m.pos = pos + start;
  }
}

// Record HBM -> LDS (the image before the generator key), key -> registers, cold per-agent fields -> registers.
// Every load of a lane is issued before the first one is waited for: a replica starts while the other ~2800 resident
// waves are streaming their observation rows out, and under that store traffic a load round trip takes several
// microseconds -- the copy loop with one round trip per 16 bytes per lane cost 22 of a wave's 93 us (tools/ose_trace.py).
// `early_perm`: SimpleLabor is the first component, so the first thing the step does with the generator is skipping its
// agent-order permutation -- which needs the key rows and the position, not the record: it runs between issuing the
// record's loads and waiting for them (~8 us of work under a ~12 us load round trip); m.pos is then the position
// behind the permutation.
__device__ __forceinline__ void ose_load_record(const Ctx& c, const uint8_t* __restrict__ arena, MT& m, OseLane& L,
                                                bool early_perm = false) {
  const uint8_t* g = arena + c.R.a_records + (int64_t)c.e * c.P.rec_bytes;
  const uint4* src = reinterpret_cast<const uint4*>(g);
  uint4* dst = reinterpret_cast<uint4*>(c.rec);
  const int nq = rec_lds_bytes(c.P) >> 4;
  const int lane = c.tid & 63;
  const uint32_t* key = reinterpret_cast<const uint32_t*>(g + c.P.o_mt);
  const double* gs = reinterpret_cast<const double*>(g + c.P.o_skill);
  const double* ge = reinterpret_cast<const double*>(g + c.P.o_esc_coin);
  mt_init(m, c.P);
  if (m.fast) {  // the counter stream: 16 bytes of state (they also travel with the image); rows only where a component
                 // draws sequentially (ose_tax_component_step)
    m.fkey = (uint32_t)uni((int)key[0]);
    m.fblk = (uint32_t)uni((int)key[1]);
    m.fsalt = (uint32_t)uni((int)key[2]);
  } else {
#pragma unroll
    for (int j = 0; j < 9; ++j) m.r[j] = key[64 * j + lane];
    m.r[9] = lane < 48 ? key[576 + lane] : 0u;
  }
  m.twists = 0;
  m.pos = early_perm ? uni(*reinterpret_cast<const int32_t*>(g + c.P.o_mt_pos)) : 0;
  L.skill0 = c.tid < c.P.n ? gs[c.tid] : 0.0;
  L.esc0 = c.tid < c.P.n ? ge[c.tid] : 0.0;
  L.skill1 = c.tid + OSE_NT < c.P.n ? gs[c.tid + OSE_NT] : 0.0;
  L.esc1 = c.tid + OSE_NT < c.P.n ? ge[c.tid + OSE_NT] : 0.0;
  L.skobs0 = (float)(L.skill0 / c.R.c.labor_pmsm);
  L.skobs1 = (float)(L.skill1 / c.R.c.labor_pmsm);
  L.met_inc0 = L.met_inc1 = L.met_paid0 = L.met_paid1 = 0.0;
  if (c.P.has_tax && c.met) {
    const double* mi = reinterpret_cast<const double*>(c.met + c.P.mo_tax_income);
    const double* mp = reinterpret_cast<const double*>(c.met + c.P.mo_tax_paid);
    if (c.tid < c.P.n) { L.met_inc0 = mi[c.tid]; L.met_paid0 = mp[c.tid]; }
    if (c.tid + OSE_NT < c.P.n) { L.met_inc1 = mi[c.tid + OSE_NT]; L.met_paid1 = mp[c.tid + OSE_NT]; }
  }
  // eight 16-byte loads in flight per lane and batch (one batch covers records up to 8 KiB); scalars rather than an
  // array: the array stayed in scratch memory
  // Fields every step overwrites before it reads them are not fetched: with tax_period == 1 every step is a tax day,
  // which rewrites last_income / last_marginal_rate (adjacent in the record) before the observations look at them.
  // HBM reads mixed into the launch's store stream cost about twice their byte share (tools/phase_overlap.hip).
  const bool dead = c.P.has_tax && c.P.c.tax_period == 1;
  const int dead_lo = (c.P.o_tax_last_income + 15) >> 4, dead_hi = (c.P.o_tax_last_marginal_rate + 8 * c.P.n) >> 4;
  const int nq_eff = OSE_SKIP(c, 32) ? 0 : nq;
#define OSE_LIVE(k) (q0 + (k) * OSE_NT < nq_eff && !(dead && q0 + (k) * OSE_NT >= dead_lo && q0 + (k) * OSE_NT < dead_hi))
#define OSE_LD(k) if (OSE_LIVE(k)) v##k = src[q0 + (k) * OSE_NT];
#define OSE_ST(k) if (OSE_LIVE(k)) dst[q0 + (k) * OSE_NT] = v##k;
  {  // the first batch, with the permutation skip between its loads and its LDS writes (wave-uniform control flow)
    const int q0 = c.tid;
    uint4 v0, v1, v2, v3, v4, v5, v6, v7;
    OSE_LD(0) OSE_LD(1) OSE_LD(2) OSE_LD(3) OSE_LD(4) OSE_LD(5) OSE_LD(6) OSE_LD(7)
    if (early_perm) rng_skip_permutation(m, lane, c.P.n);
    OSE_ST(0) OSE_ST(1) OSE_ST(2) OSE_ST(3) OSE_ST(4) OSE_ST(5) OSE_ST(6) OSE_ST(7)
  }
  for (int q0 = c.tid + 8 * OSE_NT; q0 < nq_eff; q0 += 8 * OSE_NT) {
    uint4 v0, v1, v2, v3, v4, v5, v6, v7;
    OSE_LD(0) OSE_LD(1) OSE_LD(2) OSE_LD(3) OSE_LD(4) OSE_LD(5) OSE_LD(6) OSE_LD(7)
    OSE_ST(0) OSE_ST(1) OSE_ST(2) OSE_ST(3) OSE_ST(4) OSE_ST(5) OSE_ST(6) OSE_ST(7)
  }
#undef OSE_LD
#undef OSE_ST
#undef OSE_LIVE
}

// SimpleLabor.component_step simple_labor.py:105-126.  The random agent order
// (world.py:418-422) is drawn -- it advances the stream -- but the result does not depend
// on it, so the update itself runs one lane per agent.
__device__ __forceinline__ void labor_component_step(const Ctx& c, const OseScratch& s, MT& m, const OseLane& L,
                                                     bool perm_done = false) {
  const int n = c.P.n;
  if (!perm_done) rng_skip_permutation(m, c.tid & 63, n);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int i = c.tid + OSE_NT * k, a = L.act(k);
    if (i < n && a != 0) {
      R_F64(c, o_labor)[i] = (double)a;  // hours worked this step (set, not accumulated)
      const double payoff = (double)a * L.skill(k);
      R_F64(c, o_production)[i] += payoff;
      R_F64(c, o_inv_coin)[i] += payoff;
    }
  }
  __syncthreads();
}

// WealthRedistribution.component_step, F/components/redistribution.py:46-65
__device__ __forceinline__ void ose_wealth_component_step(const Ctx& c, const OseScratch& s, const OseLane& L) {
  const int n = c.P.n;
  __syncthreads();
  OSE_MY_AGENTS(k, i, n) s.tmp[i] = R_F64(c, o_inv_coin)[i] + L.esc(k);
  __syncthreads();
  const double share = np_sum_small(s.tmp, n) / (double)n;  // every lane, same value
  OSE_MY_AGENTS(k, i, n) R_F64(c, o_inv_coin)[i] = share - L.esc(k);
  __syncthreads();
}

// PeriodicBracketTax.component_step :945-972 with enact_taxes :853-915
__device__ __forceinline__ void ose_tax_component_step(const Ctx& c, const OseScratch& s, MT& m, OseLane& L) {
  const int n = c.P.n;
  int pos = uni(*R_I32(c, o_tax_cycle_pos));
  if (pos == 1 && c.saez) {  // compute_and_set_new_period_rates_from_saez_formula (see tax_component_step)
    uint8_t* blk = saez_block(c);
    if (uni(reinterpret_cast<const int32_t*>(blk)[1])) {
      if (c.tid < c.P.NB) R_F64(c, o_tax_saez_rates)[c.tid] = reinterpret_cast<const double*>(blk + AIE_SAEZ_OFF_NEXT)[c.tid];
    } else {
      const double lo = c.R.c.tax_rate_min;
      const double hi = c.P.c.tax_annealing ? tax_curr_rate_max(c) : c.R.c.tax_rate_max;
      if (m.fast && m.pos < AIE_MT_N) mt_fast_rows(m, c.tid & 63);  // (sequential draws read the block's rows)
      for (int b = 0; b < c.P.NB; ++b) {
        const double r = lo + (hi - lo) * rng_double(m, c.tid & 63);
        if (c.tid == b) R_F64(c, o_tax_saez_rates)[b] = r;
      }
    }
    __syncthreads();
    if (c.tid < c.P.NB) R_F64(c, o_tax_saez_obs_rates)[c.tid] = tax_rate(c, c.tid);
    __syncthreads();
  }
  if (pos == 1 && c.P.c.tax_model == AIE_TAX_MODEL_WRAPPER && !c.P.c.tax_disable) {
    if (c.tid < c.P.NB) {
      const int a = c.act_p[c.tid];
      if (a > 0 && a <= c.P.c.tax_n_disc_rates) R_I32(c, o_tax_rate_idx)[c.tid] = a - 1;
    }
    __syncthreads();
  }
  if (pos >= c.P.c.tax_period) {
    int bin0 = -1, bin1 = -1;  // income bracket of this lane's agents (-1: no such agent)
    OSE_MY_AGENTS(k, i, n) {
      const double coin = R_F64(c, o_inv_coin)[i];
      const double income = (coin + L.esc(k)) - R_F64(c, o_tax_last_coin)[i];
      const double due = tax_due(c, income);
      const double eff = coin < due ? coin : due;
      R_F64(c, o_tax_last_marginal_rate)[i] = tax_marginal_rate(c, income);
      R_F64(c, o_tax_last_income)[i] = income;
      R_F64(c, o_inv_coin)[i] = coin - eff;
      s.tmp[i] = eff;
      if (c.ev) {  // dense log: one AIE_EV_TAX row per agent, in agent order
        int32_t* row = c.ev + 4 + (c.P.NB + i) * AIE_EV_WORDS;
        row[0] = AIE_EV_TAX; row[1] = i;
        for (int q = 2; q < 10; ++q) row[q] = 0;
        *reinterpret_cast<double*>(row + 10) = eff;
        if (i == 0) c.ev[0] = c.P.NB + n;
      }
      // episode accumulators for get_metrics :1141-1186: plain read-modify-write of the values fetched with the record
      int bin = 0;  // income_bin :828-835
      if (income >= 0)
        for (int b = 0; b < c.P.NB; ++b)
          if (income >= c.R.c.tax_bracket_cutoffs[b] && (b + 1 == c.P.NB || income < c.R.c.tax_bracket_cutoffs[b + 1])) { bin = b; break; }
      if (k) { L.met_inc1 += income > 0 ? income : 0.0; L.met_paid1 += eff; bin1 = bin; }
      else { L.met_inc0 += income > 0 ? income : 0.0; L.met_paid0 += eff; bin0 = bin; }
      if (!OSE_SKIP(c, 4)) {
        reinterpret_cast<double*>(c.met + c.P.mo_tax_income)[i] = k ? L.met_inc1 : L.met_inc0;
        reinterpret_cast<double*>(c.met + c.P.mo_tax_paid)[i] = k ? L.met_paid1 : L.met_paid0;
      }
    }
    {  // bracket occupancy: one ballot per bracket, lane b adds its bracket's count (NB no-return atomics per replica)
      int mine = 0;
      for (int b = 0; b < c.P.NB; ++b) {
        const int cnt = __popcll(__ballot(bin0 == b)) + __popcll(__ballot(bin1 == b));
        mine = c.tid == b ? cnt : mine;
      }
      if (c.tid < c.P.NB && mine && !OSE_SKIP(c, 4)) atomicAdd(reinterpret_cast<int32_t*>(c.met + c.P.mo_tax_occ) + c.tid, mine);
    }
    if (c.tid < c.P.NB) unsafeAtomicAdd(reinterpret_cast<double*>(c.met + c.P.mo_tax_sched) + c.tid, tax_rate(c, c.tid));
    if (c.ev && c.tid < c.P.NB) {  // the day's schedule: AIE_EV_TAX_BRACKET rows first
      int32_t* row = c.ev + 4 + c.tid * AIE_EV_WORDS;
      row[0] = AIE_EV_TAX_BRACKET; row[1] = c.tid;
      for (int q = 2; q < 10; ++q) row[q] = 0;
      *reinterpret_cast<double*>(row + 10) = tax_rate(c, c.tid);
    }
    __syncthreads();
    // running sums in agent order, as the reference accumulates them; the n divisions of the effective rates are
    // done one lane per agent first (s.sorted is free here), the sequential part is additions only
    for (int i = c.tid; i < n; i += OSE_NT) {
      const double inc = R_F64(c, o_tax_last_income)[i];
      s.sorted[i] = s.tmp[i] / (inc > 0.000001 ? inc : 0.000001);
    }
    __syncthreads();
    double net = 0, day = 0;
#pragma unroll 10
    for (int j = 0; j < n; ++j) {
      net += s.tmp[j];
      day += s.sorted[j];
    }
    if (c.tid == 0) {
      unsafeAtomicAdd(reinterpret_cast<double*>(c.met + c.P.mo_tax_eff), day);
      atomicAdd(reinterpret_cast<int32_t*>(c.met + c.P.mo_tax_days), 1);
    }
    const double lump = net / (double)n;
    OSE_MY_AGENTS(k, i, n) {
      const double v = R_F64(c, o_inv_coin)[i] + lump;
      R_F64(c, o_inv_coin)[i] = v;
      R_F64(c, o_tax_last_coin)[i] = v + L.esc(k);
    }
    if (c.tid == 0) *R_F64(c, o_tax_total_collected) += net;
    if (c.saez) {  // _update_saez_buffer :533-541 (one wavefront per replica: OSE_NT == 64)
      static_assert(OSE_NT == AIE_NT, "the in-place move below relies on one wavefront per replica");
      uint8_t* blk = saez_block(c);
      int32_t* hdr = reinterpret_cast<int32_t*>(blk);
      double* buf = reinterpret_cast<double*>(blk + AIE_SAEZ_OFF_BUF);
      int len = uni(hdr[0]);
      for (int i = c.tid; i < n; i += OSE_NT) {
        buf[2 * (len + i)] = R_F64(c, o_tax_last_income)[i];
        buf[2 * (len + i) + 1] = R_F64(c, o_tax_last_marginal_rate)[i];
      }
      len += n;
      const int size = c.R.c.saez_buffer_size;
      if (len > size) {  // drop the oldest: chunk by chunk, a chunk's loads precede its stores
        const int shift = 2 * (len - size);
        __builtin_amdgcn_s_waitcnt(0);
        for (int base = 0; base < 2 * size; base += OSE_NT) {
          const int q = base + c.tid;
          double v = 0;
          if (q < 2 * size) v = buf[q + shift];
          __builtin_amdgcn_s_waitcnt(0);
          if (q < 2 * size) buf[q] = v;
        }
        len = size;
      }
      if (c.tid == 0) {
        hdr[0] = len;
        hdr[2] += n;  // _additions_this_episode :541
      }
    }
    pos = 0;
    __syncthreads();
  }
  if (c.tid == 0) *R_I32(c, o_tax_cycle_pos) = pos + 1;
  __syncthreads();
}

// ascending (stable) sort of src[0..n) into dst by counting ranks; a lane ranks its two elements (i, i + 64) in one
// sweep, so every src[j] is read from LDS once per lane
__device__ __forceinline__ void rank_sort(const double* src, double* dst, int n, int tid) {
  // np.sort of n <= 128 doubles: a bitonic network over two values per lane (element lane and element lane + 64,
  // the tail padded with +inf), 28 compare-exchange rounds of which 27 fetch the partner with a lane permute --
  // ~350 instructions instead of the ~1 200 of counting, per element, how many others sort before it.  (A sorted
  // array does not depend on the algorithm; equal keys are interchangeable.)
  static_assert(OSE_NT == 64, "two elements per lane cover n <= 128");
  const double inf = __builtin_huge_val();
  double x0 = tid < n ? src[tid] : inf, x1 = tid + OSE_NT < n ? src[tid + OSE_NT] : inf;
#pragma unroll
  for (int k = 2; k <= 128; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j == 64) {  // partners sit in the same lane; k == 128: one ascending block
        const bool sw = x1 < x0;
        const double lo = sw ? x1 : x0, hi = sw ? x0 : x1;
        x0 = lo;
        x1 = hi;
      } else {
        const double p0 = __shfl_xor(x0, j, 64), p1 = __shfl_xor(x1, j, 64);
        const bool lower = (tid & j) == 0;                   // the lower index of its pair
        const bool up0 = (tid & k) == 0;                     // element index tid
        const bool up1 = ((tid + 64) & k) == 0;              // element index tid + 64
        const bool lt0 = p0 < x0, lt1 = p1 < x1;
        // ascending block: the lower index keeps the smaller value; descending block: the larger one
        x0 = (lower == up0) ? (lt0 ? p0 : x0) : (lt0 ? x0 : p0);
        x1 = (lower == up1) ? (lt1 ? p1 : x1) : (lt1 ? x1 : p1);
      }
    }
  }
  if (tid < n) dst[tid] = x0;
  if (tid + OSE_NT < n) dst[tid + OSE_NT] = x1;
}

// social_metrics.get_gini (social_metrics.py:10-46) of s.coin; called by every thread, the result is valid on
// thread 0; the sorted copy must be in s.sorted for n >= 30.  The running sums of the sorted-cumsum branch are
// sequential (one thread), the 100 divisions by the total and nothing else are spread over the lanes.
__device__ __forceinline__ double ose_gini(const OseScratch& s, double* cs, int n, int tid) {
  __shared__ double s_tot;
  if (n < 30) {  // the flattened n x n difference matrix in NumPy's pairwise order (np_sum_seq, whole wave)
    const double diff = np_sum_seq<3>(s.coin, n, 1, 65536u / (uint32_t)n + 1u, 0, n * n, tid);
    const double unscaled = diff / (2 * n * np_sum_small(s.coin, n) + 1e-10);
    return unscaled / ((double)(n - 1) / (double)n);
  }
  __syncthreads();
  if (tid == 0) {
    s_tot = np_sum_small(s.sorted, n) + 1e-10;
    double run = 0;
#pragma unroll 10
    for (int i = 0; i < n; ++i) {
      run += s.sorted[i];
      cs[i] = run;
    }
  }
  __syncthreads();
  const double tot = s_tot;
  for (int i = tid; i < n; i += OSE_NT) cs[i] = cs[i] / tot;
  __syncthreads();
  return tid == 0 ? 1 - (2.0 / (n + 1)) * np_sum_small(cs, n) : 0.0;
}

// get_current_optimization_metrics one_step_economy.py:280-336 -> s.part[0..n]
__device__ __forceinline__ void ose_metrics(const Ctx& c, const OseScratch& s, const OseLane& L) {
  const aie_params& P = c.P;
  const int n = P.n;
  OSE_MY_AGENTS(k, i, n) {
    const double coin = R_F64(c, o_inv_coin)[i] + L.esc(k);
    const double labor = R_F64(c, o_labor)[i];
    s.coin[i] = coin;
    double u;
    if (P.c.ose_agent_reward_type == AIE_AGENT_REW_ISOELASTIC) {
      const double eta = c.R.c.isoelastic_eta;
      const double uc = c.P.sh_eta_is_one ? aie_log_glibc(coin > 1 ? coin : 1) : (aie_pow_glibc(coin, 1 - eta) - 1) / (1 - eta);
      u = uc - labor * c.R.c.ose_labor_cost;
    } else {
      u = coin - aie_pow_glibc(labor, c.R.c.ose_labor_exponent) * c.R.c.ose_labor_cost;
    }
    s.part[i] = u;
  }
  __syncthreads();
  const int prt = P.c.planner_reward_type;
  if (prt == AIE_PLANNER_REW_COIN_EQ_TIMES_PROD) {
    if (n >= 30) rank_sort(s.coin, s.sorted, n, c.tid);
    __syncthreads();
    const double gini = ose_gini(s, s.tmp, n, c.tid);
    if (c.tid == 0) {
      const double ew = 1 - c.R.c.mixing_weight_gini_vs_coin;
      const double prod = np_sum_small(s.coin, n) / n;
      s.part[n] = (ew * (1 - gini) + (1 - ew)) * prod;
    }
  } else {
    const bool use_util = prt == AIE_PLANNER_REW_INV_INCOME_UTIL;
    for (int i = c.tid; i < n; i += OSE_NT) {
      const double base = use_util ? R_F64(c, o_production)[i] : s.coin[i];
      s.tmp[i] = 1 / (base > 1 ? base : 1);
    }
    __syncthreads();
    const double sw = np_sum_small(s.tmp, n);  // every lane, same value
    for (int i = c.tid; i < n; i += OSE_NT) s.sorted[i] = (use_util ? s.part[i] : s.coin[i]) * (s.tmp[i] / sw);
    __syncthreads();
    if (c.tid == 0) s.part[n] = np_sum_small(s.sorted, n);
  }
  __syncthreads();
}

// n rows of F floats at g (row i at float offset i * F: only dword-aligned), every row = the same template with up to
// two per-row entries patched (entry ip0 <- (float)p0[i], entry ip1 <- (float)p1[i]; -1 = none).  A lane keeps ONE
// 16-byte quad of the template in registers for the whole loop and L4 = ceil(F / 4) neighbouring lanes write one row
// with one store instruction; 64 / L4 rows go out per instruction (global dwordx4 accesses need only dword alignment
// on gfx950).  ~6 instructions per pass instead of ~45 per quad when every element is looked up and tested.
__device__ __forceinline__ void ose_store_rows(BufRsrc g, int n, int F, const float* tmpl, int lane, int ip0,
                                               const double* p0, int ip1, const double* p1) {
  const int L4 = (F + 3) >> 2;
  if (L4 > 64) return;  // (F <= 256 always: n <= 128)
  const int rpp = 64 / L4;                    // rows per pass
  const int sub = lane / L4, l = lane - sub * L4;
  const bool active = sub < rpp;
  const int j0 = 4 * l;
  float t0 = 0, t1 = 0, t2 = 0, t3 = 0;
  if (active) {
    t0 = tmpl[j0];
    if (j0 + 1 < F) t1 = tmpl[j0 + 1];
    if (j0 + 2 < F) t2 = tmpl[j0 + 2];
    if (j0 + 3 < F) t3 = tmpl[j0 + 3];
  }
  const int u0 = ip0 - j0, u1 = ip1 - j0;     // 0..3 on the lane that owns a patched entry
  const bool own0 = active && ip0 >= 0 && (unsigned)u0 < 4u, own1 = active && ip1 >= 0 && (unsigned)u1 < 4u;
  const int width = F - j0 >= 4 ? 4 : F - j0; // dwords this lane writes (the last quad of a row may be short)
  for (int r0 = 0; r0 < n; r0 += rpp) {
    const int i = r0 + sub;
    if (!active || i >= n) continue;
    float v0 = t0, v1 = t1, v2 = t2, v3 = t3;
    if (own0) {
      const float x = (float)p0[i];
      v0 = u0 == 0 ? x : v0; v1 = u0 == 1 ? x : v1; v2 = u0 == 2 ? x : v2; v3 = u0 == 3 ? x : v3;
    }
    if (own1) {
      const float x = (float)p1[i];
      v0 = u1 == 0 ? x : v0; v1 = u1 == 1 ? x : v1; v2 = u1 == 2 ? x : v2; v3 = u1 == 3 ? x : v3;
    }
    const int off = 4 * (i * F + j0);
    if (width == 4) buf_store_f32x4(g, v0, v1, v2, v3, off, 0);
    else {
      buf_store_f32(g, v0, off, 0);
      if (width > 1) buf_store_f32(g, v1, off + 4, 0);
      if (width > 2) buf_store_f32(g, v2, off + 8, 0);
    }
  }
}

// Observations + masks (one_step_economy.py:120-176, simple_labor.py:97-103,128-134,
// redistribution.py:974-1104), flat vectors in sorted-key order (base_env.py:561-612).
// The agents' action masks (simple_labor.py:97-103 through base_env.py:706-756): [NO-OP, hours...] (single-action) /
// [NO-OP, hours...] of the only subspace (multi-action) -- every agent's row is the same, and it depends on nothing but
// "is this the episode's first observation" (labor_mask_first_step).  The step kernel therefore issues these
// n x MA floats (40 KB of a replica's 88 KB at BASELINE configs[4]) right after the record arrived, so that they drain
// while the wave computes, instead of behind the flat vectors at the end of its life.
__device__ __forceinline__ void ose_store_agent_masks(const Ctx& c, const OseScratch& s, uint8_t* __restrict__ arena,
                                                      bool first_observation) {
  const aie_params& P = c.P;
  const float on = (first_observation && P.c.labor_mask_first_step) ? 0.0f : 1.0f;
  const BufRsrc g = make_rsrc(arena + c.R.a_obs_a_mask + (int64_t)c.e * P.n * P.MA * 4, (uint32_t)(P.n * P.MA * 4));
  __syncthreads();  // the row template goes through the agent template area
  for (int q = c.tid; q < P.MA; q += OSE_NT) s.tmpl_a[q] = (q == 0 || P.n_sub_a == 0) ? 1.0f : on;
  __syncthreads();
  if (!OSE_SKIP(c, 2)) ose_store_rows(g, P.n, P.MA, s.tmpl_a, c.tid, -1, nullptr, -1, nullptr);
  __syncthreads();
}

__device__ __forceinline__ void ose_write_observations(const Ctx& c, const OseScratch& s, uint8_t* __restrict__ arena,
                                                       const OseLane& L, bool at_reset = false,
                                                       bool agent_masks_done = false) {
  const aie_params& P = c.P;
  const int n = P.n, NB = P.NB, tid = c.tid;
  const int t = *R_I32(c, o_timestep);
  const float tval = (float)((double)t / (P.c.allow_observation_scaling ? (double)c.R.c.episode_length : 1.0));
  // ---- shared quantities ----
  if (P.has_tax) {
    const double per = (double)c.P.c.tax_period;
    for (int i = tid; i < n; i += OSE_NT) s.tmp[i] = R_F64(c, o_tax_last_income)[i] / per;
    __syncthreads();
    rank_sort(s.tmp, s.sorted, n, tid);
    __syncthreads();
    const int pos = *R_I32(c, o_tax_cycle_pos);
    for (int j = tid; j < NB + n + 4; j += OSE_NT) {
      float v;
      if (j < NB) v = (float)tax_rate_obs(c, j);
      else if (j == NB) v = pos == 1 ? 1.0f : 0.0f;               // is_first_day
      else if (j == NB + 1) v = pos >= c.P.c.tax_period ? 1.0f : 0.0f;  // is_tax_day
      else if (j < NB + 2 + n) v = (float)s.sorted[j - NB - 2];   // last_incomes (sorted)
      else v = (float)((double)pos / per);                        // tax_phase
      if (j != NB + 2 + n) s.tmpl_a[P.fa_tax + j] = v;            // [NB+2+n] = marginal_rate: per agent
      if (j < NB + 2 + n) s.tmpl_p[P.fp_tax + j] = v;
      else if (j == NB + 3 + n) s.tmpl_p[P.fp_tax + NB + 2 + n] = v;
    }
  }
  OSE_MY_AGENTS(k, i, n) {
    const double coin = R_F64(c, o_inv_coin)[i] + L.esc(k);
    s.coin[i] = coin;
    if (P.has_tax) s.tmp[i] = tax_marginal_rate(c, coin - R_F64(c, o_tax_last_coin)[i]);
  }
  if (tid == 0) {
    s.tmpl_a[P.fa_time] = tval;
    s.tmpl_p[P.fp_time] = tval;
    reinterpret_cast<float*>(arena + c.R.a_obs_p_time)[c.e] = tval;
  }
  __syncthreads();
  OSE_STAMP(c, 4);
  // planner world-equality / world-normalized_per_capita_productivity (:161-172)
  if (n >= 30) rank_sort(s.coin, s.sorted, n, tid);
  __syncthreads();
  const double gini = ose_gini(s, s.part, n, tid);  // s.part is free here (rewards are computed afterwards)
  if (tid == 0) {
    s.tmpl_p[P.fp_world + 0] = (float)(1 - gini);
    s.tmpl_p[P.fp_world + 1] = (float)(np_sum_small(s.coin, n) / n / 1000);
  }
  __syncthreads();
  OSE_STAMP(c, 5);
  // ---- agent flat vectors: the shared template with two per-agent entries ----
  {
    const BufRsrc g = make_rsrc(arena + c.R.a_obs_a_flat + (int64_t)c.e * n * P.FA * 4, (uint32_t)(n * P.FA * 4));
    const int i_mr = P.has_tax ? P.fa_tax + NB + 2 + n : -1;
    const int i_sk = P.has_labor ? P.fa_labor : -1;
    if (i_sk >= 0) {  // SimpleLabor-skill = skill / pmsm: n divisions, one lane per agent (s.part is free until the rewards)
      OSE_MY_AGENTS(k, i, n) s.part[i] = (double)L.skobs(k);  // (float)(skill / pmsm), simple_labor.py:128-134
      __syncthreads();
    }
    if (!OSE_SKIP(c, 1)) ose_store_rows(g, n, P.FA, s.tmpl_a, tid, i_mr, s.tmp, i_sk, s.part);
    float* gt = reinterpret_cast<float*>(arena + c.R.a_obs_a_time) + (int64_t)c.e * n;
    if (!OSE_SKIP(c, 8)) for (int i = tid; i < n; i += OSE_NT) gt[i] = tval;
    if (P.FPA && !OSE_SKIP(c, 8)) {
      float* gp = reinterpret_cast<float*>(arena + c.R.a_obs_p_agents) + (int64_t)c.e * n * P.FPA;
      for (int i = tid; i < n; i += OSE_NT) {
        gp[i * 3 + 0] = (float)s.tmp[i];
        gp[i * 3 + 1] = (float)(R_F64(c, o_tax_last_income)[i] / (double)c.P.c.tax_period);
        gp[i * 3 + 2] = (float)R_F64(c, o_tax_last_marginal_rate)[i];
      }
    }
    if (!OSE_SKIP(c, 8)) stream_out(s.tmpl_p, reinterpret_cast<float*>(arena + c.R.a_obs_p_flat) + (int64_t)c.e * P.FP, P.FP, tid);
  }
  OSE_STAMP(c, 6);
  // ---- masks ----
  {
    if (!agent_masks_done) {
      const bool first = P.has_labor && *R_I32(c, o_first_step) != 0;
      ose_store_agent_masks(c, s, arena, first);
    }
    if (at_reset && P.has_tax && P.c.tax_annealing) {  // generate_masks refreshes _last_completions after the reset's observations
      __syncthreads();
      if (tid == 0) *R_I32(c, o_tax_last_completions) = *R_I32(c, o_completions);
      __syncthreads();
    }
    const bool pmulti = P.c.multi_action_mode_planner != 0;
    const float open = (P.n_sub_p && *R_I32(c, o_tax_cycle_pos) == 1) ? 1.0f : 0.0f;
    float* gp = reinterpret_cast<float*>(arena + c.R.a_obs_p_mask) + (int64_t)c.e * P.MP;
    for (int q = tid; q < P.MP; q += OSE_NT) {
      int j = -1;  // index of the discretised rate this entry stands for (-1: a NO-OP entry)
      if (P.n_sub_p == 0) j = -1;
      else if (pmulti) j = q - udiv(q, 1 + P.sub_p_dim, P.mg_sub_p) * (1 + P.sub_p_dim) - 1;
      else if (q > 0) j = (q - 1) - udiv(q - 1, P.sub_p_dim, P.mg_sub_p_dim) * P.sub_p_dim;
      gp[q] = (j < 0 || (open != 0.0f && tax_rate_action_visible(c, j))) ? 1.0f : 0.0f;
    }
  }
  __syncthreads();
  OSE_STAMP(c, 7);
  if (tid == 0 && P.has_labor) *R_I32(c, o_first_step) = 0;
}

// The generator's rows go back to HBM only when the step twisted them (a step draws ~130 of a window's 624 words;
// the position is a record field), and as soon as the components are done: ten registers less for the rest of the step.
__device__ __forceinline__ void ose_store_key(const Ctx& c, uint8_t* __restrict__ arena, const MT& m) {
  if (m.twists == 0 || c.tid >= 64) return;  // (wave-uniform; every wave holds the same rows)
  if (m.fast) {  // the block number, in the record's LDS image (ose_store_record follows behind a barrier)
    if (c.tid == 0) R_U32(c, o_mt)[1] = m.fblk;
    return;
  }
  uint32_t* key = reinterpret_cast<uint32_t*>(arena + c.R.a_records + (int64_t)c.e * c.P.rec_bytes + c.P.o_mt);
#pragma unroll
  for (int j = 0; j < 9; ++j) key[64 * j + c.tid] = m.r[j];
  if (c.tid < 48) key[576 + c.tid] = m.r[9];
}
__device__ __forceinline__ void ose_store_record(const Ctx& c, uint8_t* __restrict__ arena) {
  uint8_t* g = arena + c.R.a_records + (int64_t)c.e * c.P.rec_bytes;
  uint4* dst = reinterpret_cast<uint4*>(g);
  const uint4* src = reinterpret_cast<const uint4*>(c.rec);
  const int nq = rec_lds_bytes(c.P) >> 4;
  if (OSE_SKIP(c, 16)) return;
  for (int q = c.tid; q < nq; q += OSE_NT) dst[q] = src[q];
}

}  // namespace aie

// reset: one_step_economy.py:99-118 + simple_labor.py:76-95 + redistribution.py:1109-1139
// + additional_reset_steps :224-241.  No random draws.  The record is in LDS; `keep_rewards`: called from the step
// kernel for a replica that just finished its episode (auto-reset): the terminal step's rewards / done stay.
namespace aie {
__device__ __forceinline__ void ose_reset_body(const Ctx& c, const OseScratch& s, uint8_t* __restrict__ arena,
                                               OseLane& L, bool keep_rewards, bool agent_masks_done = false) {
  const aie_params& P = c.P;
  const int n = P.n, tid = c.tid, e = c.e;
  for (int q = tid; q < (P.met_bytes >> 2); q += OSE_NT) reinterpret_cast<uint32_t*>(c.met)[q] = 0u;  // new episode
  if (c.ev && tid == 0) c.ev[0] = 0;
  uint8_t* grec = arena + c.R.a_records + (int64_t)e * P.rec_bytes;  // skill / escrow live behind the LDS image
  OSE_MY_AGENTS(k, i, n) {
    R_F64(c, o_inv_coin)[i] = 0; R_F64(c, o_labor)[i] = 0;
    const double sk = P.has_labor ? c.R.c.labor_skills[i] : 0;
    const float so = (float)(sk / c.R.c.labor_pmsm);
    if (k) { L.esc1 = 0; L.skill1 = sk; L.skobs1 = so; } else { L.esc0 = 0; L.skill0 = sk; L.skobs0 = so; }
    reinterpret_cast<double*>(grec + P.o_esc_coin)[i] = 0;
    reinterpret_cast<double*>(grec + P.o_skill)[i] = sk;
    R_F64(c, o_production)[i] = 0;
    if (P.has_tax) {
      R_F64(c, o_tax_last_coin)[i] = 0; R_F64(c, o_tax_last_income)[i] = 0; R_F64(c, o_tax_last_marginal_rate)[i] = 0;
    }
  }
  if (tid == 0) {
    *R_I32(c, o_timestep) = 0;
    *R_I32(c, o_error_flags) = 0;
    *R_I32(c, o_first_step) = 1;
    if (P.has_tax) {
      *R_I32(c, o_tax_cycle_pos) = 1;
      *R_F64(c, o_tax_total_collected) = 0;
    }
  }
  if (P.has_tax && tid < P.NB) R_I32(c, o_tax_rate_idx)[tid] = 0;
  __syncthreads();
  if (c.saez) {  // _curr_rates_obs first (:1123, the previous episode's rates), then the running average (:1136-1137)
    if (tid < P.NB) R_F64(c, o_tax_saez_obs_rates)[tid] = tax_rate(c, tid);
    __syncthreads();
    if (tid < P.NB) R_F64(c, o_tax_saez_rates)[tid] = reinterpret_cast<const double*>(saez_block(c) + AIE_SAEZ_OFF_AVG)[tid];
    __syncthreads();
  }
  ose_metrics(c, s, L);
  for (int i = tid; i <= n; i += OSE_NT) R_F64(c, o_util)[i] = s.part[i];
  __syncthreads();
  ose_write_observations(c, s, arena, L, true, agent_masks_done);
  if (!keep_rewards) {
    for (int i = tid; i < n; i += OSE_NT) reinterpret_cast<float*>(arena + c.R.a_rew_a)[(int64_t)e * n + i] = 0.0f;
    if (tid == 0) {
      reinterpret_cast<float*>(arena + c.R.a_rew_p)[e] = 0.0f;
      (arena + c.R.a_done)[e] = 0;
    }
  }
  __syncthreads();
}
}  // namespace aie

// BaseEnvironment.step (base_env.py:929-1032) for the one-step-economy scenario.  SPEC >= 0: compile-time instance
// (constant parameter image, see step_body in aie_kernels.hip), SPEC < 0: the generic kernel.
template <int SPEC>
__device__ __forceinline__ void ose_step_body(const aie_params* __restrict__ params, uint8_t* __restrict__ arena,
                                              const int32_t* __restrict__ act_a, const int32_t* __restrict__ act_p,
                                              const NextActions& next, uint8_t* lds) {
  using namespace aie;
  const aie_params& R = *params;
  const aie_params& P = aie_spec_params<SPEC>(params);
  OseScratch s;
  const Ctx c = ose_make_ctx(P, R, lds, replica_of_block((int)blockIdx.x, R.E), (int)threadIdx.x, s, arena);
  const int n = P.n, tid = c.tid;
  MT m;
  OseLane L;
  OSE_STAMP(c, 0);
  const bool early_perm = P.c.n_components > 0 && P.c.components[0] == AIE_COMP_SIMPLE_LABOR;
  ose_load_record(c, arena, m, L, early_perm);
  // parse_actions (base_agent.py:407-438)
  bool bad_a = false, bad_p = false;  // out-of-range indices: NO-OP here, an exception in the reference (AIE_ERR_*)
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int i = tid + OSE_NT * k;
    int a = 0;
    if (i < n && act_a && P.n_sub_a) {
      const int v = act_a[((int64_t)c.e * n + i) * P.act_a_width];
      bad_a |= v < 0 || v > P.sub_a_dim[0];
      if (P.c.multi_action_mode_agents) a = (v >= 0 && v <= P.sub_a_dim[0]) ? v : 0;
      else a = (v >= 1 && v < 1 + P.sub_a_dim[0]) ? v : 0;
    }
    if (k) L.act1 = a; else L.act0 = a;
  }
  if (tid < AIE_MAX_BRACKETS) {
    int v = 0;
    if (act_p && tid < P.n_sub_p) {
      const int32_t* a = act_p + (int64_t)c.e * P.act_p_width;
      if (P.c.multi_action_mode_planner) {
        v = a[tid];
        bad_p = v < 0 || v > P.sub_p_dim;
        if (bad_p) v = 0;
      } else {
        const int x = a[0];
        bad_p = x < 0 || x >= 1 + P.n_sub_p * P.sub_p_dim;
        if (x >= 1 && x < 1 + P.n_sub_p * P.sub_p_dim && (x - 1) / P.sub_p_dim == tid) v = (x - 1) % P.sub_p_dim + 1;
      }
    }
    c.act_p[tid] = v;
  }
  {
    const int err = (__ballot(bad_a) ? AIE_ERR_AGENT_ACTION : 0) | (__ballot(bad_p) ? AIE_ERR_PLANNER_ACTION : 0);
    if (err && tid == 0) atomicOr(R_I32(c, o_error_flags), err);
  }
  __syncthreads();
  OSE_STAMP(c, 1);
  // the observations this launch leaves behind: the step's, or -- auto-reset, episode over -- the next episode's first
  const bool will_restart = R.auto_reset && uni(*R_I32(c, o_timestep)) + 1 >= R.c.episode_length;
  ose_store_agent_masks(c, s, arena, will_restart || (P.has_labor && uni(*R_I32(c, o_first_step)) != 0));
  if (!early_perm) m.pos = uni(*R_I32(c, o_mt_pos));
  if (tid == 0) *R_I32(c, o_timestep) += 1;
  if (c.ev && tid == 0) c.ev[0] = 0;
  for (int k = 0; k < P.c.n_components; ++k) {
    if (P.c.components[k] == AIE_COMP_SIMPLE_LABOR) labor_component_step(c, s, m, L, early_perm && k == 0);
    else if (P.c.components[k] == AIE_COMP_TAX) ose_tax_component_step(c, s, m, L);
    else if (P.c.components[k] == AIE_COMP_WEALTH_REDISTRIBUTION) ose_wealth_component_step(c, s, L);
    OSE_STAMP(c, 2 + (k < 2 ? k : 1));
  }
  if (tid == 0) *R_I32(c, o_mt_pos) = m.pos;
  ose_store_key(c, arena, m);
  __syncthreads();
  const bool done = uni(*R_I32(c, o_timestep)) >= R.c.episode_length;
  // auto-reset (aie_set_auto_reset): a replica that finishes its episode in this step restarts inside this launch;
  // its terminal observations would be overwritten by the reset's before anything can read them, so they are not
  // written (rewards and `done` are the terminal step's)
  const bool restart = done && R.auto_reset;
  if (!restart) ose_write_observations(c, s, arena, L, false, true);
  __syncthreads();
  OSE_STAMP(c, 8);
  // compute_reward one_step_economy.py:195-222
  ose_metrics(c, s, L);
  OSE_STAMP(c, 9);
  // this step's slot of aie_set_reward_log (or nullptr): every thread reads the replica's slot counter, thread 0 advances
  // it behind the barrier below
  float* __restrict__ rew_log = rew_log_claim(R, R_I32(c, o_rew_slot), R_I32(c, o_rew_epoch), R.E, n, false);
  {
    double* util = R_F64(c, o_util);
    for (int i = tid; i <= n; i += OSE_NT) {
      const double r = s.part[i] - util[i];
      util[i] = s.part[i];
      if (i < n) reinterpret_cast<float*>(arena + c.R.a_rew_a)[(int64_t)c.e * n + i] = (float)r;
      else reinterpret_cast<float*>(arena + c.R.a_rew_p)[c.e] = (float)r;
      if (rew_log) rew_log[(int64_t)c.e * (n + 2) + i] = (float)r;
    }
  }
  __syncthreads();
  if (tid == 0) {
    (arena + c.R.a_done)[c.e] = (uint8_t)done;
    if (rew_log) {
      rew_log[(int64_t)c.e * (n + 2) + n + 1] = done ? 1.0f : 0.0f;
      (void)rew_log_claim(R, R_I32(c, o_rew_slot), R_I32(c, o_rew_epoch), R.E, n, true);
    }
    if (done) *R_I32(c, o_completions) += 1;
  }
  __syncthreads();
  OSE_STAMP(c, 10);
  if (restart) ose_reset_body(c, s, arena, L, true, true);
  if (next.a || next.p) {  // aie_step_sample_next: the uniform random policy's draw for the next step
    const int per_env = P.n * P.act_a_width + P.act_p_width;
    const int st = *R_I32(c, o_sample_t);
    for (int j = tid; j < per_env; j += OSE_NT) sample_action_slot(P, next.seed, next.env_offset, (int64_t)st, c.e, j, next.a, next.p);
    if (tid == 0) *R_I32(c, o_sample_t) = st + 1;  // (one wavefront per replica: every lane has read it)
  }
  ose_store_record(c, arena);
  OSE_STAMP(c, 11);
}

#ifndef AIE_JIT
extern "C" __global__ void __launch_bounds__(OSE_NT)
__attribute__((amdgpu_waves_per_eu(4, 4)))  // (128 VGPRs, as the instances: the run-time generator switch cost three more)
aie_ose_step_kernel(const aie_params* __restrict__ params, uint8_t* __restrict__ arena,
                    const int32_t* __restrict__ act_a, const int32_t* __restrict__ act_p, NextActions next) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  ose_step_body<-1>(params, arena, act_a, act_p, next, lds);
}
template <int SPEC>
__global__ void __launch_bounds__(OSE_NT)
__attribute__((amdgpu_waves_per_eu(aie_spec_image<SPEC>::waves, aie_spec_image<SPEC>::waves)))
aie_ose_step_kernel_spec(const aie_params* __restrict__ params, uint8_t* __restrict__ arena,
                         const int32_t* __restrict__ act_a, const int32_t* __restrict__ act_p, NextActions next) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  ose_step_body<SPEC>(params, arena, act_a, act_p, next, lds);
}

extern "C" __global__ void __launch_bounds__(OSE_NT)
aie_ose_reset_kernel(const aie_params* __restrict__ params, uint8_t* __restrict__ arena,
                     const uint8_t* __restrict__ mask) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  using namespace aie;
  const aie_params& P = *params;
  const int e = replica_of_block((int)blockIdx.x, P.E);
  if (mask && !mask[e]) return;
  OseScratch s;
  const Ctx c = ose_make_ctx(P, P, lds, e, (int)threadIdx.x, s, arena);
  MT m;
  OseLane L;
  ose_load_record(c, arena, m, L);
  __syncthreads();
  ose_reset_body(c, s, arena, L, false);
  ose_store_record(c, arena);
}
#endif  // !AIE_JIT

#ifdef AIE_JIT_OSE
// run-time specialisation (aie_specialize) of the one-step-economy step kernel: this environment's parameter block
// as the constant image (SimpleLabor's skills stay run-time data, as in the build's instance)
extern "C" __global__ void __launch_bounds__(OSE_NT)
__attribute__((amdgpu_waves_per_eu(aie_spec_image<0>::waves, aie_spec_image<0>::waves)))
aie_jit_ose_step(const aie_params* __restrict__ params, uint8_t* __restrict__ arena,
                 const int32_t* __restrict__ act_a, const int32_t* __restrict__ act_p, NextActions next) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  ose_step_body<0>(params, arena, act_a, act_p, next, lds);
}
#endif  // AIE_JIT_OSE
