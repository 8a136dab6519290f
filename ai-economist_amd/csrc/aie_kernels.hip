// aie_kernels.hip -- hand-written CDNA4 (gfx950) kernels for the batched Foundation
// env.step() / env.reset() of the gather-trade-build family.
//
// Execution model: ONE WAVEFRONT (64 lanes) PER ENV REPLICA, one replica per workgroup.
//   1. the replica's state record (map cells, agents, order book, tax trackers, MT19937
//      state; layout in aie_layout.h) is streamed HBM -> LDS with 16-byte lane loads,
//   2. component dynamics run out of LDS: inherently sequential parts (random agent
//      order, order matching) on lane 0, everything else (price-history decay, resource
//      regeneration incl. the MT19937 twist, observation crops, masks, utilities)
//      across the 64 lanes,
//   3. observations are written straight to their dense [E, ...] tensors with
//      lane-contiguous (coalesced) stores; small vectors are staged in LDS and
//      streamed out 4 bytes/lane contiguous,
//   4. the record is streamed back LDS -> HBM.
// The path is integer / branchy and HBM-bound on the observation writes: no MFMA.
//
// Each __device__ function cites the reference function it implements (paths relative
// to the reference tree, F/ = ai_economist/foundation/).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "aie_layout.h"

#define AIE_NT 64  // threads per replica (one wavefront)

namespace aie {

struct Ctx {
  const aie_params& P;
  uint8_t* rec;      // LDS copy of the record
  int32_t* act;      // LDS [n][AIE_N_SUB_SLOTS] decoded agent actions
  int32_t* act_p;    // LDS [AIE_MAX_BRACKETS] decoded planner actions
  int32_t* perm;     // LDS [AIE_MAX_AGENTS] random agent order
  uint8_t* locmap;   // LDS [HW] 0 = empty, i+1 = agent i
  double* fscr;      // LDS f64 scratch
  float* stage;      // LDS staging of the small observation vectors
  int tid;
  int e;
};

#define R_F64(c, off) (reinterpret_cast<double*>((c).rec + (c).P.off))
#define R_I32(c, off) (reinterpret_cast<int32_t*>((c).rec + (c).P.off))
#define R_U8(c, off) (reinterpret_cast<uint8_t*>((c).rec + (c).P.off))
#define R_CELLS(c) (reinterpret_cast<uint32_t*>((c).rec + (c).P.o_cells))

// f64 scratch slots
__device__ __forceinline__ double* scr_net_ph(const Ctx& c) { return c.fscr; }                       // [2][P]
__device__ __forceinline__ double* scr_market(const Ctx& c) { return c.fscr + 2 * 128; }             // [2]
__device__ __forceinline__ double* scr_sorted_inc(const Ctx& c) { return c.fscr + 2 * 128 + 2; }     // [n]
__device__ __forceinline__ double* scr_cmr(const Ctx& c) { return c.fscr + 2 * 128 + 2 + 64; }       // [n]
__device__ __forceinline__ double* scr_coin(const Ctx& c) { return c.fscr + 2 * 128 + 2 + 128; }     // [n]
__device__ __forceinline__ double* scr_part(const Ctx& c) { return c.fscr + 2 * 128 + 2 + 192; }     // [n+1]
#define AIE_FSCR_DOUBLES (2 * 128 + 2 + 192 + 66)

__host__ __device__ inline size_t lds_bytes(const aie_params& P) {
  size_t b = (size_t)P.rec_bytes;
  b += (size_t)P.n * AIE_N_SUB_SLOTS * 4 + AIE_MAX_BRACKETS * 4 + AIE_MAX_AGENTS * 4;
  b = (b + 15) / 16 * 16;
  b += ((size_t)P.HW + 15) / 16 * 16;
  b += AIE_FSCR_DOUBLES * 8;
  b += ((size_t)P.n * (P.FA + P.MA + P.FPA) + P.FP + P.MP) * 4 + 64;
  return (b + 15) / 16 * 16;
}

__device__ __forceinline__ Ctx make_ctx(const aie_params& P, uint8_t* lds, int e, int tid) {
  uint8_t* q = lds + P.rec_bytes;
  int32_t* act = reinterpret_cast<int32_t*>(q);
  q += P.n * AIE_N_SUB_SLOTS * 4;
  int32_t* act_p = reinterpret_cast<int32_t*>(q);
  q += AIE_MAX_BRACKETS * 4;
  int32_t* perm = reinterpret_cast<int32_t*>(q);
  q += AIE_MAX_AGENTS * 4;
  q = lds + ((q - lds) + 15) / 16 * 16;
  uint8_t* locmap = q;
  q += (P.HW + 15) / 16 * 16;
  double* fscr = reinterpret_cast<double*>(q);
  q += AIE_FSCR_DOUBLES * 8;
  float* stage = reinterpret_cast<float*>(q);
  return Ctx{P, lds, act, act_p, perm, locmap, fscr, stage, tid, e};
}

// ------------------------------------------------------------------------------------
// record streaming HBM <-> LDS (16 B per lane, fully coalesced)
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void load_record(const Ctx& c, const uint8_t* __restrict__ arena) {
  const uint4* src = reinterpret_cast<const uint4*>(arena + c.P.a_records + (int64_t)c.e * c.P.rec_bytes);
  uint4* dst = reinterpret_cast<uint4*>(c.rec);
  const int nq = c.P.rec_bytes >> 4;
  for (int q = c.tid; q < nq; q += AIE_NT) dst[q] = src[q];
}
__device__ __forceinline__ void store_record(const Ctx& c, uint8_t* __restrict__ arena) {
  uint4* dst = reinterpret_cast<uint4*>(arena + c.P.a_records + (int64_t)c.e * c.P.rec_bytes);
  const uint4* src = reinterpret_cast<const uint4*>(c.rec);
  const int nq = c.P.rec_bytes >> 4;
  for (int q = c.tid; q < nq; q += AIE_NT) dst[q] = src[q];
}

// ------------------------------------------------------------------------------------
// NumPy legacy RandomState stream (MT19937), one per replica, state in the record.
// The reference draws from the process-global np.random (F/base/base_env.py:493,
// F/base/world.py:420, F/components/move.py:138, layout_from_file.py:361-366,400).
// ------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mt_mix(uint32_t a, uint32_t b, uint32_t m) {
  uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
  return m ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

// Whole-wave twist: word k of the next state needs words k, k+1 and k+397 (mod 624);
// 64 consecutive words are independent of each other because the recurrence distance
// is 227 > 64, so the state is regenerated in ten 64-lane slices, in place.
__device__ void mt_twist_wave(const Ctx& c) {
  uint32_t* mt = reinterpret_cast<uint32_t*>(c.rec + c.P.o_mt);
  for (int base = 0; base < AIE_MT_N; base += AIE_NT) {
    const int k = base + c.tid;
    uint32_t v = 0;
    if (k < AIE_MT_N) {
      const int k1 = (k + 1 == AIE_MT_N) ? 0 : k + 1;
      const int km = (k + 397 >= AIE_MT_N) ? k + 397 - AIE_MT_N : k + 397;
      v = mt_mix(mt[k], mt[k1], mt[km]);
    }
    __syncthreads();
    if (k < AIE_MT_N) mt[k] = v;
    __syncthreads();
  }
}
// Single-lane twist for the (rare) case where the sequential draws of lane 0 run off
// the end of the current state block.
__device__ void mt_twist_serial(uint32_t* mt) {
  int i;
  for (i = 0; i < AIE_MT_N - 397; i++) mt[i] = mt_mix(mt[i], mt[i + 1], mt[i + 397]);
  for (; i < AIE_MT_N - 1; i++) mt[i] = mt_mix(mt[i], mt[i + 1], mt[i + 397 - AIE_MT_N]);
  mt[AIE_MT_N - 1] = mt_mix(mt[AIE_MT_N - 1], mt[0], mt[396]);
}
// sequential draws (lane 0 only)
__device__ uint32_t rng_u32(const Ctx& c) {
  uint32_t* mt = reinterpret_cast<uint32_t*>(c.rec + c.P.o_mt);
  int32_t* pos = R_I32(c, o_mt_pos);
  int p = *pos;
  if (p >= AIE_MT_N) {
    mt_twist_serial(mt);
    p = 0;
  }
  uint32_t y = mt[p];
  *pos = p + 1;
  return mt_temper(y);
}
__device__ __forceinline__ double u53(uint32_t a, uint32_t b) {
  return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}
// numpy float64 add.reduce over n <= 128 contiguous values (pairwise summation with an
// 8-way unrolled head, numpy/_core/src/umath/loops_utils.h.src): decisions such as
// "mean agent reward > 0" (layout_from_file.py:554-557) depend on this exact order.
__device__ double np_sum_small(const double* a, int n) {
  if (n < 8) {
    double res = -0.0;
    for (int i = 0; i < n; ++i) res += a[i];
    return res;
  }
  double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
  int i = 8;
  for (; i < n - (n % 8); i += 8) {
    r0 += a[i + 0]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
    r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
  }
  double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
  for (; i < n; ++i) res += a[i];
  return res;
}

__device__ double rng_double(const Ctx& c) {
  uint32_t a = rng_u32(c);
  uint32_t b = rng_u32(c);
  return u53(a, b);
}
// random_interval(max): masked rejection on 32-bit words
__device__ uint32_t rng_interval(const Ctx& c, uint32_t max) {
  if (max == 0) return 0;
  uint32_t mask = max, v;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
  while ((v = (rng_u32(c) & mask)) > max) {}
  return v;
}
// np.random.permutation(n): World.get_random_order_agents, F/base/world.py:418-422
__device__ void rng_permutation(const Ctx& c, int n, int32_t* out) {
  for (int i = 0; i < n; ++i) out[i] = i;
  for (int i = n - 1; i >= 1; --i) {
    int j = (int)rng_interval(c, (uint32_t)i);
    int t = out[i]; out[i] = out[j]; out[j] = t;
  }
}
__device__ double rng_gauss(const Ctx& c) {  // legacy_gauss (polar Box-Muller, cached)
  int32_t* has = R_I32(c, o_mt_has_gauss);
  double* g = R_F64(c, o_mt_gauss);
  if (*has) {
    double t = *g;
    *has = 0;
    *g = 0.0;
    return t;
  }
  double f, x1, x2, r2;
  do {
    x1 = 2.0 * rng_double(c) - 1.0;
    x2 = 2.0 * rng_double(c) - 1.0;
    r2 = x1 * x1 + x2 * x2;
  } while (r2 >= 1.0 || r2 == 0.0);
  f = sqrt(-2.0 * log(r2) / r2);
  *g = f * x1;
  *has = 1;
  return f * x2;
}
__device__ double rng_pareto(const Ctx& c, double a) { return exp(-log(1.0 - rng_double(c)) / a) - 1.0; }
__device__ double rng_lognormal(const Ctx& c, double mean, double sigma) { return exp(mean + sigma * rng_gauss(c)); }

// ------------------------------------------------------------------------------------
// World helpers (F/base/world.py)
// ------------------------------------------------------------------------------------
// World.can_agent_occupy, world.py:424-440: in bounds, no Water, House only if owner
// (world.py:213-217, 256-258, 300-305), and unoccupied.
__device__ __forceinline__ bool can_agent_occupy(const Ctx& c, int r, int col, int agent) {
  if (r < 0 || r >= c.P.H || col < 0 || col >= c.P.W) return false;
  const int cell = r * c.P.W + col;
  const uint32_t w = R_CELLS(c)[cell];
  if (AIE_CELL_FLAGS(w) & AIE_CELL_WATER) return false;
  const int o = AIE_CELL_OWNER(w);
  if (!(o < 0 || o == agent)) return false;
  const int occ = c.locmap[cell];
  return occ == 0 || occ == agent + 1;
}

// Build.agent_can_build, F/components/build.py:70-83 (+ world.py:284-293)
__device__ __forceinline__ bool agent_can_build(const Ctx& c, int i) {
  const int n = c.P.n;
  const int32_t* inv = R_I32(c, o_inv_res);
  if (inv[n + i] < 1 || inv[i] < 1) return false;
  const uint32_t w = R_CELLS(c)[R_I32(c, o_loc_r)[i] * c.P.W + R_I32(c, o_loc_c)[i]];
  // no resource, no house (owner byte 0xff = none), no water / source block
  return (w & 0xffffu) == 0 && ((w >> 16) & 0xffu) == 0xffu && (w >> 24) == 0;
}

// ------------------------------------------------------------------------------------
// Build.component_step, F/components/build.py:112-161 (lane 0)
// ------------------------------------------------------------------------------------
__device__ void build_component_step(const Ctx& c) {
  const int n = c.P.n;
  int32_t* order = c.perm;
  rng_permutation(c, n, order);  // drawn even if nobody builds (build.py:121)
  for (int k = 0; k < n; ++k) {
    const int i = order[k];
    if (c.act[i * AIE_N_SUB_SLOTS + AIE_SUB_BUILD] != 1) continue;
    if (!agent_can_build(c, i)) continue;
    R_I32(c, o_inv_res)[n + i] -= 1;
    R_I32(c, o_inv_res)[i] -= 1;
    const int cell = R_I32(c, o_loc_r)[i] * c.P.W + R_I32(c, o_loc_c)[i];
    uint32_t w = R_CELLS(c)[cell];
    R_CELLS(c)[cell] = (w & 0xff00ffffu) | ((uint32_t)i << 16);  // world.py:474-479
    R_F64(c, o_inv_coin)[i] += R_F64(c, o_build_payment)[i];
    R_F64(c, o_labor)[i] += c.P.c.build_labor;
  }
}

// ------------------------------------------------------------------------------------
// Gather.component_step, F/components/move.py:93-153 (lane 0)
// ------------------------------------------------------------------------------------
__device__ void gather_component_step(const Ctx& c) {
  const int n = c.P.n, W = c.P.W;
  int32_t* order = c.perm;
  rng_permutation(c, n, order);
  int32_t *lr = R_I32(c, o_loc_r), *lc = R_I32(c, o_loc_c);
  for (int k = 0; k < n; ++k) {
    const int i = order[k];
    const int a = c.act[i * AIE_N_SUB_SLOTS + AIE_SUB_GATHER];
    const int r = lr[i], col = lc[i];
    int nr = r, nc = col;
    if (a != 0) {
      if (a == 1) nc = col - 1;       // Left
      else if (a == 2) nc = col + 1;  // Right
      else if (a == 3) nr = r - 1;    // Up
      else nr = r + 1;                // Down
      if (can_agent_occupy(c, nr, nc, i)) {  // world.py:454-460
        c.locmap[r * W + col] = 0;
        c.locmap[nr * W + nc] = (uint8_t)(i + 1);
        lr[i] = nr;
        lc[i] = nc;
        R_F64(c, o_labor)[i] += c.P.c.move_labor;
      } else {
        nr = r;
        nc = col;
      }
    }
    // collect on the landing tile, also on a NO-OP (move.py:112-113,136)
    const int cell = nr * W + nc;
    uint32_t w = R_CELLS(c)[cell];
    const int health[2] = {(int)AIE_CELL_STONE(w), (int)AIE_CELL_WOOD(w)};
#pragma unroll
    for (int rs = 0; rs < 2; ++rs) {
      if (health[rs] >= 1) {
        // rand() is consumed even when bonus_gather_prob == 0 (move.py:138)
        const int got = 1 + (rng_double(c) < R_F64(c, o_bonus_gather_prob)[i] ? 1 : 0);
        R_I32(c, o_inv_res)[rs * n + i] += got;
        w -= (1u << (8 * rs));  // consume_resource, world.py:481-483
        R_F64(c, o_labor)[i] += c.P.c.collect_labor;
      }
    }
    R_CELLS(c)[cell] = w;
  }
}

// ------------------------------------------------------------------------------------
// ContinuousDoubleAuction, F/components/continuous_double_auction.py
// ------------------------------------------------------------------------------------
// price_history *= 0.995 for every (commodity, agent, price) -- :451, all lanes
__device__ void cda_decay_price_history(const Ctx& c) {
  double* ph = R_F64(c, o_cda_price_history);
  const int tot = 2 * c.P.n * c.P.P;
  for (int q = c.tid; q < tot; q += AIE_NT) ph[q] *= 0.995;
}

// create_bid :168-198 / create_ask :200-229 (lane 0)
__device__ void cda_create_bid(const Ctx& c, int r, int i, int price) {
  const int n = c.P.n;
  int32_t* no = R_I32(c, o_cda_n_orders) + r * n;
  if (!(no[i] < c.P.c.cda_max_num_orders) || R_F64(c, o_inv_coin)[i] < (double)price) return;
  int32_t* nb = R_I32(c, o_cda_n_bids) + r;
  R_I32(c, o_cda_bids)[r * c.P.M + *nb] = AIE_ORD_PACK(i, price, 0);
  *nb += 1;
  R_U8(c, o_cda_bid_hist)[(r * n + i) * c.P.P + price] += 1;
  no[i] += 1;
  const double inv = R_F64(c, o_inv_coin)[i];
  const double tr = inv < (double)price ? inv : (double)price;  // base_agent.py:279-299
  R_F64(c, o_inv_coin)[i] -= tr;
  R_F64(c, o_esc_coin)[i] += tr;
  R_F64(c, o_labor)[i] += c.P.c.cda_order_labor;
}
__device__ void cda_create_ask(const Ctx& c, int r, int i, int price) {
  const int n = c.P.n;
  int32_t* no = R_I32(c, o_cda_n_orders) + r * n;
  int32_t* inv = R_I32(c, o_inv_res) + r * n;
  if (!(no[i] < c.P.c.cda_max_num_orders && inv[i] > 0)) return;
  int32_t* na = R_I32(c, o_cda_n_asks) + r;
  R_I32(c, o_cda_asks)[r * c.P.M + *na] = AIE_ORD_PACK(i, price, 0);
  *na += 1;
  R_U8(c, o_cda_ask_hist)[(r * n + i) * c.P.P + price] += 1;
  no[i] += 1;
  inv[i] -= 1;
  R_I32(c, o_esc_res)[r * n + i] += 1;
  R_F64(c, o_labor)[i] += c.P.c.cda_order_labor;
}

// Sort keys: bids by (price desc, lifetime desc, book position asc), asks by
// (price asc, lifetime desc, book position asc) == Python's stable sorted() at :249-256.
__device__ __forceinline__ bool bid_before(int32_t a, int32_t b) {
  const int pa = AIE_ORD_PRICE(a), pb = AIE_ORD_PRICE(b);
  if (pa != pb) return pa > pb;
  return AIE_ORD_LIFE(a) > AIE_ORD_LIFE(b);
}
__device__ __forceinline__ bool ask_before(int32_t a, int32_t b) {
  const int pa = AIE_ORD_PRICE(a), pb = AIE_ORD_PRICE(b);
  if (pa != pb) return pa < pb;
  return AIE_ORD_LIFE(a) > AIE_ORD_LIFE(b);
}
template <bool BIDS>
__device__ void book_insertion_sort(int32_t* v, int n) {
  for (int i = 1; i < n; ++i) {
    const int32_t x = v[i];
    int j = i - 1;
    while (j >= 0 && (BIDS ? bid_before(x, v[j]) : ask_before(x, v[j]))) {
      v[j + 1] = v[j];
      --j;
    }
    v[j + 1] = x;
  }
}

// match_orders :231-350 (lane 0): best remaining bid of a not-yet-flagged buyer against
// the best ask of another agent; trade at the older order's price; restart.
__device__ void cda_match_orders(const Ctx& c) {
  const int n = c.P.n, M = c.P.M, P = c.P.P;
  for (int r = 0; r < AIE_N_RES; ++r) {
    int32_t* bids = R_I32(c, o_cda_bids) + r * M;
    int32_t* asks = R_I32(c, o_cda_asks) + r * M;
    int nb = R_I32(c, o_cda_n_bids)[r], na = R_I32(c, o_cda_n_asks)[r];
    book_insertion_sort<true>(bids, nb);
    book_insertion_sort<false>(asks, na);
    uint64_t possible = (n >= 64) ? ~0ull : ((1ull << n) - 1ull);
    bool keep_checking = true;
    while (possible != 0 && keep_checking) {
      int ib = 0, ia = 0;
      for (;;) {
        if (ib >= nb) { keep_checking = false; break; }
        const int buyer = AIE_ORD_AGENT(bids[ib]);
        if (!((possible >> buyer) & 1ull)) { ib++; continue; }
        if (ia >= na) { possible &= ~(1ull << buyer); break; }
        if (AIE_ORD_AGENT(asks[ia]) == buyer) { ia++; continue; }
        if (AIE_ORD_PRICE(bids[ib]) < AIE_ORD_PRICE(asks[ia])) { possible &= ~(1ull << buyer); break; }
        // TRADE
        const int32_t bid = bids[ib], ask = asks[ia];
        for (int k = ib; k + 1 < nb; ++k) bids[k] = bids[k + 1];
        nb--;
        for (int k = ia; k + 1 < na; ++k) asks[k] = asks[k + 1];
        na--;
        const int seller = AIE_ORD_AGENT(ask);
        const int bprice = AIE_ORD_PRICE(bid), aprice = AIE_ORD_PRICE(ask);
        const int price = (AIE_ORD_LIFE(bid) <= AIE_ORD_LIFE(ask)) ? aprice : bprice;  // :297-304
        R_U8(c, o_cda_bid_hist)[(r * n + buyer) * P + bprice] -= 1;
        R_U8(c, o_cda_ask_hist)[(r * n + seller) * P + aprice] -= 1;
        R_I32(c, o_cda_n_orders)[r * n + seller] -= 1;
        R_I32(c, o_cda_n_orders)[r * n + buyer] -= 1;
        R_F64(c, o_cda_price_history)[(r * n + seller) * P + price] += 1.0;
        R_I32(c, o_esc_res)[r * n + seller] -= 1;
        R_I32(c, o_inv_res)[r * n + buyer] += 1;
        R_F64(c, o_esc_coin)[buyer] -= (double)bprice;
        R_F64(c, o_inv_coin)[seller] += (double)price;
        R_F64(c, o_inv_coin)[buyer] += (double)(bprice - price);
        break;
      }
    }
    R_I32(c, o_cda_n_bids)[r] = nb;
    R_I32(c, o_cda_n_asks)[r] = na;
  }
}

// remove_expired_orders :352-406 (lane 0)
__device__ void cda_remove_expired(const Ctx& c) {
  const int n = c.P.n, M = c.P.M, P = c.P.P, dur = c.P.c.cda_order_duration;
  for (int r = 0; r < AIE_N_RES; ++r) {
    int32_t* bids = R_I32(c, o_cda_bids) + r * M;
    const int nb = R_I32(c, o_cda_n_bids)[r];
    int k = 0;
    for (int q = 0; q < nb; ++q) {
      const int32_t o = bids[q];
      const int life = AIE_ORD_LIFE(o) + 1, ag = AIE_ORD_AGENT(o), pr = AIE_ORD_PRICE(o);
      if (life <= dur) bids[k++] = AIE_ORD_PACK(ag, pr, life);
      else {
        const double esc = R_F64(c, o_esc_coin)[ag];
        const double tr = esc < (double)pr ? esc : (double)pr;
        R_F64(c, o_esc_coin)[ag] -= tr;
        R_F64(c, o_inv_coin)[ag] += tr;
        R_U8(c, o_cda_bid_hist)[(r * n + ag) * P + pr] -= 1;
        R_I32(c, o_cda_n_orders)[r * n + ag] -= 1;
      }
    }
    R_I32(c, o_cda_n_bids)[r] = k;
    int32_t* asks = R_I32(c, o_cda_asks) + r * M;
    const int na = R_I32(c, o_cda_n_asks)[r];
    k = 0;
    for (int q = 0; q < na; ++q) {
      const int32_t o = asks[q];
      const int life = AIE_ORD_LIFE(o) + 1, ag = AIE_ORD_AGENT(o), pr = AIE_ORD_PRICE(o);
      if (life <= dur) asks[k++] = AIE_ORD_PACK(ag, pr, life);
      else {
        R_I32(c, o_esc_res)[r * n + ag] -= 1;
        R_I32(c, o_inv_res)[r * n + ag] += 1;
        R_U8(c, o_cda_ask_hist)[(r * n + ag) * P + pr] -= 1;
        R_I32(c, o_cda_n_orders)[r * n + ag] -= 1;
      }
    }
    R_I32(c, o_cda_n_asks)[r] = k;
  }
}

// ContinuousDoubleAuction.component_step :440-489, sequential part (lane 0); the
// price-history decay of :451 has already been applied by all lanes.
__device__ void cda_component_step(const Ctx& c) {
  const int n = c.P.n;
  for (int r = 0; r < AIE_N_RES; ++r) {
    for (int i = 0; i < n; ++i) {
      int a = c.act[i * AIE_N_SUB_SLOTS + (r ? AIE_SUB_BUY1 : AIE_SUB_BUY0)];
      if (a > 0) cda_create_bid(c, r, i, a - 1);
      a = c.act[i * AIE_N_SUB_SLOTS + (r ? AIE_SUB_SELL1 : AIE_SUB_SELL0)];
      if (a > 0) cda_create_ask(c, r, i, a - 1);
    }
  }
  cda_match_orders(c);
  cda_remove_expired(c);
}

// ------------------------------------------------------------------------------------
// PeriodicBracketTax, F/components/redistribution.py
// ------------------------------------------------------------------------------------
__device__ __forceinline__ double tax_rate(const Ctx& c, int b) {  // curr_marginal_rates :396-417
  if (c.P.c.tax_model == AIE_TAX_MODEL_WRAPPER) return c.P.c.tax_disc_rates[R_I32(c, o_tax_rate_idx)[b]];
  return c.P.c.tax_fixed_rates[b];
}
__device__ double tax_marginal_rate(const Ctx& c, double income) {  // marginal_rate :837-844
  if (income < 0) return 0.0;
  const int NB = c.P.NB;
  for (int b = 0; b < NB; ++b) {
    const double lo = c.P.c.tax_bracket_cutoffs[b];
    const bool under = (b + 1 < NB) ? (income < c.P.c.tax_bracket_cutoffs[b + 1]) : (income < INFINITY);
    if (income >= lo && under) return tax_rate(c, b);
  }
  return tax_rate(c, 0);
}
__device__ __forceinline__ double tax_bin(const Ctx& c, double income, int b) {
  const int NB = c.P.NB;
  const double cut = c.P.c.tax_bracket_cutoffs[b];
  const double size = (b + 1 < NB) ? c.P.c.tax_bracket_cutoffs[b + 1] - cut : INFINITY;
  double past = income - cut;
  if (past < 0) past = 0;
  return tax_rate(c, b) * (size < past ? size : past);
}
__device__ double tax_due(const Ctx& c, double income) {  // taxes_due :846-851
  const int NB = c.P.NB;
  // np.sum of NB values (pairwise summation, loops_utils.h.src)
  if (NB < 8) {
    double res = -0.0;
    for (int b = 0; b < NB; ++b) res += tax_bin(c, income, b);
    return res;
  }
  double r0 = tax_bin(c, income, 0), r1 = tax_bin(c, income, 1), r2 = tax_bin(c, income, 2),
         r3 = tax_bin(c, income, 3), r4 = tax_bin(c, income, 4), r5 = tax_bin(c, income, 5),
         r6 = tax_bin(c, income, 6), r7 = tax_bin(c, income, 7);
  int b = 8;
  for (; b < NB - (NB % 8); b += 8) {
    r0 += tax_bin(c, income, b + 0); r1 += tax_bin(c, income, b + 1);
    r2 += tax_bin(c, income, b + 2); r3 += tax_bin(c, income, b + 3);
    r4 += tax_bin(c, income, b + 4); r5 += tax_bin(c, income, b + 5);
    r6 += tax_bin(c, income, b + 6); r7 += tax_bin(c, income, b + 7);
  }
  double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
  for (; b < NB; ++b) res += tax_bin(c, income, b);
  return res;
}
__device__ void tax_enact(const Ctx& c) {  // enact_taxes :853-915 (lane 0)
  const int n = c.P.n;
  double net = 0;
  for (int i = 0; i < n; ++i) {
    const double income = (R_F64(c, o_inv_coin)[i] + R_F64(c, o_esc_coin)[i]) - R_F64(c, o_tax_last_coin)[i];
    const double due = tax_due(c, income);
    const double inv = R_F64(c, o_inv_coin)[i];
    const double eff = inv < due ? inv : due;  // escrow is not taxed
    R_F64(c, o_tax_last_marginal_rate)[i] = tax_marginal_rate(c, income);
    R_F64(c, o_inv_coin)[i] = inv - eff;
    net += eff;
    R_F64(c, o_tax_last_income)[i] = income;
  }
  *R_F64(c, o_tax_total_collected) += net;
  const double lump = net / (double)n;
  for (int i = 0; i < n; ++i) {
    const double v = R_F64(c, o_inv_coin)[i] + lump;
    R_F64(c, o_inv_coin)[i] = v;
    R_F64(c, o_tax_last_coin)[i] = v + R_F64(c, o_esc_coin)[i];
  }
}
// component_step :945-972 + set_new_period_rates_model :419-434 (lane 0)
__device__ void tax_component_step(const Ctx& c) {
  int32_t* pos = R_I32(c, o_tax_cycle_pos);
  if (*pos == 1 && c.P.c.tax_model == AIE_TAX_MODEL_WRAPPER && !c.P.c.tax_disable) {
    for (int b = 0; b < c.P.NB; ++b) {
      const int a = c.act_p[b];
      if (a > 0 && a <= c.P.c.tax_n_disc_rates) R_I32(c, o_tax_rate_idx)[b] = a - 1;
    }
  }
  if (*pos >= c.P.c.tax_period) {
    tax_enact(c);
    *pos = 0;
  }
  *pos += 1;
}

// ------------------------------------------------------------------------------------
// LayoutFromFile.scenario_step, layout_from_file.py:372-410 (all lanes)
// regen_halfwidth == 0: p = regen_weight * max(map, src); only source blocks spawn.
// np.random.rand(H, W) is consumed for Wood, then for Stone: 2*H*W doubles = 4*H*W
// MT19937 words per step, produced 64 doubles at a time by the whole wave.
// ------------------------------------------------------------------------------------
__device__ void scenario_step_regen(const Ctx& c) {
  uint32_t* mt = reinterpret_cast<uint32_t*>(c.rec + c.P.o_mt);
  uint32_t* cells = R_CELLS(c);
  int pos = *R_I32(c, o_mt_pos);  // wave-uniform
  const int HW = c.P.HW;
  for (int pass = 0; pass < 2; ++pass) {
    const int rs = pass == 0 ? 1 : 0;  // ["Wood", "Stone"]
    const uint32_t srcbit = (rs ? AIE_CELL_WOOD_SRC : AIE_CELL_STONE_SRC) << 24;
    const double w = c.P.c.regen_weight[rs];
    const uint32_t mh = (uint32_t)c.P.c.max_health[rs];
    for (int base = 0; base < HW; base += AIE_NT) {
      const int cnt = (HW - base) < AIE_NT ? (HW - base) : AIE_NT;
      const int need = 2 * cnt;
      const bool active = c.tid < cnt;
      const int qa = 2 * c.tid, qb = qa + 1;
      uint32_t a = 0, b = 0;
      if (pos + need <= AIE_MT_N) {
        if (active) { a = mt[pos + qa]; b = mt[pos + qb]; }
        pos += need;
      } else {
        const int rem = AIE_MT_N - pos;
        if (active && qa < rem) a = mt[pos + qa];
        if (active && qb < rem) b = mt[pos + qb];
        __syncthreads();
        mt_twist_wave(c);
        if (active && qa >= rem) a = mt[qa - rem];
        if (active && qb >= rem) b = mt[qb - rem];
        pos = need - rem;
      }
      if (active) {
        const double u = u53(mt_temper(a), mt_temper(b));
        const int cell = base + c.tid;
        uint32_t cw = cells[cell];
        const uint32_t m = (cw >> (8 * rs)) & 0xffu;
        const uint32_t src = (cw & srcbit) ? 1u : 0u;
        const uint32_t health = m > src ? m : src;
        const uint32_t respawn = (src && (u < w * (double)health)) ? 1u : 0u;
        uint32_t v = m + respawn;
        v = v < mh ? v : mh;
        cells[cell] = (cw & ~(0xffu << (8 * rs))) | (v << (8 * rs));
      }
    }
  }
  __syncthreads();
  if (c.tid == 0) *R_I32(c, o_mt_pos) = pos;
}

// ------------------------------------------------------------------------------------
// Utilities / rewards
// ------------------------------------------------------------------------------------
__device__ double energy_weight(const Ctx& c) {  // layout_from_file.py:249-267
  if (c.P.c.energy_warmup_constant <= 0.0) return 1.0;
  const int v = c.P.c.energy_warmup_method == AIE_WARMUP_DECAY ? *R_I32(c, o_completions) : *R_I32(c, o_auto_warmup);
  return 1.0 - exp(-(double)v / c.P.c.energy_warmup_constant);
}

// get_current_optimization_metrics layout_from_file.py:269-318 with
// rewards.isoelastic_coin_minus_labor (F/scenarios/utils/rewards.py:12-48),
// coin_eq_times_productivity (:84-101), inv_income_weighted_* (:104-133),
// social_metrics.get_gini (social_metrics.py:10-46).
// Lane i < n computes agent i's utility; lane 0 finishes the planner's.
// Results are left in scr_part()[0..n]; must be followed by __syncthreads().
__device__ void current_metrics(const Ctx& c) {
  const int n = c.P.n, i = c.tid;
  double* coin = scr_coin(c);
  double* out = scr_part(c);
  double* tmp = scr_cmr(c);  // free at this point
  const double lcf = energy_weight(c) * c.P.c.energy_cost;
  const double eta = c.P.c.isoelastic_eta;
  if (i < n) {
    const double ci = R_F64(c, o_inv_coin)[i] + R_F64(c, o_esc_coin)[i];
    coin[i] = ci;
    double util_c;
    if (eta == 1.0) util_c = log(ci > 1 ? ci : 1);
    else util_c = (pow(ci, 1 - eta) - 1) / (1 - eta);
    out[i] = util_c - R_F64(c, o_labor)[i] * lcf;
  }
  __syncthreads();
  const int prt = c.P.c.planner_reward_type;
  if (prt == AIE_PLANNER_REW_COIN_EQ_TIMES_PROD) {
    if (n < 30) {
      if (i < n) {
        double s = 0, ci = coin[i];
        for (int j = 0; j < n; ++j) s += fabs(ci - coin[j]);
        tmp[i] = s;
      }
    }
    __syncthreads();
    if (i == 0) {
      const double tot = np_sum_small(coin, n);
      double gini;
      if (n < 30) {
        double diff = 0;
        for (int j = 0; j < n; ++j) diff += tmp[j];
        const double unscaled = diff / (2 * n * tot + 1e-10);
        gini = unscaled / ((double)(n - 1) / (double)n);
      } else {
        // sorted-cumsum branch (social_metrics.py:43-46)
        double* s = scr_sorted_inc(c);  // free here
        for (int j = 0; j < n; ++j) s[j] = coin[j];
        for (int a = 1; a < n; ++a) {
          double x = s[a]; int b = a - 1;
          while (b >= 0 && s[b] > x) { s[b + 1] = s[b]; --b; }
          s[b + 1] = x;
        }
        const double tots = np_sum_small(s, n);
        double run = 0, acc = 0;
        for (int j = 0; j < n; ++j) { run += s[j]; acc += run / (tots + 1e-10); }
        gini = 1 - (2.0 / (n + 1)) * acc;
      }
      const double ew = 1 - c.P.c.mixing_weight_gini_vs_coin;
      out[n] = (ew * (1 - gini) + (1 - ew)) * (tot / n);
    }
  } else {
    if (i == 0) {
      double sw = 0;
      for (int j = 0; j < n; ++j) sw += 1 / (coin[j] > 1 ? coin[j] : 1);
      double acc = 0;
      for (int j = 0; j < n; ++j) {
        const double w = (1 / (coin[j] > 1 ? coin[j] : 1)) / sw;
        acc += (prt == AIE_PLANNER_REW_INV_INCOME_COIN ? coin[j] : out[j]) * w;
      }
      out[n] = acc;
    }
  }
}

// compute_reward layout_from_file.py:519-559
__device__ void compute_rewards(const Ctx& c, uint8_t* __restrict__ arena) {
  const int n = c.P.n, i = c.tid;
  current_metrics(c);
  __syncthreads();
  double* cur = scr_part(c);
  double* util = R_F64(c, o_util);
  double* rew = scr_coin(c);  // reuse
  if (i <= n) {
    const double r = cur[i] - util[i];
    util[i] = cur[i];
    rew[i] = r;
    if (i < n) reinterpret_cast<float*>(arena + c.P.a_rew_a)[(int64_t)c.e * n + i] = (float)r;
    else reinterpret_cast<float*>(arena + c.P.a_rew_p)[c.e] = (float)r;
  }
  __syncthreads();
  if (i == 0) {
    if (np_sum_small(rew, n) / n > 0) *R_I32(c, o_auto_warmup) += 1;
  }
}

// ------------------------------------------------------------------------------------
// Observations
// ------------------------------------------------------------------------------------
__device__ __forceinline__ float chan_value(uint32_t w, int k, bool has_water) {
  // channel order of Maps.state (world.py:59-93): Stone, Wood, House, [Water],
  // StoneSourceBlock, WoodSourceBlock
  switch (k) {
    case 0: return (float)AIE_CELL_STONE(w);
    case 1: return (float)AIE_CELL_WOOD(w);
    case 2: return (((w >> 16) & 0xffu) != 0xffu) ? 1.0f : 0.0f;
    default: break;
  }
  const uint32_t fl = w >> 24;
  if (has_water) {
    if (k == 3) return (fl & AIE_CELL_WATER) ? 1.0f : 0.0f;
    k -= 1;
  }
  return (fl & (k == 3 ? AIE_CELL_STONE_SRC : AIE_CELL_WOOD_SRC)) ? 1.0f : 0.0f;
}

// LayoutFromFile.generate_observations layout_from_file.py:412-517: the egocentric
// (2w+1)^2 crop for every agent and the full map for the planner.  One lane per
// (agent, window cell): one LDS word read, CM+1 coalesced f32 stores, 2 i16 stores.
__device__ void write_spatial_observations(const Ctx& c, uint8_t* __restrict__ arena) {
  const int n = c.P.n, H = c.P.H, W = c.P.W, HW = c.P.HW, WV = c.P.WV, CM = c.P.CM;
  const int w = c.P.c.obs_range;
  const int WV2 = WV * WV;
  const bool has_water = c.P.c.has_water != 0;
  const uint32_t* cells = R_CELLS(c);
  const int32_t *lr = R_I32(c, o_loc_r), *lc = R_I32(c, o_loc_c);
  float* amap = reinterpret_cast<float*>(arena + c.P.a_obs_a_map) + (int64_t)c.e * n * (CM + 1) * WV2;
  int16_t* aidx = reinterpret_cast<int16_t*>(arena + c.P.a_obs_a_idx) + (int64_t)c.e * n * 2 * WV2;
  const int tot = n * WV2;
  for (int q = c.tid; q < tot; q += AIE_NT) {
    const int i = q / WV2;
    const int d = q - i * WV2;
    const int dr = d / WV, dc = d - dr * WV;
    const int r = lr[i] - w + dr, col = lc[i] - w + dc;
    const bool in = (r >= 0) & (r < H) & (col >= 0) & (col < W);
    const int cell = in ? r * W + col : 0;
    // out of the world: all channels 0, owner none, in-bounds channel 0 (:480-485)
    const uint32_t cw = in ? cells[cell] : 0x00ff0000u;
    float* o = amap + (int64_t)i * (CM + 1) * WV2 + d;
    for (int k = 0; k < CM; ++k) o[k * WV2] = chan_value(cw, k, has_water);
    o[CM * WV2] = in ? 1.0f : 0.0f;
    const int own = AIE_CELL_OWNER(cw);
    int v0 = own >= 0 ? own + 2 : 0;
    int v1 = in ? (int)c.locmap[cell] : 0;
    v1 = v1 ? v1 + 1 : 0;  // agent k -> k + 2
    if (v0 == i + 2) v0 = 1;  // :503
    if (v1 == i + 2) v1 = 1;
    int16_t* oi = aidx + (int64_t)i * 2 * WV2 + d;
    oi[0] = (int16_t)v0;
    oi[WV2] = (int16_t)v1;
  }
  if (c.P.c.planner_gets_spatial_info) {
    float* pmap = reinterpret_cast<float*>(arena + c.P.a_obs_p_map) + (int64_t)c.e * CM * HW;
    int16_t* pidx = reinterpret_cast<int16_t*>(arena + c.P.a_obs_p_idx) + (int64_t)c.e * 2 * HW;
    for (int cell = c.tid; cell < HW; cell += AIE_NT) {
      const uint32_t cw = cells[cell];
      for (int k = 0; k < CM; ++k) pmap[k * HW + cell] = chan_value(cw, k, has_water);
      const int own = AIE_CELL_OWNER(cw);
      const int occ = c.locmap[cell];
      pidx[cell] = (int16_t)(own >= 0 ? own + 2 : 0);
      pidx[HW + cell] = (int16_t)(occ ? occ + 1 : 0);
    }
  }
}

// Component + scalar observations, packed in SORTED key order (base_env.py:561-612):
// Build (build.py:163-178), CDA (continuous_double_auction.py:491-542), Gather
// (move.py:155-165), PeriodicBracketTax (redistribution.py:974-1023), time, world-*.
// Built in LDS (c.stage) then streamed out.
__device__ void write_flat_observations_and_masks(const Ctx& c, uint8_t* __restrict__ arena) {
  const aie_params& P = c.P;
  const int n = P.n, tid = c.tid, Pp = P.P, NB = P.NB;
  const double isc = P.c.allow_observation_scaling ? 0.01 : 1.0;
  float* s_aflat = c.stage;
  float* s_amask = s_aflat + n * P.FA;
  float* s_pag = s_amask + n * P.MA;
  float* s_pflat = s_pag + n * P.FPA;
  float* s_pmask = s_pflat + P.FP;

  // ---- shared CDA quantities: lanes over (commodity, price) ----
  if (P.has_cda) {
    double* net_ph = scr_net_ph(c);
    for (int q = tid; q < 2 * Pp; q += AIE_NT) {
      const int r = q / Pp, k = q - r * Pp;
      double s = 0;
      int fa = 0, fb = 0;
      for (int i = 0; i < n; ++i) {
        const double v = R_F64(c, o_cda_price_history)[(r * n + i) * Pp + k];
        s = (i == 0) ? v : s + v;
        fa += R_U8(c, o_cda_ask_hist)[(r * n + i) * Pp + k];
        fb += R_U8(c, o_cda_bid_hist)[(r * n + i) * Pp + k];
      }
      net_ph[r * 128 + k] = s;
      const float ph = (float)(s * isc);
      // planner: full_asks, full_bids, price_history
      float* g = s_pflat + P.fp_cda;
      g[0 * Pp + q] = (float)fa;
      g[2 * Pp + q] = (float)fb;
      g[4 * Pp + 2 + q] = ph;
      for (int i = 0; i < n; ++i) {
        float* f = s_aflat + i * P.FA + P.fa_cda;
        const int mya = R_U8(c, o_cda_ask_hist)[(r * n + i) * Pp + k];
        const int myb = R_U8(c, o_cda_bid_hist)[(r * n + i) * Pp + k];
        f[0 * Pp + q] = (float)(fa - mya);  // available_asks
        f[2 * Pp + q] = (float)(fb - myb);  // available_bids
        f[4 * Pp + 2 + q] = (float)mya;     // my_asks
        f[6 * Pp + 2 + q] = (float)myb;     // my_bids
        f[8 * Pp + 2 + q] = ph;             // price_history
      }
    }
    __syncthreads();
    if (tid < 2) {
      const int r = tid;
      const double* a = net_ph + r * 128;
      double dot = 0;
      for (int k = 0; k < Pp; ++k) dot += (double)k * a[k];
      // np.sum (pairwise) of P values
      double tot;
      if (Pp < 8) {
        tot = -0.0;
        for (int k = 0; k < Pp; ++k) tot += a[k];
      } else {
        double rr[8];
        int k;
        for (k = 0; k < 8; ++k) rr[k] = a[k];
        for (k = 8; k < Pp - (Pp % 8); k += 8)
          for (int j = 0; j < 8; ++j) rr[j] += a[k + j];
        tot = ((rr[0] + rr[1]) + (rr[2] + rr[3])) + ((rr[4] + rr[5]) + (rr[6] + rr[7]));
        for (; k < Pp; ++k) tot += a[k];
      }
      const float mr = (float)(dot / (tot > 0.001 ? tot : 0.001));
      s_pflat[P.fp_cda + 4 * Pp + r] = mr;
      for (int i = 0; i < n; ++i) s_aflat[i * P.FA + P.fa_cda + 4 * Pp + r] = mr;
    }
  }

  // ---- shared tax quantities ----
  double is_tax_day = 0, is_first_day = 0, tax_phase = 0;
  if (P.has_tax) {
    const int pos = *R_I32(c, o_tax_cycle_pos);
    is_tax_day = pos >= P.c.tax_period ? 1.0 : 0.0;
    is_first_day = pos == 1 ? 1.0 : 0.0;
    tax_phase = (double)pos / (double)P.c.tax_period;
    if (tid < n) {
      // last_incomes sorted ascending (redistribution.py:908-911): rank by counting
      const double per = (double)P.c.tax_period;
      const double x = R_F64(c, o_tax_last_income)[tid] / per;
      int rank = 0;
      for (int j = 0; j < n; ++j) {
        const double y = R_F64(c, o_tax_last_income)[j] / per;
        rank += (y < x || (y == x && j < tid)) ? 1 : 0;
      }
      scr_sorted_inc(c)[rank] = x;
      scr_cmr(c)[tid] = tax_marginal_rate(
          c, (R_F64(c, o_inv_coin)[tid] + R_F64(c, o_esc_coin)[tid]) - R_F64(c, o_tax_last_coin)[tid]);
    }
    __syncthreads();
    // lanes over (agent or planner, element of the tax fragment)
    const int fragA = NB + n + 4, fragP = NB + n + 3;
    for (int q = tid; q < n * fragA + fragP; q += AIE_NT) {
      const bool planner = q >= n * fragA;
      const int i = planner ? 0 : q / fragA;
      const int j = planner ? q - n * fragA : q - i * fragA;
      float v;
      if (j < NB) v = (float)tax_rate(c, j);
      else if (j == NB) v = (float)is_first_day;
      else if (j == NB + 1) v = (float)is_tax_day;
      else if (j < NB + 2 + n) v = (float)scr_sorted_inc(c)[j - NB - 2];
      else if (!planner && j == NB + 2 + n) v = (float)scr_cmr(c)[i];
      else v = (float)tax_phase;
      if (planner) s_pflat[P.fp_tax + j] = v;
      else s_aflat[i * P.FA + P.fa_tax + j] = v;
    }
  }

  // ---- per-agent scalars ----
  const int t = *R_I32(c, o_timestep);
  const float tval = (float)((double)t / (P.c.allow_observation_scaling ? (double)P.c.episode_length : 1.0));
  if (tid < n) {
    const int i = tid;
    float* f = s_aflat + i * P.FA;
    if (P.has_build) {
      f[P.fa_build + 0] = (float)(R_F64(c, o_build_payment)[i] / (double)P.c.build_payment);
      f[P.fa_build + 1] = (float)R_F64(c, o_build_skill)[i];
    }
    if (P.has_gather) f[P.fa_gather] = (float)R_F64(c, o_bonus_gather_prob)[i];
    f[P.fa_time] = tval;
    const float w0 = (float)(R_F64(c, o_inv_coin)[i] * isc);
    const float w1 = (float)((double)R_I32(c, o_inv_res)[i] * isc);
    const float w2 = (float)((double)R_I32(c, o_inv_res)[n + i] * isc);
    const float w3 = (float)((double)R_I32(c, o_loc_c)[i] / (double)P.W);
    const float w4 = (float)((double)R_I32(c, o_loc_r)[i] / (double)P.H);
    f[P.fa_world + 0] = w0; f[P.fa_world + 1] = w1; f[P.fa_world + 2] = w2;
    f[P.fa_world + 3] = w3; f[P.fa_world + 4] = w4;
    float* q = s_pag + i * P.FPA;
    if (P.has_tax) {
      q[P.fpa_tax + 0] = (float)scr_cmr(c)[i];
      q[P.fpa_tax + 1] = (float)(R_F64(c, o_tax_last_income)[i] / (double)P.c.tax_period);
      q[P.fpa_tax + 2] = (float)R_F64(c, o_tax_last_marginal_rate)[i];
    }
    q[P.fpa_world + 0] = w0; q[P.fpa_world + 1] = w1; q[P.fpa_world + 2] = w2;
    if (P.c.planner_gets_spatial_info) { q[P.fpa_world + 3] = w3; q[P.fpa_world + 4] = w4; }
    reinterpret_cast<float*>(arena + P.a_obs_a_time)[(int64_t)c.e * n + i] = tval;
  }
  if (tid == 0) {
    s_pflat[P.fp_time] = tval;
    s_pflat[P.fp_world + 0] = 0.0f;  // the planner's inventory never changes
    s_pflat[P.fp_world + 1] = 0.0f;
    s_pflat[P.fp_world + 2] = 0.0f;
    reinterpret_cast<float*>(arena + P.a_obs_p_time)[c.e] = tval;
  }

  // ---- masks: _generate_masks base_env.py:706-756 + flatten_masks base_agent.py:440-460
  // Gather move.py:167-188, Build build.py:180-193, CDA :544-580, Tax :1025-1104 ----
  {
    const bool multi = P.c.multi_action_mode_agents != 0;
    const int32_t *lr = R_I32(c, o_loc_r), *lc = R_I32(c, o_loc_c);
    for (int q = tid; q < n * P.MA; q += AIE_NT) {
      const int i = q / P.MA;
      int m = q - i * P.MA;
      float v = 1.0f;
      if (!multi) m -= 1;  // leading NO-OP entry
      if (m >= 0 && P.n_sub_a > 0) {
        int s = 0;
        for (; s < P.n_sub_a; ++s) {
          const int len = P.sub_a_dim[s] + (multi ? 1 : 0);
          if (m < len) break;
          m -= len;
        }
        if (multi) m -= 1;  // per-subspace NO-OP entry
        if (m >= 0) {
          const int slot = P.sub_a_slot[s];
          if (slot == AIE_SUB_BUILD) v = agent_can_build(c, i) ? 1.0f : 0.0f;
          else if (slot == AIE_SUB_GATHER) {
            const int ro = (m == 2) ? -1 : (m == 3) ? 1 : 0;
            const int co = (m == 0) ? -1 : (m == 1) ? 1 : 0;
            v = can_agent_occupy(c, lr[i] + ro, lc[i] + co, i) ? 1.0f : 0.0f;
          } else {
            const int r = (slot == AIE_SUB_BUY1 || slot == AIE_SUB_SELL1) ? 1 : 0;
            const bool is_buy = (slot == AIE_SUB_BUY0 || slot == AIE_SUB_BUY1);
            const bool quota = R_I32(c, o_cda_n_orders)[r * n + i] < P.c.cda_max_num_orders;
            if (is_buy) v = (quota && (double)m <= R_F64(c, o_inv_coin)[i]) ? 1.0f : 0.0f;
            else v = (quota && R_I32(c, o_inv_res)[r * n + i] > 0) ? 1.0f : 0.0f;
          }
        }
      }
      s_amask[q] = v;
    }
    const bool pmulti = P.c.multi_action_mode_planner != 0;
    const float open = (P.n_sub_p && *R_I32(c, o_tax_cycle_pos) == 1) ? 1.0f : 0.0f;
    for (int q = tid; q < P.MP; q += AIE_NT) {
      float v;
      if (P.n_sub_p == 0) v = 1.0f;
      else if (pmulti) v = (q % (1 + P.sub_p_dim) == 0) ? 1.0f : open;
      else v = (q == 0) ? 1.0f : open;
      s_pmask[q] = v;
    }
  }
  __syncthreads();

  // ---- stream the staged vectors out (lane-contiguous 4-byte stores) ----
  {
    float* g = reinterpret_cast<float*>(arena + P.a_obs_a_flat) + (int64_t)c.e * n * P.FA;
    for (int q = tid; q < n * P.FA; q += AIE_NT) g[q] = s_aflat[q];
    g = reinterpret_cast<float*>(arena + P.a_obs_a_mask) + (int64_t)c.e * n * P.MA;
    for (int q = tid; q < n * P.MA; q += AIE_NT) g[q] = s_amask[q];
    g = reinterpret_cast<float*>(arena + P.a_obs_p_agents) + (int64_t)c.e * n * P.FPA;
    for (int q = tid; q < n * P.FPA; q += AIE_NT) g[q] = s_pag[q];
    g = reinterpret_cast<float*>(arena + P.a_obs_p_flat) + (int64_t)c.e * P.FP;
    for (int q = tid; q < P.FP; q += AIE_NT) g[q] = s_pflat[q];
    g = reinterpret_cast<float*>(arena + P.a_obs_p_mask) + (int64_t)c.e * P.MP;
    for (int q = tid; q < P.MP; q += AIE_NT) g[q] = s_pmask[q];
  }
}

__device__ void rebuild_locmap(const Ctx& c) {
  uint32_t* lm = reinterpret_cast<uint32_t*>(c.locmap);
  const int nw = (c.P.HW + 3) >> 2;
  for (int q = c.tid; q < nw; q += AIE_NT) lm[q] = 0;
  __syncthreads();
  if (c.tid < c.P.n) {
    const int r = R_I32(c, o_loc_r)[c.tid], col = R_I32(c, o_loc_c)[c.tid];
    if (r >= 0 && col >= 0) c.locmap[r * c.P.W + col] = (uint8_t)(c.tid + 1);
  }
  __syncthreads();
}

// parse_actions base_env.py:552-556 -> base_agent.py:407-438 (lane i decodes agent i)
__device__ void decode_actions(const Ctx& c, const int32_t* __restrict__ aa, const int32_t* __restrict__ ap) {
  const aie_params& P = c.P;
  const int i = c.tid;
  if (i < P.n) {
    int32_t* act = c.act + i * AIE_N_SUB_SLOTS;
#pragma unroll
    for (int s = 0; s < AIE_N_SUB_SLOTS; ++s) act[s] = 0;
    if (aa) {
      const int32_t* a = aa + ((int64_t)c.e * P.n + i) * P.act_a_width;
      if (P.c.multi_action_mode_agents) {
        for (int s = 0; s < P.n_sub_a; ++s) {
          const int v = a[s];
          if (v >= 0 && v <= P.sub_a_dim[s]) act[P.sub_a_slot[s]] = v;
        }
      } else {
        const int v = a[0];
        for (int s = 0; s < P.n_sub_a; ++s)
          if (v >= P.sub_a_base[s] && v < P.sub_a_base[s] + P.sub_a_dim[s])
            act[P.sub_a_slot[s]] = v - P.sub_a_base[s] + 1;
      }
    }
  }
  if (i < AIE_MAX_BRACKETS) {
    int v = 0;
    if (ap && i < P.n_sub_p) {
      const int32_t* a = ap + (int64_t)c.e * P.act_p_width;
      if (P.c.multi_action_mode_planner) v = a[i];
      else {
        const int x = a[0];
        if (x >= 1 && x < 1 + P.n_sub_p * P.sub_p_dim && (x - 1) / P.sub_p_dim == i) v = (x - 1) % P.sub_p_dim + 1;
      }
    }
    c.act_p[i] = v;
  }
}

}  // namespace aie

// ======================================================================================
// Kernels
// ======================================================================================

// BaseEnvironment.step, F/base/base_env.py:929-1032: parse actions, timestep += 1,
// components in list order, scenario_step, observations, masks, rewards, done.
extern "C" __global__ void __launch_bounds__(AIE_NT)
aie_step_kernel(const aie_params P, uint8_t* __restrict__ arena, const int32_t* __restrict__ act_a,
                const int32_t* __restrict__ act_p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  using namespace aie;
  const Ctx c = make_ctx(P, lds, (int)blockIdx.x, (int)threadIdx.x);
  load_record(c, arena);
  decode_actions(c, act_a, act_p);
  __syncthreads();
  rebuild_locmap(c);
  if (P.has_cda) cda_decay_price_history(c);
  __syncthreads();
  const int skip = P.dev_skip_mask;
  if (c.tid == 0) {
    *R_I32(c, o_timestep) += 1;
    if (!(skip & 1)) {
      for (int k = 0; k < P.c.n_components; ++k) {
        switch (P.c.components[k]) {
          case AIE_COMP_BUILD: build_component_step(c); break;
          case AIE_COMP_CDA: cda_component_step(c); break;
          case AIE_COMP_GATHER: gather_component_step(c); break;
          case AIE_COMP_TAX: tax_component_step(c); break;
          default: break;
        }
      }
    }
  }
  __syncthreads();
  if (!(skip & 2)) scenario_step_regen(c);
  __syncthreads();
  if (!(skip & 4)) write_spatial_observations(c, arena);
  if (!(skip & 8)) write_flat_observations_and_masks(c, arena);
  __syncthreads();
  if (!(skip & 16)) compute_rewards(c, arena);
  __syncthreads();
  if (c.tid == 0) {
    const int done = *R_I32(c, o_timestep) >= P.c.episode_length;
    (arena + P.a_done)[c.e] = (uint8_t)done;
    if (done) *R_I32(c, o_completions) += 1;
  }
  __syncthreads();
  if (!(skip & 32)) store_record(c, arena);
}

// BaseEnvironment.reset, F/base/base_env.py:852-927, with LayoutFromFile
// reset_starting_layout / reset_agent_states / additional_reset_steps
// (layout_from_file.py:323-370, 564-593) and the component resets (build.py:224-254,
// move.py:193-210, continuous_double_auction.py:643-668, redistribution.py:1109-1139).
extern "C" __global__ void __launch_bounds__(AIE_NT)
aie_reset_kernel(const aie_params P, uint8_t* __restrict__ arena, const uint8_t* __restrict__ mask) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  using namespace aie;
  const int e = (int)blockIdx.x;
  if (mask && !mask[e]) return;
  const Ctx c = make_ctx(P, lds, e, (int)threadIdx.x);
  const int n = P.n, HW = P.HW, tid = c.tid;
  load_record(c, arena);
  __syncthreads();
  {  // layout_from_file.py:323-334: resources back on every source block, no houses
    uint32_t* cells = R_CELLS(c);
    for (int q = tid; q < HW; q += AIE_NT) {
      const uint32_t fl = cells[q] >> 24;
      cells[q] = AIE_CELL_PACK((fl & AIE_CELL_STONE_SRC) ? 1 : 0, (fl & AIE_CELL_WOOD_SRC) ? 1 : 0, 0xff, fl);
    }
    if (tid < n) {
      R_I32(c, o_inv_res)[tid] = 0; R_I32(c, o_inv_res)[n + tid] = 0;
      R_I32(c, o_esc_res)[tid] = 0; R_I32(c, o_esc_res)[n + tid] = 0;
      R_F64(c, o_inv_coin)[tid] = P.c.starting_agent_coin;
      R_F64(c, o_esc_coin)[tid] = 0;
      R_F64(c, o_labor)[tid] = 0;
      R_I32(c, o_loc_r)[tid] = -1;
      R_I32(c, o_loc_c)[tid] = -1;
      if (!P.has_build) { R_F64(c, o_build_payment)[tid] = 0; R_F64(c, o_build_skill)[tid] = 0; }
      if (!P.has_gather) R_F64(c, o_bonus_gather_prob)[tid] = 0;
    }
    if (P.has_cda) {
      for (int q = tid; q < 2 * P.M; q += AIE_NT) { R_I32(c, o_cda_bids)[q] = 0; R_I32(c, o_cda_asks)[q] = 0; }
      for (int q = tid; q < 2 * n * P.P; q += AIE_NT) {
        R_U8(c, o_cda_bid_hist)[q] = 0; R_U8(c, o_cda_ask_hist)[q] = 0;
        R_F64(c, o_cda_price_history)[q] = 0;
      }
      for (int q = tid; q < 2 * n; q += AIE_NT) R_I32(c, o_cda_n_orders)[q] = 0;
      if (tid < 2) { R_I32(c, o_cda_n_bids)[tid] = 0; R_I32(c, o_cda_n_asks)[tid] = 0; }
    }
  }
  __syncthreads();
  rebuild_locmap(c);  // all agents off the board
  if (tid == 0) {
    *R_I32(c, o_timestep) = 0;
    for (int i = 0; i < n; ++i) {  // layout_from_file.py:360-370
      int r = (int)rng_interval(c, (uint32_t)(P.H - 1)), col = (int)rng_interval(c, (uint32_t)(P.W - 1)), tries = 0;
      while (!can_agent_occupy(c, r, col, i)) {
        r = (int)rng_interval(c, (uint32_t)(P.H - 1));
        col = (int)rng_interval(c, (uint32_t)(P.W - 1));
        if (++tries > 200) break;  // the reference raises TimeoutError
      }
      R_I32(c, o_loc_r)[i] = r;
      R_I32(c, o_loc_c)[i] = col;
      c.locmap[r * P.W + col] = (uint8_t)(i + 1);
    }
    for (int k = 0; k < P.c.n_components; ++k) {
      switch (P.c.components[k]) {
        case AIE_COMP_BUILD:
          for (int i = 0; i < n; ++i) {
            double skill = 1, pay = 1;
            const double pm = (double)P.c.build_payment_max_skill_multiplier;
            if (P.c.build_skill_dist == AIE_SKILL_PARETO) {
              skill = rng_pareto(c, 4.0);
              pay = (pm - 1) * skill + 1; if (pm < pay) pay = pm;
            } else if (P.c.build_skill_dist == AIE_SKILL_LOGNORMAL) {
              skill = rng_lognormal(c, -1.0, 0.5);
              pay = (pm - 1) * skill + 1; if (pm < pay) pay = pm;
            }
            R_F64(c, o_build_payment)[i] = pay * (double)P.c.build_payment;
            R_F64(c, o_build_skill)[i] = skill;
          }
          break;
        case AIE_COMP_GATHER:
          for (int i = 0; i < n; ++i) {
            double b = 0.0;
            if (P.c.gather_skill_dist == AIE_SKILL_PARETO) { b = rng_pareto(c, 3.0); b = (b < 2 ? b : 2) / 2; }
            else if (P.c.gather_skill_dist == AIE_SKILL_LOGNORMAL) { b = rng_lognormal(c, -2.022, 0.938); b = (b < 2 ? b : 2) / 2; }
            R_F64(c, o_bonus_gather_prob)[i] = b;
          }
          break;
        case AIE_COMP_TAX:
          for (int b = 0; b < P.NB; ++b) R_I32(c, o_tax_rate_idx)[b] = 0;
          *R_I32(c, o_tax_cycle_pos) = 1;
          for (int i = 0; i < n; ++i) {
            R_F64(c, o_tax_last_coin)[i] = R_F64(c, o_inv_coin)[i] + R_F64(c, o_esc_coin)[i];
            R_F64(c, o_tax_last_income)[i] = 0;
            R_F64(c, o_tax_last_marginal_rate)[i] = 0;
          }
          *R_F64(c, o_tax_total_collected) = 0;
          break;
        default: break;
      }
    }
    if (P.c.fixed_four_skill_and_loc) {  // layout_from_file.py:582-586
      int32_t* order = c.perm;
      for (int i = 0; i < n; ++i) {
        c.locmap[R_I32(c, o_loc_r)[i] * P.W + R_I32(c, o_loc_c)[i]] = 0;
        R_I32(c, o_loc_r)[i] = -1;
        R_I32(c, o_loc_c)[i] = -1;
      }
      rng_permutation(c, n, order);
      for (int k = 0; k < n; ++k) {
        const int i = order[k];
        const int r = P.c.ranked_locs[k][0], col = P.c.ranked_locs[k][1];
        if (can_agent_occupy(c, r, col, i)) {
          R_I32(c, o_loc_r)[i] = r;
          R_I32(c, o_loc_c)[i] = col;
          c.locmap[r * P.W + col] = (uint8_t)(i + 1);
        }
        R_F64(c, o_build_payment)[i] = P.c.avg_ranked_skill[k];
      }
    }
  }
  __syncthreads();
  current_metrics(c);
  __syncthreads();
  if (tid <= n) R_F64(c, o_util)[tid] = scr_part(c)[tid];
  __syncthreads();
  write_spatial_observations(c, arena);
  write_flat_observations_and_masks(c, arena);
  if (tid < n) reinterpret_cast<float*>(arena + P.a_rew_a)[(int64_t)e * n + tid] = 0.0f;
  if (tid == 0) {
    reinterpret_cast<float*>(arena + P.a_rew_p)[e] = 0.0f;
    (arena + P.a_done)[e] = 0;
  }
  __syncthreads();
  store_record(c, arena);
}

// np.random.seed(base_seed + e): init_genrand (Knuth LCG), pos = 624.
// BaseEnvironment.seed, F/base/base_env.py:481-494.  One thread per replica.
extern "C" __global__ void aie_seed_kernel(const aie_params P, uint8_t* __restrict__ arena, uint32_t base_seed) {
  const int e = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (e >= P.E) return;
  uint8_t* rec = arena + P.a_records + (int64_t)e * P.rec_bytes;
  uint32_t* mt = reinterpret_cast<uint32_t*>(rec + P.o_mt);
  uint32_t x = base_seed + (uint32_t)e;
  mt[0] = x;
  for (int i = 1; i < AIE_MT_N; ++i) {
    x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i;
    mt[i] = x;
  }
  *reinterpret_cast<int32_t*>(rec + P.o_mt_pos) = AIE_MT_N;
  *reinterpret_cast<int32_t*>(rec + P.o_mt_has_gauss) = 0;
  *reinterpret_cast<double*>(rec + P.o_mt_gauss) = 0.0;
}

// Synthetic uniform random policy of the benchmark: a counter RNG
// keyed (seed, global replica id, t, agent); one thread per (replica, agent slot).
extern "C" __global__ void aie_sample_actions_kernel(const aie_params P, uint64_t seed, int64_t env_offset, int64_t t,
                                                     int32_t* __restrict__ act_a, int32_t* __restrict__ act_p) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int per_env = P.n * P.act_a_width + P.act_p_width;
  if (q >= (int64_t)P.E * per_env) return;
  const int e = (int)(q / per_env);
  const int j = (int)(q - (int64_t)e * per_env);
  const uint32_t u = aie_counter_rng(seed, (uint64_t)(env_offset + e), (uint64_t)t, (uint64_t)j);
  if (j < P.n * P.act_a_width) {
    if (!act_a) return;
    int range;
    if (P.c.multi_action_mode_agents) range = P.n_sub_a ? P.sub_a_dim[j % P.act_a_width] + 1 : 1;
    else range = P.A;
    act_a[(int64_t)e * P.n * P.act_a_width + j] = (int32_t)(((uint64_t)u * (uint64_t)range) >> 32);
  } else {
    if (!act_p) return;
    const int range = P.c.multi_action_mode_planner ? P.sub_p_dim + 1 : 1 + P.n_sub_p * P.sub_p_dim;
    act_p[(int64_t)e * P.act_p_width + (j - P.n * P.act_a_width)] = (int32_t)(((uint64_t)u * (uint64_t)range) >> 32);
  }
}
