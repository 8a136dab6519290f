/*
 * aie_oracle.c -- TEST INFRASTRUCTURE.  A plain-C, single-replica-at-a-time CPU
 * restatement of the reference Foundation env.reset()/env.step() for the
 * gather-trade-build family.  It is the checker for the HIP path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * (ai-economist_amd/) never calls into this file.
 *
 * Parity status: PINNED.  This restatement is itself checked, step by step, against
 * (a) the live reference Python (tests/test_oracle_vs_reference.py, only where
 * /root/reference exists) and (b) the committed fixtures in tests/golden/ that
 * oracle/gen_golden.py produced by running the unmodified reference.
 *
 * It steps the SAME byte records / dense tensors that the device uses
 * (ai-economist_amd/csrc/aie_layout.h), so a parity check is a byte compare for all
 * integer state and a tolerance compare for f64/f32 fields.
 *
 * Every function cites the reference file:line it follows (paths relative to the
 * reference tree; F/ = ai_economist/foundation/).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../ai-economist_amd/csrc/aie_layout.h"

typedef struct {
  const aie_params* p;
  uint8_t* rec;              /* this replica's record                                 */
  uint8_t* arena;
  int e;
  int act[AIE_MAX_AGENTS][AIE_N_SUB_SLOTS]; /* decoded per-subspace actions           */
  int act_p[AIE_MAX_BRACKETS];
  int act_wide[AIE_MAX_AGENTS_WIDE]; /* one-step-economy: SimpleLabor action per agent */
} ctx_t;

/* dense-log event rows of replicas [0, ev_replicas) (include/aie.h: AIE_EV_*) */
#define EV(c) ((c)->e < (c)->p->ev_replicas ? (int32_t*)((c)->arena + (c)->p->a_events + (int64_t)(c)->e * (c)->p->ev_stride) : (int32_t*)0)
#define MET(c) ((c)->arena + (c)->p->a_metrics + (int64_t)(c)->e * (c)->p->met_bytes)
#define F64(c, off) ((double*)((c)->rec + (c)->p->off))
#define I32(c, off) ((int32_t*)((c)->rec + (c)->p->off))
#define U8(c, off) ((uint8_t*)((c)->rec + (c)->p->off))
#define I8(c, off) ((int8_t*)((c)->rec + (c)->p->off))
#define CELLS(c) ((uint32_t*)((c)->rec + (c)->p->o_cells))
/* byte lanes of the packed cell word (little endian): stone, wood, owner, flags */
#define CB(c, cell, b) (((uint8_t*)CELLS(c))[4 * (cell) + (b)])
#define C_STONE(c, cell) CB(c, cell, 0)
#define C_WOOD(c, cell) CB(c, cell, 1)
#define C_OWNER(c, cell) (((int8_t*)CELLS(c))[4 * (cell) + 2])
#define C_FLAGS(c, cell) CB(c, cell, 3)

/* ------------------------------------------------------------------------------- */
/* NumPy legacy RandomState (MT19937) -- third-party dependency of the reference,  */
/* numpy (any version: the legacy stream is frozen); algorithms restated from the  */
/* published MT19937 reference code and numpy/random/src/{mt19937,legacy,distributions}. */
/* Used by the reference through np.random.* (F/base/base_env.py:493,              */
/* F/base/world.py:420, F/components/move.py:138, layout_from_file.py:361-366,400). */
/* ------------------------------------------------------------------------------- */
static void mt_twist(uint32_t* mt) {
  const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MAT = 0x9908b0dfu;
  int i;
  uint32_t y;
  for (i = 0; i < 624 - 397; i++) {
    y = (mt[i] & UPPER) | (mt[i + 1] & LOWER);
    mt[i] = mt[i + 397] ^ (y >> 1) ^ ((y & 1u) ? MAT : 0u);
  }
  for (; i < 623; i++) {
    y = (mt[i] & UPPER) | (mt[i + 1] & LOWER);
    mt[i] = mt[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? MAT : 0u);
  }
  y = (mt[623] & UPPER) | (mt[0] & LOWER);
  mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? MAT : 0u);
}

/* ------------------------------------------------------------------------------- */
/* rng_mode == AIE_RNG_FAST (include/aie.h): NOT the reference's generator -- the   */
/* product's throughput mode, restated here so that the device can be checked bit   */
/* for bit in that mode too.  Philox2x32-10 of Salmon, Moraes, Dror, Shaw, "Parallel */
/* random numbers: as easy as 1, 2, 3" (SC'11), constants of the authors' Random123  */
/* library (philox.h: PHILOX_M2x32_0, PHILOX_W32_0); tests/test_rng_fast.py holds    */
/* that library's known-answer vectors.  Stream word g = element g & 1 of            */
/* philox(counter = (lo32(g >> 1), hi32(g >> 1) | salt), key32); the stream is       */
/* consumed in blocks of 624 words under the position bookkeeping of the MT19937     */
/* path below (pos == 624: the next draw opens block + 1), so that every consumer -- */
/* doubles, masked-rejection integers, permutations, polar Gauss -- is shared.       */
/* ------------------------------------------------------------------------------- */
void aie_oracle_philox2x32_10(const uint32_t ctr[2], uint32_t key, uint32_t out[2]) {
  uint32_t c0 = ctr[0], c1 = ctr[1];
  for (int r = 0; r < 10; ++r) {
    const uint64_t prod = (uint64_t)0xD256D193u * (uint64_t)c0;
    const uint32_t hi = (uint32_t)(prod >> 32), lo = (uint32_t)prod;
    c0 = hi ^ key ^ c1;
    c1 = lo;
    key += 0x9E3779B9u;
  }
  out[0] = c0;
  out[1] = c1;
}
/* word `pos` of block `st[1]` of the stream with state st = (key32, block, salt, 0) */
static uint32_t fast_word(const uint32_t* st, int pos) {
  const uint64_t g = (uint64_t)st[1] * 624u + (uint64_t)pos;
  const uint64_t pair = g >> 1;
  const uint32_t ctr[2] = {(uint32_t)pair, (uint32_t)(pair >> 32) | st[2]};
  uint32_t out[2];
  aie_oracle_philox2x32_10(ctr, st[0], out);
  return out[g & 1];
}

static uint32_t rng_u32(ctx_t* c) {
  uint32_t* mt = (uint32_t*)(c->rec + c->p->o_mt);
  int32_t* pos = I32(c, o_mt_pos);
  if (c->p->c.rng_mode == AIE_RNG_FAST) {
    if (*pos >= 624) {
      mt[1] += 1u;
      *pos = 0;
    }
    return fast_word(mt, (*pos)++);
  }
  if (*pos >= 624) {
    mt_twist(mt);
    *pos = 0;
  }
  uint32_t y = mt[(*pos)++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

/* legacy random_sample: 53-bit double from two words */
static double rng_double(ctx_t* c) {
  uint32_t a = rng_u32(c) >> 5, b = rng_u32(c) >> 6;
  return (a * 67108864.0 + b) / 9007199254740992.0;
}

/* random_interval(max): masked rejection on 32-bit words (max <= 0xffffffff) */
static uint32_t rng_interval(ctx_t* c, uint32_t max) {
  if (max == 0) return 0;
  uint32_t mask = max, v;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
  while ((v = (rng_u32(c) & mask)) > max) {}
  return v;
}

/* np.random.randint(0, high) legacy (int64 default, masked rejection, rng = high-1) */
static int rng_randint(ctx_t* c, int high) { return (int)rng_interval(c, (uint32_t)(high - 1)); }

static double rng_gauss(ctx_t* c) {
  int32_t* has = I32(c, o_mt_has_gauss);
  double* g = F64(c, o_mt_gauss);
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
static double rng_std_exponential(ctx_t* c) { return -log(1.0 - rng_double(c)); }
static double rng_pareto(ctx_t* c, double a) { return exp(rng_std_exponential(c) / a) - 1.0; }
static double rng_lognormal(ctx_t* c, double mean, double sigma) {
  return exp(mean + sigma * rng_gauss(c));
}

/* np.random.permutation(n) == shuffle(arange(n)), Fisher-Yates from the top.
 * F/base/world.py:418-422 (World.get_random_order_agents). */
static void rng_permutation(ctx_t* c, int n, int* out) {
  for (int i = 0; i < n; ++i) out[i] = i;
  for (int i = n - 1; i >= 1; --i) {
    int j = (int)rng_interval(c, (uint32_t)i);
    int t = out[i]; out[i] = out[j]; out[j] = t;
  }
}

/* np.random.seed(int): init_genrand; pos = 624 (first draw twists) */
void aie_oracle_seed_one(uint32_t* mt, uint32_t seed) {
  mt[0] = seed;
  for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
}

/* numpy float64 add.reduce over a contiguous run: pairwise summation
 * (numpy/_core/src/umath/loops_utils.h.src, DOUBLE_pairwise_sum). */
static double np_sum(const double* a, int n) {
  if (n < 8) {
    double res = -0.0;
    for (int i = 0; i < n; ++i) res += a[i];
    return res;
  } else if (n <= 128) {
    double r[8];
    int i;
    for (i = 0; i < 8; ++i) r[i] = a[i];
    for (i = 8; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  } else {
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_sum(a, n2) + np_sum(a + n2, n - n2);
  }
}

/* ------------------------------------------------------------------------------- */
/* World helpers: F/base/world.py                                                   */
/* ------------------------------------------------------------------------------- */
static int occupied_by_other(ctx_t* c, int r, int col, int self) {
  const int32_t *lr = I32(c, o_loc_r), *lc = I32(c, o_loc_c);
  for (int j = 0; j < c->p->n; ++j)
    if (j != self && lr[j] == r && lc[j] == col) return 1;
  return 0;
}

/* World.can_agent_occupy (world.py:424-440): in bounds, accessible (no Water; House
 * only if owner, world.py:213-217,256-258,300-305) and unoccupied. */
static int accessible(ctx_t* c, int r, int col, int agent) {
  const aie_params* p = c->p;
  if (r < 0 || r >= p->H || col < 0 || col >= p->W) return 0;
  int cell = r * p->W + col;
  if (C_FLAGS(c, cell) & AIE_CELL_WATER) return 0;
  int8_t o = C_OWNER(c, cell);
  return o < 0 || o == agent;
}
static int can_agent_occupy(ctx_t* c, int r, int col, int agent) {
  return accessible(c, r, col, agent) && !occupied_by_other(c, r, col, agent);
}

/* ------------------------------------------------------------------------------- */
/* Build: F/components/build.py                                                     */
/* ------------------------------------------------------------------------------- */
/* Build.agent_can_build, build.py:70-83 (+ world.py:284-293) */
static int agent_can_build(ctx_t* c, int i) {
  const aie_params* p = c->p;
  const int32_t* inv = I32(c, o_inv_res);
  if (inv[1 * p->n + i] < 1 || inv[0 * p->n + i] < 1) return 0;
  int cell = I32(c, o_loc_r)[i] * p->W + I32(c, o_loc_c)[i];
  if (C_STONE(c, cell) > 0 || C_WOOD(c, cell) > 0) return 0;
  if (C_OWNER(c, cell) >= 0) return 0;
  if (C_FLAGS(c, cell) & (AIE_CELL_WATER | AIE_CELL_STONE_SRC | AIE_CELL_WOOD_SRC)) return 0;
  return 1;
}

static void log_event(ctx_t* c, int type, int a1, int a2, int a3, int a4, int a5, int a6, int a7, int a8, double f) {
  int32_t* ev = EV(c);
  if (!ev || ev[0] >= c->p->ev_cap) return;
  int32_t* row = ev + 4 + ev[0] * AIE_EV_WORDS;
  row[0] = type; row[1] = a1; row[2] = a2; row[3] = a3; row[4] = a4;
  row[5] = a5; row[6] = a6; row[7] = a7; row[8] = a8; row[9] = 0;
  memcpy(row + 10, &f, 8);
  ev[0] += 1;
}

/* Build.component_step, build.py:112-161 */
static void build_step(ctx_t* c) {
  const aie_params* p = c->p;
  int order[AIE_MAX_AGENTS];
  rng_permutation(c, p->n, order); /* drawn even if nobody builds (build.py:121) */
  for (int k = 0; k < p->n; ++k) {
    int i = order[k];
    if (c->act[i][AIE_SUB_BUILD] != 1) continue;
    if (!agent_can_build(c, i)) continue;
    I32(c, o_inv_res)[1 * p->n + i] -= 1;
    I32(c, o_inv_res)[0 * p->n + i] -= 1;
    int cell = I32(c, o_loc_r)[i] * p->W + I32(c, o_loc_c)[i];
    C_OWNER(c, cell) = (int8_t)i; /* world.py:474-479, 240-259 */
    F64(c, o_inv_coin)[i] += F64(c, o_build_payment)[i];
    F64(c, o_labor)[i] += p->c.build_labor;
    log_event(c, AIE_EV_BUILD, i, I32(c, o_loc_r)[i], I32(c, o_loc_c)[i], 0, 0, 0, 0, 0, F64(c, o_build_payment)[i]);
  }
}

/* ------------------------------------------------------------------------------- */
/* Gather: F/components/move.py                                                     */
/* ------------------------------------------------------------------------------- */
/* Gather.component_step, move.py:93-153 */
static void gather_step(ctx_t* c) {
  const aie_params* p = c->p;
  int order[AIE_MAX_AGENTS];
  rng_permutation(c, p->n, order);
  int32_t *lr = I32(c, o_loc_r), *lc = I32(c, o_loc_c);
  for (int k = 0; k < p->n; ++k) {
    int i = order[k];
    int a = c->act[i][AIE_SUB_GATHER];
    int r = lr[i], col = lc[i], nr = r, nc = col;
    if (a != 0) {
      if (a == 1) nc = col - 1;       /* Left  */
      else if (a == 2) nc = col + 1;  /* Right */
      else if (a == 3) nr = r - 1;    /* Up    */
      else nr = r + 1;                /* Down  */
      if (can_agent_occupy(c, nr, nc, i)) { /* world.py:454-460 */
        lr[i] = nr; lc[i] = nc;
      } else {
        nr = r; nc = col;
      }
      if (nr != r || nc != col) F64(c, o_labor)[i] += p->c.move_labor;
    }
    /* collect on the landing tile -- also on a NO-OP (move.py:112-113,136).
     * location_resources is evaluated once, before any consumption (world.py:284-288) */
    int cell = nr * p->W + nc;
    int health[2] = {C_STONE(c, cell), C_WOOD(c, cell)};
    for (int rsrc = 0; rsrc < 2; ++rsrc) {
      if (health[rsrc] >= 1) {
        /* rand() is consumed even if bonus_gather_prob == 0 (move.py:138) */
        int n_gathered = 1 + (rng_double(c) < F64(c, o_bonus_gather_prob)[i] ? 1 : 0);
        I32(c, o_inv_res)[rsrc * p->n + i] += n_gathered;
        CB(c, cell, rsrc) -= 1; /* consume_resource, world.py:481-483 */
        F64(c, o_labor)[i] += p->c.collect_labor;
        log_event(c, AIE_EV_GATHER, i, rsrc, n_gathered, nr, nc, 0, 0, 0, 0.0);
      }
    }
  }
}

/* ------------------------------------------------------------------------------- */
/* ContinuousDoubleAuction: F/components/continuous_double_auction.py               */
/* ------------------------------------------------------------------------------- */
/* create_bid :168-198 / create_ask :200-229 */
static void cda_create_bid(ctx_t* c, int r, int i, int price) {
  const aie_params* p = c->p;
  int32_t* no = I32(c, o_cda_n_orders) + r * p->n;
  if (!(no[i] < p->c.cda_max_num_orders) || F64(c, o_inv_coin)[i] < (double)price) return;
  int32_t* nb = I32(c, o_cda_n_bids) + r;
  I32(c, o_cda_bids)[r * p->M + *nb] = AIE_ORD_PACK(i, price, 0);
  (*nb)++;
  U8(c, o_cda_bid_hist)[(r * p->n + i) * p->P + price] += 1;
  no[i] += 1;
  /* inventory_to_escrow("Coin", price), base_agent.py:279-299 */
  double inv = F64(c, o_inv_coin)[i];
  double tr = inv < (double)price ? inv : (double)price;
  F64(c, o_inv_coin)[i] -= tr;
  F64(c, o_esc_coin)[i] += tr;
  F64(c, o_labor)[i] += p->c.cda_order_labor;
}
static void cda_create_ask(ctx_t* c, int r, int i, int price) {
  const aie_params* p = c->p;
  int32_t* no = I32(c, o_cda_n_orders) + r * p->n;
  int32_t* inv = I32(c, o_inv_res) + r * p->n;
  if (!(no[i] < p->c.cda_max_num_orders && inv[i] > 0)) return;
  int32_t* na = I32(c, o_cda_n_asks) + r;
  I32(c, o_cda_asks)[r * p->M + *na] = AIE_ORD_PACK(i, price, 0);
  (*na)++;
  U8(c, o_cda_ask_hist)[(r * p->n + i) * p->P + price] += 1;
  no[i] += 1;
  inv[i] -= 1;
  I32(c, o_esc_res)[r * p->n + i] += 1;
  F64(c, o_labor)[i] += p->c.cda_order_labor;
}

/* stable insertion sorts reproducing Python's sorted(..., key, reverse) (:249-256) */
static int bid_before(int32_t a, int32_t b) { /* (bid, lifetime) descending */
  int pa = AIE_ORD_PRICE(a), pb = AIE_ORD_PRICE(b);
  if (pa != pb) return pa > pb;
  return AIE_ORD_LIFE(a) > AIE_ORD_LIFE(b);
}
static int ask_before(int32_t a, int32_t b) { /* (ask, -lifetime) ascending */
  int pa = AIE_ORD_PRICE(a), pb = AIE_ORD_PRICE(b);
  if (pa != pb) return pa < pb;
  return AIE_ORD_LIFE(a) > AIE_ORD_LIFE(b);
}
static void stable_sort(int32_t* v, int n, int (*before)(int32_t, int32_t)) {
  for (int i = 1; i < n; ++i) {
    int32_t x = v[i];
    int j = i - 1;
    while (j >= 0 && before(x, v[j])) { v[j + 1] = v[j]; --j; }
    v[j + 1] = x;
  }
}
static void list_pop(int32_t* v, int* n, int idx) {
  for (int k = idx; k + 1 < *n; ++k) v[k] = v[k + 1];
  (*n)--;
}

/* match_orders :231-350 -- literal restatement of the restart loop */
static void cda_match(ctx_t* c) {
  const aie_params* p = c->p;
  for (int r = 0; r < AIE_N_RES; ++r) {
    int possible[AIE_MAX_AGENTS];
    for (int i = 0; i < p->n; ++i) possible[i] = 1;
    int keep_checking = 1;
    int32_t* bids = I32(c, o_cda_bids) + r * p->M;
    int32_t* asks = I32(c, o_cda_asks) + r * p->M;
    int nb = I32(c, o_cda_n_bids)[r], na = I32(c, o_cda_n_asks)[r];
    stable_sort(bids, nb, bid_before);
    stable_sort(asks, na, ask_before);
    for (;;) {
      int any = 0;
      for (int i = 0; i < p->n; ++i) any |= possible[i];
      if (!(any && keep_checking)) break;
      int ib = 0, ia = 0;
      for (;;) {
        if (ib >= nb) { keep_checking = 0; break; }
        int buyer = AIE_ORD_AGENT(bids[ib]);
        if (!possible[buyer]) { ib++; }
        else if (ia >= na) { possible[buyer] = 0; break; }
        else if (AIE_ORD_AGENT(asks[ia]) == buyer) { ia++; }
        else if (AIE_ORD_PRICE(bids[ib]) < AIE_ORD_PRICE(asks[ia])) { possible[buyer] = 0; break; }
        else {
          int32_t bid = bids[ib], ask = asks[ia];
          list_pop(bids, &nb, ib);
          list_pop(asks, &na, ia);
          int seller = AIE_ORD_AGENT(ask);
          int bprice = AIE_ORD_PRICE(bid), aprice = AIE_ORD_PRICE(ask);
          int price = (AIE_ORD_LIFE(bid) <= AIE_ORD_LIFE(ask)) ? aprice : bprice; /* :297-304 */
          U8(c, o_cda_bid_hist)[(r * p->n + buyer) * p->P + bprice] -= 1;
          U8(c, o_cda_ask_hist)[(r * p->n + seller) * p->P + aprice] -= 1;
          I32(c, o_cda_n_orders)[r * p->n + seller] -= 1;
          I32(c, o_cda_n_orders)[r * p->n + buyer] -= 1;
          F64(c, o_cda_price_history)[(r * p->n + seller) * p->P + price] += 1.0;
          { /* executed_trades -> get_metrics :585-641: per (side, commodity, agent): n_sales, sum of prices */
            int32_t* tm = (int32_t*)(MET(c) + p->mo_cda);
            int32_t* sell = tm + ((0 * AIE_N_RES + r) * p->n + seller) * 2;
            int32_t* buy = tm + ((1 * AIE_N_RES + r) * p->n + buyer) * 2;
            sell[0] += 1; sell[1] += price;
            buy[0] += 1; buy[1] += price;
          }
          log_event(c, AIE_EV_TRADE, r, seller, buyer, aprice, bprice, price, AIE_ORD_LIFE(ask), AIE_ORD_LIFE(bid), 0.0);
          I32(c, o_esc_res)[r * p->n + seller] -= 1;
          I32(c, o_inv_res)[r * p->n + buyer] += 1;
          F64(c, o_esc_coin)[buyer] -= (double)bprice;
          F64(c, o_inv_coin)[seller] += (double)price;
          F64(c, o_inv_coin)[buyer] += (double)(bprice - price);
          break;
        }
      }
    }
    I32(c, o_cda_n_bids)[r] = nb;
    I32(c, o_cda_n_asks)[r] = na;
  }
}

/* remove_expired_orders :352-406 */
static void cda_expire(ctx_t* c) {
  const aie_params* p = c->p;
  for (int r = 0; r < AIE_N_RES; ++r) {
    int32_t* bids = I32(c, o_cda_bids) + r * p->M;
    int nb = I32(c, o_cda_n_bids)[r], k = 0;
    for (int q = 0; q < nb; ++q) {
      int32_t o = bids[q];
      int life = AIE_ORD_LIFE(o) + 1, ag = AIE_ORD_AGENT(o), pr = AIE_ORD_PRICE(o);
      if (life <= p->c.cda_order_duration) bids[k++] = AIE_ORD_PACK(ag, pr, life);
      else {
        double esc = F64(c, o_esc_coin)[ag];
        double tr = esc < (double)pr ? esc : (double)pr; /* escrow_to_inventory */
        F64(c, o_esc_coin)[ag] -= tr;
        F64(c, o_inv_coin)[ag] += tr;
        U8(c, o_cda_bid_hist)[(r * p->n + ag) * p->P + pr] -= 1;
        I32(c, o_cda_n_orders)[r * p->n + ag] -= 1;
      }
    }
    I32(c, o_cda_n_bids)[r] = k;
    int32_t* asks = I32(c, o_cda_asks) + r * p->M;
    int na = I32(c, o_cda_n_asks)[r];
    k = 0;
    for (int q = 0; q < na; ++q) {
      int32_t o = asks[q];
      int life = AIE_ORD_LIFE(o) + 1, ag = AIE_ORD_AGENT(o), pr = AIE_ORD_PRICE(o);
      if (life <= p->c.cda_order_duration) asks[k++] = AIE_ORD_PACK(ag, pr, life);
      else {
        I32(c, o_esc_res)[r * p->n + ag] -= 1;
        I32(c, o_inv_res)[r * p->n + ag] += 1;
        U8(c, o_cda_ask_hist)[(r * p->n + ag) * p->P + pr] -= 1;
        I32(c, o_cda_n_orders)[r * p->n + ag] -= 1;
      }
    }
    I32(c, o_cda_n_asks)[r] = k;
  }
}

/* ContinuousDoubleAuction.component_step :440-489 */
static void cda_step(ctx_t* c) {
  const aie_params* p = c->p;
  for (int r = 0; r < AIE_N_RES; ++r) {
    for (int i = 0; i < p->n; ++i) {
      double* ph = F64(c, o_cda_price_history) + (r * p->n + i) * p->P;
      for (int k = 0; k < p->P; ++k) ph[k] *= 0.995; /* :451 */
      int a = c->act[i][r ? AIE_SUB_BUY1 : AIE_SUB_BUY0];
      if (a > 0) cda_create_bid(c, r, i, a - 1);
      a = c->act[i][r ? AIE_SUB_SELL1 : AIE_SUB_SELL0];
      if (a > 0) cda_create_ask(c, r, i, a - 1);
    }
  }
  cda_match(c);
  cda_expire(c);
}

/* ------------------------------------------------------------------------------- */
/* PeriodicBracketTax: F/components/redistribution.py                               */
/* ------------------------------------------------------------------------------- */
/* curr_marginal_rates :396-417 with curr_rate_max :390-394 (tax annealing: the limit follows
 * _last_completions, refreshed by generate_masks :1036-1046) */
static double tax_rate(ctx_t* c, int b) {
  const aie_params* p = c->p;
  if (p->c.tax_model == AIE_TAX_MODEL_WRAPPER) return p->c.tax_disc_rates[I32(c, o_tax_rate_idx)[b]];
  double r = p->c.tax_fixed_rates[b];
  if (p->c.tax_model == AIE_TAX_SAEZ) { /* np.minimum(curr_bracket_tax_rates, curr_rate_max) :406-409 */
    r = F64(c, o_tax_saez_rates)[b];
    if (!p->c.tax_annealing && p->c.tax_rate_max < r) r = p->c.tax_rate_max;
  }
  if (p->c.tax_annealing) {
    double cap = aie_annealed_tax_limit(*I32(c, o_tax_last_completions), p->c.tax_annealing_warmup,
                                        p->c.tax_annealing_slope, p->c.tax_rate_max);
    if (cap < r) r = cap;
  }
  return r;
}
/* _curr_rates_obs: what the "curr_rates" observation shows (cached at period starts and resets) */
static double tax_rate_obs(ctx_t* c, int b) {
  if (c->p->c.tax_model == AIE_TAX_SAEZ) return F64(c, o_tax_saez_obs_rates)[b];
  return tax_rate(c, b);
}
/* annealed_tax_mask utils.py:59-118 for discretised rate k */
static float tax_rate_action_mask(ctx_t* c, int k) {
  const aie_params* p = c->p;
  if (!p->c.tax_annealing) return 1.0f;
  double full = 0;
  for (int q = 0; q < p->c.tax_n_disc_rates; ++q) if (fabs(p->c.tax_disc_rates[q]) > full) full = fabs(p->c.tax_disc_rates[q]);
  double vis = aie_annealed_tax_limit(*I32(c, o_tax_last_completions), p->c.tax_annealing_warmup,
                                      p->c.tax_annealing_slope, full);
  return fabs(p->c.tax_disc_rates[k]) <= vis ? 1.0f : 0.0f;
}
/* marginal_rate :837-844 */
static double tax_marginal_rate(ctx_t* c, double income) {
  const aie_params* p = c->p;
  if (income < 0) return 0.0;
  for (int b = 0; b < p->NB; ++b) {
    double lo = p->c.tax_bracket_cutoffs[b];
    double hi = (b + 1 < p->NB) ? p->c.tax_bracket_cutoffs[b + 1] : INFINITY;
    if (income >= lo && income < hi) return tax_rate(c, b);
  }
  return tax_rate(c, 0); /* argmax of all-False == 0 */
}
/* taxes_due :846-851 */
static double tax_due(ctx_t* c, double income) {
  const aie_params* p = c->p;
  double bt[AIE_MAX_BRACKETS];
  for (int b = 0; b < p->NB; ++b) {
    double cut = p->c.tax_bracket_cutoffs[b];
    double size = (b + 1 < p->NB) ? p->c.tax_bracket_cutoffs[b + 1] - cut : INFINITY;
    double past = income - cut;
    if (past < 0) past = 0;
    double bin = size < past ? size : past;
    bt[b] = tax_rate(c, b) * bin;
  }
  return np_sum(bt, p->NB);
}
/* enact_taxes :853-915 */
static void tax_enact(ctx_t* c) {
  const aie_params* p = c->p;
  double net = 0, day_eff = 0;
  for (int b = 0; b < p->NB; ++b) log_event(c, AIE_EV_TAX_BRACKET, b, 0, 0, 0, 0, 0, 0, 0, tax_rate(c, b));
  for (int i = 0; i < p->n; ++i) {
    double income = (F64(c, o_inv_coin)[i] + F64(c, o_esc_coin)[i]) - F64(c, o_tax_last_coin)[i];
    double due = tax_due(c, income);
    double inv = F64(c, o_inv_coin)[i];
    double eff = inv < due ? inv : due; /* don't take from escrow */
    double mr = tax_marginal_rate(c, income);
    F64(c, o_inv_coin)[i] -= eff;
    net += eff;
    log_event(c, AIE_EV_TAX, i, 0, 0, 0, 0, 0, 0, 0, eff);
    F64(c, o_tax_last_income)[i] = income;
    F64(c, o_tax_last_marginal_rate)[i] = mr;
    /* bookkeeping for get_metrics :1141-1186 (redistribution.py:878-895) */
    day_eff += eff / (income > 0.000001 ? income : 0.000001);
    ((double*)(MET(c) + p->mo_tax_income))[i] += income > 0 ? income : 0.0;
    ((double*)(MET(c) + p->mo_tax_paid))[i] += eff;
    int bin = 0; /* income_bin :828-835 */
    if (income >= 0)
      for (int b = 0; b < p->NB; ++b)
        if (income >= p->c.tax_bracket_cutoffs[b] && (b + 1 == p->NB || income < p->c.tax_bracket_cutoffs[b + 1])) { bin = b; break; }
    ((int32_t*)(MET(c) + p->mo_tax_occ))[bin] += 1;
  }
  for (int b = 0; b < p->NB; ++b) ((double*)(MET(c) + p->mo_tax_sched))[b] += tax_rate(c, b);
  *(double*)(MET(c) + p->mo_tax_eff) += day_eff;
  *(int32_t*)(MET(c) + p->mo_tax_days) += 1;
  *F64(c, o_tax_total_collected) += net;
  double lump = net / p->n;
  for (int i = 0; i < p->n; ++i) {
    F64(c, o_inv_coin)[i] += lump;
    F64(c, o_tax_last_coin)[i] = F64(c, o_inv_coin)[i] + F64(c, o_esc_coin)[i];
  }
}
/* ------------------------------------------------------------------------------- */
/* tax_model == "saez", redistribution.py:436-823.  State: aie_layout.h a_saez.     */
/* ------------------------------------------------------------------------------- */
#define SAEZ(c) ((c)->arena + (c)->p->a_saez + (int64_t)(c)->e * (c)->p->saez_stride)
static double saez_curr_rate_max(ctx_t* c) { /* curr_rate_max :390-394 */
  const aie_params* p = c->p;
  if (!p->c.tax_annealing) return p->c.tax_rate_max;
  return aie_annealed_tax_limit(*I32(c, o_tax_last_completions), p->c.tax_annealing_warmup,
                                p->c.tax_annealing_slope, p->c.tax_rate_max);
}
static double saez_pareto(const aie_params* p, double z) { /* :636-643 */
  if (p->c.saez_pareto_weight_uniform) return 1.0;
  return 1.0 / (z > 1.0 ? z : 1.0);
}
static double clip01(double x) { return x < 0 ? 0 : (x > 1 ? 1 : x); }

/* estimate_uniform_income_elasticity :548-596 */
static void saez_estimate_elasticity(const double* buf, int len, double elas_tm1, double log_z0_tm1,
                                     double* elas_out, double* log_z0_out) {
  double* zs = (double*)malloc(sizeof(double) * (size_t)(2 * len + 2));
  double* taus = zs + len + 1;
  int m = 0;
  for (int k = 0; k < len; ++k)
    if (buf[2 * k] > 0 && buf[2 * k + 1] < 1) { zs[m] = buf[2 * k]; taus[m] = buf[2 * k + 1]; m++; }
  *elas_out = elas_tm1; *log_z0_out = log_z0_tm1;
  if (m >= 10) {
    double mean = np_sum(taus, m) / m; /* np.std: sqrt(mean(abs(x - mean)**2)) */
    double* d = (double*)malloc(sizeof(double) * (size_t)m);
    for (int k = 0; k < m; ++k) d[k] = (taus[k] - mean) * (taus[k] - mean);
    double sd = sqrt(np_sum(d, m) / m);
    free(d);
    if (!(sd < 1e-6)) {
      /* OLS of log income on [log(1 - marginal rate), 1]; the 2x2 normal equations in closed form
       * (the reference goes through np.linalg.inv: agreement to ~1e-12 relative, not bitwise) */
      double sxx = 0, sx = 0, sxy = 0, sy = 0;
      for (int k = 0; k < m; ++k) {
        double t1 = 1 - taus[k]; if (t1 < 1e-9) t1 = 1e-9;
        double zz = zs[k]; if (zz < 1e-9) zz = 1e-9;
        double x = log(t1), y = log(zz);
        sxx += x * x; sx += x; sxy += x * y; sy += y;
      }
      double det = sxx * (double)m - sx * sx;
      double i00 = (double)m / det, i01 = -sx / det, i11 = sxx / det;
      double elas = i00 * sxy + i01 * sy, log_z0 = i01 * sxy + i11 * sy;
      double inst = elas > 0.0 ? elas : 0.0;
      *elas_out = ((1 - 0.98) * inst) + (0.98 * elas_tm1);
      *log_z0_out = log_z0;
    }
  }
  free(zs);
}

/* the `saez_buffer` property, redistribution.py:514-525: the trainer's global buffer (if set) followed by the local
 * samples added since the buffers were last reset; else the local buffer.  Returns a malloc'd copy, *len pairs. */
static double* saez_effective_buffer(ctx_t* c, int* len) {
  const aie_params* p = c->p;
  const uint8_t* blk = SAEZ(c);
  const int32_t* hdr = (const int32_t*)blk;
  const double* lbuf = (const double*)(blk + AIE_SAEZ_OFF_BUF);
  const uint8_t* gblk = c->arena + p->a_saez_global;
  const int glen = p->saez_global_cap ? *(const int32_t*)gblk : 0;
  const int llen = hdr[0];
  const int tail = glen > 0 ? (hdr[2] < llen ? hdr[2] : llen) : llen;
  *len = glen + tail;
  double* out = (double*)malloc(sizeof(double) * 2 * (size_t)(*len + 1));
  if (glen > 0) memcpy(out, gblk + 16, sizeof(double) * 2 * (size_t)glen);
  memcpy(out + 2 * glen, lbuf + 2 * (llen - tail), sizeof(double) * 2 * (size_t)tail);
  return out;
}

/* compute_and_set_new_period_rates_from_saez_formula :437-513 (buffer already has >= min samples) */
static void saez_formula(ctx_t* c, double* rates) {
  const aie_params* p = c->p;
  uint8_t* blk = SAEZ(c);
  const int NB = p->NB, T = AIE_SAEZ_BINS;
  int len;
  double* eff = saez_effective_buffer(c, &len);
  double* el = (double*)(blk + AIE_SAEZ_OFF_ELAS); /* elas_t, elas_tm1, log_z0_t, log_z0_tm1 */
  double* avg = (double*)(blk + AIE_SAEZ_OFF_AVG);
  const double* buf = eff;
  const double* edges = p->saez_edges;
  el[1] = el[0]; el[3] = el[2];
  double elas_t, log_z0_t;
  saez_estimate_elasticity(buf, len, el[1], el[3], &elas_t, &log_z0_t);
  el[0] = elas_t; el[2] = log_z0_t;
  if (p->c.saez_fixed_elas_given) elas_t = p->c.saez_fixed_elas;

  /* get_binned_saez_welfare_weight_and_pareto_params :598-752; np.histogram: [e_i, e_i+1), last bin closed */
  double counts[AIE_SAEZ_BINS];
  for (int i = 0; i < T; ++i) counts[i] = 0;
  int n_below = 0, n_above = 0;
  double* above = (double*)malloc(sizeof(double) * (size_t)(2 * len + 2));
  double* above_w = above + len + 1;
  for (int k = 0; k < len; ++k) {
    double z = buf[2 * k];
    if (z < edges[0]) n_below++;
    else if (z > edges[T]) { above[n_above] = z; above_w[n_above] = saez_pareto(p, z); n_above++; }
    else {
      int lo = 0, hi = T; /* largest i with edges[i] <= z */
      while (hi - lo > 1) { int mid = (lo + hi) / 2; if (edges[mid] <= z) lo = mid; else hi = mid; }
      counts[lo] += 1;
    }
  }
  double w_below = (double)n_below; /* pareto(max(z, 0)) == 1 for z < 0, either weight type */
  double w_above = n_above > 0 ? np_sum(above_w, n_above) : 0.0;
  double per_bin[AIE_SAEZ_BINS + 1], dens[AIE_SAEZ_BINS + 1], pz[AIE_SAEZ_BINS + 1];
  for (int i = 0; i < T; ++i) per_bin[i] = counts[i] * saez_pareto(p, 0.5 * (edges[i] + edges[i + 1]));
  double cum = np_sum(per_bin, T);
  cum += w_below; cum += w_above;
  const double norm = cum + 1e-9;
  for (int i = 0; i < T; ++i) dens[i] = per_bin[i] / norm;
  dens[T] = w_above / norm;
  const double n_total = np_sum(counts, T) + n_below + n_above;
  for (int i = 0; i < T; ++i) pz[i] = counts[i] / n_total;
  pz[T] = n_above / n_total;
  const double p_below = n_below / n_total;
  double g[AIE_SAEZ_BINS + 1], gz[AIE_SAEZ_BINS + 1], az[AIE_SAEZ_BINS + 1], taus[AIE_SAEZ_BINS + 1];
  { /* reversed cumulative sums: weight / probability of incomes >= z */
    double cd = 0, cp = 0;
    for (int i = T; i >= 0; --i) {
      cd = (i == T) ? dens[i] : cd + dens[i];
      cp = (i == T) ? pz[i] : cp + pz[i];
      g[i] = cd / (cp + 1e-9);
    }
  }
  for (int i = 0; i < T; ++i) gz[i] = 0.5 * (g[i] + g[i + 1]);
  gz[T] = g[T];
  { /* compute_binned_a_distribution :700-744 */
    double cum_pz = pz[0] + p_below;
    for (int i = 0; i < T; ++i) {
      if (i > 0) cum_pz = clip01(cum_pz + pz[i]);
      double p_geq = 1 - cum_pz + (0.5 * pz[i]);
      if (pz[i] == 0) az[i] = NAN;
      else {
        double z = 0.5 * (edges[i] + edges[i + 1]);
        double paz = z * pz[i] / (clip01(p_geq) + 1e-9);
        az[i] = paz / (edges[i + 1] - edges[i]);
      }
    }
    if (n_above > 0) {
      double mean_above = np_sum(above, n_above) / n_above;
      az[T] = mean_above / (mean_above - edges[T] + 1e-9);
    } else az[T] = 0.0;
  }
  free(above);
  free(eff);
  /* get_saez_marginal_rates :754-790 */
  for (int i = 0; i <= T; ++i) taus[i] = (1.0 - gz[i]) / (1.0 - gz[i] + az[i] * elas_t + 1e-9);
  {
    double last_rate = 0.0; int last_idx = -1;
    for (int i = 0; i <= T; ++i) {
      if (isnan(taus[i])) continue;
      if (i - last_idx > 1) { /* np.linspace(last, tau, gap + 2)[1:-1] */
        int gap = i - last_idx - 1;
        double step = (taus[i] - last_rate) / (double)(gap + 1);
        for (int j = 1; j <= gap; ++j) taus[last_idx + j] = (double)j * step + last_rate;
      }
      last_rate = taus[i]; last_idx = i;
    }
  }
  /* bracketize_schedule :792-823 */
  double last_total = 0;
  for (int b = 0; b + 1 < NB; ++b) {
    const double income = p->c.tax_bracket_cutoffs[b + 1];
    double bin_taxes[AIE_SAEZ_BINS + 1];
    for (int i = 0; i <= T; ++i) {
      double past = income - edges[i]; if (past < 0) past = 0;
      double size = i < T ? edges[i + 1] - edges[i] : INFINITY;
      bin_taxes[i] = taus[i] * (size < past ? size : past);
    }
    double due = np_sum(bin_taxes, T + 1); if (due < 0) due = 0;
    rates[b] = (due - last_total) / (p->c.tax_bracket_cutoffs[b + 1] - p->c.tax_bracket_cutoffs[b]);
    last_total = due;
  }
  rates[NB - 1] = taus[T];
  const double lo = p->c.tax_rate_min, hi = saez_curr_rate_max(c); /* np.clip :497-506 */
  for (int b = 0; b < NB; ++b) {
    if (rates[b] < lo) rates[b] = lo;
    if (rates[b] > hi) rates[b] = hi;
    avg[b] = (avg[b] * 0.99) + (rates[b] * 0.01);
  }
}

/* period start: random rates until the buffer holds _buffer_size samples :444-458 */
static void saez_set_new_period_rates(ctx_t* c) {
  const aie_params* p = c->p;
  int32_t* hdr = (int32_t*)SAEZ(c);
  double* rates = F64(c, o_tax_saez_rates);
  if (!hdr[1]) {
    int eff_len;
    free(saez_effective_buffer(c, &eff_len));
    if (eff_len >= p->c.saez_buffer_size) hdr[1] = 1;
  }
  if (!hdr[1]) { /* np.random.uniform(low=rate_min, high=curr_rate_max, size=n_brackets): low + (high - low) * u */
    const double lo = p->c.tax_rate_min, hi = saez_curr_rate_max(c);
    for (int b = 0; b < p->NB; ++b) rates[b] = lo + (hi - lo) * rng_double(c);
    return;
  }
  double next[AIE_MAX_BRACKETS];
  saez_formula(c, next);
  for (int b = 0; b < p->NB; ++b) rates[b] = next[b];
  memcpy(SAEZ(c) + AIE_SAEZ_OFF_NEXT, next, sizeof(double) * (size_t)p->NB);
}
/* _update_saez_buffer :533-541 */
static void saez_update_buffer(ctx_t* c) {
  const aie_params* p = c->p;
  int32_t* hdr = (int32_t*)SAEZ(c);
  double* buf = (double*)(SAEZ(c) + AIE_SAEZ_OFF_BUF);
  for (int i = 0; i < p->n; ++i) {
    buf[2 * (hdr[0] + i)] = F64(c, o_tax_last_income)[i];
    buf[2 * (hdr[0] + i) + 1] = F64(c, o_tax_last_marginal_rate)[i];
  }
  hdr[0] += p->n;
  hdr[2] += p->n; /* _additions_this_episode :541 */
  if (hdr[0] > p->c.saez_buffer_size) {
    int drop = hdr[0] - p->c.saez_buffer_size;
    memmove(buf, buf + 2 * drop, sizeof(double) * 2 * (size_t)p->c.saez_buffer_size);
    hdr[0] = p->c.saez_buffer_size;
  }
}

/* component_step :945-972 (+ set_new_period_rates_model :419-434) */
static void tax_step(ctx_t* c) {
  const aie_params* p = c->p;
  int32_t* pos = I32(c, o_tax_cycle_pos);
  if (*pos == 1 && p->c.tax_model == AIE_TAX_MODEL_WRAPPER && !p->c.tax_disable) {
    for (int b = 0; b < p->NB; ++b) {
      int a = c->act_p[b];
      if (a > 0 && a <= p->c.tax_n_disc_rates) I32(c, o_tax_rate_idx)[b] = a - 1;
    }
  }
  if (*pos == 1 && p->c.tax_model == AIE_TAX_SAEZ) {
    saez_set_new_period_rates(c);
    for (int b = 0; b < p->NB; ++b) F64(c, o_tax_saez_obs_rates)[b] = tax_rate(c, b); /* _curr_rates_obs :959 */
  }
  if (*pos >= p->c.tax_period) {
    tax_enact(c);
    if (p->c.tax_model == AIE_TAX_SAEZ) saez_update_buffer(c);
    *pos = 0;
  }
  *pos += 1;
}

/* WealthRedistribution.component_step, F/components/redistribution.py:46-65: every agent's
 * inventory coin becomes (np.sum(inventory + escrow) / n) - its escrow. */
static void wealth_step(ctx_t* c) {
  const int n = c->p->n;
  double tot[AIE_MAX_AGENTS_WIDE];
  double *ic = F64(c, o_inv_coin), *ec = F64(c, o_esc_coin);
  for (int i = 0; i < n; ++i) tot[i] = ic[i] + ec[i];
  const double share = np_sum(tot, n) / n;
  for (int i = 0; i < n; ++i) ic[i] = share - ec[i];
}

/* ------------------------------------------------------------------------------- */
/* Scenario: F/scenarios/simple_wood_and_stone/layout_from_file.py                  */
/* ------------------------------------------------------------------------------- */
/* scenario_step :372-410, regen_halfwidth == 0 (1x1 kernel): p = w * max(map, src);
 * spawnable reduces to "is a source block"; rand(H,W) is drawn for Wood, then Stone. */
static void scenario_step(ctx_t* c) {
  const aie_params* p = c->p;
  uint8_t* health = NULL;
  for (int q = 0; q < 2; ++q) {
    int rsrc = q == 0 ? 1 : 0; /* ["Wood", "Stone"] */
    unsigned srcbit = rsrc ? AIE_CELL_WOOD_SRC : AIE_CELL_STONE_SRC;
    double w = p->c.regen_weight[rsrc];
    int mh = p->c.max_health[rsrc];
    const int hw = p->c.regen_halfwidth[rsrc], d = 1 + 2 * hw;
    const double kern = w / (double)(d * d); /* regen_weight * ones((d, d)) / d**2, dynamic_layout.py:446-449 */
    if (hw > 0) { /* the plane the reference convolves, BEFORE this resource's respawns: max(map, source blocks) */
      if (!health) health = (uint8_t*)malloc((size_t)p->HW);
      for (int cell = 0; cell < p->HW; ++cell) {
        int m = CB(c, cell, rsrc), src = (C_FLAGS(c, cell) & srcbit) ? 1 : 0;
        health[cell] = (uint8_t)(m > src ? m : src);
      }
    }
    for (int cell = 0; cell < p->HW; ++cell) {
      double u = rng_double(c);
      int m = CB(c, cell, rsrc), src = (C_FLAGS(c, cell) & srcbit) ? 1 : 0;
      int hl = m > src ? m : src;
      double prob = w * (double)hl;
      if (hw > 0) {
        /* signal.convolve2d(health, kernel, "same"): zero-filled window, one multiply-add per kernel element in
         * kernel row-major order, i.e. input rows r0+hw .. r0-hw, within a row columns c0+hw .. c0-hw
         * (scipy/signal/_firfilter.c pylab_convolve_2d; the order matters once max_health > 1 makes the terms
         * differ; checked bit for bit against scipy in tests/test_regen_neighbourhood.py) */
        const int r0 = cell / p->W, c0 = cell % p->W;
        prob = 0.0;
        for (int r = r0 + hw; r >= r0 - hw; --r)
          for (int cc = c0 + hw; cc >= c0 - hw; --cc) {
            if (r < 0 || r >= p->H || cc < 0 || cc >= p->W) continue; /* + kern * 0.0 */
            prob += kern * (double)health[r * p->W + cc];
          }
      }
      int respawn = (u < prob) && src > 0;
      int v = m + respawn;
      CB(c, cell, rsrc) = (uint8_t)(v < mh ? v : mh);
    }
  }
  free(health);
}

/* energy_weight :249-267 */
static double energy_weight(ctx_t* c) {
  const aie_params* p = c->p;
  if (p->c.energy_warmup_constant <= 0.0) return 1.0;
  if (p->c.energy_warmup_method == AIE_WARMUP_DECAY)
    return 1.0 - exp(-(double)(*I32(c, o_completions)) / p->c.energy_warmup_constant);
  return 1.0 - exp(-(double)(*I32(c, o_auto_warmup)) / p->c.energy_warmup_constant);
}

/* social_metrics.get_gini, F/scenarios/utils/social_metrics.py:10-46 */
static double get_gini(const double* e, int n) {
  if (n < 30) {
    double d[AIE_MAX_AGENTS * AIE_MAX_AGENTS];
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) d[i * n + j] = fabs(e[i] - e[j]);
    double diff = np_sum(d, n * n);
    double norm = 2 * n * np_sum(e, n);
    double unscaled = diff / (norm + 1e-10);
    return unscaled / ((double)(n - 1) / (double)n);
  }
  double s[AIE_MAX_AGENTS], cs[AIE_MAX_AGENTS];
  memcpy(s, e, sizeof(double) * n);
  for (int i = 1; i < n; ++i) { /* np.sort */
    double x = s[i]; int j = i - 1;
    while (j >= 0 && s[j] > x) { s[j + 1] = s[j]; --j; }
    s[j + 1] = x;
  }
  double tot = np_sum(s, n) + 1e-10, run = 0;
  for (int i = 0; i < n; ++i) { run += s[i]; cs[i] = run / tot; }
  return 1 - (2.0 / (n + 1)) * np_sum(cs, n);
}

/* same for up to AIE_MAX_AGENTS_WIDE agents (one-step-economy) */
static double get_gini_wide(const double* e, int n) {
  if (n < 30) return get_gini(e, n);
  double s[AIE_MAX_AGENTS_WIDE], cs[AIE_MAX_AGENTS_WIDE];
  memcpy(s, e, sizeof(double) * n);
  for (int i = 1; i < n; ++i) { /* np.sort */
    double x = s[i]; int j = i - 1;
    while (j >= 0 && s[j] > x) { s[j + 1] = s[j]; --j; }
    s[j + 1] = x;
  }
  double tot = np_sum(s, n) + 1e-10, run = 0;
  for (int i = 0; i < n; ++i) { run += s[i]; cs[i] = run / tot; }
  return 1 - (2.0 / (n + 1)) * np_sum(cs, n);
}

/* get_current_optimization_metrics :269-318 (rewards.py:12-48, 84-133) */
static void current_metrics(ctx_t* c, double* out /* n+1 */) {
  const aie_params* p = c->p;
  const int n = p->n;
  double coin[AIE_MAX_AGENTS];
  double lc = energy_weight(c) * p->c.energy_cost;
  double eta = p->c.isoelastic_eta;
  for (int i = 0; i < n; ++i) {
    coin[i] = F64(c, o_inv_coin)[i] + F64(c, o_esc_coin)[i];
    double util_c;
    if (eta == 1.0) util_c = log(coin[i] > 1 ? coin[i] : 1); /* rewards.py:37 ("dangerous") */
    else util_c = (pow(coin[i], 1 - eta) - 1) / (1 - eta);
    out[i] = util_c - F64(c, o_labor)[i] * lc;
  }
  if (p->c.planner_reward_type == AIE_PLANNER_REW_COIN_EQ_TIMES_PROD) {
    double ew = 1 - p->c.mixing_weight_gini_vs_coin;
    double prod = np_sum(coin, n) / n;
    double equality = ew * (1 - get_gini(coin, n)) + (1 - ew);
    out[n] = equality * prod;
  } else {
    double w[AIE_MAX_AGENTS], t[AIE_MAX_AGENTS];
    for (int i = 0; i < n; ++i) w[i] = 1 / (coin[i] > 1 ? coin[i] : 1);
    double sw = np_sum(w, n);
    for (int i = 0; i < n; ++i) {
      w[i] = w[i] / sw;
      t[i] = (p->c.planner_reward_type == AIE_PLANNER_REW_INV_INCOME_COIN ? coin[i] : out[i]) * w[i];
    }
    out[n] = np_sum(t, n);
  }
}

/* ------------------------------------------------------------------------------- */
/* Observations, masks, rewards                                                     */
/* ------------------------------------------------------------------------------- */
static double inv_scale(const aie_params* p) { return p->c.allow_observation_scaling ? 0.01 : 1.0; }

/* LayoutFromFile.generate_observations :412-517 + component obs + _package */
static void write_obs(ctx_t* c) {
  const aie_params* p = c->p;
  const int n = p->n, H = p->H, W = p->W, HW = p->HW, w = p->c.obs_range, WV = p->WV, CM = p->CM;
  const int e = c->e;
  const uint32_t* cells = CELLS(c);
  const int32_t *lr = I32(c, o_loc_r), *lc = I32(c, o_loc_c);
  const double isc = inv_scale(p);
  int16_t locmap[255 * 255];
  for (int k = 0; k < HW; ++k) locmap[k] = 0;
  for (int i = 0; i < n; ++i) locmap[lr[i] * W + lc[i]] = (int16_t)(i + 2); /* world.py:406-416, +2 */

  /* channel k of Maps.state, key order world.py:59-93 */
  #define FL(cell, bit) ((AIE_CELL_FLAGS(cells[cell]) & (bit)) ? 1 : 0)
  #define CHAN(k, cell) ((k) == 0 ? (int)AIE_CELL_STONE(cells[cell]) : (k) == 1 ? (int)AIE_CELL_WOOD(cells[cell]) : \
      (k) == 2 ? (AIE_CELL_OWNER(cells[cell]) >= 0) : \
      (p->c.has_water ? ((k) == 3 ? FL(cell, AIE_CELL_WATER) : (k) == 4 ? FL(cell, AIE_CELL_STONE_SRC) : FL(cell, AIE_CELL_WOOD_SRC)) \
                      : ((k) == 3 ? FL(cell, AIE_CELL_STONE_SRC) : FL(cell, AIE_CELL_WOOD_SRC))))
  #define OWN(cell) AIE_CELL_OWNER(cells[cell])

  float* amap = (float*)(c->arena + p->a_obs_a_map) + (int64_t)e * n * p->am_ch * p->am_h * p->am_w;
  int16_t* aidx = (int16_t*)(c->arena + p->a_obs_a_idx) + (int64_t)e * n * 2 * p->am_h * p->am_w;
  if (p->c.full_observability) { /* :466-472: the whole map + idx maps with the own id -> 1 */
    for (int i = 0; i < n; ++i) {
      for (int k = 0; k < CM; ++k)
        for (int cell = 0; cell < HW; ++cell) amap[(i * CM + k) * HW + cell] = (float)CHAN(k, cell);
      for (int cell = 0; cell < HW; ++cell) {
        int16_t v0 = (int16_t)(OWN(cell) >= 0 ? OWN(cell) + 2 : 0), v1 = locmap[cell];
        if (v0 == i + 2) v0 = 1;
        if (v1 == i + 2) v1 = 1;
        aidx[(i * 2 + 0) * HW + cell] = v0;
        aidx[(i * 2 + 1) * HW + cell] = v1;
      }
    }
  }
  for (int i = 0; i < n && !p->c.full_observability; ++i) {
    for (int dr = 0; dr < WV; ++dr)
      for (int dc = 0; dc < WV; ++dc) {
        int r = lr[i] - w + dr, col = lc[i] - w + dc;
        int in = (r >= 0 && r < H && col >= 0 && col < W);
        int cell = r * W + col;
        for (int k = 0; k < CM; ++k)
          amap[((i * (CM + 1) + k) * WV + dr) * WV + dc] = in ? (float)CHAN(k, cell) : 0.0f;
        amap[((i * (CM + 1) + CM) * WV + dr) * WV + dc] = in ? 1.0f : 0.0f; /* :480-485 */
        int16_t v0 = in ? (int16_t)(OWN(cell) >= 0 ? OWN(cell) + 2 : 0) : 0;
        int16_t v1 = in ? locmap[cell] : 0;
        if (v0 == i + 2) v0 = 1; /* :503 */
        if (v1 == i + 2) v1 = 1;
        aidx[((i * 2 + 0) * WV + dr) * WV + dc] = v0;
        aidx[((i * 2 + 1) * WV + dr) * WV + dc] = v1;
      }
  }
  if (p->c.planner_gets_spatial_info) {
    float* pmap = (float*)(c->arena + p->a_obs_p_map) + (int64_t)e * CM * HW;
    int16_t* pidx = (int16_t*)(c->arena + p->a_obs_p_idx) + (int64_t)e * 2 * HW;
    for (int k = 0; k < CM; ++k)
      for (int cell = 0; cell < HW; ++cell) pmap[k * HW + cell] = (float)CHAN(k, cell);
    for (int cell = 0; cell < HW; ++cell) {
      pidx[cell] = (int16_t)(OWN(cell) >= 0 ? OWN(cell) + 2 : 0);
      pidx[HW + cell] = locmap[cell];
    }
  }
  #undef CHAN
  #undef FL
  #undef OWN

  const int t = *I32(c, o_timestep);
  const double time_scale = p->c.allow_observation_scaling ? (double)p->c.episode_length : 1.0;
  const double tval = (double)t / time_scale;

  /* ---- CDA shared quantities (continuous_double_auction.py:491-542) ---- */
  double market_rate[2] = {0, 0};
  double net_ph[2][128];
  double full_asks[2][128], full_bids[2][128];
  if (p->has_cda) {
    for (int r = 0; r < 2; ++r) {
      for (int k = 0; k < p->P; ++k) {
        double s = 0; /* np.sum(np.stack(...), axis=0): row-by-row accumulation */
        double fa = 0, fb = 0;
        for (int i = 0; i < n; ++i) {
          double v = F64(c, o_cda_price_history)[(r * n + i) * p->P + k];
          s = (i == 0) ? v : s + v;
          fa += U8(c, o_cda_ask_hist)[(r * n + i) * p->P + k];
          fb += U8(c, o_cda_bid_hist)[(r * n + i) * p->P + k];
        }
        net_ph[r][k] = s; full_asks[r][k] = fa; full_bids[r][k] = fb;
      }
      double dot = 0;
      for (int k = 0; k < p->P; ++k) dot += (double)k * net_ph[r][k];
      double tot = np_sum(net_ph[r], p->P);
      market_rate[r] = dot / (tot > 0.001 ? tot : 0.001);
    }
  }
  /* ---- tax shared quantities (redistribution.py:974-1023) ---- */
  double is_tax_day = 0, is_first_day = 0, tax_phase = 0, sorted_inc[AIE_MAX_AGENTS];
  if (p->has_tax) {
    int pos = *I32(c, o_tax_cycle_pos);
    is_tax_day = pos >= p->c.tax_period ? 1.0 : 0.0;
    is_first_day = pos == 1 ? 1.0 : 0.0;
    tax_phase = (double)pos / (double)p->c.tax_period;
    for (int i = 0; i < n; ++i) sorted_inc[i] = F64(c, o_tax_last_income)[i] / (double)p->c.tax_period;
    for (int i = 1; i < n; ++i) {
      double x = sorted_inc[i]; int j = i - 1;
      while (j >= 0 && sorted_inc[j] > x) { sorted_inc[j + 1] = sorted_inc[j]; --j; }
      sorted_inc[j + 1] = x;
    }
  }

  float* aflat = (float*)(c->arena + p->a_obs_a_flat) + (int64_t)e * n * p->FA;
  float* atime = (float*)(c->arena + p->a_obs_a_time) + (int64_t)e * n;
  float* pag = (float*)(c->arena + p->a_obs_p_agents) + (int64_t)e * n * p->FPA;
  for (int i = 0; i < n; ++i) {
    float* f = aflat + i * p->FA;
    if (p->has_build) { /* build.py:163-178 */
      f[p->fa_build + 0] = (float)(F64(c, o_build_payment)[i] / (double)p->c.build_payment);
      f[p->fa_build + 1] = (float)F64(c, o_build_skill)[i];
    }
    if (p->has_cda) {
      float* g = f + p->fa_cda;
      const int P = p->P;
      for (int r = 0; r < 2; ++r)
        for (int k = 0; k < P; ++k) {
          double mya = U8(c, o_cda_ask_hist)[(r * n + i) * P + k];
          double myb = U8(c, o_cda_bid_hist)[(r * n + i) * P + k];
          g[0 * P + r * P + k] = (float)(full_asks[r][k] - mya);          /* available_asks */
          g[2 * P + r * P + k] = (float)(full_bids[r][k] - myb);          /* available_bids */
          g[4 * P + 2 + r * P + k] = (float)mya;                          /* my_asks        */
          g[6 * P + 2 + r * P + k] = (float)myb;                          /* my_bids        */
          g[8 * P + 2 + r * P + k] = (float)(net_ph[r][k] * isc);         /* price_history  */
        }
      g[4 * P + 0] = (float)market_rate[0];
      g[4 * P + 1] = (float)market_rate[1];
    }
    if (p->has_gather) f[p->fa_gather] = (float)F64(c, o_bonus_gather_prob)[i]; /* move.py:155-165 */
    double cmr = 0;
    if (p->has_tax) {
      float* g = f + p->fa_tax;
      for (int b = 0; b < p->NB; ++b) g[b] = (float)tax_rate_obs(c, b);
      g[p->NB + 0] = (float)is_first_day;
      g[p->NB + 1] = (float)is_tax_day;
      for (int k = 0; k < n; ++k) g[p->NB + 2 + k] = (float)sorted_inc[k];
      cmr = tax_marginal_rate(c, (F64(c, o_inv_coin)[i] + F64(c, o_esc_coin)[i]) - F64(c, o_tax_last_coin)[i]);
      g[p->NB + 2 + n] = (float)cmr;
      g[p->NB + 3 + n] = (float)tax_phase;
    }
    f[p->fa_time] = (float)tval;
    f[p->fa_world + 0] = (float)(F64(c, o_inv_coin)[i] * isc);
    f[p->fa_world + 1] = (float)((double)I32(c, o_inv_res)[0 * n + i] * isc);
    f[p->fa_world + 2] = (float)((double)I32(c, o_inv_res)[1 * n + i] * isc);
    if (!p->c.full_observability) {
      f[p->fa_world + 3] = (float)((double)lc[i] / (double)W);
      f[p->fa_world + 4] = (float)((double)lr[i] / (double)H);
    }
    atime[i] = (float)tval;
    /* planner's per-agent view p{i} */
    float* q = pag + i * p->FPA;
    if (p->has_tax) {
      q[p->fpa_tax + 0] = (float)cmr;
      q[p->fpa_tax + 1] = (float)(F64(c, o_tax_last_income)[i] / (double)p->c.tax_period);
      q[p->fpa_tax + 2] = (float)F64(c, o_tax_last_marginal_rate)[i];
    }
    if (!p->c.full_observability) { /* :508-515: only the egocentric branch builds "p<idx>" */
      q[p->fpa_world + 0] = f[p->fa_world + 0];
      q[p->fpa_world + 1] = f[p->fa_world + 1];
      q[p->fpa_world + 2] = f[p->fa_world + 2];
      if (p->c.planner_gets_spatial_info) {
        q[p->fpa_world + 3] = f[p->fa_world + 3];
        q[p->fpa_world + 4] = f[p->fa_world + 4];
      }
    }
  }
  /* planner flat */
  float* pf = (float*)(c->arena + p->a_obs_p_flat) + (int64_t)e * p->FP;
  if (p->has_cda) {
    float* g = pf + p->fp_cda;
    const int P = p->P;
    for (int r = 0; r < 2; ++r)
      for (int k = 0; k < P; ++k) {
        g[0 * P + r * P + k] = (float)full_asks[r][k];
        g[2 * P + r * P + k] = (float)full_bids[r][k];
        g[4 * P + 2 + r * P + k] = (float)(net_ph[r][k] * isc);
      }
    g[4 * P + 0] = (float)market_rate[0];
    g[4 * P + 1] = (float)market_rate[1];
  }
  if (p->has_tax) {
    float* g = pf + p->fp_tax;
    for (int b = 0; b < p->NB; ++b) g[b] = (float)tax_rate_obs(c, b);
    g[p->NB + 0] = (float)is_first_day;
    g[p->NB + 1] = (float)is_tax_day;
    for (int k = 0; k < n; ++k) g[p->NB + 2 + k] = (float)sorted_inc[k];
    g[p->NB + 2 + n] = (float)tax_phase;
  }
  pf[p->fp_time] = (float)tval;
  pf[p->fp_world + 0] = 0.0f; /* the planner's own inventory is always empty */
  pf[p->fp_world + 1] = 0.0f;
  pf[p->fp_world + 2] = 0.0f;
  ((float*)(c->arena + p->a_obs_p_time))[e] = (float)tval;
}

/* _generate_masks base_env.py:706-756 + flatten_masks base_agent.py:440-460 */
static void write_masks(ctx_t* c) {
  const aie_params* p = c->p;
  const int n = p->n, e = c->e;
  const int32_t *lr = I32(c, o_loc_r), *lc = I32(c, o_loc_c);
  float* am = (float*)(c->arena + p->a_obs_a_mask) + (int64_t)e * n * p->MA;
  const int multi = p->c.multi_action_mode_agents;
  for (int i = 0; i < n; ++i) {
    float* m = am + i * p->MA;
    int o = 0;
    if (!multi || p->n_sub_a == 0) m[o++] = 1.0f;
    for (int s = 0; s < p->n_sub_a; ++s) {
      if (multi) m[o++] = 1.0f;
      int slot = p->sub_a_slot[s];
      if (slot == AIE_SUB_BUILD) {
        m[o++] = agent_can_build(c, i) ? 1.0f : 0.0f; /* build.py:180-193 */
      } else if (slot == AIE_SUB_GATHER) {
        /* move.py:167-188: L, R, U, D neighbours free and accessible */
        static const int ro[4] = {0, 0, -1, 1}, co[4] = {-1, 1, 0, 0};
        for (int k = 0; k < 4; ++k) m[o++] = can_agent_occupy(c, lr[i] + ro[k], lc[i] + co[k], i) ? 1.0f : 0.0f;
      } else {
        /* continuous_double_auction.py:544-580 */
        int r = (slot == AIE_SUB_BUY1 || slot == AIE_SUB_SELL1) ? 1 : 0;
        int is_buy = (slot == AIE_SUB_BUY0 || slot == AIE_SUB_BUY1);
        int quota = I32(c, o_cda_n_orders)[r * n + i] < p->c.cda_max_num_orders;
        for (int k = 0; k < p->P; ++k) {
          float v;
          if (is_buy) v = (quota && (double)k <= F64(c, o_inv_coin)[i]) ? 1.0f : 0.0f;
          else v = (quota && I32(c, o_inv_res)[r * n + i] > 0) ? 1.0f : 0.0f;
          m[o++] = v;
        }
      }
    }
  }
  /* planner: redistribution.py:1025-1104 */
  float* pm = (float*)(c->arena + p->a_obs_p_mask) + (int64_t)e * p->MP;
  int o = 0;
  const int pmulti = p->c.multi_action_mode_planner;
  if (!pmulti || p->n_sub_p == 0) pm[o++] = 1.0f;
  if (p->n_sub_p) {
    float v = (*I32(c, o_tax_cycle_pos) == 1) ? 1.0f : 0.0f;
    for (int b = 0; b < p->n_sub_p; ++b) {
      if (pmulti) pm[o++] = 1.0f;
      for (int k = 0; k < p->sub_p_dim; ++k) pm[o++] = v * tax_rate_action_mask(c, k);
    }
  }
}

/* compute_reward layout_from_file.py:519-559 */
static void write_rewards(ctx_t* c) {
  const aie_params* p = c->p;
  const int n = p->n, e = c->e;
  double cur[AIE_MAX_AGENTS + 1], rew[AIE_MAX_AGENTS + 1];
  double* util = F64(c, o_util);
  current_metrics(c, cur);
  for (int i = 0; i <= n; ++i) { rew[i] = cur[i] - util[i]; util[i] = cur[i]; }
  double avg = np_sum(rew, n) / n;
  if (avg > 0) *I32(c, o_auto_warmup) += 1;
  float* ra = (float*)(c->arena + p->a_rew_a) + (int64_t)e * n;
  for (int i = 0; i < n; ++i) ra[i] = (float)rew[i];
  ((float*)(c->arena + p->a_rew_p))[e] = (float)rew[n];
}

/* parse_actions base_env.py:552-556 -> base_agent.py:407-438 */
static void decode_actions(ctx_t* c, const int32_t* aa, const int32_t* ap) {
  const aie_params* p = c->p;
  memset(c->act, 0, sizeof(c->act));
  memset(c->act_p, 0, sizeof(c->act_p));
  if (aa) {
    for (int i = 0; i < p->n; ++i) {
      const int32_t* a = aa + ((int64_t)c->e * p->n + i) * p->act_a_width;
      if (p->c.multi_action_mode_agents) {
        for (int s = 0; s < p->n_sub_a; ++s)
          if (a[s] >= 0 && a[s] <= p->sub_a_dim[s]) c->act[i][p->sub_a_slot[s]] = a[s];
      } else {
        int v = a[0];
        for (int s = 0; s < p->n_sub_a; ++s)
          if (v >= p->sub_a_base[s] && v < p->sub_a_base[s] + p->sub_a_dim[s])
            c->act[i][p->sub_a_slot[s]] = v - p->sub_a_base[s] + 1;
      }
    }
  }
  if (ap && p->n_sub_p) {
    const int32_t* a = ap + (int64_t)c->e * p->act_p_width;
    if (p->c.multi_action_mode_planner) {
      for (int b = 0; b < p->n_sub_p; ++b) c->act_p[b] = a[b];
    } else {
      int v = a[0];
      if (v >= 1 && v < 1 + p->n_sub_p * p->sub_p_dim) {
        int b = (v - 1) / p->sub_p_dim;
        c->act_p[b] = (v - 1) % p->sub_p_dim + 1;
      }
    }
  }
}

static void make_ctx(ctx_t* c, const aie_params* p, uint8_t* arena, int e) {
  c->p = p;
  c->arena = arena;
  c->e = e;
  c->rec = arena + p->a_records + (int64_t)e * p->rec_bytes;
}

/* BaseEnvironment.step, base_env.py:929-1032 */
static void step_one(const aie_params* p, uint8_t* arena, int e, const int32_t* aa, const int32_t* ap) {
  ctx_t c;
  make_ctx(&c, p, arena, e);
  decode_actions(&c, aa, ap);
  *I32(&c, o_timestep) += 1;
  if (EV(&c)) EV(&c)[0] = 0;
  for (int k = 0; k < p->c.n_components; ++k) {
    switch (p->c.components[k]) {
      case AIE_COMP_BUILD: build_step(&c); break;
      case AIE_COMP_CDA: cda_step(&c); break;
      case AIE_COMP_GATHER: gather_step(&c); break;
      case AIE_COMP_TAX: tax_step(&c); break;
      case AIE_COMP_WEALTH_REDISTRIBUTION: wealth_step(&c); break;
    }
  }
  scenario_step(&c);
  write_obs(&c);
  *I32(&c, o_obs_valid) = 1; /* device bookkeeping (incremental map observations); always full here */
  write_masks(&c);
  write_rewards(&c);
  int done = *I32(&c, o_timestep) >= p->c.episode_length;
  (arena + p->a_done)[e] = (uint8_t)done;
  if (done) *I32(&c, o_completions) += 1;
}

/* BaseEnvironment.reset, base_env.py:852-927, with LayoutFromFile
 * reset_starting_layout/reset_agent_states/additional_reset_steps
 * (layout_from_file.py:323-370, 564-593) and the component resets
 * (build.py:224-254, move.py:193-210, continuous_double_auction.py:643-668,
 * redistribution.py:1109-1139). */
/* Source layouts drawn at reset: Uniform.reset_starting_layout (dynamic_layout.py:313-392), with MultiZone's
 * per-reset zone shuffle (:778-872) and Quadrant's empty water lines (:992-1024).  NumPy / SciPy primitives restated:
 * rand / randn (legacy gauss with its cache) in row-major order, np.mean of a 0/1 plane = count / size,
 * signal.convolve2d(x, kernel, "same") = one multiply-add per kernel element in kernel row-major order over the
 * zero-filled window (checked against scipy bit for bit in tests/test_regen_neighbourhood.py). */
static void layout_generate(ctx_t* c) {
  const aie_params* p = c->p;
  const aie_config* g = &p->c;
  const int H = p->H, W = p->W, HW = p->HW;
  const double* shared_prob = (const double*)(c->arena + p->a_layout_prob); /* [2][HW]: Stone, Wood */
  double* tmp = (double*)malloc(sizeof(double) * 2 * (size_t)HW);
  double* x = tmp + HW;
  uint8_t* maybe[2];
  maybe[0] = (uint8_t*)malloc(2 * (size_t)HW);
  maybe[1] = maybe[0] + HW;
  /* multi_zone: which zone type each region is, re-drawn now (np.random.shuffle of the flat grid) */
  int grid[256];
  double mz_scale[2] = {0, 0};
  int size_r = 1, size_c = 1;
  if (g->layout_gen == AIE_LAYOUT_MULTI_ZONE) {
    const int regions = g->mz_rows * g->mz_cols;
    int k = 0;
    for (int z = 0; z < 3; ++z)
      for (int q = 0; q < g->mz_zones[z]; ++q) grid[k++] = z; /* np.repeat([0, 1, 2], counts): Wood, Stone, both */
    while (k < regions) grid[k++] = -1;
    for (int i = regions - 1; i >= 1; --i) {
      int j = (int)rng_interval(c, (uint32_t)i);
      int t = grid[i]; grid[i] = grid[j]; grid[j] = t;
    }
    size_r = (H + g->mz_rows - 1) / g->mz_rows;
    size_c = (W + g->mz_cols - 1) / g->mz_cols;
    for (int rs = 0; rs < 2; ++rs) { /* prob / np.mean(prob) * Wood's coverage (:846-863) */
      const int own = rs == 1 ? 0 : 1; /* zone index: Wood 0, Stone 1, WoodStone 2 */
      int cnt = 0;
      for (int cell = 0; cell < HW; ++cell) {
        int z = grid[(cell / W / size_r) * g->mz_cols + (cell % W) / size_c];
        cnt += (z == own || z == 2);
      }
      const double mean = (double)cnt / (double)HW;
      mz_scale[rs] = (1.0 / mean) * g->layout_coverage[1];
    }
  }
  int happy = 0;
  for (int tries = 0; tries < 100 && !happy; ++tries) {
    for (int q = 0; q < 2; ++q) {
      const int rs = q == 0 ? 1 : 0; /* ["Wood", "Stone"] */
      const double cov = g->layout_coverage[rs], clump = g->layout_clump[rs];
      uint8_t* mb = maybe[rs];
      const uint8_t* other = q == 0 ? NULL : maybe[1]; /* empty = nothing placed yet on the tile */
#define AIE_SP(cell) ((g->layout_gen == AIE_LAYOUT_MULTI_ZONE                                                    \
                           ? (((grid[((cell) / W / size_r) * g->mz_cols + ((cell) % W) / size_c] == (rs == 1 ? 0 : 1)) || \
                               (grid[((cell) / W / size_r) * g->mz_cols + ((cell) % W) / size_c] == 2))           \
                                  ? mz_scale[rs] : 0.0 * g->layout_coverage[1])                                   \
                           : shared_prob[rs * HW + (cell)]) * 0.1 * clump)
      for (int cell = 0; cell < HW; ++cell) tmp[cell] = rng_double(c);
      int count = 0;
      for (int cell = 0; cell < HW; ++cell) {
        mb[cell] = (uint8_t)((tmp[cell] < AIE_SP(cell)) && !(other && other[cell]));
        count += mb[cell];
      }
      int n_tries = 0;
      while ((double)count / (double)HW < cov * clump) {
        count = 0;
        for (int cell = 0; cell < HW; ++cell) {
          tmp[cell] *= 0.9;
          mb[cell] = (uint8_t)((tmp[cell] < AIE_SP(cell)) && !(other && other[cell]));
          count += mb[cell];
        }
        if (++n_tries > 200) break;
      }
      while ((double)count / (double)HW < cov) {
        uint8_t kern[49];
        for (int k = 0; k < 49; ++k) kern[k] = rng_gauss(c) > 0;
        for (int cell = 0; cell < HW; ++cell) x[cell] = ((double)mb[cell] + (0.2 * rng_gauss(c))) - 0.25;
        count = 0;
        for (int cell = 0; cell < HW; ++cell) tmp[cell] = (double)mb[cell]; /* old `maybe`, while mb is rewritten */
        for (int m = 0; m < H; ++m)
          for (int n2 = 0; n2 < W; ++n2) {
            double sum = 0.0;
            for (int j = 0; j < 7; ++j)
              for (int k = 0; k < 7; ++k) {
                const int i0 = m + 3 - j, i1 = n2 + 3 - k;
                if (i0 >= 0 && i0 < H && i1 >= 0 && i1 < W) sum += (double)kern[j * 7 + k] * x[i0 * W + i1];
              }
            const int cell = m * W + n2;
            mb[cell] = (uint8_t)(((sum > 0) || tmp[cell] != 0.0) && !(other && other[cell]));
            count += mb[cell];
          }
      }
#undef AIE_SP
    }
    happy = 1;
    for (int q = 0; q < 2; ++q) {
      const int rs = q == 0 ? 1 : 0;
      int count = 0;
      for (int cell = 0; cell < HW; ++cell) count += maybe[rs][cell];
      const double ratio = ((double)count / (double)HW) / g->layout_coverage[rs];
      if (!((1 / 1.4) <= ratio && ratio <= 1.4)) happy = 0;
    }
  }
  for (int cell = 0; cell < HW; ++cell) {
    const int r = cell / W, col = cell % W;
    int st = maybe[0][cell], wd = maybe[1][cell];
    if (g->layout_checker && ((r % 2) + (col % 2)) != 1) st = wd = 0;
    if (g->layout_gen == AIE_LAYOUT_QUADRANT && (col == H / 2 || r == W / 2)) st = wd = 0; /* nothing on the water lines */
    C_FLAGS(c, cell) = (uint8_t)((C_FLAGS(c, cell) & AIE_CELL_WATER) | (st ? AIE_CELL_STONE_SRC : 0) | (wd ? AIE_CELL_WOOD_SRC : 0));
  }
  free(tmp);
  free(maybe[0]);
}

static void reset_one(const aie_params* p, uint8_t* arena, int e) {
  ctx_t c;
  make_ctx(&c, p, arena, e);
  const int n = p->n, HW = p->HW;
  *I32(&c, o_timestep) = 0;
  memset(MET(&c), 0, (size_t)p->met_bytes); /* component resets clear their episode logs */
  if (EV(&c)) EV(&c)[0] = 0;
  if (p->c.layout_gen != AIE_LAYOUT_FIXED) {
    if (aie__layout_staged(&p->c)) {
      /* the counter-stream mode: the k-th reset's layout comes from a stream of its own (aie_layout.h: aie_layout_stream;
       * the device may have drawn it ahead of the reset), the replica's stream and its Gauss cache stay as they are */
      uint32_t* st = (uint32_t*)(c.rec + p->o_mt);
      uint32_t keep[4], ks[2];
      memcpy(keep, st, sizeof(keep));
      const int32_t keep_pos = *I32(&c, o_mt_pos), keep_has = *I32(&c, o_mt_has_gauss);
      const double keep_gauss = *F64(&c, o_mt_gauss);
      aie_layout_stream(keep, ks);
      st[0] = ks[0]; st[1] = 0xffffffffu; st[2] = ks[1];
      *I32(&c, o_mt_pos) = 624;
      *I32(&c, o_mt_has_gauss) = 0;
      *F64(&c, o_mt_gauss) = 0.0;
      layout_generate(&c);
      memcpy(st, keep, sizeof(keep));
      st[3] = keep[3] + 1u;
      *I32(&c, o_mt_pos) = keep_pos;
      *I32(&c, o_mt_has_gauss) = keep_has;
      *F64(&c, o_mt_gauss) = keep_gauss;
    } else {
      layout_generate(&c); /* a fresh source layout from this replica's stream */
    }
  }
  for (int cell = 0; cell < HW; ++cell) { /* layout_from_file.py:323-334 */
    unsigned fl = C_FLAGS(&c, cell);
    CELLS(&c)[cell] = AIE_CELL_PACK((fl & AIE_CELL_STONE_SRC) ? 1 : 0, (fl & AIE_CELL_WOOD_SRC) ? 1 : 0, -1, fl);
  }
  if (p->regen_conv) /* device bookkeeping: source blocks per regen window (aie_layout.h: regen_conv) */
    for (int rs = 0; rs < AIE_N_RES; ++rs)
      for (int cell = 0; cell < HW; ++cell) {
        const int hw = p->c.regen_halfwidth[rs], r0 = cell / p->W, c0 = cell % p->W;
        int cnt = 0;
        for (int r = r0 - hw; r <= r0 + hw; ++r)
          for (int cc = c0 - hw; cc <= c0 + hw; ++cc)
            if (r >= 0 && r < p->H && cc >= 0 && cc < p->W &&
                (C_FLAGS(&c, r * p->W + cc) & (rs ? AIE_CELL_WOOD_SRC : AIE_CELL_STONE_SRC)))
              cnt++;
        U8(&c, o_regen_count)[rs * HW + cell] = (uint8_t)cnt;
      }
  for (int i = 0; i < n; ++i) {
    I32(&c, o_inv_res)[i] = I32(&c, o_inv_res)[n + i] = 0;
    I32(&c, o_esc_res)[i] = I32(&c, o_esc_res)[n + i] = 0;
    F64(&c, o_inv_coin)[i] = p->c.starting_agent_coin;
    F64(&c, o_esc_coin)[i] = 0;
    F64(&c, o_labor)[i] = 0;
    I32(&c, o_loc_r)[i] = -1;
    I32(&c, o_loc_c)[i] = -1;
  }
  int place_order[AIE_MAX_AGENTS];
  if (p->c.reset_random_order) rng_permutation(&c, n, place_order); /* dynamic_layout.py:420 */
  else for (int i = 0; i < n; ++i) place_order[i] = i;
  for (int k = 0; k < n; ++k) {
    const int i = place_order[k];
    int r = rng_randint(&c, p->H), col = rng_randint(&c, p->W), tries = 0;
    while (!can_agent_occupy(&c, r, col, i)) {
      r = rng_randint(&c, p->H);
      col = rng_randint(&c, p->W);
      if (++tries > 200) break; /* reference raises TimeoutError */
    }
    I32(&c, o_loc_r)[i] = r;
    I32(&c, o_loc_c)[i] = col;
  }
  for (int k = 0; k < p->c.n_components; ++k) {
    switch (p->c.components[k]) {
      case AIE_COMP_BUILD:
        for (int i = 0; i < n; ++i) {
          double skill = 1, pay = 1, pm = (double)p->c.build_payment_max_skill_multiplier;
          if (p->c.build_skill_dist == AIE_SKILL_PARETO) {
            skill = rng_pareto(&c, 4.0);
            pay = (pm - 1) * skill + 1; if (pm < pay) pay = pm;
          } else if (p->c.build_skill_dist == AIE_SKILL_LOGNORMAL) {
            skill = rng_lognormal(&c, -1.0, 0.5);
            pay = (pm - 1) * skill + 1; if (pm < pay) pay = pm;
          }
          F64(&c, o_build_payment)[i] = pay * (double)p->c.build_payment;
          F64(&c, o_build_skill)[i] = skill;
        }
        break;
      case AIE_COMP_GATHER:
        for (int i = 0; i < n; ++i) {
          double b = 0.0;
          if (p->c.gather_skill_dist == AIE_SKILL_PARETO) { b = rng_pareto(&c, 3.0); b = (b < 2 ? b : 2) / 2; }
          else if (p->c.gather_skill_dist == AIE_SKILL_LOGNORMAL) { b = rng_lognormal(&c, -2.022, 0.938); b = (b < 2 ? b : 2) / 2; }
          F64(&c, o_bonus_gather_prob)[i] = b;
        }
        break;
      case AIE_COMP_CDA:
        memset(I32(&c, o_cda_n_bids), 0, 8);
        memset(I32(&c, o_cda_n_asks), 0, 8);
        memset(I32(&c, o_cda_bids), 0, 4 * 2 * p->M);
        memset(I32(&c, o_cda_asks), 0, 4 * 2 * p->M);
        memset(I32(&c, o_cda_n_orders), 0, 4 * 2 * n);
        memset(U8(&c, o_cda_bid_hist), 0, 2 * n * p->P);
        memset(U8(&c, o_cda_ask_hist), 0, 2 * n * p->P);
        memset(F64(&c, o_cda_price_history), 0, 8 * 2 * n * p->P);
        break;
      case AIE_COMP_TAX:
        for (int b = 0; b < p->NB; ++b) I32(&c, o_tax_rate_idx)[b] = 0;
        *I32(&c, o_tax_cycle_pos) = 1;
        for (int i = 0; i < n; ++i) {
          F64(&c, o_tax_last_coin)[i] = F64(&c, o_inv_coin)[i] + F64(&c, o_esc_coin)[i];
          F64(&c, o_tax_last_income)[i] = 0;
          F64(&c, o_tax_last_marginal_rate)[i] = 0;
        }
        *F64(&c, o_tax_total_collected) = 0;
        if (p->c.tax_model == AIE_TAX_SAEZ) { /* _curr_rates_obs first (:1123), with the previous episode's rates */
          for (int b = 0; b < p->NB; ++b) F64(&c, o_tax_saez_obs_rates)[b] = tax_rate(&c, b);
        }
        if (p->c.tax_model == AIE_TAX_SAEZ) /* curr_bracket_tax_rates = running_avg_tax_rates :1136-1137 */
          memcpy(F64(&c, o_tax_saez_rates), SAEZ(&c) + AIE_SAEZ_OFF_AVG, sizeof(double) * (size_t)p->NB);
        break;
    }
  }
  if (!p->has_build) for (int i = 0; i < n; ++i) { F64(&c, o_build_payment)[i] = 0; F64(&c, o_build_skill)[i] = 0; }
  if (!p->has_gather) for (int i = 0; i < n; ++i) F64(&c, o_bonus_gather_prob)[i] = 0;
  if (p->c.fixed_four_skill_and_loc) { /* layout_from_file.py:582-586 */
    int order[AIE_MAX_AGENTS];
    for (int i = 0; i < n; ++i) { I32(&c, o_loc_r)[i] = -1; I32(&c, o_loc_c)[i] = -1; }
    rng_permutation(&c, n, order);
    for (int k = 0; k < n; ++k) {
      int i = order[k];
      int r = p->c.ranked_locs[k][0], col = p->c.ranked_locs[k][1];
      if (can_agent_occupy(&c, r, col, i)) { I32(&c, o_loc_r)[i] = r; I32(&c, o_loc_c)[i] = col; }
      F64(&c, o_build_payment)[i] = p->c.avg_ranked_skill[k];
    }
  }
  if (p->c.split_water_line > 0) { /* SplitLayout.additional_reset_steps layout_from_file.py:759-793 */
    int order[AIE_MAX_AGENTS];
    const int wl = p->c.split_water_line;
    for (int i = 0; i < n; ++i) { I32(&c, o_loc_r)[i] = -1; I32(&c, o_loc_c)[i] = -1; }
    rng_permutation(&c, n, order);
    for (int k = 0; k < n; ++k) {
      int i = order[k];
      F64(&c, o_build_payment)[i] = p->c.avg_ranked_skill[k];
      int top = (p->c.split_top_ranks[k >> 5] >> (k & 31)) & 1u;
      int r_min = top ? 0 : wl + 1, r_max = top ? wl : p->H;
      int r = r_min + rng_randint(&c, r_max - r_min), col = rng_randint(&c, p->W), tries = 0;
      while (!can_agent_occupy(&c, r, col, i)) {
        r = r_min + rng_randint(&c, r_max - r_min);
        col = rng_randint(&c, p->W);
        if (++tries > 200) break; /* the reference raises TimeoutError */
      }
      I32(&c, o_loc_r)[i] = r;
      I32(&c, o_loc_c)[i] = col;
    }
  }
  current_metrics(&c, F64(&c, o_util));
  write_obs(&c);
  *I32(&c, o_obs_valid) = 1;
  if (p->has_tax && p->c.tax_annealing) *I32(&c, o_tax_last_completions) = *I32(&c, o_completions); /* generate_masks :1036-1046 */
  write_masks(&c);
  float* ra = (float*)(arena + p->a_rew_a) + (int64_t)e * n;
  for (int i = 0; i < n; ++i) ra[i] = 0;
  ((float*)(arena + p->a_rew_p))[e] = 0;
  (arena + p->a_done)[e] = 0;
}

/* =================================================================================== */
/* one-step-economy + SimpleLabor (BASELINE configs[4])                                 */
/* F/scenarios/one_step_economy/one_step_economy.py, F/components/simple_labor.py       */
/* =================================================================================== */
/* SimpleLabor.component_step simple_labor.py:105-126.  The random agent order is drawn
 * (world.py:418-422) although the result does not depend on it. */
static void labor_step(ctx_t* c) {
  const aie_params* p = c->p;
  int order[AIE_MAX_AGENTS_WIDE];
  for (int i = 0; i < p->n; ++i) order[i] = i;
  for (int i = p->n - 1; i >= 1; --i) {
    int j = (int)rng_interval(c, (uint32_t)i);
    int t = order[i]; order[i] = order[j]; order[j] = t;
  }
  for (int k = 0; k < p->n; ++k) {
    int i = order[k];
    int a = c->act_wide[i];
    if (a == 0) continue;
    F64(c, o_labor)[i] = (double)a; /* hours worked this step (set, not accumulated) */
    double payoff = (double)a * F64(c, o_skill)[i];
    F64(c, o_production)[i] += payoff;
    F64(c, o_inv_coin)[i] += payoff;
  }
}

/* rewards.coin_minus_labor_cost rewards.py:51-81 / isoelastic :12-48; planner SWFs
 * one_step_economy.py:300-336 (pretax incomes = production for the inverse-income SWF) */
static void ose_metrics(ctx_t* c, double* out) {
  const aie_params* p = c->p;
  const int n = p->n;
  double coin[AIE_MAX_AGENTS_WIDE];
  for (int i = 0; i < n; ++i) {
    coin[i] = F64(c, o_inv_coin)[i] + F64(c, o_esc_coin)[i];
    double labor = F64(c, o_labor)[i];
    if (p->c.ose_agent_reward_type == AIE_AGENT_REW_ISOELASTIC) {
      double eta = p->c.isoelastic_eta;
      double uc = (eta == 1.0) ? log(coin[i] > 1 ? coin[i] : 1) : (pow(coin[i], 1 - eta) - 1) / (1 - eta);
      out[i] = uc - labor * p->c.ose_labor_cost;
    } else {
      out[i] = coin[i] - pow(labor, p->c.ose_labor_exponent) * p->c.ose_labor_cost;
    }
  }
  if (p->c.planner_reward_type == AIE_PLANNER_REW_COIN_EQ_TIMES_PROD) {
    double ew = 1 - p->c.mixing_weight_gini_vs_coin;
    double prod = np_sum(coin, n) / n;
    out[n] = (ew * (1 - get_gini_wide(coin, n)) + (1 - ew)) * prod;
  } else {
    double w[AIE_MAX_AGENTS_WIDE], t[AIE_MAX_AGENTS_WIDE];
    const int use_util = p->c.planner_reward_type == AIE_PLANNER_REW_INV_INCOME_UTIL;
    for (int i = 0; i < n; ++i) {
      /* inv_income_weighted_utility is fed pretax incomes (production); the coin variant
       * is fed coin endowments (one_step_economy.py:317-333) */
      double base = use_util ? F64(c, o_production)[i] : coin[i];
      w[i] = 1 / (base > 1 ? base : 1);
    }
    double sw = np_sum(w, n);
    for (int i = 0; i < n; ++i) {
      w[i] = w[i] / sw;
      t[i] = (use_util ? out[i] : coin[i]) * w[i];
    }
    out[n] = np_sum(t, n);
  }
}

static void ose_write_obs(ctx_t* c, int at_reset) {
  const aie_params* p = c->p;
  const int n = p->n, e = c->e, NB = p->NB;
  const int t = *I32(c, o_timestep);
  const double time_scale = p->c.allow_observation_scaling ? (double)p->c.episode_length : 1.0;
  const float tval = (float)((double)t / time_scale);
  double is_tax_day = 0, is_first_day = 0, tax_phase = 0, sorted_inc[AIE_MAX_AGENTS_WIDE];
  if (p->has_tax) {
    int pos = *I32(c, o_tax_cycle_pos);
    is_tax_day = pos >= p->c.tax_period ? 1.0 : 0.0;
    is_first_day = pos == 1 ? 1.0 : 0.0;
    tax_phase = (double)pos / (double)p->c.tax_period;
    for (int i = 0; i < n; ++i) sorted_inc[i] = F64(c, o_tax_last_income)[i] / (double)p->c.tax_period;
    for (int i = 1; i < n; ++i) {
      double x = sorted_inc[i]; int j = i - 1;
      while (j >= 0 && sorted_inc[j] > x) { sorted_inc[j + 1] = sorted_inc[j]; --j; }
      sorted_inc[j + 1] = x;
    }
  }
  float* aflat = (float*)(c->arena + p->a_obs_a_flat) + (int64_t)e * n * p->FA;
  float* atime = (float*)(c->arena + p->a_obs_a_time) + (int64_t)e * n;
  float* pag = (float*)(c->arena + p->a_obs_p_agents) + (int64_t)e * n * p->FPA;
  double coin[AIE_MAX_AGENTS_WIDE];
  for (int i = 0; i < n; ++i) {
    float* f = aflat + i * p->FA;
    coin[i] = F64(c, o_inv_coin)[i] + F64(c, o_esc_coin)[i];
    if (p->has_tax) {
      float* g = f + p->fa_tax;
      for (int b = 0; b < NB; ++b) g[b] = (float)tax_rate_obs(c, b);
      g[NB + 0] = (float)is_first_day;
      g[NB + 1] = (float)is_tax_day;
      for (int k = 0; k < n; ++k) g[NB + 2 + k] = (float)sorted_inc[k];
      double cmr = tax_marginal_rate(c, coin[i] - F64(c, o_tax_last_coin)[i]);
      g[NB + 2 + n] = (float)cmr;
      g[NB + 3 + n] = (float)tax_phase;
      float* q = pag + i * p->FPA;
      q[0] = (float)cmr;
      q[1] = (float)(F64(c, o_tax_last_income)[i] / (double)p->c.tax_period);
      q[2] = (float)F64(c, o_tax_last_marginal_rate)[i];
    }
    if (p->has_labor) f[p->fa_labor] = (float)(F64(c, o_skill)[i] / p->c.labor_pmsm); /* simple_labor.py:128-134 */
    f[p->fa_time] = tval;
    atime[i] = tval;
  }
  float* pf = (float*)(c->arena + p->a_obs_p_flat) + (int64_t)e * p->FP;
  if (p->has_tax) {
    float* g = pf + p->fp_tax;
    for (int b = 0; b < NB; ++b) g[b] = (float)tax_rate_obs(c, b);
    g[NB + 0] = (float)is_first_day;
    g[NB + 1] = (float)is_tax_day;
    for (int k = 0; k < n; ++k) g[NB + 2 + k] = (float)sorted_inc[k];
    g[NB + 2 + n] = (float)tax_phase;
  }
  pf[p->fp_time] = tval;
  /* one_step_economy.py:161-172: equality, productivity / n / 1000 */
  pf[p->fp_world + 0] = (float)(1 - get_gini_wide(coin, n));
  pf[p->fp_world + 1] = (float)(np_sum(coin, n) / n / 1000);
  ((float*)(c->arena + p->a_obs_p_time))[e] = tval;

  /* masks: SimpleLabor.generate_masks simple_labor.py:97-103 (all-off on the first call
   * after a reset when mask_first_step), planner tax masks redistribution.py:1025-1104 */
  float* am = (float*)(c->arena + p->a_obs_a_mask) + (int64_t)e * n * p->MA;
  const int multi = p->c.multi_action_mode_agents;
  float on = 1.0f;
  if (p->has_labor) {
    int32_t* first = I32(c, o_first_step);
    if (*first) { *first = 0; if (p->c.labor_mask_first_step) on = 0.0f; }
  }
  for (int i = 0; i < n; ++i) {
    float* m = am + i * p->MA;
    int o = 0;
    if (!multi || p->n_sub_a == 0) m[o++] = 1.0f;
    for (int s = 0; s < p->n_sub_a; ++s) {
      if (multi) m[o++] = 1.0f;
      for (int k = 0; k < p->sub_a_dim[s]; ++k) m[o++] = on;
    }
  }
  if (at_reset && p->has_tax && p->c.tax_annealing) *I32(c, o_tax_last_completions) = *I32(c, o_completions); /* generate_masks :1036-1046 */
  float* pm = (float*)(c->arena + p->a_obs_p_mask) + (int64_t)e * p->MP;
  int o = 0;
  const int pmulti = p->c.multi_action_mode_planner;
  if (!pmulti || p->n_sub_p == 0) pm[o++] = 1.0f;
  if (p->n_sub_p) {
    float v = (*I32(c, o_tax_cycle_pos) == 1) ? 1.0f : 0.0f;
    for (int b = 0; b < p->n_sub_p; ++b) {
      if (pmulti) pm[o++] = 1.0f;
      for (int k = 0; k < p->sub_p_dim; ++k) pm[o++] = v * tax_rate_action_mask(c, k);
    }
  }
}

static void ose_decode_actions(ctx_t* c, const int32_t* aa, const int32_t* ap) {
  const aie_params* p = c->p;
  memset(c->act_wide, 0, sizeof(c->act_wide));
  memset(c->act_p, 0, sizeof(c->act_p));
  if (aa && p->n_sub_a) {
    for (int i = 0; i < p->n; ++i) {
      int v = aa[((int64_t)c->e * p->n + i) * p->act_a_width];
      if (p->c.multi_action_mode_agents) { if (v >= 0 && v <= p->sub_a_dim[0]) c->act_wide[i] = v; }
      else if (v >= 1 && v < 1 + p->sub_a_dim[0]) c->act_wide[i] = v;
    }
  }
  if (ap && p->n_sub_p) {
    const int32_t* a = ap + (int64_t)c->e * p->act_p_width;
    if (p->c.multi_action_mode_planner) { for (int b = 0; b < p->n_sub_p; ++b) c->act_p[b] = a[b]; }
    else {
      int v = a[0];
      if (v >= 1 && v < 1 + p->n_sub_p * p->sub_p_dim) c->act_p[(v - 1) / p->sub_p_dim] = (v - 1) % p->sub_p_dim + 1;
    }
  }
}

static void ose_step_one(const aie_params* p, uint8_t* arena, int e, const int32_t* aa, const int32_t* ap) {
  ctx_t c;
  make_ctx(&c, p, arena, e);
  ose_decode_actions(&c, aa, ap);
  *I32(&c, o_timestep) += 1;
  if (EV(&c)) EV(&c)[0] = 0;
  for (int k = 0; k < p->c.n_components; ++k) {
    if (p->c.components[k] == AIE_COMP_SIMPLE_LABOR) labor_step(&c);
    else if (p->c.components[k] == AIE_COMP_TAX) tax_step(&c);
    else if (p->c.components[k] == AIE_COMP_WEALTH_REDISTRIBUTION) wealth_step(&c);
  }
  ose_write_obs(&c, 0);
  /* compute_reward one_step_economy.py:195-222 */
  const int n = p->n;
  double cur[AIE_MAX_AGENTS_WIDE + 1];
  double* util = F64(&c, o_util);
  ose_metrics(&c, cur);
  float* ra = (float*)(arena + p->a_rew_a) + (int64_t)e * n;
  for (int i = 0; i < n; ++i) { ra[i] = (float)(cur[i] - util[i]); util[i] = cur[i]; }
  ((float*)(arena + p->a_rew_p))[e] = (float)(cur[n] - util[n]);
  util[n] = cur[n];
  int done = *I32(&c, o_timestep) >= p->c.episode_length;
  (arena + p->a_done)[e] = (uint8_t)done;
  if (done) *I32(&c, o_completions) += 1;
}

/* reset: one_step_economy.py:99-118 + simple_labor.py:76-95 + redistribution.py:1109-1139
 * + additional_reset_steps :224-241.  No random draws. */
static void ose_reset_one(const aie_params* p, uint8_t* arena, int e) {
  ctx_t c;
  make_ctx(&c, p, arena, e);
  const int n = p->n;
  *I32(&c, o_timestep) = 0;
  memset(MET(&c), 0, (size_t)p->met_bytes);
  if (EV(&c)) EV(&c)[0] = 0;
  for (int i = 0; i < n; ++i) {
    F64(&c, o_inv_coin)[i] = 0; F64(&c, o_esc_coin)[i] = 0; F64(&c, o_labor)[i] = 0;
    F64(&c, o_skill)[i] = p->has_labor ? p->c.labor_skills[i] : 0;
    F64(&c, o_production)[i] = 0;
  }
  *I32(&c, o_first_step) = 1;
  if (p->has_tax) {
    for (int b = 0; b < p->NB; ++b) I32(&c, o_tax_rate_idx)[b] = 0;
    *I32(&c, o_tax_cycle_pos) = 1;
    for (int i = 0; i < n; ++i) {
      F64(&c, o_tax_last_coin)[i] = 0; F64(&c, o_tax_last_income)[i] = 0; F64(&c, o_tax_last_marginal_rate)[i] = 0;
    }
    *F64(&c, o_tax_total_collected) = 0;
    if (p->c.tax_model == AIE_TAX_SAEZ) { /* _curr_rates_obs (:1123), then the running average (:1136-1137) */
      for (int b = 0; b < p->NB; ++b) F64(&c, o_tax_saez_obs_rates)[b] = tax_rate(&c, b);
      memcpy(F64(&c, o_tax_saez_rates), SAEZ(&c) + AIE_SAEZ_OFF_AVG, sizeof(double) * (size_t)p->NB);
    }
  }
  ose_metrics(&c, F64(&c, o_util));
  ose_write_obs(&c, 1);
  float* ra = (float*)(arena + p->a_rew_a) + (int64_t)e * n;
  for (int i = 0; i < n; ++i) ra[i] = 0;
  ((float*)(arena + p->a_rew_p))[e] = 0;
  (arena + p->a_done)[e] = 0;
}

/* ---- exported entry points (ctypes) ---------------------------------------------- */
int aie_oracle_params(const aie_config* cfg, aie_params* out, aie_tensor_table* tt, char* err, int errlen) {
  return aie_build_params(cfg, out, tt, err, (size_t)errlen);
}
int aie_oracle_sizeof_params(void) { return (int)sizeof(aie_params); }
int64_t aie_oracle_arena_bytes(const aie_params* p) { return p->arena_bytes; }
int aie_oracle_sizeof_table(void) { return (int)sizeof(aie_tensor_table); }

void aie_oracle_step(const aie_params* p, uint8_t* arena, const int32_t* aa, const int32_t* ap, int e0, int e1) {
  for (int e = e0; e < e1; ++e) {
    if (p->c.scenario == AIE_SCN_ONE_STEP_ECONOMY) ose_step_one(p, arena, e, aa, ap);
    else step_one(p, arena, e, aa, ap);
  }
}
/* tax_model "saez": what a period start does to the rates (random draw or formula) for every replica,
 * on whatever buffer / estimates the arena holds -- lets tests check the formula in isolation */
void aie_oracle_saez_period_start(const aie_params* p, uint8_t* arena) {
  for (int e = 0; e < p->E; ++e) {
    ctx_t c;
    make_ctx(&c, p, arena, e);
    saez_set_new_period_rates(&c);
    for (int b = 0; b < p->NB; ++b) F64(&c, o_tax_saez_obs_rates)[b] = tax_rate(&c, b);
  }
}
void aie_oracle_reset(const aie_params* p, uint8_t* arena, const uint8_t* mask, int e0, int e1) {
  for (int e = e0; e < e1; ++e)
    if (!mask || mask[e]) {
      if (p->c.scenario == AIE_SCN_ONE_STEP_ECONOMY) ose_reset_one(p, arena, e);
      else reset_one(p, arena, e);
    }
}
void aie_oracle_seed64(const aie_params* p, uint8_t* arena, uint64_t base_seed) {
  for (int e = 0; e < p->E; ++e) {
    uint8_t* rec = arena + p->a_records + (int64_t)e * p->rec_bytes;
    if (p->c.rng_mode == AIE_RNG_FAST) { /* aie_seed_fast(base_seed, 0); aie_seed: base_seed < 2^32 */
      uint32_t* st = (uint32_t*)(rec + p->o_mt);
      const uint64_t s = base_seed + (uint64_t)e;
      st[0] = (uint32_t)s; st[1] = 0u; st[2] = ((uint32_t)(s >> 32) & 0xffffu) << 16; st[3] = 0u;
    } else {
      aie_oracle_seed_one((uint32_t*)(rec + p->o_mt), (uint32_t)base_seed + (uint32_t)e);
    }
    *(int32_t*)(rec + p->o_mt_pos) = 624;
    *(int32_t*)(rec + p->o_mt_has_gauss) = 0;
    *(double*)(rec + p->o_mt_gauss) = 0.0;
  }
}
void aie_oracle_seed(const aie_params* p, uint8_t* arena, uint32_t base_seed) { aie_oracle_seed64(p, arena, (uint64_t)base_seed); }
/* aie_sample_policy_actions (include/aie.h) restated: inverse-CDF sampling over the allowed entries of every action slot,
 * float64 weights exp(logit - max) with the sampler's fixed-operation exp and prefix-sum order (aie_layout.h), one uniform
 * per slot from the counter RNG keyed (seed, global replica, draw index, slot); NaN logits count as masked, NO-OP if nothing
 * is allowed; advances `sample_t`.  Not part of the reference (its trainers sample in their own framework,
 * training_script.py:88-133): this pins the PRODUCT's sampler.  Gather-trade-build and one-step-economy layouts (COVID's
 * collated masks: a Python transcription in tests/test_gpu_parity.py). */
/* One row of the inverse-CDF sampler (aie_layout.h: the sampler's definition): chunks of 64 entries, the fixed-order scan
 * inside a chunk, a running carry from chunk to chunk. */
int32_t aie_oracle_sample_row(const float* lg, const float* mask, int mask_stride, int len, uint32_t rnd) {
  int any = 0;
  float M = 0.0f;
  for (int k = 0; k < len; ++k) {
    const float x = lg[k];
    if (!(mask[k * mask_stride] > 0.5f) || x != x) continue;
    if (!any || x > M) M = x;
    any = 1;
  }
  if (!any) return 0;
  const float u = aie_sampler_uniform(rnd);
  const int nch = (len + 63) / 64, seg = nch > 1 ? 64 : aie_sampler_segment(len);
  float T = 0.0f;
  int choice = -1, last_ok = -1;
  for (int pass = (nch > 1 ? 0 : 1); pass < 2; ++pass) {
    float carry = 0.0f;
    for (int ch = 0; ch < nch; ++ch) {
      float v[64];
      int ok[64];
      for (int r = 0; r < 64; ++r) {
        const int k = 64 * ch + r;
        v[r] = 0.0f;
        ok[r] = 0;
        if (k < len) {
          const float x = lg[k];
          ok[r] = mask[k * mask_stride] > 0.5f && x == x;
          if (ok[r]) v[r] = aie_sampler_expf(x - M);
        }
      }
      for (int d = 1; d < 16; d <<= 1)
        for (int r = 63; r >= 0; --r) v[r] = v[r] + ((r & 15) >= d ? v[r - d] : 0.0f); /* (descending r: v[r - d] is still the previous step's) */
      if (seg >= 32)
        for (int r = 0; r < 64; ++r)
          if ((r >> 4) & 1) v[r] = v[r] + v[(r & ~15) - 1];
      if (seg >= 64)
        for (int r = 32; r < 64; ++r) v[r] = v[r] + v[31];
      const float tot = carry + v[seg - 1];
      if (pass == 0) {
        carry = tot;
        continue;
      }
      if (nch == 1) T = tot;
      for (int r = 0; r < 64; ++r) {
        if (!ok[r]) continue;
        last_ok = 64 * ch + r;
        if (choice < 0 && carry + v[r] > u * T) choice = 64 * ch + r;
      }
      carry = tot;
    }
    if (pass == 0) T = carry;
  }
  return choice < 0 ? last_ok : choice;
}
void aie_oracle_sample_policy_actions(const aie_params* p, uint8_t* arena, const float* logits_a, const float* logits_p,
                                      uint64_t seed, int64_t env_offset, int32_t* act_a, int32_t* act_p) {
  const int na = p->n * p->act_a_width, per_env = na + p->act_p_width;
  for (int e = 0; e < p->E; ++e) {
    int32_t* tf = (int32_t*)(arena + p->a_records + (int64_t)e * p->rec_bytes + p->o_sample_t);
    const int64_t t = *tf;
    for (int j = 0; j < per_env; ++j) {
      const float *mask, *lg;
      int lo = 0, len;
      int32_t* dst;
      if (j < na) {
        if (!act_a || !logits_a) continue;
        const int i = j / p->act_a_width, s = j - i * p->act_a_width;
        mask = (const float*)(arena + p->a_obs_a_mask) + ((int64_t)e * p->n + i) * p->MA;
        lg = logits_a + ((int64_t)e * p->n + i) * p->MA;
        if (p->c.multi_action_mode_agents) {
          for (int k = 0; k < s; ++k) lo += 1 + p->sub_a_dim[k];
          len = p->n_sub_a ? 1 + p->sub_a_dim[s] : 1;
        } else {
          len = p->MA;
        }
        dst = act_a + (int64_t)e * na + j;
      } else {
        if (!act_p || !logits_p) continue;
        const int s = j - na;
        mask = (const float*)(arena + p->a_obs_p_mask) + (int64_t)e * p->MP;
        lg = logits_p + (int64_t)e * p->MP;
        if (p->c.multi_action_mode_planner) {
          lo = s * (1 + p->sub_p_dim);
          len = p->n_sub_p ? 1 + p->sub_p_dim : 1;
        } else {
          len = p->MP;
        }
        dst = act_p + (int64_t)e * p->act_p_width + s;
      }
      const uint32_t base = aie_counter_rng(seed, (uint64_t)(env_offset + e), (uint64_t)t, (uint64_t)per_env); /* per replica and call */
      *dst = aie_oracle_sample_row(lg + lo, mask + lo, 1, len, aie_sampler_entry_rng(base, (uint32_t)j));
    }
    *tf = (int32_t)t + 1;
  }
}
/* the sampler's building blocks (aie_layout.h), exported so that a CPU test can hold them to their Python transcription
 * and to libm */
/* the layout stream of a replica's k-th reset in the counter-stream mode (aie_layout.h): out = (key, counter high word, tag lo, tag hi) */
void aie_oracle_layout_stream(const uint32_t* st, uint32_t* out) {
  aie_layout_stream(st, out);
  const uint64_t tag = aie_layout_tag(st);
  out[2] = (uint32_t)tag;
  out[3] = (uint32_t)(tag >> 32);
}
float aie_oracle_sampler_expf(float y) { return aie_sampler_expf(y); }
float aie_oracle_sampler_uniform(uint32_t rnd) { return aie_sampler_uniform(rnd); }
uint32_t aie_oracle_sampler_entry_rng(uint32_t slot_word, uint32_t k) { return aie_sampler_entry_rng(slot_word, k); }
/* multi-threaded step for the cpu_baseline leg of bench.py */
void aie_oracle_step_mt(const aie_params* p, uint8_t* arena, const int32_t* aa, const int32_t* ap, int nthreads) {
  int E = p->E;
#pragma omp parallel for num_threads(nthreads) schedule(static)
  for (int e = 0; e < E; ++e) {
    if (p->c.scenario == AIE_SCN_ONE_STEP_ECONOMY) ose_step_one(p, arena, e, aa, ap);
    else step_one(p, arena, e, aa, ap);
  }
}
