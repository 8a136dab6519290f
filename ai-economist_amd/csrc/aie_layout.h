/*
 * aie_layout.h -- single source of truth for (a) the per-replica state record that
 * lives in HBM, (b) the dense observation / reward tensors, (c) the action-space and
 * flat-observation bookkeeping that the reference derives at construction time.
 *
 * Plain C99 (also included from HIP C++).  Included by the HIP library
 * (aie_capi.hip / aie_kernels.hip) and by the CPU restatement oracle/aie_oracle.c so
 * that both step byte-identical records.
 *
 * Reference logic restated here:
 *   action subspace registration order + single-action map
 *       F/base/base_agent.py:97-169  (_incorporate_component / register_components)
 *   per-component action counts
 *       F/components/build.py:88-98, move.py:69-79,
 *       continuous_double_auction.py:411-431, redistribution.py:920-939
 *   flat observation = concatenation in SORTED key order
 *       F/base/base_env.py:561-612 (_build_packager/_package), keys prefixed with
 *       "<Component.name>-" / "world-" at base_env.py:644-673
 *   flattened masks  F/base/base_agent.py:440-460
 *   map key order    F/base/world.py:59-93 (resources first, then sorted landmarks,
 *                    then <Res>SourceBlock appended at :66)
 */
#ifndef AIE_LAYOUT_H_
#define AIE_LAYOUT_H_

#include <math.h>
#include <stdio.h>
#include <string.h>

#include "aie.h"

#if defined(__HIPCC__)
#define AIE_HD __host__ __device__
#else
#define AIE_HD
#endif

#define AIE_MAX_TENSORS 160
#define AIE_MAX_MASK 544   /* entries of one agent's flattened action mask (4 x 127 prices + ...) */

/* internal action-subspace slots of a mobile agent */
enum {
  AIE_SUB_BUILD = 0,
  AIE_SUB_BUY0 = 1,  /* Buy_Stone  */
  AIE_SUB_SELL0 = 2, /* Sell_Stone */
  AIE_SUB_BUY1 = 3,  /* Buy_Wood   */
  AIE_SUB_SELL1 = 4, /* Sell_Wood  */
  AIE_SUB_GATHER = 5,
  AIE_SUB_LABOR = 6, /* SimpleLabor (one-step-economy) */
  AIE_N_SUB_SLOTS = 7
};

/* Everything a kernel needs, passed BY VALUE as the kernel argument. */
typedef struct aie_params {
  aie_config c;

  /* dims */
  int32_t E, n, H, W, HW;
  int32_t P;        /* price levels = max_bid_ask + 1                                  */
  int32_t NB;       /* tax brackets                                                    */
  int32_t M;        /* order-book capacity per commodity side = n * max_num_orders     */
  int32_t CM;       /* map channels in Maps.state: 5 (no water) or 6                   */
  int32_t WV;       /* egocentric window edge = 2*obs_range + 1                        */
  int32_t am_ch, am_h, am_w; /* agents' "world-map": CM+1 x WV x WV (egocentric, last channel =
                              * in-bounds) or, with full_observability, CM x H x W          */
  int32_t has_build, has_cda, has_gather, has_tax, has_labor;
  int32_t planner_acts; /* 1 if the planner has tax action subspaces                   */

  /* action spaces */
  int32_t n_sub_a;                      /* registered agent subspaces, in order        */
  int32_t sub_a_slot[AIE_MAX_SUBSPACES];/* -> AIE_SUB_*                                */
  int32_t sub_a_dim[AIE_MAX_SUBSPACES]; /* number of non-NO-OP actions                 */
  int32_t sub_a_base[AIE_MAX_SUBSPACES];/* first single-action index of the subspace   */
  int32_t A;                            /* single-action mode: total actions incl NO-OP */
  int32_t n_sub_p;                      /* planner subspaces (NB or 0)                 */
  int32_t sub_p_dim;
  int32_t act_a_width;                  /* ints per agent in d_actions_a               */
  int32_t act_p_width;                  /* ints per replica in d_actions_p             */

  /* flat observation sizes and fragment offsets (in f32 elements) */
  int32_t FA, FP, FPA;                  /* agent flat, planner flat, planner p{i}      */
  int32_t fa_build, fa_cda, fa_gather, fa_tax, fa_time, fa_world;
  int32_t fp_cda, fp_tax, fp_time, fp_world;
  int32_t fpa_tax, fpa_world;
  int32_t fa_labor;                     /* one-step-economy: SimpleLabor-skill            */
  int32_t MA, MP;                       /* flattened mask sizes                        */

  /* per-replica record: byte offsets */
  int32_t rec_bytes;
  int32_t o_cells;  /* u32 [HW]: byte0 stone, byte1 wood, byte2 house owner (i8, -1 none),
                       byte3 static flags (AIE_CELL_WATER | _STONE_SRC | _WOOD_SRC)        */
  int32_t o_loc_r, o_loc_c, o_inv_res, o_esc_res;
  int32_t o_inv_coin, o_esc_coin, o_labor, o_build_payment, o_build_skill, o_bonus_gather_prob;
  int32_t o_util;
  int32_t o_cda_n_bids, o_cda_n_asks, o_cda_bids, o_cda_asks, o_cda_n_orders;
  int32_t o_cda_bid_hist, o_cda_ask_hist, o_cda_price_history;
  int32_t o_tax_cycle_pos, o_tax_rate_idx, o_tax_last_coin, o_tax_last_income;
  int32_t o_tax_last_marginal_rate, o_tax_total_collected;
  int32_t o_timestep, o_completions, o_auto_warmup;
  int32_t o_mask_bits, o_mask_p_open; /* gather-trade-build: the per-agent mask bits (int32 [n]) and the planner's "rates may be
                          * set today" flag the action-mask tensors currently show -- a step rewrites only the masks of agents
                          * whose bits changed and the planner's when the flag flips (round 6); 0 = fields absent */
  int32_t o_skill, o_production, o_first_step; /* one-step-economy / SimpleLabor           */
  int32_t o_mt, o_mt_pos, o_mt_has_gauss, o_mt_gauss;
  int32_t o_tax_last_completions; /* PeriodicBracketTax._last_completions (tax annealing)     */
  int32_t o_error_flags; /* AIE_ERR_* bits (include/aie.h), sticky until reset */
  int32_t o_src_n, o_src_list; /* the regeneration's source doubles of THIS replica (round 6): int32 count, uint16 [AIE_SRC_CAP]
                          * ascending -- double d of a step's 2 H W np.random.rand values targets Wood cell d (d < H W) or Stone
                          * cell d - H W, and only source-block cells can respawn.  Derived from the cells' flag bytes by the
                          * reset kernel (and by aie_set_layout / aie_upload of the cells; load_state on the host), read by every
                          * step -- straight into the registers of the wave that regenerates, while the components run -- instead
                          * of a scan of the flags.  Behind the generator's state, i.e. outside the record's LDS image.  Absent
                          * (0) where the batch shares one list (a_src_list). */
  int32_t o_obs_valid;   /* 1: the map observation tensors hold this replica's current state (the step
                          * kernel then only rewrites what a step changes); cleared by anything that
                          * edits state from outside the kernels                                     */

  /* arena: byte offsets of the dense regions */
  int64_t a_records;
  int64_t a_obs_a_map, a_obs_a_idx, a_obs_a_flat, a_obs_a_mask, a_obs_a_time;
  int64_t a_obs_p_map, a_obs_p_idx, a_obs_p_flat, a_obs_p_mask, a_obs_p_time, a_obs_p_agents;
  int64_t a_rew_a, a_rew_p, a_done;
  int64_t arena_bytes;

  /* exact unsigned division by run-time constants: q / d == __umulhi(q, mg_d) for
   * q < 2^32 / d (d >= 2); see aie__magic() */
  uint32_t mg_WV2, mg_WV, mg_MA, mg_FA, mg_P, mg_2P, mg_taxA, mg_HW, mg_W, mg_sub_p, mg_sub_p_dim;

  /* development only: phases of the step kernel to skip when profiling
   * (tools/phase_profile.py); always 0 in normal operation */
  /* per-replica episode accumulators behind env.metrics (component get_metrics): touched only
   * when a trade executes / on tax days, so they live outside the streamed record */
  /* regen_halfwidth > 0 (dynamic_layout.py:446-463): the regeneration probability of a source block
   * is signal.convolve2d(max(map, source blocks), regen_weight / d^2 * ones(d, d), "same"), d = 1 + 2 *
   * halfwidth.  With max_health == 1 the convolved plane IS the source-block plane, so the
   * probability is regen_p[resource][number of source blocks in the d x d window]; the reset kernel
   * counts the window once per episode into the record plane o_regen_count. */
  int32_t regen_conv;    /* 1: some regen_halfwidth > 0 */
  int32_t regen_general; /* 1: some resource has regen_halfwidth > 0 AND max_health > 1: the convolved plane changes
                          * with the map, so its probabilities are true window sums (scipy's accumulation order)
                          * over a pre-step snapshot of max(map, source blocks); aie_step_kernel_log only */
  int32_t o_regen_count; /* u8 [AIE_N_RES][H*W] */
  double regen_p[AIE_N_RES][50];
  /* tax_model "saez" (redistribution.py:436-823): per-replica block outside the streamed record,
   * touched on tax days and period starts only.  Layout (bytes): int32 len, int32 reached_min_samples,
   * pad to 16; f64 elas[4] = {elas_t, elas_tm1, log_z0_t, log_z0_tm1}; f64 running_avg[AIE_MAX_BRACKETS];
   * f64 next_rates[AIE_MAX_BRACKETS] (formula output for the coming period start, written by
   * aie_saez_kernel); f64 buffer[saez_cap][2] = (income, marginal rate), oldest first. */
  int64_t a_saez;
  int32_t saez_stride, saez_cap;
  int64_t a_saez_global; /* shared: int32 len (16 B), then f64 [saez_global_cap][2]: the trainer's cross-replica buffer */
  int32_t saez_global_cap, saez_pad_;
  int64_t a_layout_prob; /* shared f64 [AIE_N_RES][H*W]: source probability maps of the generated layouts (layout_gen
                          * UNIFORM / QUADRANT: uploaded once by the host, tensor "layout_source_prob") */
  /* Generated layouts under AIE_RNG_FAST (aie__layout_staged): a replica's k-th reset draws its source layout from a
   * stream of its own (aie_layout_stream: keyed by the replica's stream and k, the record's fourth state word), so the
   * layout does not depend on what the episode's steps drew and can be made AHEAD of the reset.  Staging area:
   *   a_layout_stage  u8 [E][layout_stage_stride]: bit 0 Stone source, bit 1 Wood source per cell (before the checker /
   *                   water-line cuts), valid iff the tag equals the tag of the replica's coming reset;
   *   a_layout_tag    u64 [E]: aie_layout_tag of the staged layout (0: none);
   *   a_layout_ctl    int32 [4]: staged layouts consumed since the last refill, the refill's go flag, installs from the
   *                   staging area, layouts drawn inside a reset (development counters).
   * A reset installs the staged layout when the tag matches and draws the layout itself otherwise (same stream, same
   * result); behind every reset launch a one-thread kernel decides whether enough layouts were consumed for a refill
   * launch to pay (its duration is that of the slowest replica, however many it draws). */
  int64_t a_layout_stage, a_layout_tag, a_layout_ctl;
  int32_t layout_stage_stride, layout_pad_;
  int64_t a_src_list;    /* shared, fixed layouts with shared_layout only (aie__shared_src_list): int32 count (16 B), then
                          * uint16 [AIE_SRC_CAP]: the regeneration draws that target a source block (double d of a step's
                          * 2 H W np.random.rand values: Wood cell d, or Stone cell d - H W), derived once by
                          * aie_set_layout instead of by every step from the cells' flag bytes; count > AIE_SRC_CAP: no
                          * list (row-by-row regeneration) */
  int32_t o_tax_saez_rates; /* record: f64 [NB] curr_bracket_tax_rates */
  int32_t o_tax_saez_obs_rates; /* record: f64 [NB] _curr_rates_obs: the rates the "curr_rates" observation shows,
                                   refreshed at period starts and -- BEFORE the running average replaces the
                                   bracket rates -- at reset (redistribution.py:1123, 1136-1137) */
  double saez_edges[AIE_SAEZ_BINS + 1]; /* np.linspace(0, top cutoff, 101), :286-288 */
  int64_t a_events;      /* dense-log events of replicas [0, ev_replicas): int32 count (16 B), then rows */
  int32_t ev_replicas, ev_cap, ev_stride, ev_pad_;
  int64_t a_metrics;
  int32_t met_bytes;
  int32_t mo_cda;        /* int32 [2: sell, buy][AIE_N_RES][n][2: n_sales, sum of prices]         */
  int32_t mo_tax_sched;  /* f64 [NB] sum over tax days of the bracket rates                       */
  int32_t mo_tax_income; /* f64 [n]  sum over tax days of max(0, income)                          */
  int32_t mo_tax_paid;   /* f64 [n]  sum over tax days of the tax paid                            */
  int32_t mo_tax_eff;    /* f64      sum of all effective tax rates (n per tax day)               */
  int32_t mo_tax_occ;    /* int32 [NB] incomes observed per bracket                               */
  int32_t mo_tax_days;   /* int32    tax days so far                                              */
  /* flattened agent action mask, element m: allowed iff ((mask_bits >> sh) & msk) >= thr with
   * sh = test & 31, msk = (test >> 8) & 0xff, thr = test >> 16 (mask_bits: write_action_masks) */
  uint32_t mask_test[AIE_MAX_MASK];
  int32_t dev_skip_mask;
  uint64_t* dev_trace;   /* development: 8 clock stamps per workgroup (start, components.., regen, end), or NULL */
  /* aie_set_reward_log: the caller's reward log, f32 [rew_slots][E][n + 2] = agents' rewards, the planner's reward, done --
   * or NULL.  In the device-side block (not a kernel argument) so that a launch captured in a hipGraph sees a later
   * aie_set_reward_log call (round 6).  rew_epoch: bumped by every such call, restarts the replicas' slot
   * counters (record fields o_rew_slot / o_rew_epoch). */
  float* rew_log;
  int32_t rew_slots, rew_epoch;

  /* ---- COVID-19 scenario (aie__build_covid) ---- */
  int32_t cv_L;          /* filter_len                                                     */
  int32_t cv_F;          /* num_filters                                                    */
  int32_t cv_NL;         /* num_stringency_levels                                          */
  int32_t cv_NS;         /* num_subsidy_levels                                             */
  int32_t cv_nch;        /* 16-day chunks in the per-replica stringency history           */
  int32_t cv_row;        /* bytes per history chunk = 16 * n rounded up to 64              */
  int32_t cv_nrow_obs;   /* rows of the per-replica agent observation block               */
  int32_t cv_pitch;      /* lanes per row of the record's per-state rows (= n: packed; the kernel is HBM-bound and a
                          * 64-lane pitch would move 20 % idle bytes)                       */
  int32_t cv_t_first_delivery;
  int32_t o_cv_state;    /* record: AIE_CV_ST_* float32 rows of 64 lanes                  */
  int32_t o_cv_cooldown; /* record: int32 row                                              */
  int32_t o_cv_subsidy_level;
  int32_t o_cv_sums;     /* record: AIE_CV_SUM_* float64 rows of 64 lanes                 */
  int32_t o_cv_p_index;  /* record: planner Health Index, Economic Index (float32 x 2)    */
  int32_t o_cv_ring;     /* record: uint8 [32][64]: the stringency levels of the 32 most recent days, row = (filter_len + day) & 31,
                          * one 64-byte row per day (what a step reads of the recent past -- yesterday, beta_delay days
                          * ago, the lagged observation -- and writes for today are single rows, not 51 scattered bytes) */
  int32_t o_cv_acc;      /* record: float64 [F] rows of n lanes: filter_recurrence: each filter's discounted sum of stringency
                          * deltas over the current window; else: each filter's sum over the NEXT step's window without that
                          * step's own day, formed by the tail of the previous step (or by reset) */
  /* window sums (filter_recurrence == 0) over the NON-ZERO level changes only: per replica and state the change events
   * (history day tau, delta) of the episode so far + the pre-episode days, oldest first.  A day without a change adds
   * fma(0, tap, acc) == acc to the reference's sum, so the sum over the events in the same order is the SAME float64,
   * bit for bit.  A state that would exceed the list's capacity makes its replica "dense" until the next reset: the
   * kernel then streams the 600-day history as before.                                                          */
  int32_t cv_ev_groups;  /* capacity of a state's event list in groups of 4 (0: filter_recurrence)                */
  int32_t o_cv_ev_ht;    /* record: int32 row: head | tail << 16 of every state's list                           */
  int32_t o_cv_dense;    /* record: int32: 1 = this replica streams the whole window (list overflow)              */
  int32_t o_cv_tail_pending; /* record: int32: the step kernel took a step that aie_covid_window_kernel has to follow up  */
  int64_t a_cv_events;   /* uint32 [E][cv_ev_groups][64 lanes][4]: tau | (delta & 0xff) << 16                     */
  int64_t a_cv_ev0;      /* shared: the pre-episode days' lists in the same format, then int32 [64] head/tail row,
                          * then int32 dense flag (a pre-episode list overflowed); aie_covid_prepare_kernel       */
  int64_t a_cv_consts;   /* AIE_CV_K_* float64 rows of 64 (shared by all replicas)         */
  int64_t a_cv_filters;  /* float64 [pad+L+pad][F]                                              */
  int64_t a_cv_hist0;    /* uint8 [L+1][n]   stringency levels of the L days before t=0 + t=0 */
  int64_t a_cv_hist0c;   /* uint8 [nch][cv_row]: the same days in the per-replica history format (what reset copies);
                          * derived from a_cv_hist0 by aie_covid_prepare_kernel whenever that table is uploaded  */
  int64_t a_cv_acc0;     /* float64 [F][64]: what reset puts into o_cv_acc (recurrence: A_0; window sums: step 1's sums over
                          * the pre-episode days), ditto */
  int64_t a_cv_lag_obs;  /* uint8 [beta_delay][n]                                          */
  int64_t a_cv_replay_a; /* uint8 [T][64]: replay_policies: the states' stringency action of step t in row t - 1 */
  int64_t a_cv_replay_p; /* int32 [T]: replay_policies: the planner's subsidy level of step t at t - 1            */
  int64_t a_cv_replay_state; /* float64 [6: S, I, R, V, D, U][T + 1][64]: replay_data                             */
  int64_t a_cv_hist;     /* uint8 [E][nch][cv_row]                                         */
  int64_t a_cv_obs_a;    /* float32 [E][cv_nrow_obs][n]                                    */
  int64_t a_cv_obs_p;    /* float32 [E][4 + 1 + NS]                                        */
  /* properties of run-time scalars that DO shape the code (they stay in the instances' images, the values do not) */
  int32_t sh_energy_warmup; /* energy_warmup_constant > 0: the labor cost is weighted by 1 - exp(-v / constant)   */
  int32_t sh_eta_is_one;    /* isoelastic_eta == 1: log utility instead of the power form                        */
  int32_t auto_reset;    /* run-time switch (aie_set_auto_reset): replicas restart inside the launch that ends their episode */
  /* per-replica call counters (record fields, so that nothing a step needs travels by value from a host-side counter: a
   * captured hipGraph of aie_step / aie_step_sample_next replays correctly): o_sample_t = draws of the synthetic random
   * policy so far (the `t` of its counter RNG), o_rew_slot = the reward-log slot the next step fills, o_rew_epoch = the
   * aie_set_reward_log call that slot counter belongs to (a new call restarts the slots at 0 without touching memory) */
  int32_t o_sample_t, o_rew_slot, o_rew_epoch;
  int32_t dev_draw_window; /* development (tests): capacity of the components' draw window in words, 0 = stage_window_words();
                            * honoured by aie_step_kernel_log only */
} aie_params;

/* Compile-time instances of the step kernel (aie_spec_generated.h) bake a CONSTANT image of aie_params into the code.
 * An image stands for a FAMILY of configurations: everything that shapes the code (component tuple, agent count,
 * world size, observation window, book capacity, bracket / rate counts, every flag, every record offset and table
 * derived from those) is in it; what does not is read from the run-time block (Ctx::R) and is blanked in the image:
 *   - what depends on the batch: the replica count, every arena offset, the dense-log event buffer, development hooks;
 *   - the configuration's SCALARS and value tables: episode_length, starting_agent_coin, isoelastic_eta, energy_cost,
 *     energy_warmup_constant, mixing weight, every labor cost, Build payment / multiplier, order_duration, tax period
 *     (gather-trade-build), bracket cutoffs, the discretised / fixed rates, rate_min / rate_max, the annealing schedule,
 *     regeneration weights (and regen_p derived from them), fixed-four locations and skills, split-layout ranks,
 *     starting coverages / clumpiness of generated layouts, SimpleLabor's skills and multiplier, labor exponent / cost.
 * The blanked scalars are POISONED (0x5A bytes), not zeroed: a kernel line that still read one of them from the
 * image would compute with 1 515 870 810 / 2.6e130 and fail every parity test, instead of passing by accident where
 * the configuration's value happens to be 0.
 * An environment runs on an instance iff its normalised block equals the instance's image byte for byte
 * (aie_capi.hip: aie_create): same family, any scalars, any replica count, any seed. */
static inline void aie_spec_normalize(aie_params* p) {
  p->E = 0;
  p->c.n_envs = 0;
  p->a_records = 0;
  p->a_obs_a_map = p->a_obs_a_idx = p->a_obs_a_flat = p->a_obs_a_mask = p->a_obs_a_time = 0;
  p->a_obs_p_map = p->a_obs_p_idx = p->a_obs_p_flat = p->a_obs_p_mask = p->a_obs_p_time = p->a_obs_p_agents = 0;
  p->a_rew_a = p->a_rew_p = p->a_done = 0;
  p->arena_bytes = 0;
  p->a_saez = p->a_events = p->a_metrics = p->a_saez_global = 0;
  p->a_cv_consts = p->a_cv_filters = p->a_cv_hist0 = p->a_cv_lag_obs = p->a_cv_hist = p->a_cv_obs_a = p->a_cv_obs_p = 0;
  p->a_cv_hist0c = p->a_cv_acc0 = 0;
  p->a_cv_events = p->a_cv_ev0 = 0;
  p->a_cv_replay_a = p->a_cv_replay_p = p->a_cv_replay_state = 0;
  p->a_layout_prob = 0;
  p->a_layout_stage = p->a_layout_tag = p->a_layout_ctl = 0;
  p->a_src_list = 0;
  p->dev_skip_mask = 0;
  p->dev_trace = 0;
  p->rew_log = 0;
  p->rew_slots = p->rew_epoch = 0;
  p->dev_draw_window = 0;
  p->auto_reset = 0;
  p->ev_replicas = p->ev_cap = p->ev_stride = 0;  /* dense-log replicas: the fast kernels never record events */
  p->c.dense_log_replicas = 0;
  memset(p->c.labor_skills, 0, sizeof(p->c.labor_skills));  /* SimpleLabor's skills are data, read from the run-time block */
#define AIE__BLANK(field) memset(&(field), 0x5A, sizeof(field))
  {
    aie_config* c = &p->c;
    AIE__BLANK(c->episode_length); AIE__BLANK(c->starting_agent_coin); AIE__BLANK(c->isoelastic_eta);
    AIE__BLANK(c->energy_cost); AIE__BLANK(c->energy_warmup_constant); AIE__BLANK(c->mixing_weight_gini_vs_coin);
    AIE__BLANK(c->build_payment); AIE__BLANK(c->build_payment_max_skill_multiplier); AIE__BLANK(c->build_labor);
    AIE__BLANK(c->move_labor); AIE__BLANK(c->collect_labor); AIE__BLANK(c->cda_order_labor);
    AIE__BLANK(c->cda_order_duration);
    /* one-step-economy keeps its tax period in the image: with period 1 two record fields are dead and are neither
     * loaded nor stored (aie_kernels_ose.hip: ose_load_record) -- there the period shapes the code */
    if (c->scenario != AIE_SCN_ONE_STEP_ECONOMY) AIE__BLANK(c->tax_period);
    AIE__BLANK(c->tax_bracket_cutoffs); AIE__BLANK(c->tax_disc_rates); AIE__BLANK(c->tax_fixed_rates);
    AIE__BLANK(c->tax_rate_max); AIE__BLANK(c->tax_rate_min); AIE__BLANK(c->tax_annealing_warmup);
    AIE__BLANK(c->tax_annealing_slope); AIE__BLANK(c->regen_weight); AIE__BLANK(c->ranked_locs);
    AIE__BLANK(c->avg_ranked_skill); AIE__BLANK(c->split_top_ranks); AIE__BLANK(c->layout_coverage);
    AIE__BLANK(c->layout_clump); AIE__BLANK(c->ose_labor_exponent); AIE__BLANK(c->ose_labor_cost);
    AIE__BLANK(c->labor_pmsm); AIE__BLANK(c->saez_buffer_size); AIE__BLANK(c->saez_fixed_elas);
    AIE__BLANK(p->regen_p); AIE__BLANK(p->saez_edges);
  }
#undef AIE__BLANK
}

typedef struct aie_tensor_table {
  int32_t n;
  aie_tensor_desc t[AIE_MAX_TENSORS];
} aie_tensor_table;

static inline uint32_t aie__magic(int64_t d) {
  return d <= 1 ? 0u : (uint32_t)((0x100000000ll + d - 1) / d);
}

static inline int64_t aie__align(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

static inline int aie__dtype_size(int dt) {
  switch (dt) {
    case AIE_U8: case AIE_I8: return 1;
    case AIE_I16: return 2;
    case AIE_I32: case AIE_U32: case AIE_F32: return 4;
    default: return 8;
  }
}

static inline int aie__has(const aie_config* c, int comp) {
  for (int i = 0; i < c->n_components; ++i)
    if (c->components[i] == comp) return 1;
  return 0;
}

#define AIE__FAIL(...)                                   \
  do {                                                   \
    if (err) snprintf(err, errlen, __VA_ARGS__);         \
    return AIE_E_INVALID;                                \
  } while (0)

/* source-block doubles the sparse regeneration handles (else: row-by-row) */
#define AIE_SRC_CAP 128
/* One list of the regeneration's source doubles for the whole batch (aie_params.a_src_list): every replica has the same,
 * never-changing source blocks */
static inline int aie__shared_src_list(const aie_config* c) {
  return c->scenario == AIE_SCN_GTB && c->shared_layout && c->layout_gen == AIE_LAYOUT_FIXED;
}
/* ... which the kernels use with the counter stream only (aie_kernels.hip: shared_src_list); everybody else keeps a list
 * per replica in the record (o_src_n / o_src_list) */
static inline int aie__record_src_list(const aie_config* c) {
  return c->scenario == AIE_SCN_GTB && !(aie__shared_src_list(c) && c->rng_mode == AIE_RNG_FAST);
}
/* words of generator state in a replica's record ("mt"): MT19937's key, or (key32, block number, salt, 0) of the
 * counter stream (include/aie.h: AIE_RNG_FAST) */
static inline int32_t aie__rng_state_words(const aie_config* c) { return c->rng_mode == AIE_RNG_FAST ? 4 : AIE_MT_N; }
/* generated layouts come from a stream of their own, drawn ahead of the reset (aie_params: a_layout_stage) */
AIE_HD static inline int aie__layout_staged(const aie_config* c) {
  return c->scenario == AIE_SCN_GTB && c->layout_gen != AIE_LAYOUT_FIXED && c->rng_mode == AIE_RNG_FAST;
}
/* The layout stream of a replica's k-th reset, k = st[3], from its stream's state st = (key, block, salt, k): the same
 * Philox function with key + 0x9E3779B9 (k >> 15) and the counter's high word salt | 0x8000 | (k & 0x7fff) -- bit 15 is
 * never set in the replica's own stream (its pair index stays below 2^47).  out = (key, salt); the stream starts at its
 * block 0 (state block 0xffffffff, position 624: the first draw opens block 0). */
AIE_HD static inline void aie_layout_stream(const uint32_t st[4], uint32_t out[2]) {
  out[0] = st[0] + 0x9E3779B9u * (st[3] >> 15);
  out[1] = st[2] | 0x8000u | (st[3] & 0x7fffu);
}
AIE_HD static inline uint64_t aie_layout_tag(const uint32_t st[4]) {  /* never 0: bit 15 */
  uint32_t ks[2];
  aie_layout_stream(st, ks);
  return ((uint64_t)ks[0] << 32) | (uint64_t)ks[1];
}

/* record field allocator */
static inline int32_t aie__rec(int32_t* cur, int32_t bytes, int32_t align) {
  int32_t o = (int32_t)aie__align(*cur, align);
  *cur = o + bytes;
  return o;
}

static inline void aie__add(aie_tensor_table* tt, const char* name, int dtype, int64_t off,
                            int64_t env_stride, int ndim_inner, int64_t d0, int64_t d1,
                            int64_t d2, int64_t d3, int64_t E) {
  if (!tt || tt->n >= AIE_MAX_TENSORS) return;
  aie_tensor_desc* d = &tt->t[tt->n++];
  memset(d, 0, sizeof(*d));
  snprintf(d->name, sizeof(d->name), "%s", name);
  d->dtype = dtype;
  d->ndim = 1 + ndim_inner;
  int64_t dims[4] = {d0, d1, d2, d3};
  d->shape[0] = E;
  d->stride[0] = env_stride;
  int64_t st = aie__dtype_size(dtype);
  for (int i = ndim_inner - 1; i >= 0; --i) {
    d->shape[1 + i] = dims[i];
    d->stride[1 + i] = st;
    st *= dims[i];
  }
  d->arena_offset = off;
  d->data = NULL;
}

/* ---- COVID-19 (F/scenarios/covid19/covid19_env.py + F/components/covid19_components.py) ----
 * One wavefront per replica, one lane per US state.  Per-state model constants are float64
 * rows of 64 lanes (values the reference holds as float32 / int32 are exactly representable);
 * the stringency history is kept per replica as [16-day chunk][state][16 bytes] so that a
 * lane streams its own 601-day window with 16-byte loads that are contiguous across lanes. */
enum {
  AIE_CV_K_POP = 0, AIE_CV_K_BETA_SLOPE, AIE_CV_K_BETA_INTERCEPT, AIE_CV_K_UNEMP_BIAS,
  AIE_CV_K_MAX_PROD, AIE_CV_K_HEALTH_NORM, AIE_CV_K_ECON_NORM,
  AIE_CV_K_MIN_HEALTH, AIE_CV_K_MAX_HEALTH, AIE_CV_K_MIN_ECON, AIE_CV_K_MAX_ECON,
  AIE_CV_K_W_HEALTH, AIE_CV_K_W_ECON, AIE_CV_K_MAX_DAILY_SUBSIDY, AIE_CV_K_VACCINES_PER_DELIVERY,
  AIE_CV_K_S0, AIE_CV_K_I0, AIE_CV_K_R0, AIE_CV_K_D0, AIE_CV_K_U0, AIE_CV_K_V0,
  AIE_CV_K_CONV_W0,                                       /* + f, f < AIE_COVID_MAX_FILTERS */
  AIE_CV_K_COUNT = AIE_CV_K_CONV_W0 + AIE_COVID_MAX_FILTERS
};
enum { AIE_CV_ST_S = 0, AIE_CV_ST_I, AIE_CV_ST_R, AIE_CV_ST_D, AIE_CV_ST_V, AIE_CV_ST_U,
       AIE_CV_ST_PROD, AIE_CV_ST_SUBSIDY,
       AIE_CV_ST_HEALTH_INDEX, AIE_CV_ST_ECONOMIC_INDEX, /* agent.state["Health/Economic Index"], float32 running sums */
       AIE_CV_ST_COUNT };
/* float64 per-state sums over the episode's days (scenario_metrics, covid19_env.py:1613-1687) */
enum { AIE_CV_SUM_UNEMPLOYED = 0, AIE_CV_SUM_STRINGENCY, AIE_CV_SUM_PRODUCTIVITY, AIE_CV_SUM_SUBSIDY, AIE_CV_SUM_COUNT };
/* rows of the agent observation block */
enum { AIE_CV_OB_STATE = 0, AIE_CV_OB_PROD = 6, AIE_CV_OB_LAG = 7, AIE_CV_OB_TIME = 8, AIE_CV_OB_POLICY = 9,
       AIE_CV_OB_T_SUBSIDY = 10, AIE_CV_OB_SUBSIDY_LEVEL = 11, AIE_CV_OB_T_VACCINE = 12, AIE_CV_OB_MASK = 13 };

#define AIE_CV_GROUP 2            /* history chunks fetched per prefetch group                 */
#define AIE_CV_EVENT_CAP 64       /* level changes per state a replica's event list holds (multiple of 4)      */
#ifndef AIE_CV_WIN_WAVES
#define AIE_CV_WIN_WAVES 8        /* replicas per workgroup of the window-sum kernel (they share the LDS tap table) */
#endif
#define AIE_CV_TAP_PAD_FRONT 16   /* zero rows before tap 0 in the filter-tap table            */
#define AIE_CV_TAP_PAD_BACK (16 * (AIE_CV_GROUP + 3))

static inline void aie__add_shared(aie_tensor_table* tt, const char* name, int dtype, int64_t off,
                                   int nd, int64_t d0, int64_t d1) {
  aie__add(tt, name, dtype, off, 0, nd, d0, d1, 0, 0, 1);
}

static inline int aie__build_covid(const aie_config* c, aie_params* p, aie_tensor_table* tt,
                                   char* err, size_t errlen) {
  const aie_covid_config* v = &c->covid;
  if (c->n_agents > AIE_MAX_AGENTS)
    AIE__FAIL("n_agents = %d: the COVID scenario holds at most %d states per replica here (one lane each)", c->n_agents, AIE_MAX_AGENTS);
  if (c->multi_action_mode_agents || c->multi_action_mode_planner)
    AIE__FAIL("the COVID scenario uses single-action mode for agents and planner");
  static const int want[3] = {AIE_COMP_COVID_CONTROL, AIE_COMP_COVID_SUBSIDY, AIE_COMP_COVID_VACCINE};
  if (c->n_components != 3) AIE__FAIL("the COVID scenario needs exactly its three components");
  if (c->dense_log_replicas != 0) AIE__FAIL("dense logs are not available for the COVID scenario");
  /* each of the three exactly once, in any order: their steps touch disjoint state and the reference's rewards and
   * observations do not depend on the order (live reference, tests/test_covid_component_order.py); the fused kernel runs
   * them in the canonical one */
  for (int k = 0; k < 3; ++k) {
    int seen = 0;
    for (int i = 0; i < 3; ++i) seen += c->components[i] == want[k];
    if (seen != 1)
      AIE__FAIL("COVID components must be ControlUSStateOpenCloseStatus, FederalGovernmentSubsidy and VaccinationCampaign, each once");
  }
  if (v->num_stringency_levels < 1 || v->num_stringency_levels > 100) AIE__FAIL("num_stringency_levels out of range");
  if (v->beta_delay < 1 || v->beta_delay > 4096) AIE__FAIL("beta_delay out of range");
  if (v->filter_len < 1 || v->filter_len > 65536) AIE__FAIL("filter_len out of range");
  if (v->num_filters < 1 || v->num_filters > AIE_COVID_MAX_FILTERS) AIE__FAIL("num_filters out of range");
  if (v->filter_len < v->beta_delay) AIE__FAIL("filter_len < beta_delay is not supported");
  /* the window sums keep a state's level changes as events `history day | delta << 16` (aie_kernels_covid.hip): the day
   * index runs to filter_len + episode_length and must fit 16 bits (the reference has no such bound) */
  if (!v->filter_recurrence && (int64_t)v->filter_len + (int64_t)c->episode_length > 65535)
    AIE__FAIL("filter_len + episode_length = %lld: the reference-exact window sums index history days with 16 bits "
              "(<= 65535); use filter_recurrence or a shorter episode", (long long)v->filter_len + c->episode_length);
  if (v->action_cooldown_period < 1) AIE__FAIL("action_cooldown_period must be >= 1 (covid19_components.py:57)");
  if (v->subsidy_interval < 1) AIE__FAIL("subsidy_interval must be >= 1 (covid19_components.py:278)");
  if (v->num_subsidy_levels < 1 || v->num_subsidy_levels > 255) AIE__FAIL("num_subsidy_levels out of range");
  if (v->delivery_interval < 1) AIE__FAIL("delivery_interval must be >= 1 (covid19_components.py:505)");
  if (v->time_when_vaccine_delivery_begins < 0) AIE__FAIL("vaccine delivery must not begin before the start date");
  if (!(v->economic_reward_crra_eta >= 0.0)) AIE__FAIL("economic_reward_crra_eta must be >= 0");
  if (v->economic_reward_crra_eta == 1.0) AIE__FAIL("economic_reward_crra_eta == 1 divides by zero (covid19_env.py:1074)");
  if (!(v->reward_normalization_factor != 0.0)) AIE__FAIL("reward_normalization_factor must be non-zero");
  if (v->replay_data && !v->replay_policies) AIE__FAIL("replay_data (use_real_world_data) needs replay_policies (covid19_env.py:126-135)");
  if (v->filter_recurrence)
    for (int f = 0; f < v->num_filters; ++f)
      if (!(v->filter_decay[f] > 0.0 && v->filter_decay[f] < 1.0)) AIE__FAIL("filter_decay[%d] must be in (0, 1)", f);

  p->c = *c;
  p->E = c->n_envs;
  p->n = c->n_agents;
  p->H = c->world_h; p->W = c->world_w; p->HW = p->H * p->W;
  const int n = p->n;
  p->cv_L = v->filter_len;
  p->cv_F = v->num_filters;
  p->cv_NL = v->num_stringency_levels;
  p->cv_NS = v->num_subsidy_levels;
  /* + slack: the last prefetch group of the last step reads (zero-tap) chunks past day T */
  p->cv_nch = (v->filter_len + 1 + c->episode_length + 15) / 16 + AIE_CV_GROUP + 3;
  p->cv_row = (int32_t)aie__align(16 * n, 64);
  p->cv_nrow_obs = AIE_CV_OB_MASK + 1 + p->cv_NL;
  {
    int t = v->time_when_vaccine_delivery_begins;        /* covid19_components.py:640-646 */
    while (t % v->delivery_interval != 0) t++;
    p->cv_t_first_delivery = t;
  }
  p->A = 1 + p->cv_NL;  p->MA = p->A;  p->act_a_width = 1;  p->n_sub_a = 1;
  p->sub_a_dim[0] = p->cv_NL; p->sub_a_base[0] = 1;
  p->n_sub_p = 1; p->sub_p_dim = p->cv_NS; p->MP = 1 + p->cv_NS; p->act_p_width = 1;
  p->planner_acts = 1;

  int32_t cur = 0;
  p->cv_pitch = n;
  p->o_cv_state = aie__rec(&cur, 4 * n * AIE_CV_ST_COUNT, 256);
  p->o_cv_cooldown = aie__rec(&cur, 4 * n, 4);
  p->o_cv_sums = aie__rec(&cur, 8 * n * AIE_CV_SUM_COUNT, 8);
  p->o_cv_acc = aie__rec(&cur, 8 * n * v->num_filters, 8);  /* recurrence: A_t; window sums: the coming step's sums over the days before its own */
  p->o_cv_ring = aie__rec(&cur, 32 * 64, 256);
  p->cv_ev_groups = v->filter_recurrence ? 0 : AIE_CV_EVENT_CAP / 4;
  p->o_cv_ev_ht = aie__rec(&cur, p->cv_ev_groups ? 4 * n : 0, 4);
  p->o_cv_dense = aie__rec(&cur, 4, 4);
  p->o_cv_tail_pending = aie__rec(&cur, 4, 4);
  p->o_cv_subsidy_level = aie__rec(&cur, 4, 4);
  p->o_cv_p_index = aie__rec(&cur, 8, 4);
  p->o_timestep = aie__rec(&cur, 4, 4);
  p->o_completions = aie__rec(&cur, 4, 4);
  p->o_sample_t = aie__rec(&cur, 4, 4);
  p->o_rew_slot = aie__rec(&cur, 4, 4);
  p->o_rew_epoch = aie__rec(&cur, 4, 4);
  p->rec_bytes = (int32_t)aie__align(cur, 256);

  const int64_t E = p->E;
  const int64_t no = (int64_t)p->cv_nrow_obs * n * 4, po = (int64_t)(4 + p->MP) * 4;
  int64_t a = 0;
  p->a_records = a;    a = aie__align(a + E * (int64_t)p->rec_bytes, 256);
  p->a_cv_consts = a;  a = aie__align(a + (int64_t)AIE_CV_K_COUNT * 64 * 8, 256);
  p->a_cv_filters = a;
  a = aie__align(a + (int64_t)(AIE_CV_TAP_PAD_FRONT + p->cv_L + AIE_CV_TAP_PAD_BACK) * p->cv_F * 8, 256);
  p->a_cv_hist0 = a;   a = aie__align(a + (int64_t)(p->cv_L + 1) * n, 256);
  p->a_cv_hist0c = a;  a = aie__align(a + (int64_t)p->cv_nch * p->cv_row, 256);
  p->a_cv_acc0 = a;    a = aie__align(a + (int64_t)AIE_COVID_MAX_FILTERS * 64 * 8, 256);
  p->a_cv_lag_obs = a; a = aie__align(a + (int64_t)v->beta_delay * n, 256);
  p->a_cv_replay_a = p->a_cv_replay_p = p->a_cv_replay_state = 0;
  if (v->replay_policies) {
    p->a_cv_replay_a = a; a = aie__align(a + (int64_t)c->episode_length * 64, 256);
    p->a_cv_replay_p = a; a = aie__align(a + (int64_t)c->episode_length * 4, 256);
  }
  if (v->replay_data) {
    p->a_cv_replay_state = a; a = aie__align(a + (int64_t)6 * (c->episode_length + 1) * 64 * 8, 256);
  }
  p->a_cv_hist = a;    a = aie__align(a + E * (int64_t)p->cv_nch * p->cv_row, 256);
  p->a_cv_ev0 = a;     a = aie__align(a + (int64_t)p->cv_ev_groups * 1024 + 256 + 16, 256);
  p->a_cv_events = a;  a = aie__align(a + E * (int64_t)p->cv_ev_groups * 1024, 256);
  p->a_cv_obs_a = a;   a = aie__align(a + E * no, 256);
  p->a_cv_obs_p = a;   a = aie__align(a + E * po, 256);
  p->a_rew_a = a; a = aie__align(a + E * n * 4, 256);
  p->a_rew_p = a; a = aie__align(a + E * 4, 256);
  p->a_done = a;  a = aie__align(a + E, 256);
  p->arena_bytes = a;

  if (tt) {
    const int64_t rs = p->rec_bytes, r0 = p->a_records;
    static const char* st_name[AIE_CV_ST_COUNT] = {"susceptible", "infected", "recovered", "deaths", "vaccinated",
                                                   "unemployed", "postsubsidy_productivity", "subsidy",
                                                   "health_index", "economic_index"};
    static const char* sum_name[AIE_CV_SUM_COUNT] = {"sum_unemployed", "sum_stringency_level",
                                                     "sum_postsubsidy_productivity", "sum_subsidy"};
    for (int k = 0; k < AIE_CV_SUM_COUNT; ++k)
      aie__add(tt, sum_name[k], AIE_F64, r0 + p->o_cv_sums + 8 * n * k, rs, 1, n, 0, 0, 0, E);
    aie__add(tt, "planner_health_economic_index", AIE_F32, r0 + p->o_cv_p_index, rs, 1, 2, 0, 0, 0, E);
    /* recurrence: each filter's discounted delta sum A_t; window sums: each filter's sum over the NEXT step's window
     * without that step's own day (formed at the end of a step, when the registers are free; the step adds its day) */
    aie__add(tt, v->filter_recurrence ? "filter_discounted_delta_sums" : "filter_window_sums_before_today", AIE_F64,
             r0 + p->o_cv_acc, rs, 2, p->cv_F, n, 0, 0, E);
    tt->t[tt->n - 1].stride[1] = 8 * n;
    for (int k = 0; k < AIE_CV_ST_COUNT; ++k)
      aie__add(tt, st_name[k], AIE_F32, r0 + p->o_cv_state + 4 * n * k, rs, 1, n, 0, 0, 0, E);
    aie__add(tt, "cooldown_until", AIE_I32, r0 + p->o_cv_cooldown, rs, 1, n, 0, 0, 0, E);
    aie__add(tt, "subsidy_level", AIE_I32, r0 + p->o_cv_subsidy_level, rs, 0, 0, 0, 0, 0, E);
    aie__add(tt, "timestep", AIE_I32, r0 + p->o_timestep, rs, 0, 0, 0, 0, 0, E);
    aie__add(tt, "completions", AIE_I32, r0 + p->o_completions, rs, 0, 0, 0, 0, 0, E);
    aie__add(tt, "sample_t", AIE_I32, r0 + p->o_sample_t, rs, 0, 0, 0, 0, 0, E);
    aie__add(tt, "rew_log_slot", AIE_I32, r0 + p->o_rew_slot, rs, 0, 0, 0, 0, 0, E);
    /* stringency level of the 32 most recent days: row (filter_len + day) & 31 (always current) */
    aie__add(tt, "stringency_ring", AIE_U8, r0 + p->o_cv_ring, rs, 2, 32, n, 0, 0, E);
    tt->t[tt->n - 1].stride[1] = 64;
    /* stringency level on day (16*chunk + j - filter_len) of the episode, per state.  With filter_recurrence a chunk
     * is written when its 16th day is over (from the ring); without, every day (the window sums stream it) */
    if (p->cv_ev_groups) {
      /* level-change events per state: list positions [head, tail) are inside the current filter window */
      aie__add(tt, "stringency_change_head_tail", AIE_I32, r0 + p->o_cv_ev_ht, rs, 1, n, 0, 0, 0, E);
      aie__add(tt, "stringency_change_events", AIE_U32, p->a_cv_events, (int64_t)p->cv_ev_groups * 1024, 3, p->cv_ev_groups,
               n, 4, 0, E);
      tt->t[tt->n - 1].stride[1] = 1024;
      tt->t[tt->n - 1].stride[2] = 16;
    }
    aie__add(tt, "window_streams_whole_history", AIE_I32, r0 + p->o_cv_dense, rs, 0, 0, 0, 0, 0, E);
    aie__add(tt, "stringency_history_chunks", AIE_U8, p->a_cv_hist, (int64_t)p->cv_nch * p->cv_row, 3,
             p->cv_nch, n, 16, 0, E);
    tt->t[tt->n - 1].stride[1] = p->cv_row;

    static const char* k_name[AIE_CV_K_CONV_W0] = {
        "model_us_state_population", "model_beta_slopes", "model_beta_intercepts", "model_unemployment_bias",
        "model_maximum_productivity", "model_agents_health_norm", "model_agents_economic_norm",
        "model_min_marginal_agent_health_index", "model_max_marginal_agent_health_index",
        "model_min_marginal_agent_economic_index", "model_max_marginal_agent_economic_index",
        "model_weightage_on_marginal_agent_health_index", "model_weightage_on_marginal_agent_economic_index",
        "model_max_daily_subsidy_per_state", "model_num_vaccines_per_delivery",
        "model_susceptible_0", "model_infected_0", "model_recovered_0", "model_deaths_0", "model_unemployed_0",
        "model_vaccinated_0"};
    for (int k = 0; k < AIE_CV_K_CONV_W0; ++k)
      aie__add_shared(tt, k_name[k], AIE_F64, p->a_cv_consts + (int64_t)k * 512, 1, n, 0);
    aie__add_shared(tt, "model_conv_weights", AIE_F64, p->a_cv_consts + (int64_t)AIE_CV_K_CONV_W0 * 512, 2, p->cv_F, n);
    tt->t[tt->n - 1].stride[1] = 512;
    /* stored tap-major ([L][F]) so that the F taps of one day, and 16 days, are contiguous;
     * zero rows before and after (the arena is zero-initialised) */
    aie__add_shared(tt, "model_unemp_conv_filters", AIE_F64,
                    p->a_cv_filters + (int64_t)AIE_CV_TAP_PAD_FRONT * p->cv_F * 8, 2, p->cv_F, p->cv_L);
    tt->t[tt->n - 1].stride[1] = 8;
    tt->t[tt->n - 1].stride[2] = 8 * (int64_t)p->cv_F;
    aie__add_shared(tt, "model_stringency_level_history_0", AIE_U8, p->a_cv_hist0, 2, p->cv_L + 1, n);
    aie__add_shared(tt, "model_policy_before_start_obs", AIE_U8, p->a_cv_lag_obs, 2, v->beta_delay, n);
    if (v->replay_policies) {
      aie__add_shared(tt, "replay_stringency_policy", AIE_U8, p->a_cv_replay_a, 2, c->episode_length, n);
      tt->t[tt->n - 1].stride[1] = 64;
      aie__add_shared(tt, "replay_subsidy_level", AIE_I32, p->a_cv_replay_p, 1, c->episode_length, 0);
    }
    if (v->replay_data) {
      aie__add(tt, "replay_state", AIE_F64, p->a_cv_replay_state, 0, 3, 6, c->episode_length + 1, n, 0, 1);
      tt->t[tt->n - 1].stride[1] = (int64_t)(c->episode_length + 1) * 64 * 8;
      tt->t[tt->n - 1].stride[2] = 64 * 8;
    }

#define OBA(name, row, nd, d0) aie__add(tt, name, AIE_F32, p->a_cv_obs_a + (int64_t)(row) * n * 4, no, nd, d0, (nd) == 2 ? n : 0, 0, 0, E)
    OBA("obs_a_world-agent_state", AIE_CV_OB_STATE, 2, 6);
    OBA("obs_a_world-agent_postsubsidy_productivity", AIE_CV_OB_PROD, 1, n);
    OBA("obs_a_world-lagged_stringency_level", AIE_CV_OB_LAG, 1, n);
    OBA("obs_a_time", AIE_CV_OB_TIME, 1, n);
    OBA("obs_a_ControlUSStateOpenCloseStatus-agent_policy_indicators", AIE_CV_OB_POLICY, 1, n);
    OBA("obs_a_FederalGovernmentSubsidy-t_until_next_subsidy", AIE_CV_OB_T_SUBSIDY, 1, n);
    OBA("obs_a_FederalGovernmentSubsidy-current_subsidy_level", AIE_CV_OB_SUBSIDY_LEVEL, 1, n);
    OBA("obs_a_VaccinationCampaign-t_until_next_vaccines", AIE_CV_OB_T_VACCINE, 1, n);
    OBA("obs_a_action_mask", AIE_CV_OB_MASK, 2, 1 + p->cv_NL);
    /* the planner sees the same per-state arrays (covid19_env.py:986-992, covid19_components.py:167-168) */
    OBA("obs_p_world-agent_state", AIE_CV_OB_STATE, 2, 6);
    OBA("obs_p_world-agent_postsubsidy_productivity", AIE_CV_OB_PROD, 1, n);
    OBA("obs_p_world-lagged_stringency_level", AIE_CV_OB_LAG, 1, n);
    OBA("obs_p_ControlUSStateOpenCloseStatus-agent_policy_indicators", AIE_CV_OB_POLICY, 1, n);
#undef OBA
#define OBP(name, idx, nd, d0) aie__add(tt, name, AIE_F32, p->a_cv_obs_p + 4 * (idx), po, nd, d0, 0, 0, 0, E)
    OBP("obs_p_time", 0, 1, 1);
    OBP("obs_p_FederalGovernmentSubsidy-t_until_next_subsidy", 1, 0, 0);
    OBP("obs_p_FederalGovernmentSubsidy-current_subsidy_level", 2, 0, 0);
    OBP("obs_p_VaccinationCampaign-t_until_next_vaccines", 3, 0, 0);
    OBP("obs_p_action_mask", 4, 1, p->MP);
#undef OBP
    aie__add(tt, "rewards_a", AIE_F32, p->a_rew_a, (int64_t)n * 4, 1, n, 0, 0, 0, E);
    aie__add(tt, "rewards_p", AIE_F32, p->a_rew_p, 4, 0, 0, 0, 0, 0, E);
    aie__add(tt, "done", AIE_U8, p->a_done, 1, 0, 0, 0, 0, 0, E);
  }
  return AIE_OK;
}

/* per-replica episode accumulators (see aie_params.a_metrics): allocation + tensor views */
static inline void aie__alloc_metrics(aie_params* p, int64_t* a) {
  const int n = p->n;
  int32_t m = 0;
  p->mo_tax_sched = m;  m += 8 * (p->has_tax ? p->NB : 0);
  p->mo_tax_income = m; m += 8 * (p->has_tax ? n : 0);
  p->mo_tax_paid = m;   m += 8 * (p->has_tax ? n : 0);
  p->mo_tax_eff = m;    m += 8;
  p->mo_cda = m;        m += 4 * (p->has_cda ? 2 * AIE_N_RES * n * 2 : 0);
  p->mo_tax_occ = m;    m += 4 * (p->has_tax ? p->NB : 0);
  p->mo_tax_days = m;   m += 4;
  p->met_bytes = (int32_t)aie__align(m, 16);
  p->a_metrics = *a;
  *a = aie__align(*a + (int64_t)p->E * p->met_bytes, 256);
}
#define AIE_SAEZ_OFF_ELAS 16
#define AIE_SAEZ_OFF_AVG (AIE_SAEZ_OFF_ELAS + 32)
#define AIE_SAEZ_OFF_NEXT (AIE_SAEZ_OFF_AVG + 8 * AIE_MAX_BRACKETS)
#define AIE_SAEZ_OFF_BUF (AIE_SAEZ_OFF_NEXT + 8 * AIE_MAX_BRACKETS)
static inline void aie__alloc_saez(const aie_config* c, aie_params* p, int64_t* a) {
  p->a_saez = 0; p->saez_stride = 0; p->saez_cap = 0; p->a_saez_global = 0; p->saez_global_cap = 0;
  if (!p->has_tax || c->tax_model != AIE_TAX_SAEZ) return;
  p->saez_cap = c->saez_buffer_size + p->n; /* a tax day appends n pairs before the oldest are dropped */
  p->saez_stride = (int32_t)aie__align(AIE_SAEZ_OFF_BUF + (int64_t)p->saez_cap * 16, 64);
  p->a_saez = *a;
  *a = aie__align(*a + (int64_t)p->E * p->saez_stride, 256);
  p->saez_global_cap = c->saez_global_capacity > 0 ? c->saez_global_capacity : 0;
  p->a_saez_global = *a;
  *a = aie__align(*a + 16 + (int64_t)p->saez_global_cap * 16, 256);
  const double top = c->tax_bracket_cutoffs[p->NB - 1], step = top / (double)AIE_SAEZ_BINS;
  for (int i = 0; i <= AIE_SAEZ_BINS; ++i) p->saez_edges[i] = (double)i * step + 0.0; /* np.linspace */
  p->saez_edges[AIE_SAEZ_BINS] = top;
}
static inline void aie__add_saez_tensors(const aie_params* p, aie_tensor_table* tt) {
  if (!p->saez_stride) return;
  const int64_t s0 = p->a_saez, ss = p->saez_stride, E = p->E;
  aie__add(tt, "saez_buffer_len", AIE_I32, s0, ss, 0, 0, 0, 0, 0, E);
  aie__add(tt, "saez_reached_min_samples", AIE_I32, s0 + 4, ss, 0, 0, 0, 0, 0, E);
  aie__add(tt, "saez_additions", AIE_I32, s0 + 8, ss, 0, 0, 0, 0, 0, E);  /* _additions_this_episode :541 */
  aie__add_shared(tt, "saez_global_len", AIE_I32, p->a_saez_global, 1, 1, 0);
  if (p->saez_global_cap)
    aie__add_shared(tt, "saez_global_buffer", AIE_F64, p->a_saez_global + 16, 2, p->saez_global_cap, 2);
  aie__add(tt, "saez_elas", AIE_F64, s0 + AIE_SAEZ_OFF_ELAS, ss, 1, 4, 0, 0, 0, E);
  aie__add(tt, "saez_running_avg_tax_rates", AIE_F64, s0 + AIE_SAEZ_OFF_AVG, ss, 1, p->NB, 0, 0, 0, E);
  aie__add(tt, "saez_next_rates", AIE_F64, s0 + AIE_SAEZ_OFF_NEXT, ss, 1, p->NB, 0, 0, 0, E);
  aie__add(tt, "saez_buffer", AIE_F64, s0 + AIE_SAEZ_OFF_BUF, ss, 2, p->saez_cap, 2, 0, 0, E);
}

/* dense-log event rows (include/aie.h: AIE_EV_*): at most n builds, 2n gathers, NB + n tax rows
 * and one trade per resting order of a commodity in a step */
static inline void aie__alloc_events(const aie_config* c, aie_params* p, int64_t* a) {
  const int n = p->n;
  p->ev_replicas = c->dense_log_replicas;
  p->ev_cap = p->ev_stride = 0;
  p->a_events = 0;
  if (p->ev_replicas <= 0) return;
  p->ev_cap = 4 * n + (p->has_tax ? p->NB : 0) + (p->has_cda ? AIE_N_RES * n * c->cda_max_num_orders : 0);
  p->ev_stride = (int32_t)aie__align(16 + (int64_t)p->ev_cap * AIE_EV_WORDS * 4, 64);
  p->a_events = *a;
  *a = aie__align(*a + (int64_t)p->ev_replicas * p->ev_stride, 256);
}
static inline void aie__add_event_tensors(const aie_params* p, aie_tensor_table* tt) {
  if (p->ev_replicas <= 0) return;
  aie__add(tt, "log_event_count", AIE_I32, p->a_events, p->ev_stride, 0, 0, 0, 0, 0, p->ev_replicas);
  aie__add(tt, "log_events", AIE_I32, p->a_events + 16, p->ev_stride, 2, p->ev_cap, AIE_EV_WORDS, 0, 0,
           p->ev_replicas);
}
static inline void aie__add_metrics_tensors(const aie_params* p, aie_tensor_table* tt) {
  const int64_t ms = p->met_bytes, m0 = p->a_metrics, E = p->E;
  const int n = p->n;
  if (p->has_cda) aie__add(tt, "metrics_cda_trades", AIE_I32, m0 + p->mo_cda, ms, 4, 2, AIE_N_RES, n, 2, E);
  if (p->has_tax) {
    aie__add(tt, "metrics_tax_schedule_sum", AIE_F64, m0 + p->mo_tax_sched, ms, 1, p->NB, 0, 0, 0, E);
    aie__add(tt, "metrics_tax_income_sum", AIE_F64, m0 + p->mo_tax_income, ms, 1, n, 0, 0, 0, E);
    aie__add(tt, "metrics_tax_paid_sum", AIE_F64, m0 + p->mo_tax_paid, ms, 1, n, 0, 0, 0, E);
    aie__add(tt, "metrics_tax_effective_rate_sum", AIE_F64, m0 + p->mo_tax_eff, ms, 0, 0, 0, 0, 0, E);
    aie__add(tt, "metrics_tax_bracket_occupancy", AIE_I32, m0 + p->mo_tax_occ, ms, 1, p->NB, 0, 0, 0, E);
    aie__add(tt, "metrics_tax_days", AIE_I32, m0 + p->mo_tax_days, ms, 0, 0, 0, 0, 0, E);
  }
}

/* one-step-economy (F/scenarios/one_step_economy/one_step_economy.py): no map, agents
 * hold coin / labor / skill / production; flat observations in sorted-key order:
 *   agent   PeriodicBracketTax-{curr_rates,is_first_day,is_tax_day,last_incomes,
 *           marginal_rate,tax_phase}, SimpleLabor-skill, time
 *   planner PeriodicBracketTax-{curr_rates,is_first_day,is_tax_day,last_incomes,tax_phase},
 *           time, world-equality, world-normalized_per_capita_productivity
 *   p{i}    PeriodicBracketTax-{curr_marginal_rate,last_income,last_marginal_rate}      */
static inline int aie__build_one_step_economy(const aie_config* c, aie_params* p, aie_tensor_table* tt) {
  const int n = p->n;
  int f = 0;
  p->fa_tax = f;   if (p->has_tax) f += p->NB + n + 4;
  p->fa_labor = f; if (p->has_labor) f += 1;
  p->fa_time = f;  f += 1;
  p->fa_world = f;
  p->FA = f;
  f = 0;
  p->fp_tax = f;   if (p->has_tax) f += p->NB + n + 3;
  p->fp_time = f;  f += 1;
  p->fp_world = f; f += 2;
  p->FP = f;
  p->fpa_tax = 0;
  p->fpa_world = p->has_tax ? 3 : 0;
  p->FPA = p->has_tax ? 3 : 0;
  p->mg_FA = aie__magic(p->FA);
  p->mg_MA = aie__magic(p->MA);

  int32_t cur = 0;
  p->o_inv_coin = aie__rec(&cur, 8 * n, 16);
  p->o_labor = aie__rec(&cur, 8 * n, 8);
  p->o_production = aie__rec(&cur, 8 * n, 8);
  p->o_util = aie__rec(&cur, 8 * (n + 1), 8);
  if (p->has_tax) {
    p->o_tax_last_coin = aie__rec(&cur, 8 * n, 8);
    p->o_tax_last_income = aie__rec(&cur, 8 * n, 8);
    p->o_tax_last_marginal_rate = aie__rec(&cur, 8 * n, 8);
    /* aie_kernels_ose.hip (ose_load_record) skips [o_tax_last_income, o_tax_last_marginal_rate + 8 n) as one dead range
     * when every step is a tax day: the two fields have to stay adjacent */
    if (p->o_tax_last_marginal_rate != p->o_tax_last_income + 8 * n) return AIE_E_INVALID;
    p->o_tax_total_collected = aie__rec(&cur, 8, 8);
    p->o_tax_cycle_pos = aie__rec(&cur, 4, 4);
    p->o_tax_last_completions = aie__rec(&cur, 4, 4);
    p->o_tax_rate_idx = aie__rec(&cur, 4 * p->NB, 4);
    if (c->tax_model == AIE_TAX_SAEZ) {
      p->o_tax_saez_rates = aie__rec(&cur, 8 * p->NB, 8);
      p->o_tax_saez_obs_rates = aie__rec(&cur, 8 * p->NB, 8);
    }
  }
  p->o_timestep = aie__rec(&cur, 4, 4);
  p->o_completions = aie__rec(&cur, 4, 4);
  p->o_auto_warmup = aie__rec(&cur, 4, 4);
  p->o_first_step = aie__rec(&cur, 4, 4);
  p->o_error_flags = aie__rec(&cur, 4, 4);
  p->o_sample_t = aie__rec(&cur, 4, 4);
  p->o_rew_slot = aie__rec(&cur, 4, 4);
  p->o_rew_epoch = aie__rec(&cur, 4, 4);
  p->o_mt_gauss = aie__rec(&cur, 8, 8);
  p->o_mt_pos = aie__rec(&cur, 4, 4);
  p->o_mt_has_gauss = aie__rec(&cur, 4, 4);
  p->o_mt = aie__rec(&cur, 4 * aie__rng_state_words(c), 16);
  /* behind the generator: per-agent fields a step only reads (SimpleLabor's skills; the escrow account, which no
   * component of this scenario moves) -- the kernels keep them in registers, they are not part of the LDS image
   * (everything before o_mt), which is what decides how many replicas a CU holds at once */
  p->o_skill = aie__rec(&cur, 8 * n, 16);
  p->o_esc_coin = aie__rec(&cur, 8 * n, 8);
  p->rec_bytes = (int32_t)aie__align(cur, 16);

  const int64_t E = p->E;
  int64_t a = 0;
  p->a_records = a;      a = aie__align(a + E * (int64_t)p->rec_bytes, 256);
  p->a_obs_a_flat = a;   a = aie__align(a + E * n * p->FA * 4, 256);
  p->a_obs_a_mask = a;   a = aie__align(a + E * n * p->MA * 4, 256);
  p->a_obs_a_time = a;   a = aie__align(a + E * n * 4, 256);
  p->a_obs_p_flat = a;   a = aie__align(a + E * p->FP * 4, 256);
  p->a_obs_p_mask = a;   a = aie__align(a + E * p->MP * 4, 256);
  p->a_obs_p_time = a;   a = aie__align(a + E * 4, 256);
  p->a_obs_p_agents = a; a = aie__align(a + E * n * (p->FPA ? p->FPA : 1) * 4, 256);
  p->a_rew_a = a; a = aie__align(a + E * n * 4, 256);
  p->a_rew_p = a; a = aie__align(a + E * 4, 256);
  p->a_done = a;  a = aie__align(a + E, 256);
  aie__alloc_metrics(p, &a);
  aie__alloc_events(c, p, &a);
  aie__alloc_saez(c, p, &a);
  p->arena_bytes = a;

  if (tt) {
    const int64_t rs = p->rec_bytes, r0 = p->a_records;
    aie__add_metrics_tensors(p, tt);
    aie__add_event_tensors(p, tt);
    aie__add_saez_tensors(p, tt);
#define REC(name, dt, off, nd, d0) aie__add(tt, name, dt, r0 + (off), rs, nd, d0, 0, 0, 0, E)
    REC("inv_coin", AIE_F64, p->o_inv_coin, 1, n);
    REC("esc_coin", AIE_F64, p->o_esc_coin, 1, n);
    REC("labor", AIE_F64, p->o_labor, 1, n);
    REC("skill", AIE_F64, p->o_skill, 1, n);
    REC("production", AIE_F64, p->o_production, 1, n);
    REC("util", AIE_F64, p->o_util, 1, n + 1);
    if (p->has_tax) {
      REC("tax_cycle_pos", AIE_I32, p->o_tax_cycle_pos, 0, 0);
      REC("tax_last_completions", AIE_I32, p->o_tax_last_completions, 0, 0);
      REC("tax_rate_idx", AIE_I32, p->o_tax_rate_idx, 1, p->NB);
      if (c->tax_model == AIE_TAX_SAEZ) {
        REC("tax_saez_bracket_rates", AIE_F64, p->o_tax_saez_rates, 1, p->NB);
        REC("tax_saez_observed_rates", AIE_F64, p->o_tax_saez_obs_rates, 1, p->NB);
      }
      REC("tax_last_coin", AIE_F64, p->o_tax_last_coin, 1, n);
      REC("tax_last_income", AIE_F64, p->o_tax_last_income, 1, n);
      REC("tax_last_marginal_rate", AIE_F64, p->o_tax_last_marginal_rate, 1, n);
      REC("tax_total_collected", AIE_F64, p->o_tax_total_collected, 0, 0);
    }
    REC("timestep", AIE_I32, p->o_timestep, 0, 0);
    REC("completions", AIE_I32, p->o_completions, 0, 0);
    REC("sample_t", AIE_I32, p->o_sample_t, 0, 0);
    REC("rew_log_slot", AIE_I32, p->o_rew_slot, 0, 0);
    REC("labor_first_step", AIE_I32, p->o_first_step, 0, 0);
    REC("error_flags", AIE_I32, p->o_error_flags, 0, 0);
    REC("mt", AIE_U32, p->o_mt, 1, aie__rng_state_words(&p->c));
    REC("mt_pos", AIE_I32, p->o_mt_pos, 0, 0);
    REC("mt_has_gauss", AIE_I32, p->o_mt_has_gauss, 0, 0);
    REC("mt_gauss", AIE_F64, p->o_mt_gauss, 0, 0);
#undef REC
#define DENSE(name, dt, off, nd, d0, d1)                                                    \
  do {                                                                                      \
    int64_t dd[2] = {d0, d1};                                                               \
    int64_t es = aie__dtype_size(dt);                                                       \
    for (int q = 0; q < nd; ++q) es *= dd[q];                                               \
    aie__add(tt, name, dt, off, es, nd, d0, d1, 0, 0, E);                                   \
  } while (0)
    DENSE("obs_a_flat", AIE_F32, p->a_obs_a_flat, 2, n, p->FA);
    DENSE("obs_a_action_mask", AIE_F32, p->a_obs_a_mask, 2, n, p->MA);
    DENSE("obs_a_time", AIE_F32, p->a_obs_a_time, 2, n, 1);
    DENSE("obs_p_flat", AIE_F32, p->a_obs_p_flat, 1, p->FP, 0);
    DENSE("obs_p_action_mask", AIE_F32, p->a_obs_p_mask, 1, p->MP, 0);
    DENSE("obs_p_time", AIE_F32, p->a_obs_p_time, 1, 1, 0);
    if (p->FPA) DENSE("obs_p_agents", AIE_F32, p->a_obs_p_agents, 2, n, p->FPA);
    DENSE("rewards_a", AIE_F32, p->a_rew_a, 1, n, 0);
    DENSE("rewards_p", AIE_F32, p->a_rew_p, 0, 0, 0);
    DENSE("done", AIE_U8, p->a_done, 0, 0, 0);
#undef DENSE
  }
  (void)c;
  return AIE_OK;
}

/* Validates the config (mirrors the reference's constructor asserts) and derives
 * every dimension / offset.  `tt` may be NULL. */
static inline int aie_build_params(const aie_config* c, aie_params* p, aie_tensor_table* tt,
                                   char* err, size_t errlen) {
  memset(p, 0, sizeof(*p));
  if (tt) tt->n = 0;
  if (c->abi_version != AIE_ABI_VERSION) AIE__FAIL("abi_version %d != %d", c->abi_version, AIE_ABI_VERSION);
  if (c->n_envs < 1) AIE__FAIL("n_envs must be >= 1");
  if (c->n_agents < 2) AIE__FAIL("n_agents must be >= 2 (base_env.py:223)");
  if (c->scenario != AIE_SCN_GTB && c->scenario != AIE_SCN_ONE_STEP_ECONOMY && c->scenario != AIE_SCN_COVID)
    AIE__FAIL("unknown scenario %d", c->scenario);
  if (c->rng_mode != AIE_RNG_NUMPY && c->rng_mode != AIE_RNG_FAST) AIE__FAIL("unknown rng_mode %d", c->rng_mode);
  if (c->scenario == AIE_SCN_COVID) {
    if (c->episode_length < 1) AIE__FAIL("episode_length must be >= 1 (base_env.py:254)");
    return aie__build_covid(c, p, tt, err, errlen);
  }
  if (c->scenario == AIE_SCN_GTB && c->n_agents > AIE_MAX_AGENTS - 2)
    AIE__FAIL("n_agents = %d: the spatial scenarios hold at most %d mobile agents per replica here (one lane of a 64-lane "
              "wavefront each; the reference has no upper bound, base_env.py:221-224)", c->n_agents, AIE_MAX_AGENTS - 2);
  if (c->scenario == AIE_SCN_ONE_STEP_ECONOMY && c->n_agents > AIE_MAX_AGENTS_WIDE)
    AIE__FAIL("n_agents = %d: one-step-economy holds at most %d agents per replica here (the reference has no upper "
              "bound, base_env.py:221-224)", c->n_agents, AIE_MAX_AGENTS_WIDE);
  if (c->world_h < 1 || c->world_w < 1 || c->world_h > 255 || c->world_w > 255)
    AIE__FAIL("world_size out of range");
  if (c->episode_length < 1) AIE__FAIL("episode_length must be >= 1 (base_env.py:254)");
  if (c->dense_log_replicas < 0 || c->dense_log_replicas > c->n_envs)
    AIE__FAIL("dense_log_replicas must be in [0, n_envs]");
  if (c->n_components < 0 || c->n_components > AIE_MAX_COMPONENTS) AIE__FAIL("bad n_components");
  for (int i = 0; i < c->n_components; ++i) {
    int k = c->components[i];
    if (k < AIE_COMP_BUILD || (k > AIE_COMP_SIMPLE_LABOR && k != AIE_COMP_WEALTH_REDISTRIBUTION))
      AIE__FAIL("unknown component id %d", k);
    if (c->scenario == AIE_SCN_GTB && k == AIE_COMP_SIMPLE_LABOR)
      AIE__FAIL("SimpleLabor is only supported with the one-step-economy scenario");
    if (c->scenario == AIE_SCN_ONE_STEP_ECONOMY && k != AIE_COMP_SIMPLE_LABOR && k != AIE_COMP_TAX &&
        k != AIE_COMP_WEALTH_REDISTRIBUTION)
      AIE__FAIL("one-step-economy supports SimpleLabor, PeriodicBracketTax and WealthRedistribution only");
    for (int j = 0; j < i; ++j)
      if (c->components[j] == k) AIE__FAIL("component %d listed twice", k);
  }
  const int gtb = c->scenario == AIE_SCN_GTB;
  if (gtb && (c->obs_range < 0 || c->obs_range > 15)) AIE__FAIL("mobile_agent_observation_range out of range");
  for (int r = 0; gtb && r < AIE_N_RES; ++r) {
    if (c->regen_halfwidth[r] < 0 || c->regen_halfwidth[r] > 3)
      AIE__FAIL("regen_halfwidth must be in [0, 3] (dynamic_layout.py:152-153)");
    if (!(c->regen_weight[r] >= 0.0 && c->regen_weight[r] <= 1.0)) AIE__FAIL("regen_weight not in [0,1]");
    if (c->max_health[r] < 1 || c->max_health[r] > 255) AIE__FAIL("max_health out of range");
  }
  if (c->layout_gen != AIE_LAYOUT_FIXED) {
    if (!gtb) AIE__FAIL("layout_gen needs a gather-trade-build scenario");
    if (c->layout_gen < 0 || c->layout_gen > AIE_LAYOUT_MULTI_ZONE) AIE__FAIL("unknown layout_gen %d", c->layout_gen);
    if (c->shared_layout) AIE__FAIL("generated layouts are per replica: shared_layout must be 0");
    if (c->world_h * c->world_w > 4096) {  /* (the generator's planes live in LDS: 18 B per cell beside the record image) */
      if (err) snprintf(err, errlen, "layouts are generated on the device for worlds of up to 4096 cells (64 x 64)");
      return AIE_E_UNSUPPORTED;
    }
    for (int r = 0; r < AIE_N_RES; ++r) {
      if (!(c->layout_coverage[r] > 0.0 && c->layout_coverage[r] < 1.0)) AIE__FAIL("layout_coverage not in (0, 1)");
      if (!(c->layout_clump[r] > 0.0 && c->layout_clump[r] <= 1.0)) AIE__FAIL("layout_clump not in (0, 1]");
    }
    if (c->layout_gen == AIE_LAYOUT_MULTI_ZONE) {
      const int regions = c->mz_rows * c->mz_cols, zones = c->mz_zones[0] + c->mz_zones[1] + c->mz_zones[2];
      if (c->mz_rows < 1 || c->mz_cols < 1 || regions > 256 || zones > regions || c->mz_zones[0] < 0 ||
          c->mz_zones[1] < 0 || c->mz_zones[2] < 0)
        AIE__FAIL("multi_zone: partitions / zone counts out of range");
    }
  }
  if (!(c->starting_agent_coin >= 0.0)) AIE__FAIL("starting_agent_coin must be >= 0");
  if (!(c->isoelastic_eta >= 0.0 && c->isoelastic_eta <= 1.0)) AIE__FAIL("isoelastic_eta not in [0,1]");
  if (gtb && !(c->energy_cost >= 0.0)) AIE__FAIL("energy_cost must be >= 0");
  if (gtb && !(c->energy_warmup_constant >= 0.0)) AIE__FAIL("energy_warmup_constant must be >= 0");
  if (!(c->mixing_weight_gini_vs_coin >= 0.0 && c->mixing_weight_gini_vs_coin <= 1.0))
    AIE__FAIL("mixing_weight_gini_vs_coin not in [0,1]");

  p->c = *c;
  p->sh_energy_warmup = c->energy_warmup_constant > 0.0 ? 1 : 0;
  p->sh_eta_is_one = c->isoelastic_eta == 1.0 ? 1 : 0;
  p->E = c->n_envs;
  p->n = c->n_agents;
  p->H = c->world_h;
  p->W = c->world_w;
  p->HW = c->world_h * c->world_w;
  p->has_build = aie__has(c, AIE_COMP_BUILD);
  p->has_cda = aie__has(c, AIE_COMP_CDA);
  p->has_gather = aie__has(c, AIE_COMP_GATHER);
  p->has_tax = aie__has(c, AIE_COMP_TAX);
  p->has_labor = aie__has(c, AIE_COMP_SIMPLE_LABOR);
  if (p->has_labor) {
    if (c->labor_num_hours < 1 || c->labor_num_hours > 1000) AIE__FAIL("SimpleLabor.num_labor_hours out of range");
    if (!(c->labor_pmsm > 0.0)) AIE__FAIL("SimpleLabor.payment_max_skill_multiplier must be > 0");
  }
  if (!gtb) {
    if (c->ose_agent_reward_type == AIE_AGENT_REW_COIN_MINUS_LABOR_COST && !(c->ose_labor_exponent > 1.0))
      AIE__FAIL("labor_exponent must be > 1 (rewards.py:69)");
    if (c->ose_agent_reward_type != AIE_AGENT_REW_COIN_MINUS_LABOR_COST && c->ose_agent_reward_type != AIE_AGENT_REW_ISOELASTIC)
      AIE__FAIL("unknown agent_reward_type");
  }
  p->CM = c->has_water ? 6 : 5;
  p->WV = 2 * c->obs_range + 1;

  if (p->has_build) {
    if (c->build_payment < 0) AIE__FAIL("Build.payment must be >= 0");
    if (c->build_payment_max_skill_multiplier < 1) AIE__FAIL("Build.payment_max_skill_multiplier must be >= 1");
    if (!(c->build_labor >= 0.0)) AIE__FAIL("Build.build_labor must be >= 0");
    if (c->build_skill_dist < 0 || c->build_skill_dist > 2) AIE__FAIL("Build.skill_dist invalid");
  }
  if (p->has_gather) {
    if (!(c->move_labor >= 0.0) || !(c->collect_labor >= 0.0)) AIE__FAIL("Gather labor must be >= 0");
    if (c->gather_skill_dist < 0 || c->gather_skill_dist > 2) AIE__FAIL("Gather.skill_dist invalid");
  }
  if (c->split_water_line) {
    if (c->scenario != AIE_SCN_GTB) AIE__FAIL("split_water_line only applies to the gather-trade-build scenarios");
    if (c->split_water_line <= 0 || c->split_water_line >= c->world_h - 1) AIE__FAIL("water_row out of range (layout_from_file.py:722)");
    if (c->fixed_four_skill_and_loc) AIE__FAIL("The split layout scenario does not support fixed_four_skill_and_loc (layout_from_file.py:712-716)");
    if (!(p->has_build && c->build_skill_dist == AIE_SKILL_PARETO)) AIE__FAIL("split layout requires Build with skill_dist='pareto' (layout_from_file.py:748)");
    if (!c->has_water) AIE__FAIL("split layout needs the Water landmark");
  }
  if (c->fixed_four_skill_and_loc && !(p->has_build && c->build_skill_dist == AIE_SKILL_PARETO))
    AIE__FAIL("fixed_four_skill_and_loc requires Build with skill_dist='pareto' (layout_from_file.py:177-178)");
  if (p->has_cda) {
    if (c->cda_max_bid_ask < 1 || c->cda_max_bid_ask > 126) AIE__FAIL("CDA.max_bid_ask out of range");
    if (c->cda_order_duration < 1 || c->cda_order_duration > 32000) AIE__FAIL("CDA.order_duration out of range");
    if (c->cda_max_num_orders < 1 || c->cda_max_num_orders > 255) AIE__FAIL("CDA.max_num_orders out of range");
    if (!(c->cda_order_labor >= 0.0)) AIE__FAIL("CDA.order_labor must be >= 0");
    p->P = c->cda_max_bid_ask + 1;
    p->M = c->n_agents * c->cda_max_num_orders;
  }
  if (p->has_tax) {
    if (c->tax_period < 1) AIE__FAIL("Tax.period must be > 0");
    if (c->tax_n_brackets < 2 || c->tax_n_brackets > AIE_MAX_BRACKETS) AIE__FAIL("Tax.n_brackets out of range");
    if (c->tax_model < AIE_TAX_MODEL_WRAPPER || c->tax_model > AIE_TAX_SAEZ) AIE__FAIL("unknown tax_model");
    if (c->tax_model == AIE_TAX_SAEZ) {
      if (c->saez_buffer_size < 1 || c->saez_buffer_size > 4096) AIE__FAIL("saez_buffer_size out of range");
      if (!(c->tax_rate_min >= 0.0 && c->tax_rate_min <= c->tax_rate_max)) AIE__FAIL("rate_min / rate_max");
      if (c->saez_fixed_elas_given && !(c->saez_fixed_elas >= 0.0)) AIE__FAIL("saez_fixed_elas must be >= 0");
    }
    if (c->tax_bracket_cutoffs[0] != 0.0) AIE__FAIL("bracket_cutoffs[0] must be 0 (redistribution.py:243)");
    if (c->tax_model == AIE_TAX_MODEL_WRAPPER && !c->tax_disable) {
      if (c->tax_n_disc_rates < 2 || c->tax_n_disc_rates > AIE_MAX_RATES) AIE__FAIL("Tax.n_disc_rates out of range");
      p->planner_acts = 1;
    } else if (c->tax_model == AIE_TAX_MODEL_WRAPPER) {
      if (c->tax_n_disc_rates < 1 || c->tax_n_disc_rates > AIE_MAX_RATES) AIE__FAIL("Tax.n_disc_rates out of range");
    }
    p->NB = c->tax_n_brackets;
  }

  /* ---- action subspaces, registration order = component order ----------------- */
  int ns = 0, base = 1;
  for (int i = 0; i < c->n_components; ++i) {
    switch (c->components[i]) {
      case AIE_COMP_BUILD:
        p->sub_a_slot[ns] = AIE_SUB_BUILD; p->sub_a_dim[ns] = 1; ns++; break;
      case AIE_COMP_CDA:
        for (int r = 0; r < AIE_N_RES; ++r) {
          p->sub_a_slot[ns] = r ? AIE_SUB_BUY1 : AIE_SUB_BUY0; p->sub_a_dim[ns] = p->P; ns++;
          p->sub_a_slot[ns] = r ? AIE_SUB_SELL1 : AIE_SUB_SELL0; p->sub_a_dim[ns] = p->P; ns++;
        }
        break;
      case AIE_COMP_GATHER:
        p->sub_a_slot[ns] = AIE_SUB_GATHER; p->sub_a_dim[ns] = 4; ns++; break;
      case AIE_COMP_SIMPLE_LABOR:
        p->sub_a_slot[ns] = AIE_SUB_LABOR; p->sub_a_dim[ns] = c->labor_num_hours; ns++; break;
      default: break;
    }
  }
  p->n_sub_a = ns;
  for (int s = 0; s < ns; ++s) { p->sub_a_base[s] = base; base += p->sub_a_dim[s]; }
  p->A = base;
  if (c->multi_action_mode_agents) {
    if (ns == 0) { p->act_a_width = 1; }      /* PassiveAgentPlaceholder, base_agent.py:158-161 */
    else p->act_a_width = ns;
  } else {
    p->act_a_width = 1;
  }
  p->n_sub_p = p->planner_acts ? p->NB : 0;
  p->sub_p_dim = p->planner_acts ? c->tax_n_disc_rates : 0;
  if (c->multi_action_mode_planner) p->act_p_width = p->n_sub_p ? p->n_sub_p : 1;
  else p->act_p_width = 1;

  /* ---- flattened masks (base_agent.py:440-460) -------------------------------- */
  if (c->multi_action_mode_agents) {
    p->MA = 0;
    for (int s = 0; s < ns; ++s) p->MA += 1 + p->sub_a_dim[s];
    if (ns == 0) p->MA = 1;
  } else {
    p->MA = p->A;
  }
  if (c->multi_action_mode_planner) p->MP = p->n_sub_p ? p->n_sub_p * (1 + p->sub_p_dim) : 1;
  else p->MP = 1 + p->n_sub_p * p->sub_p_dim;
  if (p->MA > AIE_MAX_MASK) AIE__FAIL("flattened action mask too long (%d > %d)", p->MA, AIE_MAX_MASK);
  {
    /* mask bits (kernel): 0 build | 1..4 move L,R,U,D | 5,6 sell Stone,Wood | 8..15, 16..23:
     * number of affordable bid prices for Stone, Wood */
    const int multi = c->multi_action_mode_agents ? 1 : 0;
    int m = 0;
    if (!multi) p->mask_test[m++] = 0;                                 /* leading NO-OP: always 1 */
    for (int sub = 0; sub < ns; ++sub) {
      if (multi) p->mask_test[m++] = 0;                                /* the subspace's NO-OP   */
      for (int l = 0; l < p->sub_a_dim[sub]; ++l) {
        uint32_t sh = 0, msk = 1, thr = 1;
        switch (p->sub_a_slot[sub]) {
          case AIE_SUB_BUILD: sh = 0; break;
          case AIE_SUB_GATHER: sh = 1 + (uint32_t)l; break;
          case AIE_SUB_SELL0: sh = 5; break;
          case AIE_SUB_SELL1: sh = 6; break;
          case AIE_SUB_BUY0: sh = 8; msk = 0xff; thr = (uint32_t)l + 1; break;
          case AIE_SUB_BUY1: sh = 16; msk = 0xff; thr = (uint32_t)l + 1; break;
          default: msk = 0; thr = 0; break;                            /* unmasked subspace      */
        }
        p->mask_test[m++] = sh | (msk << 8) | (thr << 16);
      }
    }
    if (ns == 0 && multi) p->mask_test[m++] = 0;
  }

  /* ---- flat observations: sorted keys ----------------------------------------- */
  {
    int f = 0;
    p->fa_build = f;  if (p->has_build) f += 2;
    p->fa_cda = f;    if (p->has_cda) f += 10 * p->P + 2;
    p->fa_gather = f; if (p->has_gather) f += 1;
    p->fa_tax = f;    if (p->has_tax) f += p->NB + p->n + 4;
    p->fa_time = f;   f += 1;
    /* inventory-Coin,-Stone,-Wood, loc-col, loc-row; with full_observability the location is
     * only conveyed through the maps (layout_from_file.py:466-472) */
    p->fa_world = f;  f += c->full_observability ? 3 : 5;
    p->FA = f;
    f = 0;
    p->fp_cda = f;    if (p->has_cda) f += 6 * p->P + 2;
    p->fp_tax = f;    if (p->has_tax) f += p->NB + p->n + 3;
    p->fp_time = f;   f += 1;
    p->fp_world = f;  f += 3;
    p->FP = f;
    f = 0;
    p->fpa_tax = f;   if (p->has_tax) f += 3;
    /* the scenario's per-agent planner fragments only exist with egocentric observations (:508-515) */
    p->fpa_world = f; if (!c->full_observability) f += 3 + (c->planner_gets_spatial_info ? 2 : 0);
    p->FPA = f;
  }

  p->mg_WV2 = aie__magic((int64_t)p->WV * p->WV);
  p->mg_WV = aie__magic(p->WV);
  p->mg_MA = aie__magic(p->MA);
  p->mg_FA = aie__magic(p->FA);
  p->mg_P = aie__magic(p->P);
  p->mg_2P = aie__magic(2 * p->P);
  p->mg_taxA = aie__magic(p->NB + p->n + 4);
  p->mg_HW = aie__magic(p->HW);
  p->mg_W = aie__magic(p->W);
  p->mg_sub_p = aie__magic(1 + p->sub_p_dim);
  p->mg_sub_p_dim = aie__magic(p->sub_p_dim > 0 ? p->sub_p_dim : 1);

  if (!gtb) return aie__build_one_step_economy(c, p, tt);

  /* ---- per-replica record ------------------------------------------------------ */
  const int n = p->n, HW = p->HW, R = AIE_N_RES;
  int32_t cur = 0;
  p->o_cells = aie__rec(&cur, 4 * HW, 16);
  p->regen_conv = c->regen_halfwidth[0] > 0 || c->regen_halfwidth[1] > 0;
  p->regen_general = (c->regen_halfwidth[0] > 0 && c->max_health[0] > 1) || (c->regen_halfwidth[1] > 0 && c->max_health[1] > 1);
  if (p->regen_conv) p->o_regen_count = aie__rec(&cur, AIE_N_RES * HW, 16);
  for (int r = 0; r < AIE_N_RES; ++r) {
    /* convolve2d accumulates sum += health * kernel over the window: m equal terms k, in sequence */
    const int d = 1 + 2 * c->regen_halfwidth[r];
    const double k = c->regen_weight[r] / (double)(d * d);
    p->regen_p[r][0] = 0.0;
    for (int m = 1; m < 50; ++m) p->regen_p[r][m] = p->regen_p[r][m - 1] + k;
  }
  p->o_inv_coin = aie__rec(&cur, 8 * n, 8);
  p->o_esc_coin = aie__rec(&cur, 8 * n, 8);
  p->o_labor = aie__rec(&cur, 8 * n, 8);
  p->o_build_payment = aie__rec(&cur, 8 * n, 8);
  p->o_build_skill = aie__rec(&cur, 8 * n, 8);
  p->o_bonus_gather_prob = aie__rec(&cur, 8 * n, 8);
  p->o_util = aie__rec(&cur, 8 * (n + 1), 8);
  p->o_loc_r = aie__rec(&cur, 4 * n, 4);
  p->o_loc_c = aie__rec(&cur, 4 * n, 4);
  p->o_inv_res = aie__rec(&cur, 4 * R * n, 4);
  p->o_esc_res = aie__rec(&cur, 4 * R * n, 4);
  if (p->has_cda) {
    p->o_cda_price_history = aie__rec(&cur, 8 * R * n * p->P, 8);
    p->o_cda_n_bids = aie__rec(&cur, 4 * R, 4);
    p->o_cda_n_asks = aie__rec(&cur, 4 * R, 4);
    p->o_cda_bids = aie__rec(&cur, 4 * R * p->M, 4);
    p->o_cda_asks = aie__rec(&cur, 4 * R * p->M, 4);
    p->o_cda_n_orders = aie__rec(&cur, 4 * R * n, 4);
    p->o_cda_bid_hist = aie__rec(&cur, R * n * p->P, 4);
    p->o_cda_ask_hist = aie__rec(&cur, R * n * p->P, 4);
  }
  if (p->has_tax) {
    p->o_tax_last_coin = aie__rec(&cur, 8 * n, 8);
    p->o_tax_last_income = aie__rec(&cur, 8 * n, 8);
    p->o_tax_last_marginal_rate = aie__rec(&cur, 8 * n, 8);
    p->o_tax_total_collected = aie__rec(&cur, 8, 8);
    p->o_tax_cycle_pos = aie__rec(&cur, 4, 4);
    p->o_tax_last_completions = aie__rec(&cur, 4, 4);
    p->o_tax_rate_idx = aie__rec(&cur, 4 * p->NB, 4);
    if (c->tax_model == AIE_TAX_SAEZ) {
      p->o_tax_saez_rates = aie__rec(&cur, 8 * p->NB, 8);
      p->o_tax_saez_obs_rates = aie__rec(&cur, 8 * p->NB, 8);
    }
  }
  p->o_timestep = aie__rec(&cur, 4, 4);
  p->o_completions = aie__rec(&cur, 4, 4);
  p->o_auto_warmup = aie__rec(&cur, 4, 4);
  p->o_obs_valid = aie__rec(&cur, 4, 4);
  p->o_error_flags = aie__rec(&cur, 4, 4);
  p->o_sample_t = aie__rec(&cur, 4, 4);
  p->o_rew_slot = aie__rec(&cur, 4, 4);
  p->o_rew_epoch = aie__rec(&cur, 4, 4);
  p->o_mask_bits = aie__rec(&cur, 4 * n, 4);
  p->o_mask_p_open = aie__rec(&cur, 4, 4);
  p->o_mt_gauss = aie__rec(&cur, 8, 8);
  p->o_mt_pos = aie__rec(&cur, 4, 4);
  p->o_mt_has_gauss = aie__rec(&cur, 4, 4);
  p->o_mt = aie__rec(&cur, 4 * aie__rng_state_words(c), 16);
  p->o_src_n = p->o_src_list = 0;
  if (aie__record_src_list(c)) {  /* behind the generator's state: not part of the LDS image (everything before o_mt) */
    p->o_src_n = aie__rec(&cur, 16, 16);
    p->o_src_list = aie__rec(&cur, 2 * AIE_SRC_CAP, 16);
  }
  p->rec_bytes = (int32_t)aie__align(cur, 16);

  /* ---- arena ------------------------------------------------------------------- */
  const int64_t E = p->E;
  int64_t a = 0;
  p->a_records = a; a = aie__align(a + E * (int64_t)p->rec_bytes, 256); /* (0: the step kernel's record copies rely on it) */
  p->am_ch = c->full_observability ? p->CM : p->CM + 1;
  p->am_h = c->full_observability ? p->H : p->WV;
  p->am_w = c->full_observability ? p->W : p->WV;
  const int64_t wv2 = (int64_t)p->am_h * p->am_w;
  p->a_obs_a_map = a;  a = aie__align(a + E * n * p->am_ch * wv2 * 4, 256);
  p->a_obs_a_idx = a;  a = aie__align(a + E * n * 2 * wv2 * 2, 256);
  p->a_obs_a_flat = a; a = aie__align(a + E * n * p->FA * 4, 256);
  p->a_obs_a_mask = a; a = aie__align(a + E * n * p->MA * 4, 256);
  p->a_obs_a_time = a; a = aie__align(a + E * n * 4, 256);
  if (c->planner_gets_spatial_info) {
    p->a_obs_p_map = a; a = aie__align(a + E * p->CM * HW * 4, 256);
    p->a_obs_p_idx = a; a = aie__align(a + E * 2 * HW * 2, 256);
  }
  p->a_obs_p_flat = a;   a = aie__align(a + E * p->FP * 4, 256);
  p->a_obs_p_mask = a;   a = aie__align(a + E * p->MP * 4, 256);
  p->a_obs_p_time = a;   a = aie__align(a + E * 4, 256);
  p->a_obs_p_agents = a; a = aie__align(a + E * n * (p->FPA ? p->FPA : 1) * 4, 256);
  p->a_rew_a = a; a = aie__align(a + E * n * 4, 256);
  p->a_rew_p = a; a = aie__align(a + E * 4, 256);
  p->a_done = a;  a = aie__align(a + E, 256);
  aie__alloc_metrics(p, &a);
  aie__alloc_events(c, p, &a);
  aie__alloc_saez(c, p, &a);
  p->a_layout_prob = a;
  if (c->layout_gen != AIE_LAYOUT_FIXED) a = aie__align(a + (int64_t)AIE_N_RES * HW * 8, 256);
  p->layout_stage_stride = (HW + 15) / 16 * 16;
  p->layout_pad_ = 0;
  p->a_layout_stage = p->a_layout_tag = p->a_layout_ctl = a;
  if (aie__layout_staged(c)) {
    p->a_layout_stage = a; a = aie__align(a + E * p->layout_stage_stride, 256);
    p->a_layout_tag = a;   a = aie__align(a + E * 8, 256);
    p->a_layout_ctl = a;   a = aie__align(a + 16, 256);
  }
  p->a_src_list = a;
  if (aie__shared_src_list(c)) a = aie__align(a + 16 + 2 * AIE_SRC_CAP, 256);
  p->arena_bytes = a;

  /* ---- tensor table ------------------------------------------------------------ */
  if (tt) {
    const int64_t rs = p->rec_bytes, r0 = p->a_records;
#define REC(name, dt, off, nd, d0, d1, d2) aie__add(tt, name, dt, r0 + (off), rs, nd, d0, d1, d2, 0, E)
    REC("cells", AIE_U32, p->o_cells, 2, p->H, p->W, 0);
    /* byte-plane views into the packed cell words (element stride 4) */
    {
      static const char* nm[4] = {"stone", "wood", "house_owner", "cell_flags"};
      for (int b = 0; b < 4; ++b) {
        REC(nm[b], b == 2 ? AIE_I8 : AIE_U8, p->o_cells + b, 2, p->H, p->W, 0);
        aie_tensor_desc* d = &tt->t[tt->n - 1];
        d->stride[2] = 4;
        d->stride[1] = 4 * (int64_t)p->W;
      }
    }
    if (p->regen_conv) REC("regen_source_count", AIE_U8, p->o_regen_count, 3, R, p->H, p->W);
    REC("loc_r", AIE_I32, p->o_loc_r, 1, n, 0, 0);
    REC("loc_c", AIE_I32, p->o_loc_c, 1, n, 0, 0);
    REC("inv_res", AIE_I32, p->o_inv_res, 2, R, n, 0);
    REC("esc_res", AIE_I32, p->o_esc_res, 2, R, n, 0);
    REC("inv_coin", AIE_F64, p->o_inv_coin, 1, n, 0, 0);
    REC("esc_coin", AIE_F64, p->o_esc_coin, 1, n, 0, 0);
    REC("labor", AIE_F64, p->o_labor, 1, n, 0, 0);
    REC("build_payment", AIE_F64, p->o_build_payment, 1, n, 0, 0);
    REC("build_skill", AIE_F64, p->o_build_skill, 1, n, 0, 0);
    REC("bonus_gather_prob", AIE_F64, p->o_bonus_gather_prob, 1, n, 0, 0);
    REC("util", AIE_F64, p->o_util, 1, n + 1, 0, 0);
    if (p->has_cda) {
      REC("cda_n_bids", AIE_I32, p->o_cda_n_bids, 1, R, 0, 0);
      REC("cda_n_asks", AIE_I32, p->o_cda_n_asks, 1, R, 0, 0);
      REC("cda_bids", AIE_I32, p->o_cda_bids, 2, R, p->M, 0);
      REC("cda_asks", AIE_I32, p->o_cda_asks, 2, R, p->M, 0);
      REC("cda_n_orders", AIE_I32, p->o_cda_n_orders, 2, R, n, 0);
      REC("cda_bid_hist", AIE_U8, p->o_cda_bid_hist, 3, R, n, p->P);
      REC("cda_ask_hist", AIE_U8, p->o_cda_ask_hist, 3, R, n, p->P);
      REC("cda_price_history", AIE_F64, p->o_cda_price_history, 3, R, n, p->P);
    }
    if (p->has_tax) {
      REC("tax_cycle_pos", AIE_I32, p->o_tax_cycle_pos, 0, 0, 0, 0);
    REC("tax_last_completions", AIE_I32, p->o_tax_last_completions, 0, 0, 0, 0);
      REC("tax_rate_idx", AIE_I32, p->o_tax_rate_idx, 1, p->NB, 0, 0);
      if (c->tax_model == AIE_TAX_SAEZ) {
        REC("tax_saez_bracket_rates", AIE_F64, p->o_tax_saez_rates, 1, p->NB, 0, 0);
        REC("tax_saez_observed_rates", AIE_F64, p->o_tax_saez_obs_rates, 1, p->NB, 0, 0);
      }
      REC("tax_last_coin", AIE_F64, p->o_tax_last_coin, 1, n, 0, 0);
      REC("tax_last_income", AIE_F64, p->o_tax_last_income, 1, n, 0, 0);
      REC("tax_last_marginal_rate", AIE_F64, p->o_tax_last_marginal_rate, 1, n, 0, 0);
      REC("tax_total_collected", AIE_F64, p->o_tax_total_collected, 0, 0, 0, 0);
    }
    REC("timestep", AIE_I32, p->o_timestep, 0, 0, 0, 0);
    REC("completions", AIE_I32, p->o_completions, 0, 0, 0, 0);
    REC("sample_t", AIE_I32, p->o_sample_t, 0, 0, 0, 0);
    REC("rew_log_slot", AIE_I32, p->o_rew_slot, 0, 0, 0, 0);
    REC("auto_warmup", AIE_I32, p->o_auto_warmup, 0, 0, 0, 0);
    REC("obs_valid", AIE_I32, p->o_obs_valid, 0, 0, 0, 0);
    REC("error_flags", AIE_I32, p->o_error_flags, 0, 0, 0, 0);
    if (p->o_src_list) {
      REC("regen_src_n", AIE_I32, p->o_src_n, 0, 0, 0, 0);
      REC("regen_src_list", AIE_I16, p->o_src_list, 1, AIE_SRC_CAP, 0, 0);
    }
    REC("mt", AIE_U32, p->o_mt, 1, aie__rng_state_words(&p->c), 0, 0);
    REC("mt_pos", AIE_I32, p->o_mt_pos, 0, 0, 0, 0);
    REC("mt_has_gauss", AIE_I32, p->o_mt_has_gauss, 0, 0, 0, 0);
    REC("mt_gauss", AIE_F64, p->o_mt_gauss, 0, 0, 0, 0);
#undef REC
#define DENSE(name, dt, off, nd, d0, d1, d2, d3)                                            \
  do {                                                                                      \
    int64_t dd[4] = {d0, d1, d2, d3};                                                       \
    int64_t es = aie__dtype_size(dt);                                                       \
    for (int q = 0; q < nd; ++q) es *= dd[q];                                               \
    aie__add(tt, name, dt, off, es, nd, d0, d1, d2, d3, E);                                 \
  } while (0)
    DENSE("obs_a_world-map", AIE_F32, p->a_obs_a_map, 4, n, p->am_ch, p->am_h, p->am_w);
    DENSE("obs_a_world-idx_map", AIE_I16, p->a_obs_a_idx, 4, n, 2, p->am_h, p->am_w);
    DENSE("obs_a_flat", AIE_F32, p->a_obs_a_flat, 2, n, p->FA, 0, 0);
    DENSE("obs_a_action_mask", AIE_F32, p->a_obs_a_mask, 2, n, p->MA, 0, 0);
    DENSE("obs_a_time", AIE_F32, p->a_obs_a_time, 2, n, 1, 0, 0);
    if (c->planner_gets_spatial_info) {
      DENSE("obs_p_world-map", AIE_F32, p->a_obs_p_map, 3, p->CM, p->H, p->W, 0);
      DENSE("obs_p_world-idx_map", AIE_I16, p->a_obs_p_idx, 3, 2, p->H, p->W, 0);
    }
    DENSE("obs_p_flat", AIE_F32, p->a_obs_p_flat, 1, p->FP, 0, 0, 0);
    DENSE("obs_p_action_mask", AIE_F32, p->a_obs_p_mask, 1, p->MP, 0, 0, 0);
    DENSE("obs_p_time", AIE_F32, p->a_obs_p_time, 1, 1, 0, 0, 0);
    if (p->FPA) DENSE("obs_p_agents", AIE_F32, p->a_obs_p_agents, 2, n, p->FPA, 0, 0);
    DENSE("rewards_a", AIE_F32, p->a_rew_a, 1, n, 0, 0, 0);
    DENSE("rewards_p", AIE_F32, p->a_rew_p, 0, 0, 0, 0, 0);
    DENSE("done", AIE_U8, p->a_done, 0, 0, 0, 0, 0);
    aie__add_metrics_tensors(p, tt);
    aie__add_event_tensors(p, tt);
    aie__add_saez_tensors(p, tt);
    if (c->layout_gen != AIE_LAYOUT_FIXED)
      aie__add_shared(tt, "layout_source_prob", AIE_F64, p->a_layout_prob, 2, AIE_N_RES, HW);
    if (aie__layout_staged(c)) aie__add_shared(tt, "layout_stage_ctl", AIE_I32, p->a_layout_ctl, 1, 4, 0);
#undef DENSE
  }
  return AIE_OK;
}

/* Counter-based RNG of the benchmark's synthetic uniform random policy
 * (SURVEY.md 8(d)): splitmix64 finaliser over (seed, global replica id, t, slot). */
/* annealed_tax_limit, F/components/utils.py:10-56 */
AIE_HD static inline double aie_annealed_tax_limit(int completions, double warmup, double slope, double final_max) {
  double pv = slope * ((double)completions - warmup);
  pv = pv < 1.0 ? pv : 1.0;
  pv = pv > 0.0 ? pv : 0.0;
  return pv * final_max;
}

/* ---- the policy sampler (aie_sample_policy_actions; round 6: inverse CDF in float32) ---------------------------------
 * A row (one action slot: `len` entries k = 0 .. len - 1 with logits x_k, of which some are allowed: mask > 0.5 and x_k
 * not a NaN) is sampled as
 *     M = max of the allowed x_k;  w_k = aie_sampler_expf(x_k - M) for allowed k, 0 otherwise (and for k >= len);
 *     c_k = inclusive prefix sums of w in the FIXED order below;  T = the total in that order;
 *     u = aie_sampler_uniform(aie_sampler_entry_rng(base, slot)) in (0, 1), 23 bits;
 *     choice = the first allowed k with c_k > u T  (the last allowed k if rounding leaves none; NO-OP if nothing is allowed),
 * base = one 64-bit counter hash per replica and call (aie_counter_rng), slot = the row's index in the replica.  Every
 * operation is an IEEE float32 add / multiply / fused multiply-add (fmaf: exact, rounded once) in a fixed order -- no
 * libm exponential, no table -- so the kernel, the CPU restatement (oracle/) and a Python transcription pick the same entry:
 *   * prefix sums: rows are cut into chunks of 64 entries; inside a chunk, over the entry's index r in its chunk and the
 *     row's segment size seg = aie_sampler_segment(len) (16, 32 or 64: what the kernel's cross-lane primitives span; 64
 *     for every chunk of a row of more than 64 entries):
 *     steps d = 1, 2, 4, 8: v_r <- v_r + v_{r-d} for (r mod 16) >= d (all r at once, inputs = the previous step's values);
 *     seg >= 32: v_r <- v_r + v_{16 (r div 16) - 1} for r div 16 odd;  seg = 64: v_r <- v_r + v_31 for r >= 32;
 *     c_k = carry + v_r;  the chunk's total = carry + v_{seg-1};  carry = 0 for the first chunk, then the previous total;
 *   * aie_sampler_expf(y), y <= 0: 0 at or below -80; n = rint(y log2(e)); r = fma(n, -ln2_lo, fma(n, -ln2_hi, y)); the
 *     Taylor polynomial of e^r through r^6 by Horner in fma (|r| <= 0.35: the first dropped term is 1.3e-7 of the sum);
 *     times 2^n (ldexp: exact, n >= -116).
 * Per entry: one exponential of 14 vector instructions, a row maximum and a prefix sum of 6 - 10 data-parallel-primitive
 * instructions each (DPP row rotates / shifts / broadcasts and gfx950's permlane swaps: no LDS), one comparison -- where
 * round 5's Gumbel-max spent two float64 logarithms (each with an IEEE division) and a 64-bit key per entry.  float32
 * resolves a probability to 2^-24 of the row's total; u itself has 23 bits. */
AIE_HD static inline uint32_t aie_sampler_entry_rng(uint32_t base, uint32_t k) {  /* "lowbias32", C. Wellons' hash prospector */
  uint32_t h = base + k * 0x9E3779B1u;
  h ^= h >> 16;
  h *= 0x7feb352du;
  h ^= h >> 15;
  h *= 0x846ca68bu;
  h ^= h >> 16;
  return h;
}
AIE_HD static inline float aie_sampler_uniform(uint32_t rnd) {  /* ((rnd >> 9) + 1/2) / 2^23: exact in float32 */
  return (float)(rnd >> 9) * 0x1p-23f + 0x1p-24f;
}
AIE_HD static inline float aie_sampler_expf(float y) {
  /* (the guard is a select at the end: one straight line of 14 instructions on the device) */
  const float n = rintf(y * 0x1.715476p+0f);      /* round half to even */
  float r = fmaf(n, -0x1.62e4p-1f, y);            /* ln 2 = 0x1.62e4p-1 + 0x1.7f7d1cp-20: the first product is exact */
  r = fmaf(n, -0x1.7f7d1cp-20f, r);
  float p = 0x1.6c16c2p-10f;                      /* 1/6! */
  p = fmaf(p, r, 0x1.111112p-7f);                 /* 1/5! */
  p = fmaf(p, r, 0x1.555556p-5f);                 /* 1/4! */
  p = fmaf(p, r, 0x1.555556p-3f);                 /* 1/3! */
  p = fmaf(p, r, 0.5f);
  p = fmaf(p, r, 1.0f);
  p = fmaf(p, r, 1.0f);
  const int ni = y > -80.0f ? (int)n : 0;
  return y > -80.0f ? ldexpf(p, ni) : 0.0f;
}
/* the aligned lane segment a row of `len` entries is scanned in, and how many such rows share a wavefront */
AIE_HD static inline int aie_sampler_segment(int len) { return len <= 16 ? 16 : len <= 32 ? 32 : 64; }
AIE_HD static inline int aie_sampler_rows_per_wave(int len) { return 64 / aie_sampler_segment(len); }
/* What the sampler kernel needs of the parameter block, as its kernel argument (aie_sampler_args_of fills it).  A group
 * is a replica's agent rows or its planner rows: row r's entry k has its logit at logits[e lg_estride + r lrs + k] and its
 * mask at the arena's float mk_off / 4 + e mk_estride + r mrs + k mks (COVID's collated agent masks: mrs 1, mks n). */
typedef struct aie_sampler_group {
  int64_t mk_off;
  uint32_t mk_estride, lg_estride;       /* (32 bits: a replica's rows x entries) */
  int32_t len, lrs, mrs, mks, lsh, rows; /* entries per row; strides; log2 of the lanes a row takes; rows per replica */
} aie_sampler_group;
typedef struct aie_sampler_args {
  aie_sampler_group agents, planner;
  int64_t t_off, rec_bytes;   /* replica e's draw index: the arena's int32 at t_off + e rec_bytes */
  const aie_params* params;   /* the device copy: read for multi-action agents only (rows of different lengths) */
  int32_t E, ragged, act_a_width, pad_;
} aie_sampler_args;
static inline aie_sampler_args aie_sampler_args_of(const aie_params* p, const aie_params* d_params) {
  aie_sampler_args S;
  memset(&S, 0, sizeof(S));
  const int covid = p->c.scenario == AIE_SCN_COVID;
  const int wa = covid ? 1 + p->cv_NL : p->MA; /* logits per agent, in the mask's own (flattened) layout */
  aie_sampler_group* A = &S.agents;
  aie_sampler_group* Q = &S.planner;
  S.ragged = p->c.multi_action_mode_agents != 0;
  A->len = wa;
  A->lsh = S.ragged ? 6 : (aie_sampler_segment(wa) == 16 ? 4 : aie_sampler_segment(wa) == 32 ? 5 : 6);
  A->rows = p->n * p->act_a_width;
  A->lrs = wa;
  A->lg_estride = (uint32_t)(p->n * wa);
  if (covid) {
    A->mk_off = p->a_cv_obs_a + 4 * (int64_t)AIE_CV_OB_MASK * p->n;
    A->mk_estride = (uint32_t)(p->cv_nrow_obs * p->n);
    A->mks = p->n;
    A->mrs = 1;
  } else {
    A->mk_off = p->a_obs_a_mask;
    A->mk_estride = (uint32_t)(p->n * p->MA);
    A->mks = 1;
    A->mrs = p->MA;
  }
  Q->len = p->c.multi_action_mode_planner ? (p->n_sub_p ? 1 + p->sub_p_dim : 1) : p->MP;
  Q->lsh = aie_sampler_segment(Q->len) == 16 ? 4 : aie_sampler_segment(Q->len) == 32 ? 5 : 6;
  Q->rows = p->act_p_width;
  Q->lrs = Q->mrs = p->c.multi_action_mode_planner ? 1 + p->sub_p_dim : p->MP; /* (a multi-action planner's rows: 1 + sub_p_dim apart) */
  Q->mks = 1;
  Q->lg_estride = (uint32_t)p->MP;
  Q->mk_off = covid ? p->a_cv_obs_p + 16 : p->a_obs_p_mask;
  Q->mk_estride = (uint32_t)(covid ? 4 + p->MP : p->MP);
  S.t_off = p->a_records + p->o_sample_t;
  S.rec_bytes = p->rec_bytes;
  S.params = d_params;
  S.E = p->E;
  S.act_a_width = p->act_a_width;
  return S;
}
AIE_HD static inline uint32_t aie_counter_rng(uint64_t seed, uint64_t env, uint64_t t, uint64_t slot) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (env + 1ull);
  z ^= (t + 1ull) * 0xBF58476D1CE4E5B9ull;
  z ^= (slot + 1ull) * 0x94D049BB133111EBull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (uint32_t)(z >> 32);
}

/* packed map cell */
#define AIE_CELL_WATER 1u
#define AIE_CELL_STONE_SRC 2u
#define AIE_CELL_WOOD_SRC 4u
#define AIE_CELL_STONE(w) ((w) & 0xffu)
#define AIE_CELL_WOOD(w) (((w) >> 8) & 0xffu)
#define AIE_CELL_OWNER(w) ((int)(int8_t)(((w) >> 16) & 0xffu))
#define AIE_CELL_FLAGS(w) (((w) >> 24) & 0xffu)
#define AIE_CELL_PACK(st, wd, own, fl) \
  ((uint32_t)(st) | ((uint32_t)(wd) << 8) | (((uint32_t)(own) & 0xffu) << 16) | ((uint32_t)(fl) << 24))

/* order word packing: agent | price << 8 | lifetime << 16 */
#define AIE_ORD_AGENT(o) ((o) & 0xff)
#define AIE_ORD_PRICE(o) (((o) >> 8) & 0xff)
#define AIE_ORD_LIFE(o) (((o) >> 16) & 0xffff)
#define AIE_ORD_PACK(a, p, l) ((int32_t)((a) | ((p) << 8) | ((l) << 16)))

#endif /* AIE_LAYOUT_H_ */
