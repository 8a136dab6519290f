/*
 * aie.h -- C ABI of the MI355X-native batched Foundation env.step().
 *
 * One `aie_env` owns E independent replicas of one Foundation environment on one
 * GPU.  The reference has no such seam for gather-trade-build; the closest thing it
 * has is the WarpDrive wrapper used by the COVID scenario, and every entry point below
 * cites the reference interface it stands in for (paths relative to the reference
 * tree, F/ = ai_economist/foundation/):
 *
 *   aie_create            F/__init__.py:16-18 (make_env_instance) +
 *                         F/base/base_env.py:178-366 (BaseEnvironment.__init__) +
 *                         F/env_wrapper.py:96-265 (FoundationEnvWrapper.__init__)
 *   aie_seed              F/base/base_env.py:481-494 (BaseEnvironment.seed ->
 *                         np.random.seed), one legacy MT19937 stream PER replica
 *   aie_seed_fast         the same seam for environments created with rng_mode = AIE_RNG_FAST: a counter-based
 *                         stream (Philox2x32-10) per replica instead of NumPy's MT19937 -- a throughput mode the
 *                         reference does not have (its trainers only ever call np.random.seed, base_env.py:481-494);
 *                         NOT stream-compatible with NumPy
 *   aie_set_rng_state     F/base/base_env.py:871-881 / 968-978 (seed_state injection)
 *   aie_reset             F/base/base_env.py:852-927 (reset) +
 *                         F/env_wrapper.py:267-353 (reset_all_envs/reset_only_done_envs)
 *   aie_step              F/base/base_env.py:929-1032 (step) +
 *                         F/env_wrapper.py:355-377 (step_all_envs)
 *   aie_get_tensor        F/env_wrapper.py:297-326 (get_data_dictionary /
 *                         get_tensor_dictionary -> CUDADataManager.push_data_to_device,
 *                         torch_accessible=True)
 *   aie_upload/_download  F/env_wrapper.py:291-329 (one-time host -> device push)
 *
 * Conventions: every function returns 0 on success or a negative AIE_E_* code;
 * aie_last_error() gives the message.  The library owns all state / observation /
 * reward buffers (one arena, optionally caller-provided so that it can be a
 * torch-owned allocation); the caller owns action buffers.  All kernels are enqueued
 * asynchronously on the caller's HIP stream; there are no hidden synchronisations
 * except in aie_upload/aie_download/aie_destroy.  Calls on one handle are not
 * thread-safe; distinct handles are independent.  No process-global RNG.
 */
#ifndef AIE_H_
#define AIE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: exactly the functions declared in this header are exported
 * (tests/test_cabi_symbols.py compares `nm -D` with this file). */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define AIE_ABI_VERSION 10

#define AIE_MAX_AGENTS 64      /* mobile agents per replica, spatial scenarios (one lane each) */
#define AIE_MAX_AGENTS_WIDE 128 /* mobile agents per replica, map-less one-step-economy        */
#define AIE_MAX_COMPONENTS 8
#define AIE_MAX_BRACKETS 16
#define AIE_MAX_RATES 64       /* discretised tax rates per bracket                     */
#define AIE_MAX_SUBSPACES 16   /* action subspaces per agent class                      */
#define AIE_N_RES 2            /* collectible resources, sorted: 0 = Stone, 1 = Wood    */
#define AIE_MT_N 624
/* aie_config.rng_mode: which generator feeds the replicas' np.random.* draws (agent orders, pick-up bonus, resource
 * regeneration, reset placement / skills / layouts).
 *   AIE_RNG_NUMPY  NumPy's legacy MT19937 stream per replica, bit for bit (the parity mode; the default).
 *   AIE_RNG_FAST   a counter-based stream per replica: 32-bit word number g of replica e is element g & 1 of
 *                  Philox2x32-10(counter = (lo32(g >> 1), hi32(g >> 1) | salt), key = key32) with
 *                  key32 = lo32(seed + e_global) and salt = (bits 32..47 of seed + e_global) << 16 (Salmon et al.,
 *                  "Parallel random numbers: as easy as 1, 2, 3", SC'11; the Random123 constants).  Every consumer of
 *                  the stream (53-bit doubles from two words, masked-rejection integers, Fisher-Yates permutations,
 *                  polar Gauss) is unchanged, so a run differs from the parity mode ONLY in the words drawn.  The
 *                  record carries 16 bytes of generator state instead of 2 496, and a step computes only the words it
 *                  uses (the resource regeneration addresses the words of its source cells directly instead of
 *                  advancing MT19937 through all 4 H W of them).  Checked bit for bit against oracle/'s restatement of
 *                  the same generator; not stream-compatible with NumPy.  COVID draws no random numbers: ignored there.
 *                  Scenarios that draw a new source layout at every reset (uniform/, quadrant/, multi_zone/) take the
 *                  layout of a replica's k-th reset (k = 0, 1, ... since seeding) from a stream of its own -- the same
 *                  function with key32 + 0x9E3779B9 (k >> 15) and the counter's high word salt | 0x8000 | (k & 0x7fff),
 *                  from its word 0 -- so the layout does not depend on what the episode's steps drew, and the library
 *                  draws layouts AHEAD of their resets (one refill launch behind a reset once a quarter of the
 *                  replicas have used theirs up; a reset whose layout is not there yet draws it itself: same result).
 * The position bookkeeping is shared: the stream is consumed in blocks of AIE_MT_N words ("mt_pos" counts inside the
 * block, 624 = block exhausted); in fast mode tensor "mt" is uint32 [E, 4] = key32, block number, salt, resets so far. */
#define AIE_RNG_NUMPY 0
#define AIE_RNG_FAST 1
#define AIE_COVID_MAX_FILTERS 8 /* unemployment filter bank size (covid19_env.py:242)    */

/* ---- error codes ---------------------------------------------------------------- */
#define AIE_OK 0
#define AIE_E_INVALID (-1)     /* bad argument / config (reference: assert/ValueError)  */
#define AIE_E_NOTFOUND (-2)    /* unknown tensor name (reference: KeyError)             */
#define AIE_E_HIP (-3)         /* HIP runtime error                                     */
#define AIE_E_NOMEM (-4)
#define AIE_E_UNSUPPORTED (-5)

/* Per-replica `error_flags` tensor (int32 [E]): conditions on which the reference RAISES from inside step() / reset()
 * while a batched launch cannot (it carries on as documented and leaves the evidence here; sticky until the replica is
 * reset).  The host mirror's env.check_errors() turns them into the reference's exceptions. */
#define AIE_ERR_AGENT_ACTION 1    /* an agent's action index outside its action space: treated as NO-OP (reference:
                                     ValueError, F/components/move.py:133-134, build.py:158-159; TypeError from
                                     base_agent.py:407-438 in single-action mode)                                   */
#define AIE_ERR_PLANNER_ACTION 2  /* likewise for the planner (redistribution.py:953-960)                             */
#define AIE_ERR_RESET_PLACEMENT 4 /* reset could not find a free tile for an agent in 200 tries (reference: TimeoutError,
                                     layout_from_file.py:366-368); the agent was put on the last tile tried           */

/* ---- component ids (registry names in the reference, F/components/*.py) -------- */
enum {
  AIE_COMP_BUILD = 1,          /* "Build"                     F/components/build.py:15        */
  AIE_COMP_CDA = 2,            /* "ContinuousDoubleAuction"   F/components/continuous_double_auction.py:16 */
  AIE_COMP_GATHER = 3,         /* "Gather"                    F/components/move.py:16         */
  AIE_COMP_TAX = 4,            /* "PeriodicBracketTax"        F/components/redistribution.py:78 */
  AIE_COMP_SIMPLE_LABOR = 5,   /* "SimpleLabor"               F/components/simple_labor.py:15  */
  AIE_COMP_COVID_CONTROL = 6,  /* "ControlUSStateOpenCloseStatus" F/components/covid19_components.py:32  */
  AIE_COMP_COVID_SUBSIDY = 7,  /* "FederalGovernmentSubsidy"      F/components/covid19_components.py:244 */
  AIE_COMP_COVID_VACCINE = 8,  /* "VaccinationCampaign"           F/components/covid19_components.py:472 */
  AIE_COMP_WEALTH_REDISTRIBUTION = 9 /* "WealthRedistribution"     F/components/redistribution.py:21-75 */
};

/* ---- scenario families (registry names in the reference, F/scenarios) ----------- */
enum {
  AIE_SCN_GTB = 0,             /* "layout_from_file/simple_wood_and_stone" (+ uniform: next)   */
  AIE_SCN_ONE_STEP_ECONOMY = 1,/* "one-step-economy"  F/scenarios/one_step_economy/one_step_economy.py:15 */
  AIE_SCN_COVID = 2            /* "CovidAndEconomySimulation"  F/scenarios/covid19/covid19_env.py:33 */
};
enum { AIE_AGENT_REW_COIN_MINUS_LABOR_COST = 0, AIE_AGENT_REW_ISOELASTIC = 1 };

enum { AIE_SKILL_NONE = 0, AIE_SKILL_PARETO = 1, AIE_SKILL_LOGNORMAL = 2 };
enum {
  AIE_TAX_MODEL_WRAPPER = 0,   /* "model_wrapper": planner actions pick the rates       */
  AIE_TAX_US_FEDERAL = 1,      /* "us-federal-single-filer-2018-scaled"                 */
  AIE_TAX_FIXED = 2,           /* "fixed-bracket-rates"                                 */
  AIE_TAX_SAEZ = 3             /* "saez": rates from the Saez formula on a buffer of observed
                                  (income, marginal rate) pairs, redistribution.py:436-823 */
};
#define AIE_SAEZ_BINS 100      /* _saez_n_estimation_bins, redistribution.py:284         */
enum { AIE_WARMUP_DECAY = 0, AIE_WARMUP_AUTO = 1 };
enum {
  AIE_PLANNER_REW_COIN_EQ_TIMES_PROD = 0,
  AIE_PLANNER_REW_INV_INCOME_COIN = 1,
  AIE_PLANNER_REW_INV_INCOME_UTIL = 2
};

enum {
  AIE_U8 = 0, AIE_I8 = 1, AIE_I16 = 2, AIE_I32 = 3, AIE_U32 = 4, AIE_F32 = 5, AIE_F64 = 6
};

/* ---- COVID-19 scenario: the scalar constants CovidAndEconomyEnvironment.__init__ derives
 * (covid19_env.py:68-330) + the kwargs of its three components.  Per-state constants and
 * tables are named tensors ("model_*", leading dim 1) filled with aie_upload after
 * aie_create -- the counterpart of the reference's get_data_dictionary() push
 * (covid19_env.py:436-560, covid19_components.py:110-143,330-359,560-591).  Floating-point
 * constants the reference holds as float32 are passed as doubles holding the float32 value. */
typedef struct aie_covid_config {
  int32_t num_stringency_levels;          /* model_constants NUM_STRINGENCY_LEVELS (10)        */
  int32_t beta_delay;                     /* fitted BETA_DELAY (days)                          */
  int32_t filter_len;                     /* fitted FILTER_LEN (600)                           */
  int32_t num_filters;                    /* len(CONV_LAMBDAS) <= AIE_COVID_MAX_FILTERS        */
  int32_t action_cooldown_period;         /* ControlUSStateOpenCloseStatus kwarg               */
  int32_t subsidy_interval;               /* FederalGovernmentSubsidy kwargs                   */
  int32_t num_subsidy_levels;
  int32_t delivery_interval;              /* VaccinationCampaign kwarg                         */
  int32_t time_when_vaccine_delivery_begins; /* days from start_date to vaccine_delivery_start_date */
  int32_t filter_recurrence;              /* 1: the unemployment filters are exp(-age / lambda_f) (covid19_env.py:242-247)
                                             and filter_decay[] holds exp(-1 / lambda_f): the step updates each
                                             filter's discounted delta sum in O(1) (A_t = r (A_{t-1} - r^{L-1} d_old)
                                             + d_new) instead of re-summing the filter_len-day window; 0: direct sum
                                             over the uploaded taps (any filter shape)                              */
  double death_rate, gamma;               /* SIR_MORTALITY, SIR_GAMMA                          */
  double value_of_life;
  double daily_production_per_worker;
  double infection_too_sick_to_work_rate;
  double population_between_age_18_65;
  double risk_free_interest_rate;
  double economic_reward_crra_eta;
  double planner_health_norm, planner_economic_norm;
  double min_marginal_planner_health_index, max_marginal_planner_health_index;
  double min_marginal_planner_economic_index, max_marginal_planner_economic_index;
  double weightage_on_marginal_planner_health_index, weightage_on_marginal_planner_economic_index;
  double reward_normalization_factor;
  double filter_decay[AIE_COVID_MAX_FILTERS]; /* r_f = exp(-1 / CONV_LAMBDAS[f]), used iff filter_recurrence      */
  double filter_tail[AIE_COVID_MAX_FILTERS];  /* r_f^(filter_len - 1): the weight with which a delta leaves the window */
  int32_t replay_policies;                /* use_real_world_policies (covid19_env.py:56-60, covid19_components.py:181-186,
                                             394-425): actions are ignored; the states' stringency actions come from tensor
                                             "replay_stringency_policy" [episode_length][n] (row t-1 acts at step t), the
                                             planner's subsidy level of every step from "replay_subsidy_level"
                                             [episode_length]; every action mask is fully open                       */
  int32_t replay_data;                    /* use_real_world_data (covid19_env.py:52-55, 734-757, 815-818; implies
                                             replay_policies): susceptible / infected / recovered / vaccinated / deaths /
                                             unemployed of day t are read from tensor "replay_state" (float64 [6][episode_length + 1][n],
                                             clamped at 0) instead of being simulated                                 */
} aie_covid_config;

/* ---- configuration: the kwargs of make_env_instance + component kwargs ---------- */
typedef struct aie_config {
  int32_t abi_version;
  int32_t n_envs;                    /* E replicas on this device                      */
  int32_t n_agents;                  /* base_env.py:221-224 (>= 2)                      */
  int32_t world_h, world_w;          /* base_env.py:215-219                             */
  int32_t episode_length;            /* base_env.py:253                                 */
  int32_t multi_action_mode_agents;  /* base_env.py:258                                 */
  int32_t multi_action_mode_planner; /* base_env.py:259                                 */
  int32_t allow_observation_scaling; /* base_env.py:262 -> inv_scale 0.01 / time scale  */
  int32_t n_components;
  int32_t components[AIE_MAX_COMPONENTS]; /* AIE_COMP_*, in config order (=dynamics order) */

  /* scenario: layout_from_file / uniform simple_wood_and_stone
   * (F/scenarios/simple_wood_and_stone/layout_from_file.py:68-167) */
  int32_t has_water;                 /* "Water" landmark registered (LayoutFromFile)    */
  int32_t shared_layout;             /* 1: source/water planes identical in all replicas */
  int32_t planner_gets_spatial_info;
  int32_t full_observability;
  int32_t obs_range;                 /* mobile_agent_observation_range                  */
  int32_t fixed_four_skill_and_loc;
  int32_t reset_random_order;        /* agents are placed in a random order at reset
                                      * (uniform/..., dynamic_layout.py:420-431)        */
  int32_t energy_warmup_method;
  int32_t planner_reward_type;
  int32_t regen_halfwidth[AIE_N_RES];/* 0..3 (dynamic_layout.py:150-153); > 0 needs max_health == 1 */
  int32_t max_health[AIE_N_RES];
  double regen_weight[AIE_N_RES];
  double starting_agent_coin;
  double isoelastic_eta;
  double energy_cost;
  double energy_warmup_constant;
  double mixing_weight_gini_vs_coin;
  int32_t ranked_locs[AIE_MAX_AGENTS][2]; /* fixed_four_skill_and_loc, :196-247         */
  double avg_ranked_skill[AIE_MAX_AGENTS];/* already multiplied by Build.payment, :192  */

  /* Build (F/components/build.py:41-68) */
  int32_t build_payment;
  int32_t build_payment_max_skill_multiplier;
  int32_t build_skill_dist;
  double build_labor;

  /* Gather (F/components/move.py:41-64) */
  int32_t gather_skill_dist;
  double move_labor, collect_labor;

  /* ContinuousDoubleAuction (continuous_double_auction.py:42-77) */
  int32_t cda_max_bid_ask;
  int32_t cda_order_duration;
  int32_t cda_max_num_orders;
  double cda_order_labor;

  /* PeriodicBracketTax (redistribution.py:137-346) */
  int32_t tax_disable;
  int32_t tax_model;
  int32_t tax_period;
  int32_t tax_n_brackets;
  int32_t tax_n_disc_rates;
  double tax_bracket_cutoffs[AIE_MAX_BRACKETS];
  double tax_disc_rates[AIE_MAX_RATES];       /* np.arange(rate_min, rate_max+disc, disc) */
  double tax_fixed_rates[AIE_MAX_BRACKETS];   /* us-federal / fixed models, <= rate_max  */

  /* scenario family + one-step-economy (one_step_economy.py:55-78) */
  int32_t scenario;                  /* AIE_SCN_*                                       */
  int32_t ose_agent_reward_type;     /* AIE_AGENT_REW_*                                 */
  double ose_labor_exponent;
  double ose_labor_cost;

  /* SimpleLabor (F/components/simple_labor.py:41-74) */
  int32_t labor_mask_first_step;
  int32_t labor_num_hours;           /* 100                                             */
  double labor_pmsm;                 /* payment_max_skill_multiplier                    */
  double labor_skills[AIE_MAX_AGENTS_WIDE]; /* per-agent skill (sorted Pareto means, :66-74) */

  aie_covid_config covid;            /* scenario == AIE_SCN_COVID only                  */

  /* "split_layout/simple_wood_and_stone" (layout_from_file.py:653-800): a water row; at reset the
   * agents get the ranked build skills (avg_ranked_skill[], highest first) in a random order and
   * are placed above the water row if their skill rank is listed, else below it. */
  int32_t split_water_line;          /* 0: not a split layout; else 0 < row < world_h - 1 */
  uint32_t split_top_ranks[2];       /* bit k: skill rank k starts in the top part        */
  int32_t rng_mode;                  /* AIE_RNG_NUMPY (0, parity with the reference) or AIE_RNG_FAST              */

  /* PeriodicBracketTax tax_annealing_schedule=[warmup, slope] (redistribution.py:311-330,
   * utils.py:10-118): the highest allowed rate grows with the number of completed episodes */
  int32_t tax_annealing;             /* 1: schedule given                                 */
  /* dense logs (base_env.py:984-1016, component get_dense_log): replicas [0, dense_log_replicas)
   * record the component events of the current step (AIE_EV_*) in the tensors
   * "log_event_count" / "log_events"; 0 = off. */
  int32_t dense_log_replicas;
  double tax_annealing_warmup, tax_annealing_slope;
  double tax_rate_max;               /* rate_max kwarg (0 when taxes are disabled)        */

  /* tax_model="saez" (redistribution.py:262-296) */
  double tax_rate_min;               /* rate_min kwarg (0 when taxes are disabled)        */
  int32_t saez_buffer_size;          /* _buffer_size: samples before the formula is used (500) */
  int32_t saez_pareto_weight_uniform;/* pareto_weight_type: 0 "inverse_income", 1 "uniform" */
  int32_t saez_fixed_elas_given;     /* saez_fixed_elas is not None                        */
  int32_t saez_global_capacity;      /* > 0: room (pairs) for the cross-replica sample buffer of the reference's trainer
                                      * (set_global_saez_buffer, redistribution.py:530-533); 0: feature off        */
  double saez_fixed_elas;

  /* Source layouts drawn at every reset, on the device, from the replica's own stream ("uniform/", "quadrant/",
   * "multi_zone/simple_wood_and_stone": Uniform.reset_starting_layout, dynamic_layout.py:313-392; MultiZone
   * :778-872; Quadrant :992-1024).  Index 0 = Stone, 1 = Wood as everywhere else. */
  int32_t layout_gen;                /* AIE_LAYOUT_FIXED (planes from aie_set_layout), _UNIFORM, _QUADRANT, _MULTI_ZONE */
  int32_t layout_checker;            /* checker_source_blocks                                                        */
  double layout_coverage[AIE_N_RES]; /* layout_specs[r]["starting_coverage"] (already doubled under checker)           */
  double layout_clump[AIE_N_RES];    /* 1 - clip(clumpiness, 0, 0.99)                                                */
  int32_t mz_rows, mz_cols;          /* multi_zone: num_partitions_row / _col                                        */
  int32_t mz_zones[3];               /* multi_zone: number of Wood, Stone, Wood+Stone zones                          */
  int32_t layout_pad_;
} aie_config;
#define AIE_LAYOUT_FIXED 0
#define AIE_LAYOUT_UNIFORM 1
#define AIE_LAYOUT_QUADRANT 2
#define AIE_LAYOUT_MULTI_ZONE 3

/* ---- dense-log events: one row of "log_events" int32 [L, cap, AIE_EV_WORDS] ------- */
#define AIE_EV_WORDS 12 /* [0] type, [1..8] integers, [10..11] one float64 (bit pattern)          */
enum aie_event {
  AIE_EV_BUILD = 1,  /* builder, row, col; f64 income              build.py:149-155                */
  AIE_EV_TRADE = 2,  /* commodity (0 Stone, 1 Wood), seller, buyer, ask, bid, price, ask_lifetime,
                        bid_lifetime                               continuous_double_auction.py:289-305 */
  AIE_EV_GATHER = 3, /* agent, resource (0 Stone, 1 Wood), n, row, col   move.py:141-149          */
  AIE_EV_TAX = 4,    /* agent; f64 tax_paid (income, marginal rate: tensors "tax_last_*")
                                                                  redistribution.py:878-883        */
  AIE_EV_TAX_BRACKET = 5 /* bracket; f64 marginal rate in force (the tax day's "schedule"; precedes
                        that day's AIE_EV_TAX rows)              redistribution.py:856-859        */
};

/* ---- tensor descriptor ---------------------------------------------------------- */
typedef struct aie_tensor_desc {
  char name[64];
  void* data;            /* device pointer of element [0,...,0] (NULL before aie_create)*/
  int32_t dtype;         /* AIE_U8 ...                                                  */
  int32_t ndim;          /* includes the leading env dimension                         */
  int64_t shape[6];
  int64_t stride[6];     /* in BYTES (state fields live in per-env records => strided)  */
  int64_t arena_offset;  /* byte offset of element [0,...] inside the arena             */
} aie_tensor_desc;

typedef struct aie_env aie_env;

/* Validates cfg and returns the arena size in bytes (> 0) or a negative error code. */
int64_t aie_arena_bytes(const aie_config* cfg);

/* Creates E replicas on HIP device `device`.  If `arena` is non-NULL it must be a
 * device allocation of at least aie_arena_bytes(cfg) bytes, 256-byte aligned, that
 * outlives the env (e.g. a torch uint8 tensor); otherwise the library allocates it: hipMalloc, or -- from
 * $AIE_ARENA_VMM_MIN_MB (default 1024) MiB on -- a virtual range backed by physical pieces of $AIE_ARENA_PIECE_MB
 * (default 64) MiB each (hipMemCreate / hipMemMap), on which the store-bound one-step-economy launch runs 14 % faster
 * than on one big allocation.  For a configuration outside every compile-time instance's family aie_create also starts
 * the run-time specialisation in the background (see aie_specialize; $AIE_JIT_AUTO=0 switches that off). */
int aie_create(const aie_config* cfg, int device, void* arena, int64_t arena_bytes,
               aie_env** out);
int aie_destroy(aie_env* env);
const char* aie_last_error(const aie_env* env /* NULL: last create error */);

/* Number of exported tensors and their descriptors (index or name lookup).  Names (aie_num_tensors /
 * aie_tensor_at enumerate what a configuration has; csrc/aie_layout.h is where they are defined):
 *   observations  obs_a_world-map, obs_a_world-idx_map, obs_a_flat, obs_a_action_mask, obs_a_time, obs_p_* (flat,
 *                 action_mask, time, world-map, world-idx_map, agents = the planner's per-agent p{i} fragments)
 *   outputs       rewards_a [E, n], rewards_p [E], done [E]
 *   state         cells (+ byte-plane views stone / wood / house_owner / cell_flags), loc_r/c, inv_res, esc_res,
 *                 inv_coin, esc_coin, labor, build_payment, build_skill, bonus_gather_prob, util, cda_* (books,
 *                 histograms, price history), tax_* (cycle position, rate indices, last coin / income / marginal
 *                 rate, total collected; tax_saez_bracket_rates / tax_saez_observed_rates), timestep, completions,
 *                 mt / mt_pos / mt_has_gauss / mt_gauss (the replica's NumPy-legacy generator), regen_source_count
 *   episode       metrics_cda_trades, metrics_tax_* (accumulators behind env.metrics)
 *   Saez          saez_buffer, saez_buffer_len, saez_reached_min_samples, saez_elas, saez_running_avg_tax_rates,
 *                 saez_next_rates
 *   dense log     log_event_count [L], log_events [L, cap, AIE_EV_WORDS]
 *   COVID         susceptible ... economic_index state rows, stringency_history_chunks, model_* constants */
int aie_num_tensors(const aie_env* env);
int aie_tensor_at(const aie_env* env, int index, aie_tensor_desc* out);
int aie_get_tensor(const aie_env* env, const char* name, aie_tensor_desc* out);

/* Host <-> device copies of one named tensor (dense, C-order, leading dim = E).
 * Synchronous; meant for initial state injection and parity dumps.  The COVID tables "model_stringency_level_history_0"
 * and "model_unemp_conv_filters" have to be written through aie_upload (not through a view of the arena): the library
 * derives data from them at that point (the history-format image and the pre-episode change events every reset copies;
 * whether the taps are float32 values, which selects the tap-table format of the window-sum kernel). */
int aie_upload(aie_env* env, const char* name, const void* host, int64_t bytes);
int aie_download(aie_env* env, const char* name, void* host, int64_t bytes);

/* Source-block / water planes (HOST pointers, u8 [H*W] each; with shared_layout=1 one
 * replica's planes which are broadcast, else E of them).  They are packed into the
 * static flag byte of every map cell (layout_from_file.py:103-112, 323-334).  With shared_layout=1 (fixed layouts) the
 * call also derives the batch's one list of regeneration draws that target a source block (rng_mode AIE_RNG_FAST: the
 * step kernels read it instead of scanning the flag bytes every step): change such a layout through this call, not by
 * writing the `cell_flags` tensor. */
int aie_set_layout(aie_env* env, const uint8_t* stone_src, const uint8_t* wood_src,
                   const uint8_t* water);

/* Replica e gets the stream np.random.seed(base_seed + e) would give. */
int aie_seed(aie_env* env, uint32_t base_seed, void* stream);
/* rng_mode == AIE_RNG_FAST only (AIE_E_UNSUPPORTED otherwise): replica e gets the counter stream keyed by
 * seed + global_env_offset + e (48 bits are used); aie_seed(env, s, ...) is aie_seed_fast(env, s, 0, ...) there. */
int aie_seed_fast(aie_env* env, uint64_t seed, int64_t global_env_offset, void* stream);
/* Raw legacy-MT19937 state per replica: key[E][624], pos[E] (np.random.get_state()); rng_mode == AIE_RNG_FAST:
 * key[E][4] (key32, block number, salt, resets so far), pos[E]. */
int aie_set_rng_state(aie_env* env, const uint32_t* key, const int32_t* pos);

/* Resets the replicas whose env_mask byte is non-zero (NULL = all); env_mask is a
 * DEVICE pointer [E] u8 (e.g. the `done` tensor).  Writes reset observations. */
int aie_reset(aie_env* env, const uint8_t* d_env_mask, void* stream);

/* One env.step() for all replicas.  d_actions_a: device int32 [E, n_agents]
 * (single-action mode) or [E, n_agents, n_subspaces] (multi-action mode);
 * d_actions_p: device int32 [E, n_planner_subspaces] (multi-action planner) or [E].
 * Either may be NULL (= all NO-OP, base_env.py:964-966). */
int aie_step(aie_env* env, const int32_t* d_actions_a, const int32_t* d_actions_p,
             void* stream);

/* Fills caller-owned action buffers with the bench's synthetic uniform random policy:
 * counter RNG keyed (seed, global env id, t, agent) -- SURVEY.md section 8(d). */
int aie_sample_random_actions(aie_env* env, uint64_t seed, int64_t global_env_offset,
                              int32_t* d_actions_a, int32_t* d_actions_p, void* stream);

/* aie_step + aie_sample_random_actions for the NEXT step in one launch: steps with
 * d_actions_a/p and, while the replica's first wavefront runs the serial dynamics, lets the
 * otherwise idle second wavefront fill d_next_a/p (caller-owned, must differ from the current
 * action buffers) with exactly what aie_sample_random_actions would write next.  A rollout loop
 * of the uniform random policy then needs one launch per step instead of two. */
int aie_step_sample_next(aie_env* env, const int32_t* d_actions_a, const int32_t* d_actions_p, uint64_t seed,
                         int64_t global_env_offset, int32_t* d_next_a, int32_t* d_next_p, void* stream);

/* The same with the NEXT actions drawn as aie_sample_masked_actions would draw them from the masks this step writes
 * (a state's stringency levels only outside its cooldown, the planner's subsidy levels only on the first day of an
 * interval, NO-OP always): the random policy of a trainer that applies `action_mask` to its logits, one launch per
 * step.  COVID scenario; AIE_E_UNSUPPORTED elsewhere (there: aie_step, then aie_sample_masked_actions). */
int aie_step_sample_next_masked(aie_env* env, const int32_t* d_actions_a, const int32_t* d_actions_p, uint64_t seed,
                                int64_t global_env_offset, int32_t* d_next_a, int32_t* d_next_p, void* stream);

/* Part of a step, for components that live on the HOST (round 6: ai_economist_amd.foundation.BatchedComponent -- a user's
 * registered component whose component_step runs as torch code on the state tensors between two launches; the
 * reference's registries are open, F/base/base_component.py:378, F/base/registrar.py:48-66).  Runs the built-in components
 * [comp_lo, comp_hi) of aie_config.components, in order, and of the rest of a step what `phases` names:
 *   AIE_STEP_HEAD     timestep += 1 (base_env.py:981: before the first component of a step)
 *   AIE_STEP_TAIL     scenario_step (regeneration), observations, masks, rewards, done (+ auto-reset) -- the end of a step
 *   AIE_STEP_OBSERVE  observations and masks of the state as it stands, nothing else (after a host-side edit, e.g. a
 *                     component's additional_reset_steps); with AIE_STEP_REBASE also the utilities the next rewards are
 *                     measured from, as a reset leaves them (layout_from_file.py:347-349 runs behind the components' resets)
 * (phases == 0: components only, a stretch in the middle of a step.)
 * One step = any sequence of calls whose first carries HEAD, whose last carries TAIL and whose ranges tile the list;
 * aie_step(a, p) == aie_step_range(a, p, 0, n_components, HEAD | TAIL).  Every call takes the same action buffers.
 * Between two calls of a step the caller may edit state tensors; the TAIL call rewrites all observations.  Always the
 * full-featured kernel (no compile-time / run-time instance).  Gather-trade-build scenarios; AIE_E_UNSUPPORTED elsewhere,
 * with tax_model "saez" and while a dense-log replica records. */
#define AIE_STEP_HEAD 1
#define AIE_STEP_TAIL 2
#define AIE_STEP_OBSERVE 4
#define AIE_STEP_REBASE 8
#define AIE_STEP_RETAX 16 /* with OBSERVE: PeriodicBracketTax's reset-time snapshot of the agents' coin (redistribution.py:1106-1110) taken
                           * again -- a host component listed AHEAD of the tax component edited coin in its reset hook */
int aie_step_range(aie_env* env, const int32_t* d_actions_a, const int32_t* d_actions_p, int32_t comp_lo, int32_t comp_hi,
                   int32_t phases, void* stream);

/* Reward log for learners on another device: every following aie_step / aie_step_sample_next ALSO
 * writes replica e's (agent rewards [n_agents], planner reward, done as 0/1) as n_agents + 2 floats to
 * d_log[((slot * n_envs) + e) * (n_agents + 2) + ...], slot = 0, 1, ... n_slots - 1, 0, ... advancing by one per
 * step (this call resets it to 0).  The caller owns d_log (n_slots * n_envs * (n_agents + 2) floats) and ships
 * whole ranges of slots to the learner rank with one collective per many steps instead of one per step
 * (ai_economist_amd/sharding.py).  d_log == NULL switches the log off.  All scenarios.
 * The log's address, slot count and restart live in device memory, not in the step's kernel arguments: step launches
 * captured in a hipGraph BEFORE this call write to the log this call names when they are replayed after it, and a
 * replica that sits a launch out (COVID: stepped past its episode's end without a reset) still moves its slot with the
 * batch's.  The call waits for the device to go idle (call it between steps, not under stream capture). */
int aie_set_reward_log(aie_env* env, float* d_log, int32_t n_slots);

/* Auto-reset (the vectorised-trainer convention, reference analogue: F/env_wrapper.py:341-353 reset_only_done_envs):
 * with on != 0 a replica whose episode ends in a step restarts before the step call returns control of the stream --
 * its state and observations are those of the fresh episode, `done` and the rewards stay the terminal step's.
 * one-step-economy does it inside the step launch (the terminal observations, which nothing could read before they
 * are overwritten, are not written at all); the other scenarios enqueue their reset kernel, masked with `done`,
 * right behind the step (uniform/, quadrant/ and multi_zone/ layouts are drawn inside that reset kernel).
 * AIE_E_UNSUPPORTED only when the HOST supplies a new layout per episode (per-replica layouts passed to
 * aie_set_layout: worlds too large for the device-side generator).  Episode metrics of a finished episode are gone
 * once it restarts. */
int aie_set_auto_reset(aie_env* env, int on);

/* tax_model "saez": the cross-replica sample buffer (reference: PeriodicBracketTax.set_global_saez_buffer,
 * redistribution.py:515-533; filled by the trainer with the concatenation of every replica's local buffer,
 * tutorials/rllib/utils/remote.py:56-73).  d_pairs: n_pairs (income, marginal rate) float64 pairs in DEVICE memory
 * (n_pairs <= aie_config.saez_global_capacity); n_pairs == 0 clears it.  From then on every replica's period start
 * uses global + its own samples added since the buffers were last reset (`saez_additions` tensor), as the
 * reference's `saez_buffer` property does. */
int aie_set_global_saez_buffer(aie_env* env, const double* d_pairs, int64_t n_pairs);

/* Dense logs (aie_config.dense_log_replicas > 0; reference: F/base/base_env.py:273-283, 883-891 -- only every
 * `dense_log_frequency`-th episode is logged): with on == 0 the dense-log replicas stop recording AIE_EV_* rows and
 * step with the rest of the batch on the environment's fast kernel; with on != 0 (the default) they -- and only they
 * -- take the full-featured kernel.  State and observations are bit-identical either way; the host mirror switches
 * it per episode (foundation/base_env.py: reset).  Gather-trade-build only: the one-step-economy kernel records its
 * (few) events for the dense-log replicas regardless of this switch -- there is no separate full-featured kernel to
 * leave -- and COVID has no event rows. */
int aie_set_dense_log_active(aie_env* env, int on);

/* Where the arena lives: *bytes = its size, *allocator = AIE_ARENA_CALLER (passed to aie_create), AIE_ARENA_HIPMALLOC
 * (one hipMalloc) or AIE_ARENA_VMM (a virtual range backed by physical pieces of *piece_bytes each; see aie_create).
 * Any out pointer may be NULL.  (Launch times of the store-bound kernels depend on it: a measurement names it.) */
#define AIE_ARENA_CALLER 0
#define AIE_ARENA_HIPMALLOC 1
#define AIE_ARENA_VMM 2
int aie_arena_info(const aie_env* env, int64_t* bytes, int32_t* allocator, int64_t* piece_bytes);

/* sizeof(aie_config) as this library was built: a binding checks its mirror of the struct against it. */
int aie_sizeof_config(void);

/* Which step kernel runs this environment: 0 .. 999 = a compile-time instance (the code-shaping part of the
 * configuration's parameter block folded into the code, csrc/aie_spec_generated.h; an instance stands for the family of
 * configurations that differ in scalars only), AIE_KERNEL_INSTANCE_JIT = the run-time specialisation, -1 = the generic
 * kernel. */
int aie_step_kernel_instance(aie_env* env);

/* Chooses between the step kernels that can run this environment: AIE_KERNEL_AUTO (default) = the specialised
 * instance when one matches the configuration's family (compile-time) or once its run-time specialisation is ready,
 * AIE_KERNEL_GENERIC = pinned to the generic kernel that reads the parameter block at run time.  Both produce the same arena bit for bit (tests/test_gpu_parity.py steps them side by side); the
 * switch exists so that a user can check exactly that on their own configuration. */
#define AIE_KERNEL_AUTO 0
#define AIE_KERNEL_GENERIC 1
int aie_select_step_kernel(aie_env* env, int which);

/* Specialises the step and reset kernels on THIS environment's configuration family at run time, the way the build does
 * for the BASELINE configurations: the code-shaping part of the parameter block (component tuple, agent count, world
 * size, capacities, flags; NOT scalars such as starting coin, eta, episode length, labor costs, tax period / cutoffs,
 * which every specialised kernel reads at run time) becomes a compile-time constant of the kernels (hiprtc compiles
 * csrc/aie_kernels.hip with it as a constant image, a few seconds once; the code object is cached under
 * $AIE_JIT_CACHE / ~/.cache/ai_economist_amd -- private directories only -- keyed by the image, the sources, the
 * architecture, the compiler version and options).  aie_create already starts this in the background and a later
 * aie_reset (an episode boundary, outside stream capture; never aie_step) adopts the result; this call WAITS for it.  Afterwards AIE_KERNEL_AUTO runs
 * the specialised kernels (aie_step_kernel_instance() == AIE_KERNEL_INSTANCE_JIT); results are bit-identical to the
 * generic kernel's.  Gather-trade-build and one-step-economy environments.  AIE_E_UNSUPPORTED -- and the environment simply keeps the
 * generic kernel -- when hiprtc, the kernel sources beside the library ($AIE_JIT_SOURCE_DIR) or the toolchain headers
 * are not there, or when the configuration needs the full-featured kernel (tax_model "saez", order books beyond a
 * wavefront, general regeneration; dense-log replicas do NOT: they take the full-featured kernel alone and only while an
 * episode is being logged, aie_set_dense_log_active). */
#define AIE_KERNEL_INSTANCE_JIT 1000
int aie_specialize(aie_env* env);

/* Categorical sampling from the CALLER's policy logits under the current action masks (the step a trainer
 * performs between two env steps -- the reference's trainers apply `action_mask` to the logits and sample,
 * base_env.py:141-145, tutorials/rllib/env_wrapper.py:50-211, training_script.py:88-133).
 *   d_logits_a  float32 [E, n, MA]  in the layout of obs_a_action_mask (single-action agents: one row of MA entries,
 *               entry 0 = NO-OP; multi-action: the subspaces' (1 + dim) entries back to back; COVID: [E, n, 1 + levels])
 *   d_logits_p  float32 [E, MP]     in the layout of obs_p_action_mask
 * Sub-action = a draw from softmax(logits) restricted to the mask by inverse CDF: weights exp(logit_k - max) of the allowed
 * entries, their prefix sums in a fixed order, the first entry whose sum passes u x total, u in (0, 1) from a counter hash
 * keyed (seed, global env id, the replica's draw index, slot); NaN logits count as masked, NO-OP if nothing is allowed.
 * Everything is float32 and a fixed sequence of IEEE add / multiply / fused multiply-add operations (csrc/aie_layout.h:
 * aie_sampler_expf, aie_sampler_uniform, aie_sampler_entry_rng, the scan's order), so the CPU restatement (oracle/:
 * aie_oracle_sample_policy_actions) picks the same entries; a probability is resolved to 2^-24 of its row's total and u
 * has 23 bits.  Rows of up to 64 entries with single-action agents (every BASELINE configuration, COVID) run on
 * instances with compile-time row shapes, anything else on a generic kernel with the same results.  One launch; the
 * draw index is the replica's record field `sample_t`, advanced by the kernel (replayable from a hipGraph).  Either pair
 * (logits, actions) may be NULL. */
int aie_sample_policy_actions(aie_env* env, const float* d_logits_a, const float* d_logits_p, uint64_t seed,
                              int64_t global_env_offset, int32_t* d_actions_a, int32_t* d_actions_p, void* stream);

/* Same counter RNG, but each sub-action is drawn uniformly among the entries that the
 * CURRENT action masks allow (obs_a_action_mask / obs_p_action_mask; NO-OP is always
 * allowed).  This is the random policy a trainer starts from when it applies the
 * `action_mask` observation to its logits (F/base/base_env.py:141-145,
 * tutorials/rllib/env_wrapper.py:50-211 hand the mask to the policy network). */
int aie_sample_masked_actions(aie_env* env, uint64_t seed, int64_t global_env_offset, int32_t* d_actions_a,
                              int32_t* d_actions_p, void* stream);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* AIE_H_ */
