"""ctypes mirror of include/aie.h (the C ABI of the batched env.step()).

Only plain C types cross this boundary: pointers, sizes, ints, doubles.  torch is used
by env.py purely as the owner of device memory / streams.
"""
import ctypes as C

ABI_VERSION = 10
MAX_AGENTS = 64
MAX_AGENTS_WIDE = 128
MAX_COMPONENTS = 8
MAX_BRACKETS = 16
MAX_RATES = 64
N_RES = 2
MT_N = 624
RNG_NUMPY, RNG_FAST = 0, 1  # aie_config.rng_mode (AIE_RNG_*)
RNG_FAST_STATE_WORDS = 4     # tensor "mt" in fast mode: key32, block number, salt, resets so far

COMP_BUILD, COMP_CDA, COMP_GATHER, COMP_TAX, COMP_SIMPLE_LABOR = 1, 2, 3, 4, 5
COMP_COVID_CONTROL, COMP_COVID_SUBSIDY, COMP_COVID_VACCINE = 6, 7, 8
COMP_WEALTH_REDISTRIBUTION = 9
# flag byte of the packed map cell ("cell_flags" tensor; csrc/aie_layout.h: AIE_CELL_*)
CELL_WATER, CELL_STONE_SRC, CELL_WOOD_SRC = 1, 2, 4
SCN_GTB, SCN_ONE_STEP_ECONOMY, SCN_COVID = 0, 1, 2
LAYOUT_FIXED, LAYOUT_UNIFORM, LAYOUT_QUADRANT, LAYOUT_MULTI_ZONE = 0, 1, 2, 3
COVID_MAX_FILTERS = 8
MAX_TENSORS = 160  # AIE_MAX_TENSORS (csrc/aie_layout.h)
AGENT_REWARD = {"coin_minus_labor_cost": 0, "isoelastic_coin_minus_labor": 1}
SKILL = {"none": 0, "pareto": 1, "lognormal": 2}
TAX_MODEL = {
    "model_wrapper": 0,
    "us-federal-single-filer-2018-scaled": 1,
    "fixed-bracket-rates": 2,
    "saez": 3,
}
WARMUP = {"decay": 0, "auto": 1}
PLANNER_REWARD = {
    "coin_eq_times_productivity": 0,
    "inv_income_weighted_coin_endowments": 1,
    "inv_income_weighted_utility": 2,
}

E_INVALID, E_NOTFOUND, E_HIP, E_NOMEM, E_UNSUPPORTED = -1, -2, -3, -4, -5

DTYPES = ["uint8", "int8", "int16", "int32", "uint32", "float32", "float64"]


class AieCovidConfig(C.Structure):
    _fields_ = [(k, C.c_int32) for k in (
        "num_stringency_levels", "beta_delay", "filter_len", "num_filters", "action_cooldown_period",
        "subsidy_interval", "num_subsidy_levels", "delivery_interval", "time_when_vaccine_delivery_begins",
        "filter_recurrence")] + [(k, C.c_double) for k in (
            "death_rate", "gamma", "value_of_life", "daily_production_per_worker",
            "infection_too_sick_to_work_rate", "population_between_age_18_65", "risk_free_interest_rate",
            "economic_reward_crra_eta", "planner_health_norm", "planner_economic_norm",
            "min_marginal_planner_health_index", "max_marginal_planner_health_index",
            "min_marginal_planner_economic_index", "max_marginal_planner_economic_index",
            "weightage_on_marginal_planner_health_index", "weightage_on_marginal_planner_economic_index",
            "reward_normalization_factor")] + [("filter_decay", C.c_double * COVID_MAX_FILTERS), ("filter_tail", C.c_double * COVID_MAX_FILTERS),
                                             ("replay_policies", C.c_int32), ("replay_data", C.c_int32)]


class AieConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("n_envs", C.c_int32),
        ("n_agents", C.c_int32),
        ("world_h", C.c_int32),
        ("world_w", C.c_int32),
        ("episode_length", C.c_int32),
        ("multi_action_mode_agents", C.c_int32),
        ("multi_action_mode_planner", C.c_int32),
        ("allow_observation_scaling", C.c_int32),
        ("n_components", C.c_int32),
        ("components", C.c_int32 * MAX_COMPONENTS),
        ("has_water", C.c_int32),
        ("shared_layout", C.c_int32),
        ("planner_gets_spatial_info", C.c_int32),
        ("full_observability", C.c_int32),
        ("obs_range", C.c_int32),
        ("fixed_four_skill_and_loc", C.c_int32),
        ("reset_random_order", C.c_int32),
        ("energy_warmup_method", C.c_int32),
        ("planner_reward_type", C.c_int32),
        ("regen_halfwidth", C.c_int32 * N_RES),
        ("max_health", C.c_int32 * N_RES),
        ("regen_weight", C.c_double * N_RES),
        ("starting_agent_coin", C.c_double),
        ("isoelastic_eta", C.c_double),
        ("energy_cost", C.c_double),
        ("energy_warmup_constant", C.c_double),
        ("mixing_weight_gini_vs_coin", C.c_double),
        ("ranked_locs", (C.c_int32 * 2) * MAX_AGENTS),
        ("avg_ranked_skill", C.c_double * MAX_AGENTS),
        ("build_payment", C.c_int32),
        ("build_payment_max_skill_multiplier", C.c_int32),
        ("build_skill_dist", C.c_int32),
        ("build_labor", C.c_double),
        ("gather_skill_dist", C.c_int32),
        ("move_labor", C.c_double),
        ("collect_labor", C.c_double),
        ("cda_max_bid_ask", C.c_int32),
        ("cda_order_duration", C.c_int32),
        ("cda_max_num_orders", C.c_int32),
        ("cda_order_labor", C.c_double),
        ("tax_disable", C.c_int32),
        ("tax_model", C.c_int32),
        ("tax_period", C.c_int32),
        ("tax_n_brackets", C.c_int32),
        ("tax_n_disc_rates", C.c_int32),
        ("tax_bracket_cutoffs", C.c_double * MAX_BRACKETS),
        ("tax_disc_rates", C.c_double * MAX_RATES),
        ("tax_fixed_rates", C.c_double * MAX_BRACKETS),
        ("scenario", C.c_int32),
        ("ose_agent_reward_type", C.c_int32),
        ("ose_labor_exponent", C.c_double),
        ("ose_labor_cost", C.c_double),
        ("labor_mask_first_step", C.c_int32),
        ("labor_num_hours", C.c_int32),
        ("labor_pmsm", C.c_double),
        ("labor_skills", C.c_double * MAX_AGENTS_WIDE),
        ("covid", AieCovidConfig),
        ("split_water_line", C.c_int32),
        ("split_top_ranks", C.c_uint32 * 2),
        ("rng_mode", C.c_int32),
        ("tax_annealing", C.c_int32),
        ("dense_log_replicas", C.c_int32),
        ("tax_annealing_warmup", C.c_double),
        ("tax_annealing_slope", C.c_double),
        ("tax_rate_max", C.c_double),
        ("tax_rate_min", C.c_double),
        ("saez_buffer_size", C.c_int32),
        ("saez_pareto_weight_uniform", C.c_int32),
        ("saez_fixed_elas_given", C.c_int32),
        ("saez_global_capacity", C.c_int32),
        ("saez_fixed_elas", C.c_double),
        ("layout_gen", C.c_int32),
        ("layout_checker", C.c_int32),
        ("layout_coverage", C.c_double * N_RES),
        ("layout_clump", C.c_double * N_RES),
        ("mz_rows", C.c_int32),
        ("mz_cols", C.c_int32),
        ("mz_zones", C.c_int32 * 3),
        ("layout_pad_", C.c_int32),
    ]


class AieTensorDesc(C.Structure):
    _fields_ = [
        ("name", C.c_char * 64),
        ("data", C.c_void_p),
        ("dtype", C.c_int32),
        ("ndim", C.c_int32),
        ("shape", C.c_int64 * 6),
        ("stride", C.c_int64 * 6),
        ("arena_offset", C.c_int64),
    ]


def bind(lib):
    """Declares the prototypes of every symbol include/aie.h exports."""
    vp = C.c_void_p
    lib.aie_sizeof_config.restype = C.c_int
    if lib.aie_sizeof_config() != C.sizeof(AieConfig):
        raise RuntimeError("ctypes mirror of aie_config is %d bytes, the library's struct %d: _cabi.py and include/aie.h "
                           "are out of step" % (C.sizeof(AieConfig), lib.aie_sizeof_config()))
    lib.aie_arena_bytes.restype = C.c_int64
    lib.aie_arena_bytes.argtypes = [C.POINTER(AieConfig)]
    lib.aie_create.restype = C.c_int
    lib.aie_create.argtypes = [C.POINTER(AieConfig), C.c_int, vp, C.c_int64, C.POINTER(vp)]
    lib.aie_destroy.restype = C.c_int
    lib.aie_destroy.argtypes = [vp]
    lib.aie_last_error.restype = C.c_char_p
    lib.aie_last_error.argtypes = [vp]
    lib.aie_num_tensors.restype = C.c_int
    lib.aie_num_tensors.argtypes = [vp]
    lib.aie_tensor_at.restype = C.c_int
    lib.aie_tensor_at.argtypes = [vp, C.c_int, C.POINTER(AieTensorDesc)]
    lib.aie_get_tensor.restype = C.c_int
    lib.aie_get_tensor.argtypes = [vp, C.c_char_p, C.POINTER(AieTensorDesc)]
    lib.aie_upload.restype = C.c_int
    lib.aie_upload.argtypes = [vp, C.c_char_p, vp, C.c_int64]
    lib.aie_download.restype = C.c_int
    lib.aie_download.argtypes = [vp, C.c_char_p, vp, C.c_int64]
    lib.aie_set_layout.restype = C.c_int
    lib.aie_set_layout.argtypes = [vp, vp, vp, vp]
    lib.aie_seed.restype = C.c_int
    lib.aie_seed.argtypes = [vp, C.c_uint32, vp]
    lib.aie_seed_fast.restype = C.c_int
    lib.aie_seed_fast.argtypes = [vp, C.c_uint64, C.c_int64, vp]
    lib.aie_set_rng_state.restype = C.c_int
    lib.aie_set_rng_state.argtypes = [vp, vp, vp]
    lib.aie_reset.restype = C.c_int
    lib.aie_reset.argtypes = [vp, vp, vp]
    lib.aie_step.restype = C.c_int
    lib.aie_step.argtypes = [vp, vp, vp, vp]
    lib.aie_sample_policy_actions.restype = C.c_int
    lib.aie_sample_policy_actions.argtypes = [vp, vp, vp, C.c_uint64, C.c_int64, vp, vp, vp]
    lib.aie_sample_random_actions.restype = C.c_int
    lib.aie_sample_random_actions.argtypes = [vp, C.c_uint64, C.c_int64, vp, vp, vp]
    lib.aie_set_reward_log.restype = C.c_int
    lib.aie_set_reward_log.argtypes = [vp, vp, C.c_int32]
    lib.aie_step_range.restype = C.c_int
    lib.aie_step_range.argtypes = [vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, vp]
    lib.aie_step_sample_next.restype = C.c_int
    lib.aie_step_sample_next.argtypes = [vp, vp, vp, C.c_uint64, C.c_int64, vp, vp, vp]
    lib.aie_step_sample_next_masked.restype = C.c_int
    lib.aie_step_sample_next_masked.argtypes = [vp, vp, vp, C.c_uint64, C.c_int64, vp, vp, vp]
    lib.aie_set_global_saez_buffer.restype = C.c_int
    lib.aie_set_global_saez_buffer.argtypes = [vp, vp, C.c_int64]
    lib.aie_set_auto_reset.restype = C.c_int
    lib.aie_set_auto_reset.argtypes = [vp, C.c_int]
    lib.aie_set_dense_log_active.restype = C.c_int
    lib.aie_set_dense_log_active.argtypes = [vp, C.c_int]
    lib.aie_step_kernel_instance.restype = C.c_int
    lib.aie_step_kernel_instance.argtypes = [vp]
    lib.aie_sample_masked_actions.restype = C.c_int
    lib.aie_sample_masked_actions.argtypes = [vp, C.c_uint64, C.c_int64, vp, vp, vp]
    lib.aie_select_step_kernel.restype = C.c_int
    lib.aie_select_step_kernel.argtypes = [vp, C.c_int]
    lib.aie_specialize.restype = C.c_int
    lib.aie_specialize.argtypes = [vp]
    lib.aie_arena_info.restype = C.c_int
    lib.aie_arena_info.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    return lib


EXPORTED_SYMBOLS = [
    "aie_arena_bytes", "aie_create", "aie_destroy", "aie_last_error", "aie_num_tensors",
    "aie_tensor_at", "aie_get_tensor", "aie_upload", "aie_download", "aie_set_layout",
    "aie_seed", "aie_seed_fast", "aie_set_rng_state", "aie_reset", "aie_step", "aie_sample_random_actions",
    "aie_sample_masked_actions", "aie_sample_policy_actions", "aie_step_sample_next", "aie_step_sample_next_masked", "aie_set_reward_log", "aie_set_auto_reset",
    "aie_set_dense_log_active", "aie_step_kernel_instance", "aie_select_step_kernel", "aie_specialize", "aie_set_global_saez_buffer", "aie_sizeof_config",
    "aie_arena_info", "aie_step_range",
]
STEP_HEAD, STEP_TAIL, STEP_OBSERVE, STEP_REBASE, STEP_RETAX = 1, 2, 4, 8, 16  # AIE_STEP_*
ARENA_ALLOCATORS = ["caller", "hipMalloc", "vmm"]  # AIE_ARENA_*
KERNEL_AUTO, KERNEL_GENERIC = 0, 1
KERNEL_INSTANCE_JIT = 1000
