"""Construction-time numerics of `CovidAndEconomySimulation` (reference:
F/scenarios/covid19/covid19_env.py:98-380 constructor, :1517-1612 loaders, :1175-1293
reset constants, and the three COVID components' constants,
F/components/covid19_components.py:303-307, 524-546).

Everything here is a function of the fitted parameters / real-world data files and the
scenario kwargs; it runs once on the host and produces the static tables the kernels
consume (`build_model` -> dict of NumPy arrays / scalars, keys == names of the reference's
device data dictionary where one exists, covid19_env.py:388-641).
"""
import os
from datetime import datetime

import numpy as np

F32 = np.float32
I32 = np.int32
_DATA = None


def load_data(path=None):
    """The exported data bundle (oracle/export_covid_data.py) or a directory holding the
    reference's three data files."""
    global _DATA
    if path:
        import json

        mc = json.load(open(os.path.join(path, "model_constants.json")))
        fp = json.load(open(os.path.join(path, "fitted_params.json")))
        rw = np.load(os.path.join(path, "real_world_data.npz"))
        d = {"mc_" + k: np.array(v) for k, v in mc.items()}
        d.update({"fp_" + k: np.array(v) for k, v in fp.items() if k != "settings"})
        d.update({"rw_" + k: rw[k] for k in rw.files})
        return d
    if _DATA is None:
        here = os.path.dirname(os.path.abspath(__file__))
        with np.load(os.path.join(here, "covid19_data.npz")) as z:
            _DATA = {k: z[k] for k in z.files}
    return _DATA


def _softplus(x):
    return np.log(1 + np.exp(x)) * (x <= 20) + x * (x > 20)


def build_model(start_date="2020-03-22", pop_between_age_18_65=0.6, infection_too_sick_to_work_rate=0.1,
                risk_free_interest_rate=0.03, economic_reward_crra_eta=2, health_priority_scaling_agents=1,
                health_priority_scaling_planner=1, reward_normalization_factor=1, episode_length=540,
                path_to_data_and_fitted_params=""):
    d = load_data(path_to_data_and_fitted_params or None)
    m = {}
    fmt = str(d["mc_DATE_FORMAT"])
    pop = I32(d["mc_US_STATE_POPULATION"])
    n = len(pop)
    us_pop = I32(d["mc_US_POPULATION"])
    m["us_state_population"] = pop
    m["us_state_names"] = [str(x) for x in np.asarray(d["mc_US_STATE_IDX_TO_STATE_NAME"]).reshape(-1)] \
        if np.asarray(d["mc_US_STATE_IDX_TO_STATE_NAME"]).ndim else None
    m["us_population"] = us_pop
    m["num_stringency_levels"] = int(d["mc_NUM_STRINGENCY_LEVELS"])
    m["death_rate"] = F32(d["mc_SIR_MORTALITY"])
    m["gamma"] = F32(d["mc_SIR_GAMMA"])
    gdp_per_capita = F32(d["mc_GDP_PER_CAPITA"])
    policy_start = datetime.strptime(str(d["fp_POLICY_START_DATE"]), fmt)
    m["value_of_life"] = I32(d["fp_VALUE_OF_LIFE"])
    m["beta_delay"] = int(d["fp_BETA_DELAY"])
    m["beta_slopes"] = np.array(d["fp_BETA_SLOPES"], dtype=F32)
    m["beta_intercepts"] = np.array(d["fp_BETA_INTERCEPTS"], dtype=F32)
    for k in ("MIN_MARGINAL_AGENT_HEALTH_INDEX", "MAX_MARGINAL_AGENT_HEALTH_INDEX",
              "MIN_MARGINAL_AGENT_ECONOMIC_INDEX", "MAX_MARGINAL_AGENT_ECONOMIC_INDEX"):
        m[k.lower()] = np.array(d["fp_" + k], dtype=F32)
    for k in ("MIN_MARGINAL_PLANNER_HEALTH_INDEX", "MAX_MARGINAL_PLANNER_HEALTH_INDEX",
              "MIN_MARGINAL_PLANNER_ECONOMIC_INDEX", "MAX_MARGINAL_PLANNER_ECONOMIC_INDEX"):
        m[k.lower()] = F32(d["fp_" + k])
    w_agent = np.array(d["fp_INFERRED_WEIGHTAGE_ON_AGENT_HEALTH_INDEX"], dtype=F32)
    w_planner = F32(d["fp_INFERRED_WEIGHTAGE_ON_PLANNER_HEALTH_INDEX"])
    filter_len = int(d["fp_FILTER_LEN"])
    lambdas = np.array(d["fp_CONV_LAMBDAS"], dtype=F32)
    m["filter_len"] = filter_len
    m["num_filters"] = len(lambdas)
    m["conv_lambdas"] = lambdas
    m["unemployment_bias"] = np.array(d["fp_UNEMPLOYMENT_BIAS"], dtype=F32)
    gw = np.array(d["fp_GROUPED_CONVOLUTIONAL_FILTER_WEIGHTS"], dtype=F32)
    m["conv_weights"] = gw.reshape(n, len(lambdas))

    start = datetime.strptime(start_date, fmt)
    assert start >= policy_start
    sidx = (start - policy_start).days
    policy = np.asarray(d["rw_policy"]).astype(np.int64)
    assert 0 <= sidx < len(policy)
    m["start_date"] = start
    m["start_date_index"] = sidx

    # exp(-age / lambda) filters; newest stringency change last (covid19_env.py:225-231)
    f_ts = np.tile(np.flip(np.arange(filter_len), (0,))[None, None], (1, len(lambdas), 1)).astype(F32)
    m["unemp_conv_filters"] = np.exp(-f_ts / lambdas[None, :, None])[0]  # [num_filters, filter_len] f32

    # unemployment at "all ones" = softplus(0) + bias (:247-249, :1405-1441 at t = 0)
    unemp_rate0 = _softplus(np.zeros(n)) + m["unemployment_bias"]
    unemployed_lvl1 = unemp_rate0 * pop / 100
    risk_free = F32(risk_free_interest_rate)
    workforce = (us_pop * pop_between_age_18_65 - np.sum(unemployed_lvl1)).astype(I32)
    workers_per_capita = (workforce / us_pop).astype(F32)
    gdp_per_worker = (gdp_per_capita / workers_per_capita).astype(F32)
    m["num_days_in_an_year"] = 365
    m["daily_production_per_worker"] = (gdp_per_worker / 365).astype(F32)
    m["infection_too_sick_to_work_rate"] = F32(infection_too_sick_to_work_rate)
    m["population_between_age_18_65"] = F32(pop_between_age_18_65)
    assert 0 <= m["infection_too_sick_to_work_rate"] <= 1 and 0 <= m["population_between_age_18_65"] <= 1
    m["risk_free_interest_rate"] = risk_free
    # economy_step at zero infections / deaths (:268-276, :1444-1475)
    incap = (m["infection_too_sick_to_work_rate"] * np.zeros(n, I32)) + np.zeros(n, I32)
    cant = (incap * m["population_between_age_18_65"]) + unemployed_lvl1
    workers = pop * m["population_between_age_18_65"]
    m["maximum_productivity"] = (np.maximum(0, workers - cant) * m["daily_production_per_worker"]).astype(F32)
    m["economic_reward_crra_eta"] = F32(economic_reward_crra_eta)
    assert 0.0 <= m["economic_reward_crra_eta"] < 20.0
    m["agents_health_norm"] = m["maximum_productivity"] * 365
    m["planner_health_norm"] = np.sum(m["agents_health_norm"])
    m["agents_economic_norm"] = m["maximum_productivity"] * 365
    m["planner_economic_norm"] = np.sum(m["agents_economic_norm"])

    def scale(h, alphas):  # :320-330
        z = alphas / (1 - alphas)
        sz = h * z
        return sz / (1 + sz)

    m["weightage_on_marginal_agent_health_index"] = scale(health_priority_scaling_agents, w_agent)
    m["weightage_on_marginal_agent_economic_index"] = 1 - m["weightage_on_marginal_agent_health_index"]
    m["weightage_on_marginal_planner_health_index"] = scale(health_priority_scaling_planner, w_planner)
    m["weightage_on_marginal_planner_economic_index"] = 1 - m["weightage_on_marginal_planner_health_index"]
    m["reward_normalization_factor"] = reward_normalization_factor

    # ---- reset constants (additional_reset_steps :1175-1293) ----
    rw = {k: np.asarray(d["rw_" + k]) for k in ("susceptible", "infected", "recovered", "unemployed", "vaccinated")}
    m["susceptible_0"] = rw["susceptible"][sidx].astype(F32)
    m["infected_0"] = rw["infected"][sidx].astype(F32)
    m["recovered_0"] = rw["recovered"][sidx].astype(F32)
    m["deaths_0"] = (rw["recovered"][sidx] * m["death_rate"]).astype(F32)
    m["unemployed_0"] = rw["unemployed"][sidx].astype(F32)
    m["vaccinated_0"] = rw["vaccinated"][sidx].astype(F32)
    m["stringency_0"] = policy[sidx].astype(F32)
    # agent.state at reset is filled from the table itself, before any float32 cast (:1237-1256; dense logs)
    prev = max(0, sidx - 1)
    m["reset_agent_state"] = {
        "Total Susceptible": rw["susceptible"][sidx].astype(I32),
        "New Infections": (rw["infected"][sidx] - rw["infected"][prev]).astype(I32),
        "Total Infected": rw["infected"][sidx].astype(I32),
        "Total Recovered": rw["recovered"][sidx].astype(I32),
        "New Deaths": (rw["recovered"][sidx] * m["death_rate"] - rw["recovered"][prev] * m["death_rate"]).astype(I32),
        "Total Deaths": (rw["recovered"][sidx] * m["death_rate"]).astype(I32),
        "Total Vaccinated": np.asarray(rw["vaccinated"][sidx]),
    }
    hist = np.pad(policy[: sidx + 1], [(filter_len, 0), (0, 0)], constant_values=1)[-(filter_len + 1):]
    m["stringency_level_history_0"] = hist  # [filter_len + 1, n]
    # stringency levels of the days before the episode, for the beta delay (:744-760, :951-960)
    bd = m["beta_delay"]
    pre = np.ones((bd, n), np.int64)
    for k in range(bd):  # row k <-> day (sidx - bd + k)
        day = sidx - bd + k
        if day >= 0:
            pre[k] = policy[day]
    m["policy_before_start"] = pre
    # generate_observations() (covid19_env.py:958-962) indexes the same table WITHOUT the
    # "before the data begins -> level 1" guard of the step (:759-768): a negative day index
    # wraps to the end of the real-world policy table (Python indexing).  Kept as-is for parity.
    pre_obs = np.ones((bd, n), dtype=np.int64)
    for k in range(bd):
        pre_obs[k] = policy[sidx - bd + k]
    m["policy_before_start_obs"] = pre_obs
    m["episode_length"] = int(episode_length)
    m["real_world_policy_length"] = int(len(policy) - sidx)
    return m


def replay_tables(m, subsidy_kwargs, use_real_world_data, path_to_data_and_fitted_params=""):
    """What `use_real_world_policies` / `use_real_world_data` replay (covid19_env.py:188-231, 734-757, 815-818;
    covid19_components.py:181-186, 394-425), as tables indexed by the episode's day:
      stringency_policy [T, n]  the action ControlUSStateOpenCloseStatus takes at step t sits in row t - 1
                                (yesterday's recorded level);
      subsidy_level [T]         FederalGovernmentSubsidy's level at step t in slot t - 1: every recorded payment is
                                rounded to a number of levels and spread over the `subsidy_interval` days from its
                                date on.  (The reference keeps accumulating into the same array in later episodes of
                                the same object; the table is the first episode's, i.e. what a fresh environment does.)
      state [6, T + 1, n]       susceptible, infected, recovered, vaccinated, deaths, unemployed of day t (float64: the
                                global state stores their float32 cast, the economy step works on the table values)."""
    d = load_data(path_to_data_and_fitted_params or None)
    T, sidx = int(m["episode_length"]), int(m["start_date_index"])
    n = len(m["us_state_population"])
    if T > m["real_world_policy_length"]:
        raise AssertionError("The real-world policies are only available for {0} timesteps; so the 'episode_length' "
                             "in the environment configuration can only be at most {0}".format(m["real_world_policy_length"]))
    policy = np.asarray(d["rw_policy"]).astype(np.int64)[sidx:]
    out = {"stringency_policy": policy[:T].astype(np.uint8)}
    assert (out["stringency_policy"] <= m["num_stringency_levels"]).all()
    interval, levels = int(subsidy_kwargs["subsidy_interval"]), int(subsidy_kwargs["num_subsidy_levels"])
    per_level = (m["us_population"] * subsidy_kwargs["max_annual_subsidy_per_person"] / levels * interval / 365)
    subsidy = np.asarray(d["rw_subsidy"], np.float64).reshape(-1)[sidx:]
    arr = np.zeros(T + 1)
    for t in range(1, T + 1):
        amount = subsidy[t - 1]
        if amount > 0:
            lvl = np.round(amount / per_level)
            for k in range(t - 1, min(len(arr), t - 1 + interval)):
                arr[k] += lvl
    lv = arr[:T]
    assert ((0 <= lv) & (lv <= levels)).all(), "recorded subsidies exceed num_subsidy_levels (covid19_components.py:428)"
    out["subsidy_level"] = lv.astype(np.int32)
    if use_real_world_data:
        keys = ("susceptible", "infected", "recovered", "vaccinated", "deaths", "unemployed")
        if T + 1 > m["real_world_policy_length"]:
            raise AssertionError("use_real_world_data reads day episode_length of the recorded tables")
        out["state"] = np.stack([np.asarray(d["rw_" + k], np.float64)[sidx:sidx + T + 1] for k in keys])
        assert out["state"].shape == (6, T + 1, n)
    return out


def component_constants(m, subsidy_kwargs, vaccine_kwargs):
    """FederalGovernmentSubsidy / VaccinationCampaign constants."""
    out = {}
    pop = m["us_state_population"]
    out["max_daily_subsidy_per_state"] = pop * float(subsidy_kwargs.get("max_annual_subsidy_per_person", 20000)) / 365
    dv = int(vaccine_kwargs.get("daily_vaccines_per_million_people", 4500))
    di = int(vaccine_kwargs.get("delivery_interval", 1))
    out["num_vaccines_per_delivery"] = np.array(np.floor(di * (pop / 1e6) * dv), dtype=I32)
    vstart = datetime.strptime(vaccine_kwargs.get("vaccine_delivery_start_date", "2020-12-22"), "%Y-%m-%d")
    out["time_when_vaccine_delivery_begins"] = (vstart - m["start_date"]).days
    t_first = int(out["time_when_vaccine_delivery_begins"])
    while t_first % di != 0:
        t_first += 1
    out["t_first_delivery"] = t_first
    return out
