"""`env.metrics` for the batched backend: the reference's scenario_metrics + component
get_metrics (F/base/base_env.py:420-432), every value an array over the E replicas.

Nothing here is on the step path: the numbers are derived, whenever asked for, from the state
tensors and from the few per-episode accumulators the kernels keep (`metrics_*` tensors:
trades per agent / commodity / side, tax-day sums).  The social metrics follow
F/scenarios/utils/social_metrics.py:10-80 and rewards.py:84-133, batched over the first axis.
"""
import numpy as np


# ---- social_metrics.py / rewards.py, batched: e [E, n] -> [E] ----
def get_gini(e):
    e = np.asarray(e, np.float64)
    n = e.shape[1]
    if n < 30:
        diff = np.abs(e[:, :, None] - e[:, None, :]).reshape(e.shape[0], -1).sum(axis=1)
        norm = 2 * n * e.sum(axis=1)
        return (diff / (norm + 1e-10)) / ((n - 1) / n)
    s = np.sort(e, axis=1)
    return 1 - (2 / (n + 1)) * np.sum(np.cumsum(s, axis=1) / (np.sum(s, axis=1, keepdims=True) + 1e-10), axis=1)


def get_equality(e):
    return 1 - get_gini(e)


def get_productivity(coin):
    return np.sum(coin, axis=1)


def coin_eq_times_productivity(coin, equality_weight):
    n = coin.shape[1]
    prod = get_productivity(coin) / n
    return (equality_weight * get_equality(coin) + (1 - equality_weight)) * prod


def _pareto_weights(coin):
    w = 1 / np.maximum(coin, 1)
    return w / np.sum(w, axis=1, keepdims=True)


def inv_income_weighted_coin_endowments(coin):
    return np.sum(coin * _pareto_weights(coin), axis=1)


def inv_income_weighted_utility(coin, utilities):
    return np.sum(utilities * _pareto_weights(coin), axis=1)


# ---- scenario_metrics: layout_from_file.py:595-650 / dynamic_layout.py (same body) ----
def gtb_scenario_metrics(env, t):
    """t: {tensor name: ndarray with leading E}.  Returns {key: ndarray [E]}."""
    n = env.n_agents
    coin = t["inv_coin"] + t["esc_coin"]
    util = t["util"]
    m = {}
    m["social/productivity"] = get_productivity(coin)
    m["social/equality"] = get_equality(coin)
    m["social_welfare/coin_eq_times_productivity"] = coin_eq_times_productivity(coin, 1.0)
    m["social_welfare/inv_income_weighted_coin_endow"] = inv_income_weighted_coin_endowments(coin)
    m["social_welfare/inv_income_weighted_utility"] = inv_income_weighted_utility(coin, util[:, :n])
    res = t["inv_res"] + t["esc_res"]  # [E, 2, n]: Stone, Wood
    zeros = np.zeros(coin.shape[0])
    for i in range(n):
        m["endow/%d/Coin" % i] = coin[:, i]
        m["endow/%d/Stone" % i] = res[:, 0, i].astype(np.float64)
        m["endow/%d/Wood" % i] = res[:, 1, i].astype(np.float64)
        m["endogenous/%d/Labor" % i] = t["labor"][:, i]
        m["util/%d" % i] = util[:, i]
    for r in ("Coin", "Stone", "Wood"):  # the planner never holds anything
        m["endow/p/%s" % r] = zeros
    m["util/p"] = util[:, n]
    # labor cost annealing (layout_from_file.py:249-267)
    if env.energy_warmup_constant <= 0:
        w = np.ones(coin.shape[0])
    else:
        v = t["completions"] if env.energy_warmup_method == "decay" else t["auto_warmup"]
        w = 1.0 - np.exp(-v.astype(np.float64) / env.energy_warmup_constant)
    m["labor/weighted_cost"] = env.energy_cost * w
    m["labor/warmup_integrator"] = t["auto_warmup"].astype(np.int64)
    return m


# ---- one-step-economy scenario_metrics: one_step_economy.py:207-277 ----
def ose_scenario_metrics(env, t):
    n = env.n_agents
    coin = t["inv_coin"] + t["esc_coin"]
    util = t["util"]
    m = {}
    m["social/productivity"] = get_productivity(coin)
    m["social/equality"] = get_equality(coin)
    m["social_welfare/coin_eq_times_productivity"] = coin_eq_times_productivity(coin, 1.0)
    # (the reference weights by pre-tax income here, :239-243)
    m["social_welfare/inv_income_weighted_utility"] = inv_income_weighted_utility(t["production"], util[:, :n])
    m["endow/avg_agent/Coin"] = np.mean(coin, axis=1)
    m["endogenous/avg_agent/Labor"] = np.mean(t["labor"], axis=1)
    m["util/avg_agent"] = np.mean(util[:, :n], axis=1)
    m["endow/p/Coin"] = np.zeros(coin.shape[0])
    m["util/p"] = util[:, n]
    return m


# ---- COVID-19 scenario_metrics: covid19_env.py:1613-1687 ----
def covid_scenario_metrics(env, t):
    """From the current state + the per-state day sums the kernel keeps.  Note that the
    reference averages over ALL episode_length days (days not simulated yet count as zeros)."""
    m = env.model
    names = m["us_state_names"]
    pop = np.asarray(m["us_state_population"], np.float64)
    us_pop = float(m["us_population"])
    T = float(env.episode_length)
    out = {}
    as_int = lambda x: np.trunc(np.asarray(x, np.float64))  # noqa: E731  (.astype(np.int32) of a float)
    for i, nm in enumerate(names):
        out["%s/infected (millions)" % nm] = as_int(t["infected"][:, i]) / 1e6
        out["%s/recovered (millions)" % nm] = as_int(t["recovered"][:, i]) / 1e6
        out["%s/deaths (millions)" % nm] = as_int(t["deaths"][:, i]) / 1e6
        out["%s/mean_unemployment_rate (%%)" % nm] = t["sum_unemployed"][:, i] / T / pop[i] * 100
        out["%s/mean_open_close_stringency_level" % nm] = t["sum_stringency_level"][:, i] / T
        out["%s/total_productivity (billion $)" % nm] = t["sum_postsubsidy_productivity"][:, i] / 1e9
        out["%s/health_index_at_end_of_episode" % nm] = t["health_index"][:, i].astype(np.float64)
        out["%s/economic_index_at_end_of_episode" % nm] = t["economic_index"][:, i].astype(np.float64)
    out["usa/vaccinated (% of population)"] = t["vaccinated"].astype(np.float64).sum(axis=1) / us_pop * 100
    out["usa/deaths (thousands)"] = t["deaths"].astype(np.float64).sum(axis=1) / 1e3
    out["usa/mean_unemployment_rate (%)"] = t["sum_unemployed"].sum(axis=1) / us_pop / T * 100
    out["usa/total_amount_subsidized (trillion $)"] = t["sum_subsidy"].sum(axis=1) / 1e12
    out["usa/total_productivity (trillion $)"] = t["sum_postsubsidy_productivity"].sum(axis=1) / 1e12
    out["usa/health_index_at_end_of_episode"] = t["planner_health_economic_index"][:, 0].astype(np.float64)
    out["usa/economic_index_at_end_of_episode"] = t["planner_health_economic_index"][:, 1].astype(np.float64)
    return out


# ---- component get_metrics ----
def build_metrics(comp, env, t):  # build.py:198-222
    owner = t["house_owner"].reshape(t["house_owner"].shape[0], -1)
    out = {}
    for i in range(env.n_agents):
        out["%d/n_builds" % i] = (owner == i).sum(axis=1)  # houses last for the whole episode
    out["total_builds"] = (owner >= 0).sum(axis=1)
    return out


def cda_metrics(comp, env, t):  # continuous_double_auction.py:585-641
    tr = t["metrics_cda_trades"].astype(np.float64)  # [E, side, commodity, agent, (n, sum of prices)]
    out = {}
    with np.errstate(invalid="ignore", divide="ignore"):
        for i in range(env.n_agents):
            for ci, cname in enumerate(("Stone", "Wood")):
                for side, prefix in ((0, "Sell"), (1, "Buy")):
                    cnt, tot = tr[:, side, ci, i, 0], tr[:, side, ci, i, 1]
                    avg = np.where(cnt > 0, tot / np.where(cnt > 0, cnt, 1), np.nan)
                    for k in ("price", "cost", "income"):  # one and the same number per trade (:301-308)
                        out["%d/%s%s/%s" % (i, prefix, cname, k)] = avg
                    out["%d/%s%s/n_sales" % (i, prefix, cname)] = cnt.astype(np.int64)
    out["n_trades"] = tr[:, 0, :, :, 0].sum(axis=(1, 2)).astype(np.int64)
    return out


def tax_metrics(comp, env, t):  # redistribution.py:1141-1186
    out = {}
    days = t["metrics_tax_days"].astype(np.float64)
    occ = t["metrics_tax_bracket_occupancy"].astype(np.float64)
    n_obs = np.maximum(1, occ.sum(axis=1))
    with np.errstate(invalid="ignore", divide="ignore"):
        for b, c in enumerate(comp.bracket_cutoffs):
            k = "{:03d}".format(int(c))
            out["avg_bracket_rate/%s" % k] = np.where(days > 0, t["metrics_tax_schedule_sum"][:, b] / days, np.nan)
            out["bracket_occupancy/%s" % k] = occ[:, b] / n_obs
        if not comp.disable_taxes:
            n = env.n_agents
            out["avg_effective_tax_rate"] = np.where(days > 0, t["metrics_tax_effective_rate_sum"] / (days * n), np.nan)
            out["total_collected_taxes"] = t["tax_total_collected"].astype(np.float64)
            coin = t["inv_coin"] + t["esc_coin"]
            rows = np.arange(coin.shape[0])
            for idx, tag in ((np.argmin(coin, axis=1), "poorest"), (np.argmax(coin, axis=1), "richest")):
                inc = t["metrics_tax_income_sum"][rows, idx]
                paid = t["metrics_tax_paid_sum"][rows, idx]
                out["avg_tax_rate/%s" % tag] = paid / np.maximum(0.001, inc)
            if comp.tax_model == "saez":  # running elasticity estimate, :1183-1185
                out["saez/estimated_elasticity"] = t["saez_elas"][:, 1].astype(np.float64)
    return out


COMPONENT_METRICS = {"Build": build_metrics, "ContinuousDoubleAuction": cda_metrics,
                     "PeriodicBracketTax": tax_metrics}


def env_metrics(env, tensors):
    """tensors: {name: ndarray [E, ...]} (device tensors already copied to the host)."""
    m = dict(env.scenario_metrics(tensors) or {})
    for comp in env.components:
        fn = COMPONENT_METRICS.get(comp.name)
        if fn is None:
            continue
        for k, v in fn(comp, env, tensors).items():
            m["%s/%s" % (comp.shorthand, k)] = v
    return m
