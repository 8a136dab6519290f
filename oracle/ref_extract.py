"""Test infrastructure ONLY (see oracle/ref_harness.py).

Flattens the state of a live *reference* Foundation env object (Python dict-of-dicts,
see /root/reference/ai_economist/foundation/base/world.py:13-329 and
base_agent.py:62) into the named flat arrays that `include/aie.h` defines for the
device record, and packs reference observations into the batched tensor naming.

Used by oracle/gen_golden.py (fixtures) and by the live-oracle parity tests.
"""
import numpy as np

RES = ["Stone", "Wood"]  # sorted collectible resources == commodity order


def pack_order(agent, price, lifetime):
    return np.int32(int(agent) | (int(price) << 8) | (int(lifetime) << 16))


def _saez_state(c, s, n_agents):
    """tax_model "saez": sample buffer (zero-padded to the device capacity), estimates, rates."""
    if c.tax_model != "saez":
        return
    buf = np.array(c._local_saez_buffer, np.float64).reshape(-1, 2)
    pad = np.zeros((c._buffer_size + n_agents, 2))
    pad[: len(buf)] = buf
    s["saez_buffer_len"] = np.array(len(buf), np.int32)
    s["saez_reached_min_samples"] = np.array(int(c._reached_min_samples), np.int32)
    s["saez_buffer_filled"] = pad
    s["saez_elas"] = np.array([c.elas_t, c.elas_tm1, c.log_z0_t, c.log_z0_tm1], np.float64)
    s["saez_running_avg_tax_rates"] = np.array(c.running_avg_tax_rates, np.float64)
    s["tax_saez_bracket_rates"] = np.array(c.curr_bracket_tax_rates, np.float64)


def extract_state_one_step_economy(env):
    w = env.world
    ag = w.agents
    s = {}
    s["inv_coin"] = np.array([a.inventory["Coin"] for a in ag], np.float64)
    s["esc_coin"] = np.array([a.escrow["Coin"] for a in ag], np.float64)
    s["labor"] = np.array([a.endogenous["Labor"] for a in ag], np.float64)
    s["skill"] = np.array([a.state.get("skill", 0.0) for a in ag], np.float64)
    s["production"] = np.array([a.state.get("production", 0.0) for a in ag], np.float64)
    com = env.curr_optimization_metrics
    s["util"] = np.array([com.get(a.idx, com.get(str(a.idx), 0.0)) for a in ag]
                         + [com.get(w.planner.idx, 0.0)], np.float64)
    comps = {c.name: c for c in env.components}
    if "PeriodicBracketTax" in comps:
        c = comps["PeriodicBracketTax"]
        s["tax_cycle_pos"] = np.array(c.tax_cycle_pos, np.int32)
        s["tax_last_completions"] = np.array(c._last_completions, np.int32)
        s["tax_rate_idx"] = np.array(c.curr_rate_indices, np.int32)
        s["tax_last_coin"] = np.array(c.last_coin, np.float64)
        s["tax_last_income"] = np.array(c.last_income, np.float64)
        s["tax_last_marginal_rate"] = np.array(c.last_marginal_rate, np.float64)
        s["tax_total_collected"] = np.array(c.total_collected_taxes, np.float64)
        _saez_state(c, s, len(ag))
    if "SimpleLabor" in comps:
        s["labor_first_step"] = np.array(int(comps["SimpleLabor"].is_first_step), np.int32)
    s["timestep"] = np.array(w.timestep, np.int32)
    s["completions"] = np.array(env._completions, np.int32)
    st = np.random.get_state()
    s["mt"] = np.array(st[1], np.uint32)
    s["mt_pos"] = np.array(st[2], np.int32)
    s["mt_has_gauss"] = np.array(st[3], np.int32)
    s["mt_gauss"] = np.array(st[4], np.float64)
    return s


def extract_state(env):
    """Returns {field_name: ndarray} for ONE reference env (no leading env dim)."""
    if env.name == "one-step-economy":
        return extract_state_one_step_economy(env)
    w = env.world
    n = env.n_agents
    H, W = env.world_size
    maps = w.maps
    keys = list(maps.keys())
    s = {}
    s["stone"] = maps.get("Stone").astype(np.uint8)
    s["wood"] = maps.get("Wood").astype(np.uint8)
    assert np.all(maps.get("Stone") == s["stone"]) and np.all(maps.get("Wood") == s["wood"])
    owner = maps.get("House", owner=True)
    health = maps.get("House")
    assert np.all((health > 0) == (owner >= 0))
    s["house_owner"] = owner.astype(np.int8)
    s["stone_src"] = maps.get("StoneSourceBlock").astype(np.uint8)
    s["wood_src"] = maps.get("WoodSourceBlock").astype(np.uint8)
    s["water"] = (
        maps.get("Water").astype(np.uint8) if "Water" in keys else np.zeros((H, W), np.uint8)
    )

    ag = w.agents
    s["loc_r"] = np.array([a.loc[0] for a in ag], np.int32)
    s["loc_c"] = np.array([a.loc[1] for a in ag], np.int32)
    inv = np.array([[a.inventory[r] for a in ag] for r in RES], np.float64)
    esc = np.array([[a.escrow[r] for a in ag] for r in RES], np.float64)
    assert np.all(inv == np.round(inv)) and np.all(esc == np.round(esc))
    s["inv_res"] = inv.astype(np.int32)
    s["esc_res"] = esc.astype(np.int32)
    s["inv_coin"] = np.array([a.inventory["Coin"] for a in ag], np.float64)
    s["esc_coin"] = np.array([a.escrow["Coin"] for a in ag], np.float64)
    s["labor"] = np.array([a.endogenous["Labor"] for a in ag], np.float64)
    s["build_payment"] = np.array([a.state.get("build_payment", 0.0) for a in ag], np.float64)
    s["build_skill"] = np.array([a.state.get("build_skill", 0.0) for a in ag], np.float64)
    s["bonus_gather_prob"] = np.array(
        [a.state.get("bonus_gather_prob", 0.0) for a in ag], np.float64
    )
    com = env.curr_optimization_metric
    s["util"] = np.array([com[a.idx] for a in ag] + [com[w.planner.idx]], np.float64)

    comps = {c.name: c for c in env.components}
    if "ContinuousDoubleAuction" in comps:
        c = comps["ContinuousDoubleAuction"]
        assert c.commodities == RES
        M = n * c.max_num_orders
        P = c.max_bid_ask + 1
        s["cda_n_bids"] = np.array([len(c.bids[r]) for r in RES], np.int32)
        s["cda_n_asks"] = np.array([len(c.asks[r]) for r in RES], np.int32)
        bids = np.zeros((2, M), np.int32)
        asks = np.zeros((2, M), np.int32)
        for ri, r in enumerate(RES):
            for k, b in enumerate(c.bids[r]):
                bids[ri, k] = pack_order(b["buyer"], b["bid"], b["bid_lifetime"])
            for k, a in enumerate(c.asks[r]):
                asks[ri, k] = pack_order(a["seller"], a["ask"], a["ask_lifetime"])
        s["cda_bids"] = bids
        s["cda_asks"] = asks
        s["cda_n_orders"] = np.array(
            [[c.n_orders[r][i] for i in range(n)] for r in RES], np.int32
        )
        s["cda_bid_hist"] = np.array(
            [[c.bid_hists[r][i] for i in range(n)] for r in RES], np.float64
        ).astype(np.uint8)
        s["cda_ask_hist"] = np.array(
            [[c.ask_hists[r][i] for i in range(n)] for r in RES], np.float64
        ).astype(np.uint8)
        s["cda_price_history"] = np.array(
            [[c.price_history[r][i] for i in range(n)] for r in RES], np.float64
        ).reshape(2, n, P)
    if "PeriodicBracketTax" in comps:
        c = comps["PeriodicBracketTax"]
        s["tax_cycle_pos"] = np.array(c.tax_cycle_pos, np.int32)
        s["tax_last_completions"] = np.array(c._last_completions, np.int32)
        s["tax_rate_idx"] = np.array(c.curr_rate_indices, np.int32)
        s["tax_last_coin"] = np.array(c.last_coin, np.float64)
        s["tax_last_income"] = np.array(c.last_income, np.float64)
        s["tax_last_marginal_rate"] = np.array(c.last_marginal_rate, np.float64)
        s["tax_total_collected"] = np.array(c.total_collected_taxes, np.float64)
        _saez_state(c, s, len(ag))
    s["timestep"] = np.array(w.timestep, np.int32)
    s["completions"] = np.array(env._completions, np.int32)
    s["auto_warmup"] = np.array(env._auto_warmup_integrator, np.int32)
    st = np.random.get_state()
    s["mt"] = np.array(st[1], np.uint32)
    s["mt_pos"] = np.array(st[2], np.int32)
    s["mt_has_gauss"] = np.array(st[3], np.int32)
    s["mt_gauss"] = np.array(st[4], np.float64)
    return s


def extract_obs(env, obs):
    """Reference obs dict -> {tensor_name: ndarray} in the batched naming (no env dim).

    obs_a_<key> stacks agents on a leading axis; obs_p_<key> is the planner's.
    `p{i}` planner sub-observations are stacked into obs_p_agents [n, k].
    """
    n = env.n_agents
    out = {}
    for k in obs["0"].keys():
        out["obs_a_" + k] = np.stack([np.asarray(obs[str(i)][k]) for i in range(n)])
    p = obs["p"]
    pa = []
    for k, v in p.items():
        if k.startswith("p") and k[1:].isdigit():
            continue
        out["obs_p_" + k] = np.asarray(v)
    if "p0" in p:
        pa = np.stack([np.asarray(p["p%d" % i], np.float32) for i in range(n)])
        out["obs_p_agents"] = pa
    for k in list(out.keys()):
        if k.endswith("time"):
            out[k] = out[k].astype(np.float32)
    return out


def rewards_array(env, rew):
    n = env.n_agents
    return np.array([rew[str(i)] for i in range(n)] + [rew["p"]], np.float64)
