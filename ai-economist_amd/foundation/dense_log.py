"""Dense logs of one replica (reference: F/base/base_env.py:763-814, 984-1016 and the
components' get_dense_log: build.py:256-266, continuous_double_auction.py:670-679,
move.py:212-222, redistribution.py:1188-1202).

The device records the component events of each step for the logged replicas (tensors
"log_event_count" / "log_events", include/aie.h: AIE_EV_*); this module turns them and the
state tensors into the reference's dense-log dictionary:

    {"world": [...], "states": [...], "actions": [...], "rewards": [...],
     "Build": [...], "Trade": [...], "Gather": [...], "PeriodicTax": [...]}

Everything here is presentation of device results (reads, no dynamics).
"""
import numpy as np

EV_BUILD, EV_TRADE, EV_GATHER, EV_TAX, EV_TAX_BRACKET = 1, 2, 3, 4, 5
_RES = ["Stone", "Wood"]


def _host(t, e):
    return t[e].cpu().numpy()


class DenseLogger:
    """Collects the dense log of replica `e` of a batched environment."""

    def __init__(self, env, e=0):
        self.env = env
        self.e = int(e)
        self.names_a, self.names_p = env.action_subspace_names()
        self.log = None
        self.component_logs = None

    # ---- episode boundaries ----
    def begin_episode(self):
        self.log = {"world": [], "states": [], "actions": [], "rewards": []}
        self.component_logs = {}
        for comp in self.env.components:
            if comp.name in ("Build", "ContinuousDoubleAuction", "Gather"):
                self.component_logs[comp.shorthand] = []
            elif comp.name == "PeriodicBracketTax" and not comp.disable_taxes:
                self.component_logs[comp.shorthand] = []

    def finalize(self):
        """base_env.py:763-814: the closing world / states snapshot, then the component logs."""
        self.log["world"].append(self.world_snapshot())
        self.log["states"].append(self.states_snapshot())
        for k, v in self.component_logs.items():
            self.log[k] = list(v)
        return self.log

    # ---- snapshots of the state tensors ----
    def world_snapshot(self):
        t, e = self.env.backend.tensors, self.e
        if "cell_flags" not in t:  # map-less scenario: Maps(size, n_agents, [], []) has no entries
            return {}
        from .. import _cabi

        flags = _host(t["cell_flags"], e)
        owner = _host(t["house_owner"], e).astype(np.int16)
        # key order = landmark / resource registration order of the reference's Maps (world.py:33-77)
        return {
            "Stone": _host(t["stone"], e).astype(np.float64).tolist(),
            "Wood": _host(t["wood"], e).astype(np.float64).tolist(),
            "House": {"owner": owner.tolist(), "health": (owner >= 0).astype(np.float64).tolist()},
            "Water": ((flags & _cabi.CELL_WATER) != 0).astype(np.float64).tolist(),
            "StoneSourceBlock": ((flags & _cabi.CELL_STONE_SRC) != 0).astype(np.float64).tolist(),
            "WoodSourceBlock": ((flags & _cabi.CELL_WOOD_SRC) != 0).astype(np.float64).tolist(),
        }

    def states_snapshot(self):
        env, e = self.env, self.e
        t = env.backend.tensors
        n = env.n_agents
        coin, esc = _host(t["inv_coin"], e), _host(t["esc_coin"], e)
        labor = _host(t["labor"], e)
        spatial = "loc_r" in t
        if spatial:
            lr, lc = _host(t["loc_r"], e), _host(t["loc_c"], e)
            inv, escr = _host(t["inv_res"], e), _host(t["esc_res"], e)
        extra = {}
        for comp in env.components:
            for field, tensor in comp.agent_state_fields().items():
                extra[field] = _host(t[tensor], e)
        resources = [r for r in env.resources if r != "Coin"]
        out = {}
        for i in range(n):
            inventory, escrow = {"Coin": float(coin[i])}, {"Coin": float(esc[i])}
            for r in resources:
                k = _RES.index(r)
                inventory[r] = int(inv[k, i]) if spatial else 0
                escrow[r] = int(escr[k, i]) if spatial else 0
            st = {"loc": [int(lr[i]), int(lc[i])] if spatial else [-1, -1],  # never placed
                  "inventory": inventory, "escrow": escrow,
                  "endogenous": {"Labor": float(labor[i])}}
            for field, arr in extra.items():
                st[field] = float(arr[i])
            out[str(i)] = st
        zero = {r: 0 for r in env.resources}
        out["p"] = {"inventory": dict(zero), "escrow": dict(zero), "endogenous": {}}
        return out

    # ---- one step ----
    def before_step(self, actions_a, actions_p):
        """World / states / actions as they are when step() is entered (base_env.py:984-998)."""
        env, e = self.env, self.e
        t = int(env.backend.tensors["timestep"][e].item())
        self.log["world"].append(self.world_snapshot() if t % env.world_dense_log_frequency == 0 else {})
        self.log["states"].append(self.states_snapshot())
        acts = {}
        a = None if actions_a is None else actions_a[e].cpu().numpy().reshape(env.n_agents, -1)
        p = None if actions_p is None else actions_p[e].cpu().numpy().reshape(-1)
        for i in range(env.n_agents):
            acts[str(i)] = self._decode(self.names_a, None if a is None else a[i],
                                        env.multi_action_mode_agents)
        acts["p"] = self._decode(self.names_p, p, env.multi_action_mode_planner)
        self.log["actions"].append(acts)

    def reference_actions(self, actions_a, actions_p):
        """The logged replica's actions as the dictionary the reference's step() takes: {"0": index or list of
        sub-action indices, ..., "p": ...}."""
        env, e = self.env, self.e
        out = {}
        a = None if actions_a is None else actions_a[e].cpu().numpy().reshape(env.n_agents, -1)
        p = None if actions_p is None else actions_p[e].cpu().numpy().reshape(-1)
        for i in range(env.n_agents):
            if a is None:
                out[str(i)] = 0
            else:
                out[str(i)] = [int(x) for x in a[i]] if env.multi_action_mode_agents else int(a[i, 0])
        if p is None:
            out["p"] = 0
        else:
            out["p"] = [int(x) for x in p] if env.multi_action_mode_planner else int(p[0])
        return out

    @staticmethod
    def _decode(names, vec, multi):
        """{subspace name: chosen index > 0} (base_agent.py:97-114, 407-438)."""
        if vec is None or not names:
            return {}
        if multi:
            return {nm: int(v) for (nm, _), v in zip(names, vec) if v > 0}
        a = int(vec[0])
        if a <= 0:
            return {}
        base = 1
        for nm, dim in names:
            if a < base + dim:
                return {nm: a - base + 1}
            base += dim
        return {}

    def after_step(self):
        """Rewards and this step's component events (base_env.py:1015-1016)."""
        env, e = self.env, self.e
        t = env.backend.tensors
        ra, rp = _host(t["rewards_a"], e), t["rewards_p"][e].item()
        rew = {str(i): float(ra[i]) for i in range(env.n_agents)}
        rew["p"] = float(rp)
        self.log["rewards"].append(rew)

        cnt = int(t["log_event_count"][e].item())
        rows = _host(t["log_events"], e)[:cnt]
        vals = np.ascontiguousarray(rows[:, 10:12]).view(np.float64).reshape(-1) if cnt else np.zeros(0)
        builds, trades, gathers, tax_rows, schedule = [], [], [], [], []
        for row, f in zip(rows, vals):
            kind = int(row[0])
            if kind == EV_BUILD:
                builds.append({"builder": int(row[1]), "loc": [int(row[2]), int(row[3])], "income": float(f)})
            elif kind == EV_TRADE:
                price = int(row[6])
                trades.append({"commodity": _RES[int(row[1])], "buyer": int(row[3]), "bid": int(row[5]),
                               "bid_lifetime": int(row[8]), "seller": int(row[2]), "ask": int(row[4]),
                               "ask_lifetime": int(row[7]), "price": price, "cost": price, "income": price})
            elif kind == EV_GATHER:
                gathers.append({"agent": int(row[1]), "resource": _RES[int(row[2])], "n": int(row[3]),
                                "loc": [int(row[4]), int(row[5])]})
            elif kind == EV_TAX:
                tax_rows.append((int(row[1]), float(f)))
            elif kind == EV_TAX_BRACKET:
                schedule.append(float(f))
        logs = self.component_logs
        if "Build" in logs:
            logs["Build"].append(builds)
        if "Trade" in logs:
            logs["Trade"].append(trades)
        if "Gather" in logs:
            logs["Gather"].append(gathers)
        if "PeriodicTax" in logs:
            logs["PeriodicTax"].append(self._tax_entry(tax_rows, schedule) if tax_rows else [])

    def _tax_entry(self, tax_rows, schedule):
        """enact_taxes' tax_dict, redistribution.py:853-915."""
        env, e = self.env, self.e
        t = env.backend.tensors
        tax = env.get_component("PeriodicBracketTax")
        income = _host(t["tax_last_income"], e)
        marginal = _host(t["tax_last_marginal_rate"], e)
        entry = {"schedule": schedule, "cutoffs": [float(x) for x in tax.bracket_cutoffs]}
        net = 0
        for i, paid in tax_rows:
            net += np.float64(paid)
        lump = float(net / env.n_agents)
        for i, paid in tax_rows:
            inc = float(income[i])
            entry[str(i)] = {"income": inc, "tax_paid": paid, "marginal_rate": float(marginal[i]),
                             "effective_rate": float(np.float64(paid) / np.maximum(0.000001, inc)),
                             "lump_sum": lump}
        return entry


class CovidDenseLogger(DenseLogger):
    """Dense log of one replica of CovidAndEconomySimulation: the agents' and the planner's `state`
    dictionaries as covid19_env.py:848-922 (scenario_step), :1237-1288 (additional_reset_steps), :1124-1161
    (compute_reward) and covid19_components.py:180-221, 425-443, 615-627 leave them after every step -- rebuilt on the
    host from the replica's device tensors, one read per logged step.  The world log holds empty dictionaries (the
    scenario has no maps), the components have no dense logs, rewards are collated ({"a": [...], "p": x}).

    The reference's quirks are kept: fields a reset does not touch ("R0", "Total Unemployed", "New Subsidy Received",
    "Postsubsidy Productivity", "Current Open Close Stringency Level") carry over from the previous episode of the
    same environment object and are absent in the first snapshot of its first episode; "New Infections" / "New
    Deaths" difference against the previous snapshot's integer-cast totals."""

    STICKY = ("R0", "Total Unemployed", "New Subsidy Received", "Postsubsidy Productivity",
              "Current Open Close Stringency Level")

    def __init__(self, env, e=0):
        super().__init__(env, e)
        self.agent_states = None
        self.planner_state = None

    def begin_episode(self):
        import datetime

        self.log = {"world": [], "states": [], "actions": [], "rewards": []}
        self.component_logs = {}
        env, m = self.env, self.env.model
        n = env.n_agents
        rs = m["reset_agent_state"]
        date = datetime.datetime.strftime(m["start_date"], "%Y-%m-%d")
        old = self.agent_states
        self.agent_states = []
        for i in range(n):
            st = {"loc": [-1, -1], "inventory": {"Coin": 0}, "escrow": {"Coin": 0}, "endogenous": {"Labor": 0},
                  "Total Vaccinated": 0, "Vaccines Available": 0}
            for k in ("Total Susceptible", "New Infections", "Total Infected", "Total Recovered", "New Deaths",
                      "Total Deaths"):
                st[k] = int(rs[k][i])
            st["Health Index"] = [0.0]
            st["Economic Index"] = [0.0]
            st["Date"] = date
            if old is not None:
                for k in self.STICKY:
                    if k in old[i]:
                        st[k] = old[i][k]
            self.agent_states.append(st)
        oldp = self.planner_state
        p = {"inventory": {"Coin": 0}, "escrow": {"Coin": 0}, "endogenous": {}, "Total Subsidy": 0,
             "Current Subsidy Level": 0}
        for k in ("Total Susceptible", "New Infections", "Total Infected", "Total Recovered", "New Deaths",
                  "Total Deaths"):
            p[k] = int(np.sum([st[k] for st in self.agent_states]).astype(np.int32))
        p["Total Vaccinated"] = int(np.sum(rs["Total Vaccinated"]).astype(np.int32))
        p["Health Index"] = [0.0]
        p["Economic Index"] = [0.0]
        p["Date"] = date
        if oldp is not None:
            for k in ("Total Unemployed", "New Subsidy Provided", "Postsubsidy Productivity"):
                if k in oldp:
                    p[k] = oldp[k]
        self.planner_state = p
        self._total_subsidy = 0

    def world_snapshot(self):
        return {}

    def states_snapshot(self):
        import copy

        out = {str(i): copy.deepcopy(st) for i, st in enumerate(self.agent_states)}
        out["p"] = copy.deepcopy(self.planner_state)
        return out

    def after_step(self):
        import datetime

        env, e, m = self.env, self.e, self.env.model
        t = env.backend.tensors
        n = env.n_agents
        ra, rp = _host(t["rewards_a"], e), t["rewards_p"][e].item()
        self.log["rewards"].append({"a": [float(x) for x in ra], "p": float(rp)})

        ts = int(t["timestep"][e].item())
        f32 = {k: _host(t[k], e).astype(np.float32) for k in
               ("susceptible", "infected", "recovered", "deaths", "vaccinated", "unemployed", "subsidy",
                "postsubsidy_productivity", "health_index", "economic_index")}
        level_now = np.asarray(env.stringency_level(e, ts), np.float32)
        # sir_step (:1477-1497): beta from the stringency level beta_delay days back
        level_tmk = np.asarray(env.stringency_level(e, ts - int(m["beta_delay"]))).astype(np.int32)
        beta = (m["beta_intercepts"] * 1 + m["beta_slopes"] * 1 * level_tmk).astype(np.float32)
        r0 = beta / m["gamma"]
        date = datetime.datetime.strftime(m["start_date"] + datetime.timedelta(days=ts), "%Y-%m-%d")
        i32 = lambda x: x.astype(np.int32)  # noqa: E731
        S, I, R, V, U = (i32(f32[k]) for k in ("susceptible", "infected", "recovered", "vaccinated", "unemployed"))
        D = f32["deaths"]
        replay_data = bool(getattr(env, "use_real_world_data", False))
        cc = env.component_constants
        delivery = (ts >= int(cc["time_when_vaccine_delivery_begins"])
                    and ts % int(env.get_component("VaccinationCampaign").delivery_interval) == 0)
        for i, st in enumerate(self.agent_states):
            st["Current Open Close Stringency Level"] = float(level_now[i])
            if replay_data:  # nothing consumes the deliveries and sir_step (R0) does not run (covid19_env.py:734-757)
                if delivery:
                    st["Vaccines Available"] += int(cc["num_vaccines_per_delivery"][i])
            else:
                st["Vaccines Available"] = 0
                st["R0"] = float(r0[i])
            st["Total Susceptible"] = int(S[i])
            st["New Infections"] = int(np.asarray(f32["infected"][i] - st["Total Infected"]).astype(np.int32))
            st["Total Infected"] = int(I[i])
            st["Total Recovered"] = int(R[i])
            st["New Deaths"] = float(D[i] - np.int32(st["Total Deaths"]))
            st["Total Deaths"] = int(D[i].astype(np.int32))
            st["Total Vaccinated"] = int(V[i])
            st["Total Unemployed"] = int(U[i])
            st["New Subsidy Received"] = float(f32["subsidy"][i])
            st["Postsubsidy Productivity"] = float(f32["postsubsidy_productivity"][i])
            st["Date"] = date
            st["Health Index"] = [float(f32["health_index"][i])]
            st["Economic Index"] = [float(f32["economic_index"][i])]
        p = self.planner_state
        self._total_subsidy = self._total_subsidy + np.sum(f32["subsidy"])
        p["Total Subsidy"] = float(self._total_subsidy)
        p["Current Subsidy Level"] = int(t["subsidy_level"][e].item())
        p["Total Susceptible"] = int(np.sum(f32["susceptible"]).astype(np.int32))
        p["New Infections"] = int(np.asarray(np.sum(f32["infected"]) - p["Total Infected"]).astype(np.int32))
        p["Total Infected"] = int(np.sum(f32["infected"]).astype(np.int32))
        p["Total Recovered"] = int(np.sum(f32["recovered"]).astype(np.int32))
        p["New Deaths"] = int(np.asarray(np.sum(D) - p["Total Deaths"]).astype(np.int32))
        p["Total Deaths"] = int(np.sum(D).astype(np.int32))
        p["Total Vaccinated"] = int(np.sum(f32["vaccinated"]).astype(np.int32))
        p["Total Unemployed"] = int(np.sum(f32["unemployed"]).astype(np.int32))
        p["New Subsidy Provided"] = float(np.sum(f32["subsidy"]))
        p["Postsubsidy Productivity"] = float(np.sum(f32["postsubsidy_productivity"]))
        p["Date"] = date
        hp = _host(t["planner_health_economic_index"], e).astype(np.float32)
        p["Health Index"] = [float(hp[0])]
        p["Economic Index"] = [float(hp[1])]

    def finalize(self):
        self.log["world"].append(self.world_snapshot())
        self.log["states"].append(self.states_snapshot())
        return self.log
