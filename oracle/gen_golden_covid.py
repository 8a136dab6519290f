"""TEST INFRASTRUCTURE ONLY.  Generates the COVID-19 golden fixtures by running the LIVE
reference (CPU path) in this container:

    python oracle/gen_golden_covid.py        # writes tests/golden/c4_covid_*.npz

Each fixture holds the env kwargs (JSON), the random action sequence, and what the reference
produced: per-step rewards, per-step SIR / unemployment / productivity state, masks, and the
full observation dict at a few timesteps."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_harness import load_reference_foundation  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")

BASE = dict(
    collate_agent_step_and_reset_data=True,
    components=[["ControlUSStateOpenCloseStatus", {"action_cooldown_period": 28}],
                ["FederalGovernmentSubsidy", {"num_subsidy_levels": 20, "subsidy_interval": 90,
                                              "max_annual_subsidy_per_person": 20000}],
                ["VaccinationCampaign", {"daily_vaccines_per_million_people": 3000, "delivery_interval": 1,
                                         "vaccine_delivery_start_date": "2021-01-12"}]],
    economic_reward_crra_eta=2, episode_length=540, flatten_masks=True, flatten_observations=False,
    health_priority_scaling_agents=0.3, health_priority_scaling_planner=0.45,
    infection_too_sick_to_work_rate=0.1, multi_action_mode_agents=False, multi_action_mode_planner=False,
    n_agents=51, path_to_data_and_fitted_params="", pop_between_age_18_65=0.6, risk_free_interest_rate=0.03,
    world_size=[1, 1], start_date="2020-03-22", use_real_world_data=False, use_real_world_policies=False)

CASES = {
    # the shipped run config (training/run_configs/covid_and_economy_environment.yaml), 330 days:
    # three subsidy rounds and the start of the vaccination campaign (day 296)
    "c4_covid_51ag": (dict(), 330, 0.6, 11),
    # earliest dates with complete real-world data, short cooldown,
    # vaccines every 3 days from day 19, no masking of actions at all, episode runs to `done`
    "c4_covid_variant": (dict(start_date="2020-02-25", episode_length=100, economic_reward_crra_eta=0.5,
                              health_priority_scaling_agents=1, health_priority_scaling_planner=2,
                              reward_normalization_factor=3, pop_between_age_18_65=0.55,
                              infection_too_sick_to_work_rate=0.2, risk_free_interest_rate=0.05,
                              components=[["ControlUSStateOpenCloseStatus", {"action_cooldown_period": 5}],
                                          ["FederalGovernmentSubsidy", {"num_subsidy_levels": 7, "subsidy_interval": 30,
                                                                        "max_annual_subsidy_per_person": 12000}],
                                          ["VaccinationCampaign", {"daily_vaccines_per_million_people": 4500,
                                                                   "delivery_interval": 3,
                                                                   "vaccine_delivery_start_date": "2020-03-15"}]]),
                         100, 0.0, 12),
}
STATE = [("susceptible", "Susceptible"), ("infected", "Infected"), ("recovered", "Recovered"), ("deaths", "Deaths"),
         ("vaccinated", "Vaccinated"), ("unemployed", "Unemployed"),
         ("postsubsidy_productivity", "Postsubsidy Productivity"), ("subsidy", "Subsidy"),
         ("stringency_level", "Stringency Level")]


def flat_obs(obs):
    out = {}
    for grp in ("a", "p"):
        for k, v in obs[grp].items():
            if k == "world-agent_index":
                continue
            out["obs_%s_%s" % (grp, k)] = np.asarray(v, np.float32)
    return out


def main():
    f = load_reference_foundation()
    for name, (over, steps, p_noop, seed) in CASES.items():
        cfg = json.loads(json.dumps(BASE))
        cfg.update(over)
        env = f.make_env_instance("CovidAndEconomySimulation", **cfg)
        obs = env.reset()
        rng = np.random.RandomState(seed)
        n, ns = 51, cfg["components"][1][1]["num_subsidy_levels"]
        acts_a = rng.randint(0, 11, size=(steps, n)).astype(np.int32)
        acts_a[rng.rand(steps, n) < p_noop] = 0
        acts_p = rng.randint(0, ns + 1, size=steps).astype(np.int32)
        snap_at = sorted(set([0, 1, 2, 28, 29, 30, 90, 91, steps // 2, steps - 1, steps]))
        out = {"config_json": np.array(json.dumps(cfg)), "actions_a": acts_a, "actions_p": acts_p,
               "snap_at": np.array(snap_at, np.int32)}
        rew = np.zeros((steps, n + 1), np.float64)
        masks_a = np.zeros((steps + 1, 11, n), np.uint8)
        masks_p = np.zeros((steps + 1, ns + 1), np.uint8)
        done = np.zeros(steps, np.uint8)

        def record(t, obs):
            masks_a[t] = np.asarray(obs["a"]["action_mask"])
            masks_p[t] = np.asarray(obs["p"]["action_mask"])
            if t in snap_at:
                for k, v in flat_obs(obs).items():
                    out["snap%d_%s" % (t, k)] = v

        record(0, obs)
        for t in range(1, steps + 1):
            a = {str(i): int(acts_a[t - 1, i]) for i in range(n)}
            a["p"] = int(acts_p[t - 1])
            obs, r, d, _ = env.step(a)
            rew[t - 1, :n] = np.asarray(r["a"], np.float64)
            rew[t - 1, n] = float(r["p"])
            done[t - 1] = d["__all__"]
            record(t, obs)
        gs = env.world.global_state
        for key, ref_key in STATE:
            out["state_" + key] = np.asarray(gs[ref_key][: steps + 1], np.float32)
        out["rewards"] = rew
        out["masks_a"] = masks_a
        out["masks_p"] = masks_p
        out["done"] = done
        path = os.path.join(GOLDEN, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
