"""COVID-19 scenario (BASELINE configs[3]): the NumPy oracle and the HIP path against the
fixtures the live reference produced (oracle/gen_golden_covid.py), and against each other on
batched random rollouts.

Tolerances (the reference's own CPU<->CUDA tolerance lives in un-vendored WarpDrive, so these
are ours): the SIR state is float32 arithmetic with IEEE basic operations only and is compared
at rtol 1e-6 (HIP vs oracle: a few float32 ulps would show a misplaced cast); unemployment /
productivity go through exp/log and a reordered float64 filter sum: rtol 1e-5; rewards are
min-max normalised differences of nearly equal float32 numbers: atol 1e-5 (the 1e-5 of BASELINE's north star;
measured at 8192 replicas x 64 days: max 2.4e-6, tests/test_gpu_full_size.py)."""
import os
import sys

import numpy as np
import pytest

from helpers import ROOT, covid_golden_names, load_covid_golden

sys.path.insert(0, os.path.join(ROOT, "oracle"))

STATE_TOL = dict(susceptible=1e-6, infected=1e-6, recovered=1e-6, deaths=1e-6, vaccinated=1e-6,
                 unemployed=1e-5, postsubsidy_productivity=1e-5, subsidy=1e-6)


def model_for(cfg):
    """Constants from the product's host-side builder (itself pinned against the live
    reference's attributes in tests/test_covid_reference.py)."""
    import ai_economist_amd  # noqa: F401
    from ai_economist_amd.foundation.scenarios.covid19_model import build_model, component_constants

    kw = {k: cfg[k] for k in ("start_date", "pop_between_age_18_65", "infection_too_sick_to_work_rate",
                              "risk_free_interest_rate", "economic_reward_crra_eta",
                              "health_priority_scaling_agents", "health_priority_scaling_planner",
                              "episode_length")}
    kw["reward_normalization_factor"] = cfg.get("reward_normalization_factor", 1)
    m = build_model(**kw)
    comps = dict(cfg["components"])
    c = component_constants(m, comps["FederalGovernmentSubsidy"], comps["VaccinationCampaign"])
    return m, c


def make_oracle(cfg, n_envs):
    from covid_oracle import CovidOracle

    m, c = model_for(cfg)
    comps = dict(cfg["components"])
    return CovidOracle(m, c, n_envs=n_envs,
                       action_cooldown_period=comps["ControlUSStateOpenCloseStatus"]["action_cooldown_period"],
                       subsidy_interval=comps["FederalGovernmentSubsidy"]["subsidy_interval"],
                       num_subsidy_levels=comps["FederalGovernmentSubsidy"]["num_subsidy_levels"],
                       delivery_interval=comps["VaccinationCampaign"]["delivery_interval"],
                       episode_length=cfg["episode_length"])


def check_against_golden(g, t, state, obs, rew_a, rew_p, done, where):
    """state/obs: dicts of arrays for ONE replica at timestep t."""
    for k, tol in STATE_TOL.items():
        np.testing.assert_allclose(np.asarray(state[k], np.float64), g["state_" + k][t], rtol=tol, atol=1e-3,
                                   err_msg="%s t=%d state %s" % (where, t, k))
    assert np.array_equal(np.asarray(state["stringency_level"]).astype(np.int64),
                          g["state_stringency_level"][t].astype(np.int64)), "%s t=%d stringency" % (where, t)
    assert np.array_equal(np.asarray(obs["obs_a_action_mask"]).astype(np.uint8), g["masks_a"][t]), \
        "%s t=%d agent masks" % (where, t)
    assert np.array_equal(np.asarray(obs["obs_p_action_mask"]).astype(np.uint8), g["masks_p"][t]), \
        "%s t=%d planner mask" % (where, t)
    if t in g["snap_at"]:
        for k in g:
            pre = "snap%d_" % t
            if k.startswith(pre):
                np.testing.assert_allclose(np.asarray(obs[k[len(pre):]], np.float64).reshape(g[k].shape), g[k],
                                           rtol=1e-5, atol=1e-6, err_msg="%s t=%d %s" % (where, t, k))
    if t > 0:
        want = g["rewards"][t - 1]
        np.testing.assert_allclose(np.asarray(rew_a, np.float64), want[:-1], rtol=0, atol=1e-5,
                                   err_msg="%s t=%d agent rewards" % (where, t))
        np.testing.assert_allclose(float(rew_p), want[-1], rtol=0, atol=1e-5, err_msg="%s t=%d planner reward" % (where, t))
        assert int(done) == int(g["done"][t - 1]), "%s t=%d done" % (where, t)


@pytest.mark.parametrize("name", covid_golden_names())
def test_covid_oracle_matches_reference_golden(name):
    g = load_covid_golden(name)
    o = make_oracle(g["cfg"], n_envs=1)
    obs = o.reset()
    steps = len(g["actions_p"])
    pick = lambda d: {k: v[0] for k, v in d.items()}  # noqa: E731
    st = pick(o.state())
    st["stringency_level"] = o.stringency[0, 0]
    check_against_golden(g, 0, st, pick(obs), None, None, 0, name + "/oracle")
    for t in range(1, steps + 1):
        obs = o.step(g["actions_a"][t - 1][None], g["actions_p"][t - 1][None])
        st = pick(o.state())
        check_against_golden(g, t, st, pick(obs), o.rew_a[0], o.rew_p[0], o.done[0], name + "/oracle")


# ------------------------------------------------------------------------------------------
# GPU: the HIP path through the C ABI
# ------------------------------------------------------------------------------------------
def hip_env(cfg, n_envs, **extra):
    from ai_economist_amd import foundation

    return foundation.make_env_instance("CovidAndEconomySimulation", n_envs=n_envs, **dict(cfg, **extra))


# Both instantiations of the step kernel are compared with the oracle / the reference's fixtures: the default
# (aie_covid_step_kernel<F, false>: the reference's 600-tap window sums) and the opt-in O(1) recurrence
# (filter_recurrence=True, <F, true>: what bench.py's C4 line runs).
FILTER_MODES = pytest.mark.parametrize("recurrence", [False, True], ids=["window-sums", "recurrence"])


def hip_state(env, e):
    t = env.tensors
    ts = int(t["timestep"][e].item())
    L = env.model["filter_len"]
    tau = L + ts
    st = {k: t[k][e].cpu().numpy() for k in STATE_TOL}
    st["stringency_level"] = t["stringency_ring"][e, tau % 32].cpu().numpy()  # (the ring holds the 32 most recent days)
    assert np.array_equal(st["stringency_level"], env.stringency_level(e, ts))
    return st


def hip_obs(env, e):
    return {k: v[e].cpu().numpy() for k, v in env.tensors.items() if k.startswith("obs_")}


@pytest.mark.gpu
@FILTER_MODES
@pytest.mark.parametrize("name", covid_golden_names())
def test_covid_hip_matches_reference_golden(name, recurrence):
    import torch

    g = load_covid_golden(name)
    env = hip_env(g["cfg"], n_envs=3, filter_recurrence=recurrence)
    assert ("filter_discounted_delta_sums" in env.tensors) == recurrence
    env.reset()
    t = env.tensors
    steps = len(g["actions_p"])
    check_against_golden(g, 0, hip_state(env, 1), hip_obs(env, 1), None, None, 0, name + "/hip")
    for k in range(1, steps + 1):
        a = torch.as_tensor(np.repeat(g["actions_a"][k - 1][None], 3, axis=0), dtype=torch.int32, device="cuda")
        p = torch.as_tensor(np.repeat(g["actions_p"][k - 1][None, None], 3, axis=0), dtype=torch.int32, device="cuda")
        env.step({"a": a, "p": p})
        if k % 7 == 0 or k < 35 or k in g["snap_at"] or k > steps - 3:
            check_against_golden(g, k, hip_state(env, 1), hip_obs(env, 1), t["rewards_a"][1].cpu().numpy(),
                                 t["rewards_p"][1].item(), t["done"][1].item(), name + "/hip")
    # replicas fed identical actions stay identical
    for key in ("susceptible", "unemployed", "rewards_a", "obs_a_world-agent_state"):
        assert torch.equal(t[key][0], t[key][2]), key


@pytest.mark.gpu
@FILTER_MODES
@pytest.mark.parametrize("name,E,T", [("c4_covid_51ag", 64, 130), ("c4_covid_variant", 33, 100)])
def test_covid_hip_matches_oracle_on_batched_rollouts(name, E, T, recurrence):
    """Every replica gets its own action stream; masked resets mid-run; HIP vs oracle."""
    import torch

    g = load_covid_golden(name)
    cfg = g["cfg"]
    env = hip_env(cfg, n_envs=E, filter_recurrence=recurrence)
    o = make_oracle(cfg, n_envs=E)
    env.reset()
    o.reset()
    t = env.tensors
    rng = np.random.RandomState(5)
    ns = dict(cfg["components"])["FederalGovernmentSubsidy"]["num_subsidy_levels"]

    def compare(where):
        st = o.state()
        for k, tol in STATE_TOL.items():
            np.testing.assert_allclose(t[k].cpu().numpy().astype(np.float64), st[k], rtol=tol, atol=1e-3,
                                       err_msg="%s %s" % (where, k))
        assert np.array_equal(t["cooldown_until"].cpu().numpy(), st["cooldown_until"]), where
        assert np.array_equal(t["subsidy_level"].cpu().numpy(), st["subsidy_level"]), where
        for k in ("health_index", "economic_index", "planner_health_economic_index", "sum_unemployed",
                  "sum_stringency_level", "sum_postsubsidy_productivity", "sum_subsidy"):
            np.testing.assert_allclose(t[k].cpu().numpy().astype(np.float64), st[k], rtol=2e-5, atol=2e-4,
                                       err_msg="%s %s" % (where, k))
        # env.metrics from the device state vs the same formulas on the oracle's state
        from ai_economist_amd.foundation.metrics import covid_scenario_metrics

        got, want = env.metrics, covid_scenario_metrics(env, st)
        assert sorted(got) == sorted(want) and len(got) == 51 * 8 + 7
        for k, v in want.items():
            np.testing.assert_allclose(got[k], v, rtol=2e-5, atol=2e-4, err_msg="%s metric %s" % (where, k))
        oo = o.observe()
        for k, v in oo.items():
            np.testing.assert_allclose(t[k].cpu().numpy().reshape(v.shape), v, rtol=1e-5, atol=1e-6,
                                       err_msg="%s %s" % (where, k))

    compare("reset")
    for k in range(1, T + 1):
        a = rng.randint(0, 11, size=(E, 51)).astype(np.int32)
        a[rng.rand(E, 51) < 0.5] = 0
        p = rng.randint(0, ns + 1, size=(E,)).astype(np.int32)
        env.step({"a": torch.as_tensor(a, device="cuda"), "p": torch.as_tensor(p[:, None], device="cuda")})
        o.step(a, p)
        np.testing.assert_allclose(t["rewards_a"].cpu().numpy(), o.rew_a, rtol=0, atol=1e-5, err_msg="step %d" % k)
        np.testing.assert_allclose(t["rewards_p"].cpu().numpy(), o.rew_p, rtol=0, atol=1e-5, err_msg="step %d" % k)
        assert np.array_equal(t["done"].cpu().numpy(), o.done), "done at step %d" % k
        if k % 10 == 0 or k == T:
            compare("step %d" % k)


@pytest.mark.gpu
@FILTER_MODES
@pytest.mark.parametrize("seed", range(10))
def test_covid_hip_matches_oracle_on_random_configs(seed, recurrence):
    """helpers.random_covid_config (seeds 0..7 are pinned against the live reference on CPU):
    HIP vs oracle over a whole episode, then a masked reset and a few more days."""
    import torch
    from helpers import random_covid_config

    cfg = random_covid_config(seed)
    E = 12
    env = hip_env(cfg, n_envs=E, filter_recurrence=recurrence)
    o = make_oracle(cfg, n_envs=E)
    env.reset()
    o.reset()
    t = env.tensors
    rng = np.random.RandomState(seed)
    ns = dict(cfg["components"])["FederalGovernmentSubsidy"]["num_subsidy_levels"]

    def compare(where):
        st = o.state()
        for k, tol in STATE_TOL.items():
            np.testing.assert_allclose(t[k].cpu().numpy().astype(np.float64), st[k], rtol=tol, atol=1e-3,
                                       err_msg="%s %s %r" % (where, k, cfg))
        assert np.array_equal(t["cooldown_until"].cpu().numpy(), st["cooldown_until"]), where
        for k, v in o.observe().items():
            np.testing.assert_allclose(t[k].cpu().numpy().reshape(v.shape), v, rtol=1e-5, atol=1e-6,
                                       err_msg="%s %s" % (where, k))

    compare("reset")
    T = cfg["episode_length"]
    for k in range(1, T + 1):
        a = rng.randint(0, 11, size=(E, 51)).astype(np.int32)
        a[rng.rand(E, 51) < 0.5] = 0
        p = rng.randint(0, ns + 1, size=(E,)).astype(np.int32)
        env.step({"a": torch.as_tensor(a, device="cuda"), "p": torch.as_tensor(p[:, None], device="cuda")})
        o.step(a, p)
        np.testing.assert_allclose(t["rewards_a"].cpu().numpy(), o.rew_a, rtol=0, atol=1e-5, err_msg="step %d" % k)
        np.testing.assert_allclose(t["rewards_p"].cpu().numpy(), o.rew_p, rtol=0, atol=1e-5, err_msg="step %d" % k)
        assert np.array_equal(t["done"].cpu().numpy(), o.done), "done at step %d" % k
        if k % 15 == 0 or k == T:
            compare("step %d" % k)
    assert bool(t["done"].all())
    env.reset(t["done"])
    o.reset()
    compare("second reset")


@pytest.mark.gpu
@FILTER_MODES
def test_covid_hip_follows_the_reference_consistency_procedure(recurrence):
    """The reference's CPU<->GPU consistency check (tests/run_covid19_cpu_gpu_consistency_checks.py:43-101: run config
    covid_and_economy_environment.yaml, 3 environments x 2 episodes x 540 steps, uniform random actions), with this
    library in the place of its CUDA path and the NumPy oracle -- which tests/test_covid_reference.py follows through
    the same procedure beside the LIVE reference -- in the place of its CPU path.  Its comparator's tolerance lives in
    un-vendored WarpDrive; ours are the module docstring's."""
    import torch

    cfg = load_covid_golden("c4_covid_51ag")["cfg"]
    assert cfg["episode_length"] == 540
    E, EPISODES, T = 3, 2, 540
    env = hip_env(cfg, n_envs=E, filter_recurrence=recurrence)
    o = make_oracle(cfg, n_envs=E)
    env.reset()
    o.reset()
    t = env.tensors
    rng = np.random.RandomState(17)

    def compare(where):
        st = o.state()
        for k, tol in STATE_TOL.items():
            np.testing.assert_allclose(t[k].cpu().numpy().astype(np.float64), st[k], rtol=tol, atol=1e-3,
                                       err_msg="%s %s" % (where, k))
        assert np.array_equal(t["cooldown_until"].cpu().numpy(), st["cooldown_until"]), where
        assert np.array_equal(t["subsidy_level"].cpu().numpy(), st["subsidy_level"]), where
        for k, v in o.observe().items():
            np.testing.assert_allclose(t[k].cpu().numpy().reshape(v.shape), v, rtol=1e-5, atol=1e-6,
                                       err_msg="%s %s" % (where, k))

    for ep in range(EPISODES):
        compare("episode %d reset" % ep)
        for k in range(1, T + 1):
            a = rng.randint(0, 11, size=(E, 51)).astype(np.int32)
            p = rng.randint(0, 21, size=(E,)).astype(np.int32)
            env.step({"a": torch.as_tensor(a, device="cuda"), "p": torch.as_tensor(p[:, None], device="cuda")})
            o.step(a, p)
            np.testing.assert_allclose(t["rewards_a"].cpu().numpy(), o.rew_a, rtol=0, atol=1e-5,
                                       err_msg="episode %d day %d" % (ep, k))
            np.testing.assert_allclose(t["rewards_p"].cpu().numpy(), o.rew_p, rtol=0, atol=1e-5,
                                       err_msg="episode %d day %d" % (ep, k))
            assert np.array_equal(t["done"].cpu().numpy(), o.done) and bool(o.done.all()) == (k == T)
            if k % 30 == 0 or k == T:
                compare("episode %d day %d" % (ep, k))
        env.reset(t["done"])
        o.reset()


@pytest.mark.gpu
def test_covid_masked_action_sampler_respects_masks():
    """aie_sample_masked_actions on the collated COVID observations (states' masks are rows [1 + levels, n] of the
    replica's block, base_env.py:141-145 semantics): a stringency level under cool-down is never drawn, the planner's
    subsidy level only on the days its mask opens, every allowed entry is reached, and the masked rollout still
    follows the oracle."""
    import torch

    g = load_covid_golden("c4_covid_variant")
    cfg = g["cfg"]
    E, T = 48, 70
    env = hip_env(cfg, n_envs=E)
    o = make_oracle(cfg, n_envs=E)
    env.reset()
    o.reset()
    t = env.tensors
    be = env.backend
    seen_a = np.zeros(t["obs_a_action_mask"].shape[1], bool)
    seen_p = np.zeros(t["obs_p_action_mask"].shape[1], bool)
    closed_some_day = False
    for k in range(1, T + 1):
        a, p = be.sample_masked_actions(seed=11)
        torch.cuda.synchronize()
        ma = t["obs_a_action_mask"].cpu().numpy()  # [E, 1 + levels, n]
        mp = t["obs_p_action_mask"].cpu().numpy()  # [E, 1 + subsidy levels]
        an, pn = a.cpu().numpy()[:, :, 0], p.cpu().numpy()[:, 0]
        sel = np.take_along_axis(ma, an[:, None, :], axis=1)[:, 0, :]
        assert np.all(sel == 1.0), "day %d: a masked stringency level was sampled" % k
        assert np.all(np.take_along_axis(mp, pn[:, None], axis=1) == 1.0), "day %d: a masked subsidy level was sampled" % k
        closed_some_day |= bool((ma[:, 1:, :] == 0).any())
        seen_a[np.unique(an)] = True
        seen_p[np.unique(pn)] = True
        env.step({"a": a, "p": p})
        o.step(an.astype(np.int32), pn.astype(np.int32))
        np.testing.assert_allclose(t["rewards_a"].cpu().numpy(), o.rew_a, rtol=0, atol=1e-5, err_msg="day %d" % k)
    assert closed_some_day, "the rollout never met a closed mask entry"
    assert seen_a.all() and seen_p.all(), "entries never sampled: %s %s" % (np.where(~seen_a)[0], np.where(~seen_p)[0])
    st = o.state()
    assert np.array_equal(t["cooldown_until"].cpu().numpy(), st["cooldown_until"])
    assert np.array_equal(t["subsidy_level"].cpu().numpy(), st["subsidy_level"])


@pytest.mark.gpu
def test_covid_masked_reset_and_config_errors():
    import torch

    g = load_covid_golden("c4_covid_variant")
    cfg = g["cfg"]
    env = hip_env(cfg, n_envs=4)
    env.reset()
    t = env.tensors
    s0 = t["susceptible"].clone()
    for k in range(12):
        a = torch.full((4, 51), (k % 10) + 1, dtype=torch.int32, device="cuda")
        env.step({"a": a, "p": torch.full((4, 1), 3, dtype=torch.int32, device="cuda")})
    assert not torch.equal(t["susceptible"], s0)
    mask = torch.tensor([0, 1, 0, 1], dtype=torch.uint8, device="cuda")
    before = t["susceptible"].clone()
    env.reset(mask)
    assert torch.equal(t["susceptible"][1], s0[1]) and torch.equal(t["susceptible"][3], s0[3])
    assert torch.equal(t["susceptible"][0], before[0]) and torch.equal(t["susceptible"][2], before[2])
    assert t["timestep"].cpu().tolist() == [12, 0, 12, 0]
    # constructor checks mirror the reference's
    bad = dict(cfg)
    bad["n_agents"] = 50
    with pytest.raises(AssertionError):
        hip_env(bad, 1)
    bad = dict(cfg)
    bad["components"] = [("ControlUSStateOpenCloseStatus", {"n_stringency_levels": 5})] + list(cfg["components"][1:])
    with pytest.raises(ValueError):
        hip_env(bad, 1)
    bad = dict(cfg)
    bad["use_real_world_data"] = True  # covid19_env.py:126-135: needs use_real_world_policies too
    with pytest.raises(AssertionError):
        hip_env(bad, 1)


@pytest.mark.gpu
def test_covid_filter_recurrence_vs_exact_window_sums():
    """`filter_recurrence=True` updates each unemployment filter's discounted delta sum in O(1) (the taps are
    exp(-age/lambda), covid19_env.py:242-247); the default re-sums the 600-day window over the reference's float32 taps.
    Same actions through both: identical integer state, `unemployed` within 4e-6 relative (the reference's taps are
    float32 exp(-float32(age) / lambda): up to ~1.3e-6 off the exponential law they sample), everything downstream
    within the suite's tolerances."""
    import torch

    g = load_covid_golden("c4_covid_51ag")
    cfg = g["cfg"]
    E, T = 96, 200
    env_r = hip_env(cfg, n_envs=E, filter_recurrence=True)
    env_x = hip_env(cfg, n_envs=E)  # the default: window sums
    assert env_x.exact_filter_sums and not env_r.exact_filter_sums
    assert "filter_discounted_delta_sums" in env_r.tensors and "filter_discounted_delta_sums" not in env_x.tensors
    env_r.reset()
    env_x.reset()
    rng = np.random.RandomState(3)
    worst = 0.0
    for k in range(1, T + 1):
        a = rng.randint(0, 11, size=(E, 51)).astype(np.int32)
        a[rng.rand(E, 51) < 0.5] = 0
        p = rng.randint(0, 21, size=(E, 1)).astype(np.int32)
        for env in (env_r, env_x):
            env.step({"a": torch.as_tensor(a, device="cuda"), "p": torch.as_tensor(p, device="cuda")})
        if k % 20 == 0 or k == T:
            tr, tx = env_r.tensors, env_x.tensors
            for name in ("cooldown_until", "subsidy_level", "timestep", "stringency_ring"):
                assert torch.equal(tr[name], tx[name]), name
            # the long history: every day without the recurrence, whole 16-day chunks with it
            done_chunks = (int(env_r.model["filter_len"]) + k + 1) // 16
            assert torch.equal(tr["stringency_history_chunks"][:, :done_chunks], tx["stringency_history_chunks"][:, :done_chunks])
            for day in (k, k - 17, k - 40, 0, -5):
                assert np.array_equal(env_r.stringency_level(3, day), env_x.stringency_level(3, day)), day
            u_r, u_x = tr["unemployed"].double(), tx["unemployed"].double()
            worst = max(worst, float(((u_r - u_x).abs() / u_x.abs().clamp_min(1.0)).max()))
            for name, tol in STATE_TOL.items():
                np.testing.assert_allclose(tr[name].cpu().numpy(), tx[name].cpu().numpy(), rtol=tol, atol=1e-3, err_msg=name)
            np.testing.assert_allclose(tr["rewards_a"].cpu().numpy(), tx["rewards_a"].cpu().numpy(), rtol=0, atol=1e-5)
    assert worst < 4e-6, "unemployed: recurrence vs exact window sums differ by %.3g relative" % worst
    print("covid filter recurrence: max relative deviation of `unemployed` from the exact window sums: %.3g" % worst)


def _covid_state_tensors(env):
    # (sample_t: the policy's draw index, a record field since round 5 -- only the environment the test draws from advances it)
    skip = ("stringency_change_", "window_streams_whole_history", "stringency_history_chunks", "sample_t")
    return {k: v for k, v in env.tensors.items() if not k.startswith(skip)}


@pytest.mark.gpu
@pytest.mark.parametrize("policy", ["masked", "unmasked"])
def test_covid_window_sums_over_change_events_equal_the_streamed_window(policy):
    """The default (reference-exact) filter sums add only the NON-ZERO level changes of the 600-day window, kept per
    state as an event list; a day without a change contributes fma(0, tap, acc) == acc to the streamed sum, so both
    must give the SAME float64 -- every tensor bit for bit.  Environment `ev` keeps the lists, `st` is made to stream
    its whole window from the first step (the exported flag).  "masked": the policy respects the cool-down masks, the
    lists never overflow over a whole episode; "unmasked": levels change on most days, a list overflows mid-episode
    and the replica carries on streaming (the step that switches builds the open history chunk from the recent-days
    ring).  Masked resets in between."""
    import torch

    g = load_covid_golden("c4_covid_51ag")
    cfg = g["cfg"]
    E = 40
    ev, st = hip_env(cfg, n_envs=E), hip_env(cfg, n_envs=E)
    for env in (ev, st):
        env.reset()
    st.tensors["window_streams_whole_history"].fill_(1)
    T = int(cfg["episode_length"])
    steps = T + 30 if policy == "masked" else 220
    went_dense_at = None
    for k in range(1, steps + 1):
        if policy == "masked":
            a, p = ev.backend.sample_masked_actions(seed=5)
        else:
            a, p = ev.backend.sample_random_actions(seed=5)
        for env in (ev, st):
            env.backend.step(a, p)
        if k % 90 == 0 or k == T:
            mask = ev.tensors["done"].clone() if k == T else (torch.arange(E, device="cuda") % 3 == 0).to(torch.uint8)
            for env in (ev, st):
                env.backend.reset(mask)
            st.tensors["window_streams_whole_history"].fill_(1)
        if k % 15 == 0 or k in (1, 2, T - 1, T, T + 1, steps):
            torch.cuda.synchronize()
            a_t, b_t = _covid_state_tensors(ev), _covid_state_tensors(st)
            for name in a_t:
                assert torch.equal(a_t[name], b_t[name]), "day %d: %s differs (event lists vs streamed window)" % (k, name)
            dense = ev.tensors["window_streams_whole_history"]
            if policy == "masked":
                assert int(dense.sum()) == 0, "day %d: a list overflowed under the masked policy" % k
                ht = ev.tensors["stringency_change_head_tail"]
                assert int((ht >> 16).max()) <= 64 and bool(((ht & 0xffff) <= (ht >> 16)).all())
            elif went_dense_at is None and int(dense.sum()) > 0:
                went_dense_at = k
    if policy == "unmasked":
        assert went_dense_at is not None and went_dense_at > 15, "the unmasked rollout was meant to overflow a list mid-episode"
        # whole 16-day chunks of the long history agree as well once a replica streams
        e = int(torch.nonzero(ev.tensors["window_streams_whole_history"])[0])
        L, ts = int(ev.model["filter_len"]), int(ev.tensors["timestep"][e])
        done_chunks = (L + ts + 1) // 16
        assert torch.equal(ev.tensors["stringency_history_chunks"][e, :done_chunks], st.tensors["stringency_history_chunks"][e, :done_chunks])


@pytest.mark.gpu
@FILTER_MODES
def test_covid_step_sample_next_masked_equals_two_launches(recurrence):
    """aie_step_sample_next_masked: the next actions drawn inside the step launch are what aie_sample_masked_actions
    draws from the masks that step wrote."""
    import torch

    g = load_covid_golden("c4_covid_variant")
    E = 64
    one, two = [hip_env(g["cfg"], n_envs=E, filter_recurrence=recurrence) for _ in range(2)]
    for env in (one, two):
        env.reset()
    cur = one.backend.sample_masked_actions(seed=9, slot=0)
    slot = 0
    for k in range(1, int(g["cfg"]["episode_length"]) + 1):  # (a step past the episode's end does nothing, draws included)
        a, p = two.backend.sample_masked_actions(seed=9)
        torch.cuda.synchronize()
        assert torch.equal(a, cur[0]) and torch.equal(p, cur[1]), "day %d: fused masked draw differs" % k
        two.backend.step(a, p)
        cur = one.backend.step_sample_next(cur[0], cur[1], seed=9, next_slot=slot ^ 1, masked=True)
        slot ^= 1
        if k % 40 == 0:
            torch.cuda.synchronize()
            for name in _covid_state_tensors(one):
                assert torch.equal(one.tensors[name], two.tensors[name]), "day %d: %s" % (k, name)
    with pytest.raises(Exception):  # COVID only
        from helpers import C2, make_env

        gtb = make_env(C2, n_envs=4, device="cuda:0")
        gtb.reset()
        a, p = gtb.backend.sample_random_actions(seed=1)
        gtb.backend.step_sample_next(a, p, seed=1, masked=True)


@pytest.mark.gpu
@pytest.mark.parametrize("order", ["filters_last", "history_last"])
def test_covid_window_sums_with_taps_that_are_not_float32_values(order):
    """The window kernel keeps its LDS tap table in float32 only when every uploaded tap is a float32 value (the
    reference's are); any other table takes the float64 one.  Taps perturbed in their low mantissa bits, uploaded through
    aie_upload in either order relative to the pre-episode history (both uploads refresh what reset copies: the first
    step's sums need the taps): the sums over the event lists (LDS table, float64 this time) against the streamed window
    (which reads the uploaded float64 taps from memory), every tensor bit for bit."""
    import torch

    g = load_covid_golden("c4_covid_51ag")
    cfg = g["cfg"]
    E = 16
    envs = [hip_env(cfg, n_envs=E), hip_env(cfg, n_envs=E)]
    for env in envs:
        be = env.backend
        taps = np.asarray(env.model["unemp_conv_filters"], np.float64) * (1.0 + 3e-9)  # [F, L]: no longer float32 values
        assert not np.array_equal(taps.astype(np.float32).astype(np.float64), taps)
        hist = np.asarray(env.model["stringency_level_history_0"])
        first, second = ("model_stringency_level_history_0", hist), ("model_unemp_conv_filters", taps)
        if order == "history_last":
            first, second = second, first
        be.upload(first[0], first[1][None])
        be.upload(second[0], second[1][None])
        env.reset()
    ev, st = envs
    st.tensors["window_streams_whole_history"].fill_(1)
    for k in range(1, 41):
        a, p = ev.backend.sample_masked_actions(seed=21)
        for env in envs:
            env.backend.step(a, p)
        torch.cuda.synchronize()
        for name in _covid_state_tensors(ev):
            assert torch.equal(ev.tensors[name], st.tensors[name]), "day %d: %s differs (float64 tap table)" % (k, name)
    assert int(ev.tensors["window_streams_whole_history"].sum()) == 0
    # and the perturbation is visible: the same rollout on the reference's own taps gives another `unemployed`
    ref = hip_env(cfg, n_envs=E)
    ref.reset()
    for k in range(1, 41):
        a, p = ref.backend.sample_masked_actions(seed=21)
        ref.backend.step(a, p)
    torch.cuda.synchronize()
    assert not torch.equal(ref.tensors["unemployed"], ev.tensors["unemployed"])
