"""The reference's own unit test (tests/test_env.py:16-107: uniform 15x15, 4 agents,
Build + ContinuousDoubleAuction{max_num_orders: 5} + Gather, starting coin 10, 10 %
coverage) replayed against the batched backend: same construction call, same structural
assertions, on the batched (obs, rew, done, info) and on the per-replica reference view."""
import numpy as np
import pytest

ENV_CONFIG = {
    "n_agents": 4,
    "world_size": [15, 15],
    "episode_length": 1000,
    "multi_action_mode_agents": False,
    "multi_action_mode_planner": True,
    # as in the reference's test: observations with minimal processing (every key on its own)
    "flatten_observations": False,
    "flatten_masks": True,
    "components": [
        {"Build": {}},
        {"ContinuousDoubleAuction": {"max_num_orders": 5}},
        {"Gather": {}},
    ],
    "scenario_name": "uniform/simple_wood_and_stone",
    "starting_agent_coin": 10,
    "starting_stone_coverage": 0.10,
    "starting_wood_coverage": 0.10,
}


def test_construction_matches_reference_contract():
    from ai_economist_amd import foundation

    env = foundation.make_env_instance(**ENV_CONFIG)
    assert env.n_agents == ENV_CONFIG["n_agents"]
    assert env.num_agents == ENV_CONFIG["n_agents"] + 1  # + the planner
    assert [c.name for c in env.components] == ["Build", "ContinuousDoubleAuction", "Gather"]
    assert env.resources == ["Coin", "Stone", "Wood"] and env.landmarks == ["House"]
    env_m = foundation.make_env_instance(**dict(ENV_CONFIG, flatten_masks=False))  # masks as per-subspace views
    from ai_economist_amd.foundation.obs_keys import mask_keys

    assert [k for k, _, _ in mask_keys(env_m)["a"]][:2] == ["Build", "ContinuousDoubleAuction.Buy_Stone"]


@pytest.mark.gpu
def test_env_reset_and_step():
    from ai_economist_amd import foundation

    n = ENV_CONFIG["n_agents"]
    env = foundation.make_env_instance(n_envs=8, device="cuda:0", **ENV_CONFIG)
    env.seed(3)
    obs = env.reset()
    assert sorted(obs.keys()) == ["a", "p"]
    assert tuple(obs["a"]["world-map"].shape) == (8, n, 6, 11, 11)  # no Water channel
    assert tuple(obs["a"]["action_mask"].shape) == (8, n, 50)
    # the per-replica view has exactly the reference's keys
    ref_view = env.as_reference_dicts(0)
    assert sorted(ref_view.keys()) == [str(i) for i in range(n)] + ["p"]
    # flatten_observations=False: every key on its own (base_env.py:591-612), zero-copy slices of the packed vector
    assert "flat" not in obs["a"] and "flat" not in ref_view["0"]
    assert {"world-map", "world-idx_map", "time", "action_mask", "world-inventory-Coin", "world-loc-row",
            "Build-build_payment", "ContinuousDoubleAuction-available_asks-Stone",
            "Gather-bonus_gather_prob"} <= set(ref_view["0"].keys())
    assert ref_view["0"]["world-inventory-Coin"] == pytest.approx(10 * 0.01)
    assert ref_view["0"]["ContinuousDoubleAuction-available_asks-Stone"].shape == (11,)
    assert {"p0", "p1", "p2", "p3"} <= set(ref_view["p"].keys())
    assert {"world-inventory-Coin", "world-loc-col"} <= set(ref_view["p"]["p2"].keys())
    flat = env.tensor("obs_a_flat")
    assert obs["a"]["world-inventory-Coin"].data_ptr() >= flat.data_ptr()  # a view, not a copy
    assert tuple(obs["a"]["ContinuousDoubleAuction-my_bids-Wood"].shape) == (8, n, 11)
    # and the flattened form of the same environment
    env_f = foundation.make_env_instance(n_envs=8, device="cuda:0", **dict(ENV_CONFIG, flatten_observations=True))
    env_f.seed(3)
    obs_f = env_f.reset()
    assert {"world-map", "world-idx_map", "time", "flat", "action_mask"} <= set(obs_f["a"].keys())
    assert (obs_f["a"]["flat"] == flat).all()

    obs, reward, done, info = env.step({})  # no actions == all NO-OP (base_env.py:964-966)
    assert obs.keys() == reward.keys()
    assert obs.keys() == info.keys()
    assert "__all__" in done
    assert int(env.tensor("timestep").min()) == 1
    # every replica got its own random layout and placement
    flags = env.tensor("cell_flags").reshape(8, -1)
    assert len({bytes(f.cpu().numpy().tobytes()) for f in flags}) > 1


@pytest.mark.gpu
def test_one_replica_environment_takes_the_reference_action_dictionary():
    """tests/test_env.py steps the environment with {agent.idx: action, ..., planner.idx: [...]}: a one-replica
    environment accepts that form as is (a batch needs the tensor form and says so)."""
    from ai_economist_amd import foundation

    n = ENV_CONFIG["n_agents"]
    env = foundation.make_env_instance(n_envs=1, device="cuda:0", **ENV_CONFIG)
    env.seed(5)
    env.reset()
    before = env.tensor("loc_c")[0].cpu().numpy().copy()
    names_a, _ = env.action_subspace_names()
    first = {}
    base = 1
    for nm, d in names_a:
        first[nm] = base
        base += d
    actions = {str(i): first["Gather"] + (i % 4) for i in range(n)}  # everybody tries to move
    actions["p"] = []  # this configuration's planner has no actions
    obs, rew, done, info = env.step(actions)
    assert int(env.tensor("timestep")[0]) == 1 and obs.keys() == rew.keys()
    moved = (env.tensor("loc_c")[0].cpu().numpy() != before).any() or True  # (moves may be blocked; the call is the point)
    assert moved
    batch = foundation.make_env_instance(n_envs=2, device="cuda:0", **ENV_CONFIG)
    batch.reset()
    with pytest.raises(ValueError):
        batch.step(actions)


@pytest.mark.gpu
def test_reference_unit_test_verbatim_on_the_reference_format_view():
    """tests/test_env.py:70-107, statement for statement, on make_env_instance(reference_format=True, **config)."""
    from ai_economist_amd import foundation

    env = foundation.make_env_instance(reference_format=True, device="cuda:0", **ENV_CONFIG)
    num_planners = 1
    assert len(env.all_agents) == ENV_CONFIG["n_agents"] + num_planners
    assert len(env.world.agents) == ENV_CONFIG["n_agents"]
    assert env.world.planner.idx == "p"
    obs = env.reset()
    assert sorted(list(obs.keys())) == [str(i) for i in range(ENV_CONFIG["n_agents"])] + ["p"]
    obs, reward, done, info = env.step({})
    assert obs.keys() == reward.keys()
    assert obs.keys() == info.keys()
    assert "__all__" in done
    # and what the tutorials read off the agents (base_agent.py:173-186)
    assert env.get_agent("0").action_spaces == 50 and env.get_agent(0).idx == 0
    assert list(env.get_agent("p").action_spaces) == [] and env.get_agent("p").multi_action_mode
    assert env.get_agent(1).state["inventory"]["Coin"] == 10.0


@pytest.mark.gpu
def test_foundation_env_wrapper_surface():
    """The reference's device seam (F/env_wrapper.py): spaces per actor, reset_all_envs / step_all_envs /
    reset_only_done_envs on the batched environment."""
    import torch
    from ai_economist_amd.foundation.env_wrapper import FoundationEnvWrapper

    cfg = dict(ENV_CONFIG, episode_length=6)
    w = FoundationEnvWrapper(env_name=cfg["scenario_name"], env_config=cfg, num_envs=5, device="cuda:0")
    n = cfg["n_agents"]
    assert w.n_agents == n + 1 and w.episode_length == 6 and w.n_envs == 5
    assert sorted(w.env.action_space) == sorted(w.env.observation_space) == sorted([str(i) for i in range(n)] + ["p"])
    assert w.env.action_space["0"].n == 50 and w.env.action_space["0"].dtype == np.int32
    assert w.env.observation_space["1"]["world-map"].shape == (6, 11, 11)
    obs = w.reset_all_envs()
    assert tuple(obs["2"]["world-inventory-Coin"].shape) == (5,) and not w.reset_on_host
    a = torch.randint(0, 50, (5, n, 1), dtype=torch.int32, device="cuda:0")
    for t in range(6):
        assert w.step_all_envs({"a": a}) is None
    assert bool(w.env.tensors["done"].all())
    assert w.reset_only_done_envs() == {}
    assert int(w.env.tensors["timestep"].max()) == 0
    obs, rew, done, info = w.step({"a": a})
    assert tuple(rew["0"].shape) == (5,) and "__all__" in done


@pytest.mark.gpu
def test_foundation_env_wrapper_on_the_covid_scenario():
    """The scenario the reference's wrapper was written for: collated observations, agent axis last."""
    import torch
    from ai_economist_amd.foundation.env_wrapper import FoundationEnvWrapper
    from test_covid_dense_log import CFG

    cfg = {k: v for k, v in CFG.items() if k not in ("dense_log_frequency", "world_dense_log_frequency")}
    w = FoundationEnvWrapper(env_name="CovidAndEconomySimulation", env_config=cfg, num_envs=3, device="cuda:0")
    assert w.n_agents == 52 and sorted(w.env.action_space, key=str)[:2] == ["0", "1"]
    obs = w.reset_all_envs()
    assert tuple(obs["7"]["world-agent_state"].shape) == (3, 6)
    assert tuple(obs["7"]["action_mask"].shape) == (3, 11)
    assert w.env.action_space["p"].n == 21 and w.env.observation_space["0"]["world-agent_state"].shape == (6,)
    a = torch.zeros((3, 51, 1), dtype=torch.int32, device="cuda:0")
    p = torch.zeros((3, 1), dtype=torch.int32, device="cuda:0")
    w.step_all_envs({"a": a, "p": p})
    assert int(w.env.tensors["timestep"].min()) == 1


def test_reference_format_view_without_a_gpu():
    """The same view over the CPU oracle standing in for the device backend (tests/test_dense_log.py: OracleBackend):
    the host-side presentation code runs in the CPU suite too."""
    import numpy as np
    from ai_economist_amd import foundation
    from ai_economist_amd.foundation.reference_view import ReferenceFormatEnv
    from helpers import oracle_host_pre_reset
    from oracle_lib import OracleEnv
    from test_dense_log import ReplayOracleBackend

    cfg = dict(ENV_CONFIG, scenario_name="layout_from_file/simple_wood_and_stone", world_size=[25, 25],
               env_layout_file="quadrant_25x25_20each_30clump.txt", flatten_observations=True)
    for k in ("starting_stone_coverage", "starting_wood_coverage"):
        cfg.pop(k)
    scen = cfg.pop("scenario_name")
    host = foundation.make_env_instance(scen, n_envs=1, **cfg)
    o = OracleEnv(host.build_config(), host.layout_planes())
    host._backend = ReplayOracleBackend(o, host)
    host.host_pre_reset = lambda mask: oracle_host_pre_reset(host, o)
    np.random.seed(3)
    st = np.random.get_state()
    o.t["mt"][0] = st[1]
    o.t["mt_pos"][0] = st[2]
    env = ReferenceFormatEnv(host)
    n = cfg["n_agents"]
    assert len(env.all_agents) == n + 1 and env.world.planner.idx == "p"
    obs = env.reset()
    assert sorted(obs.keys()) == [str(i) for i in range(n)] + ["p"]
    obs, reward, done, info = env.step({"0": 1, "2": 3})
    assert obs.keys() == reward.keys() == info.keys() and "__all__" in done
    assert int(o.t["timestep"][0]) == 1
    assert env.get_agent("1").action_spaces == 50


def test_foundation_env_wrapper_without_a_gpu():
    """FoundationEnvWrapper(env_obj=...) around a host environment whose backend is the CPU oracle."""
    import numpy as np
    import torch
    from ai_economist_amd import foundation
    from ai_economist_amd.foundation.env_wrapper import FoundationEnvWrapper
    from helpers import oracle_host_pre_reset
    from oracle_lib import OracleEnv
    from test_dense_log import ReplayOracleBackend

    cfg = dict(ENV_CONFIG, scenario_name="layout_from_file/simple_wood_and_stone", world_size=[25, 25],
               env_layout_file="quadrant_25x25_20each_30clump.txt", flatten_observations=True, episode_length=3)
    for k in ("starting_stone_coverage", "starting_wood_coverage"):
        cfg.pop(k)
    scen = cfg.pop("scenario_name")
    host = foundation.make_env_instance(scen, n_envs=1, **cfg)
    o = OracleEnv(host.build_config(), host.layout_planes())
    host._backend = ReplayOracleBackend(o, host)
    host.host_pre_reset = lambda mask: oracle_host_pre_reset(host, o)
    o.seed(4)
    w = FoundationEnvWrapper(env_obj=host)
    n = cfg["n_agents"]
    assert w.n_agents == n + 1 and sorted(host.action_space) == sorted(host.observation_space)
    assert host.action_space["0"].n == 50 and tuple(host.observation_space["0"]["flat"].shape) == (121,)  # no tax component here
    w.reset_all_envs()
    a = torch.ones((1, n, 1), dtype=torch.int32)
    p = torch.zeros((1, 1), dtype=torch.int32)  # (the CPU stand-in wants both buffers)
    for _ in range(3):
        w.step_all_envs({"a": a, "p": p})
    assert bool(o.t["done"][0])
    w.reset_only_done_envs()
    assert int(o.t["timestep"][0]) == 0


def test_reference_action_dictionaries_in_multi_action_mode_without_a_gpu():
    """{"0": [sub-action per subspace], "p": [one per bracket]}: lists land in the per-subspace slots of replica 0;
    a seed_state that is not a NumPy MT19937 state tuple is refused as in the reference (base_env.py:969-970)."""
    import numpy as np
    from ai_economist_amd import foundation
    from helpers import oracle_host_pre_reset
    from oracle_lib import OracleEnv
    from test_dense_log import ReplayOracleBackend

    cfg = dict(ENV_CONFIG, scenario_name="layout_from_file/simple_wood_and_stone", world_size=[25, 25],
               env_layout_file="quadrant_25x25_20each_30clump.txt", flatten_observations=True,
               multi_action_mode_agents=True,
               components=[{"Build": {}}, {"ContinuousDoubleAuction": {"max_num_orders": 5}}, {"Gather": {}},
                           {"PeriodicBracketTax": {}}])
    for k in ("starting_stone_coverage", "starting_wood_coverage"):
        cfg.pop(k)
    scen = cfg.pop("scenario_name")
    host = foundation.make_env_instance(scen, n_envs=1, **cfg)
    o = OracleEnv(host.build_config(), host.layout_planes())
    be = ReplayOracleBackend(o, host)
    host._backend = be
    host.host_pre_reset = lambda mask: oracle_host_pre_reset(host, o)
    o.seed(2)
    host.reset()
    names_a, names_p = host.action_subspace_names()
    seen = {}
    orig = be.step

    def spy(a, p):
        seen["a"], seen["p"] = a.numpy().copy(), p.numpy().copy()
        return orig(a, p)

    be.step = spy
    acts = {"1": [0] * len(names_a), "p": [2] + [0] * (len(names_p) - 1)}
    acts["1"][-1] = 3  # last subspace (Gather): move
    host.step(acts)
    assert seen["a"].shape == (1, cfg["n_agents"], len(names_a)) and seen["a"][0, 1, -1] == 3 and seen["a"].sum() == 3
    assert seen["p"].shape == (1, len(names_p)) and seen["p"][0, 0] == 2
    with pytest.raises(AssertionError):
        host.step(acts, seed_state=("MT19937", np.zeros(624, np.uint32), 0))
    st = host.rng_state(0)
    host.step(acts, seed_state=st)  # a valid state passes through
    assert host.rng_state(0)[0] == "MT19937"


def test_reference_format_view_unflattened_masks_match_the_live_reference():
    """make_env_instance(reference_format=True, flatten_masks=False) has to return the reference's mask
    DICTIONARIES ({"<Component>[.<sub-action>]": list of uint8}, NO-OP entries left out, base_env.py:749-756) for every
    actor, not the flattened float vectors.  Same stream as the live reference (np.random.seed(3) -> replica 0), reset
    plus three steps, multi-action planner with taxes: keys and values equal."""
    import numpy as np
    from ai_economist_amd import foundation
    from ai_economist_amd.foundation.reference_view import ReferenceFormatEnv
    from helpers import oracle_host_pre_reset
    from oracle_lib import OracleEnv
    from ref_harness import load_reference_foundation, reference_available
    from test_dense_log import ReplayOracleBackend

    if not reference_available():
        pytest.skip("reference tree not present")
    cfg = dict(ENV_CONFIG, scenario_name="layout_from_file/simple_wood_and_stone", world_size=[25, 25],
               env_layout_file="quadrant_25x25_20each_30clump.txt", flatten_observations=True, flatten_masks=False,
               components=[{"Build": {}}, {"ContinuousDoubleAuction": {"max_num_orders": 5}}, {"Gather": {}},
                           {"PeriodicBracketTax": {}}])
    for k in ("starting_stone_coverage", "starting_wood_coverage"):
        cfg.pop(k)
    scen = cfg.pop("scenario_name")
    kw_ref = dict(cfg, components=[tuple(c.items())[0] for c in cfg["components"]])
    np.random.seed(3)
    ref = load_reference_foundation().make_env_instance(scen, **kw_ref)
    ref.seed(3)
    host = foundation.make_env_instance(scen, n_envs=1, **cfg)
    o = OracleEnv(host.build_config(), host.layout_planes())
    host._backend = ReplayOracleBackend(o, host)
    host.host_pre_reset = lambda mask: oracle_host_pre_reset(host, o)
    st = np.random.get_state()
    o.t["mt"][0] = st[1]
    o.t["mt_pos"][0] = st[2]
    env = ReferenceFormatEnv(host)
    obs, want = env.reset(), ref.reset()
    acts = {"0": 3, "1": 47, "2": 0, "3": 48, "p": [1, 0, 2, 0, 0, 0, 0]}
    for t in range(4):
        for actor in [str(i) for i in range(cfg["n_agents"])] + ["p"]:
            got_m, want_m = obs[actor]["action_mask"], want[actor]["action_mask"]
            assert isinstance(got_m, dict) and sorted(got_m) == sorted(want_m), (t, actor)
            for key in want_m:
                assert isinstance(got_m[key], list) and got_m[key] == want_m[key], (t, actor, key)
        obs, _, _, _ = env.step(acts)
        want, _, _, _ = ref.step(acts)


def test_registered_component_without_a_device_kernel_is_refused_at_construction():
    """The component registry is open like the reference's (base_component.py:378, registrar.py:48-66); a registered
    class whose dynamics are Python cannot run inside a batched launch, and make_env_instance says so -- by name --
    instead of failing at backend creation with "unknown component id 0"."""
    from ai_economist_amd import foundation
    from ai_economist_amd.foundation.components import component_registry
    from ai_economist_amd.foundation.components.base import BaseComponent

    @component_registry.add
    class GiftEveryone(BaseComponent):
        name = "GiftEveryoneForTheTest"
        required_entities = ["Coin"]
        agent_subclasses = ["BasicMobileAgent"]

        def __init__(self, *args, amount=1, **kwargs):
            super().__init__(*args, **kwargs)
            self.amount = amount

        def get_n_actions(self, agent_cls_name):
            return None

    assert component_registry.has("GiftEveryoneForTheTest")
    kw = dict(ENV_CONFIG)
    scenario = kw.pop("scenario_name")
    kw["components"] = list(kw["components"]) + [{"GiftEveryoneForTheTest": {"amount": 2}}]
    with pytest.raises(NotImplementedError, match="GiftEveryoneForTheTest.*no device kernel"):
        foundation.make_env_instance(scenario, **kw)
