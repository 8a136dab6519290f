"""CPU tests: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/aie.h declares (no compute calls without a GPU); host-side config validation."""
import ctypes
import os
import re

import pytest

from helpers import ROOT, make_env


def _lib_path():
    from ai_economist_amd import _build

    return _build.build()


def test_library_exports_every_declared_symbol():
    path = _lib_path()
    lib = ctypes.CDLL(path)
    hdr = open(os.path.join(ROOT, "include", "aie.h")).read()
    declared = set(re.findall(r"\b(aie_[a-z_]+)\s*\(", hdr))
    declared -= {"aie_env"}
    assert len(declared) >= 15
    for sym in sorted(declared):
        assert hasattr(lib, sym), "missing export: " + sym
    from ai_economist_amd import _cabi

    assert set(_cabi.EXPORTED_SYMBOLS) == declared
    # ... and nothing else: the shipping library is built with -fvisibility=hidden, so no kernel stub, development
    # hook (aie_dev_*, -DAIE_DEV build only) or helper leaks into its dynamic symbol table
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", path], check=True, capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    exported -= {"_init", "_fini", "__bss_start", "_edata", "_end"}
    assert exported == declared, "undeclared exports: %s; missing: %s" % (sorted(exported - declared),
                                                                          sorted(declared - exported))


def test_arena_bytes_and_validation_without_gpu():
    from ai_economist_amd import _cabi

    lib = _cabi.bind(ctypes.CDLL(_lib_path()))
    cfg = dict(scenario_name="layout_from_file/simple_wood_and_stone", n_agents=4,
               world_size=[25, 25], episode_length=1000,
               components=[["Build", {}], ["ContinuousDoubleAuction", {"max_num_orders": 5}],
                           ["Gather", {}], ["PeriodicBracketTax", {}]], starting_agent_coin=10)
    env = make_env(cfg, n_envs=4096)
    c = env.build_config()
    nbytes = lib.aie_arena_bytes(ctypes.byref(c))
    # 4096 replicas: ~37 KB of observations + ~7 KB of state each
    assert 4096 * 40000 < nbytes < 4096 * 60000
    c.n_agents = 1
    assert lib.aie_arena_bytes(ctypes.byref(c)) == _cabi.E_INVALID
    assert b"n_agents" in lib.aie_last_error(None)
    c = env.build_config()
    c.regen_halfwidth[0] = 1  # one more byte plane pair per record (source blocks per regen window)
    assert lib.aie_arena_bytes(ctypes.byref(c)) > nbytes
    c.max_health[0] = 2  # neighbourhood regeneration with max_health > 1: true window sums (supported since round 2)
    assert lib.aie_arena_bytes(ctypes.byref(c)) > nbytes
    c.layout_gen = _cabi.LAYOUT_UNIFORM  # layouts drawn at reset need per-replica planes and coverage settings
    assert lib.aie_arena_bytes(ctypes.byref(c)) == _cabi.E_INVALID
    c.regen_halfwidth[0] = 4
    assert lib.aie_arena_bytes(ctypes.byref(c)) == _cabi.E_INVALID
    c = env.build_config()
    c.full_observability = 1  # every agent sees the whole map: n x 6 x 25 x 25 floats instead of n x 7 x 11 x 11
    assert lib.aie_arena_bytes(ctypes.byref(c)) > nbytes
    # layouts drawn on the device: up to 64 x 64 cells; larger worlds are for the host procedure (layout_gen stays FIXED)
    for side, ok in ((48, True), (64, True), (65, False)):
        u = make_env(dict(scenario_name="uniform/simple_wood_and_stone", n_agents=4, world_size=[side, side],
                          episode_length=10, components=[["Build", {}], ["Gather", {}]], starting_agent_coin=3,
                          starting_stone_coverage=0.05, starting_wood_coverage=0.05), n_envs=2)
        assert bool(u.layouts_on_device) == ok
        uc = u.build_config()
        assert (uc.layout_gen != _cabi.LAYOUT_FIXED) == ok and lib.aie_arena_bytes(ctypes.byref(uc)) > 0
        uc.layout_gen = _cabi.LAYOUT_UNIFORM
        assert (lib.aie_arena_bytes(ctypes.byref(uc)) > 0) == ok
        if not ok:
            assert b"4096 cells" in lib.aie_last_error(None)


def test_host_registry_and_kwargs_validation():
    from ai_economist_amd import foundation

    assert foundation.scenarios.has("LAYOUT_FROM_FILE/simple_wood_and_stone")  # case-insensitive
    builtin = ["Build", "ContinuousDoubleAuction", "ControlUSStateOpenCloseStatus", "FederalGovernmentSubsidy", "Gather",
               "PeriodicBracketTax", "SimpleLabor", "VaccinationCampaign", "WealthRedistribution"]
    # (the registry is open: tests/test_batched_component.py adds its toy components to it in the same process)
    toys = {"ActingToy", "CoinSubsidy", "LaborRelief", "PlainToy"}
    assert [c for c in foundation.components.entries if c not in toys] == builtin
    assert foundation.scenarios.entries == ["CovidAndEconomySimulation", "layout_from_file/simple_wood_and_stone",
                                            "multi_zone/simple_wood_and_stone", "one-step-economy",
                                            "quadrant/simple_wood_and_stone", "split_layout/simple_wood_and_stone",
                                            "uniform/simple_wood_and_stone"]
    with pytest.raises(KeyError):
        foundation.make_env_instance("no/such_scenario")
    base = dict(n_agents=4, world_size=[25, 25], components=[("Build", {}), ("Gather", {})])
    with pytest.raises(AssertionError):
        foundation.make_env_instance("layout_from_file/simple_wood_and_stone",
                                     **dict(base, n_agents=1))
    with pytest.raises(AssertionError):
        foundation.make_env_instance("layout_from_file/simple_wood_and_stone",
                                     **dict(base, components=[("Build", {"payment": -1})]))
    with pytest.raises(KeyError):
        foundation.make_env_instance("layout_from_file/simple_wood_and_stone",
                                     **dict(base, components=[("Nope", {})]))
    env = foundation.make_env_instance("layout_from_file/simple_wood_and_stone", **base)
    assert env.resources == ["Coin", "Stone", "Wood"]
    assert env.landmarks == ["House", "Water"]
    assert env.get_component("Build").payment == 10
    # product path must fail loudly without a GPU / HIP runtime
    import torch

    if not torch.cuda.is_available():
        with pytest.raises(Exception):
            env.reset()
