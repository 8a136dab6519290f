"""CPU tests: the C restatement (oracle/aie_oracle.c) against the committed golden
vectors that oracle/gen_golden.py produced from the UNMODIFIED reference."""
import zlib

import numpy as np
import pytest

from helpers import (compare_state, golden_names, load_golden, make_env, oracle_host_pre_reset,
                     state_from_golden)
from oracle_lib import OracleEnv

OBS_TOL = 2e-6  # f32 observations: values are f64 in the reference, rounded once to f32


def _obs_check(o, g, k, where):
    for name in [x for x in g.keys() if x.startswith("ob_")]:
        t = name[3:]
        want = g[name][k]
        got = o.t[t][0]
        if want.dtype.kind in "iu":
            assert np.array_equal(got, want), "%s: obs %s differs" % (where, t)
        else:
            np.testing.assert_allclose(got, want, rtol=OBS_TOL, atol=OBS_TOL,
                                       err_msg="%s: obs %s" % (where, t))


@pytest.mark.parametrize("name", golden_names())
def test_oracle_step_matches_reference_golden(name):
    g = load_golden(name)
    env = make_env(g["cfg"])
    o = OracleEnv(env.build_config(), env.layout_planes())
    o.load_state(state_from_golden(g, "s0_"), e=0)
    T = g["actions_a"].shape[0]
    obs_steps = list(g["obs_steps"])
    resets = {int(t): i for i, t in enumerate(g.get("reset_at", []))}
    for t in range(T):
        ap = g["actions_p"][t][None] if g["actions_p"].shape[1] else None
        o.step(g["actions_a"][t][None], ap)
        got = {k: v[0] for k, v in o.t.items()}
        compare_state(got, state_from_golden(g, "st_", t), where="%s step %d" % (name, t + 1))
        assert zlib.crc32(o.t["mt"][0].tobytes()) == int(g["st_mt_crc"][t]), "MT state, step %d" % (t + 1)
        rew = np.concatenate([o.t["rewards_a"][0], o.t["rewards_p"][[0]]])
        np.testing.assert_allclose(rew, g["rew"][t], rtol=2e-7, atol=1e-5)  # f32 storage
        assert int(o.t["done"][0]) == int(g["done"][t])
        if (t + 1) in obs_steps:
            _obs_check(o, g, obs_steps.index(t + 1), "%s step %d" % (name, t + 1))
        if (t + 1) in resets:
            oracle_host_pre_reset(env, o)
            o.reset()
            compare_state({k: v[0] for k, v in o.t.items()},
                          state_from_golden(g, "rs_", resets[t + 1]),
                          where="%s reset after step %d" % (name, t + 1))
    assert np.array_equal(o.t["mt"][0], g["final_mt"])


@pytest.mark.parametrize("name", golden_names())
def test_oracle_reset_matches_reference_golden(name):
    """reset() from the injected pre-reset MT state reproduces the reference's
    placements, skills, trackers and reset observations."""
    g = load_golden(name)
    env = make_env(g["cfg"])
    o = OracleEnv(env.build_config(), env.layout_planes())
    o.t["mt"][0] = g["pre_reset_mt"]
    o.t["mt_pos"][0] = g["pre_reset_pos"]
    oracle_host_pre_reset(env, o)
    o.reset()
    got = {k: v[0] for k, v in o.t.items()}
    want = state_from_golden(g, "s0_")
    compare_state(got, want, where=name + " reset")
    assert np.array_equal(got["mt"], want["mt"])
    if 0 in list(g["obs_steps"]):
        _obs_check(o, g, list(g["obs_steps"]).index(0), name + " reset obs")
