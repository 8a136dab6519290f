"""aie_sample_policy_actions, CPU side: the sampler's building blocks (csrc/aie_layout.h: aie_sampler_log,
aie_sampler_key, aie_sampler_entry_rng) against their Python transcription and libm, and -- through oracle/'s restatement,
which the -m gpu tests hold the kernel to entry for entry -- that the sampler draws from softmax(logits) restricted to the
action mask (what a trainer expects of it: base_env.py:141-145, tutorials/rllib/env_wrapper.py:50-211)."""
import math
import struct

import numpy as np
import pytest

from helpers import make_env

GTB = [["Build", {}], ["ContinuousDoubleAuction", {"max_num_orders": 5}], ["Gather", {}], ["PeriodicBracketTax", {}]]
C2 = dict(scenario_name="layout_from_file/simple_wood_and_stone", n_agents=4, world_size=[25, 25],
          episode_length=1000, components=GTB, starting_agent_coin=10,
          env_layout_file="quadrant_25x25_20each_30clump.txt")


def _slog(v):
    m, ex = math.frexp(v)
    m, ex = m * 2.0, ex - 1
    if m > 1.4142135623730951:
        m, ex = m * 0.5, ex + 1
    s = (m - 1.0) / (m + 1.0)
    z = s * s
    p = 0.076923076923076927
    for c in (0.090909090909090912, 0.1111111111111111, 0.14285714285714285, 0.2, 0.33333333333333331, 1.0):
        p = p * z + c
    return float(ex) * 0.69314718055994529 + (2.0 * s) * p


def test_sampler_log_is_the_same_bits_everywhere_and_close_to_libm():
    from oracle_lib import lib

    L = lib()
    rs = np.random.RandomState(0)
    worst = 0.0
    vals = list(np.exp(rs.uniform(-23, 4, size=20000))) + [1.0, 2.0, 0.5, 1.4142135623730951, 1.4142135623730954, 1e-10, 22.18]
    for v in vals:
        got = L.aie_oracle_sampler_log(float(v))
        assert struct.pack("<d", got) == struct.pack("<d", _slog(float(v))), v
        worst = max(worst, abs(got - math.log(v)) / max(1.0, abs(math.log(v))))
    assert worst < 1e-12, worst


def test_sampler_key_orders_scores_then_prefers_the_lower_entry():
    from oracle_lib import lib

    L = lib()
    rs = np.random.RandomState(1)
    sc = np.concatenate([rs.randn(2000) * 5, [-1e30, 1e30, 0.0, -0.0, 1e-300, -1e-300]])
    for a, b in zip(sc[:-1], sc[1:]):
        ka, kb = L.aie_oracle_sampler_key(float(a), 7), L.aie_oracle_sampler_key(float(b), 7)
        if abs(a - b) > 1e-9 * max(1.0, abs(a), abs(b)):  # (scores closer than the 11 dropped bits count as tied)
            assert (ka > kb) == (a > b), (a, b)
    assert L.aie_oracle_sampler_key(1.5, 3) > L.aie_oracle_sampler_key(1.5, 4) > 0
    assert L.aie_oracle_sampler_key(-1e30, 2047) > 0  # 0 is reserved for "nothing allowed"
    assert (L.aie_oracle_sampler_key(2.25, 1234) & 0x7ff) == 2047 - 1234


def test_sampler_entry_hash_is_uniform():
    from oracle_lib import lib

    L = lib()
    words = np.array([L.aie_oracle_sampler_entry_rng(0x1234567 + 977 * s, k) for s in range(256) for k in range(256)], np.uint32)
    h = np.bincount(words.view(np.uint8), minlength=256).astype(np.float64)
    exp = words.size * 4 / 256.0
    assert float(((h - exp) ** 2 / exp).sum()) < 330.5  # 255 degrees of freedom, 99.9 %
    assert len(np.unique(words)) > 0.999 * words.size


def test_sampler_draws_from_the_masked_softmax():
    """20 000 draws of every action slot of one replica with fixed logits (the draw index advances per call): the
    frequencies of each agent's 50 entries and of the planner's seven 22-entry rows against softmax(logits) over the
    allowed entries (chi-square, 99.9 %); a masked entry is never drawn."""
    from oracle_lib import OracleEnv

    env = make_env(C2, n_envs=1)
    o = OracleEnv(env.build_config(), env.layout_planes())
    o.seed(3)
    o.reset()
    rs = np.random.RandomState(4)
    MA, MP = o.t["obs_a_action_mask"].shape[-1], o.t["obs_p_action_mask"].shape[-1]
    la = (rs.randn(1, 4, MA) * 1.5).astype(np.float32)
    lp = (rs.randn(1, MP) * 1.5).astype(np.float32)
    N = 20000
    ca = np.zeros((4, MA))
    cp = np.zeros((7, MP // 7))
    for _ in range(N):
        a, p = o.sample_policy_actions(la, lp, seed=99, env_offset=0, width_a=1, width_p=7)
        for i in range(4):
            ca[i, a[0, i, 0]] += 1
        for b in range(7):
            cp[b, p[0, b]] += 1
    assert int(o.t["sample_t"][0]) == N
    ma, mp = o.t["obs_a_action_mask"][0], o.t["obs_p_action_mask"][0].reshape(7, -1)
    assert (ma < 0.5).any(), "the reset state masks some agent actions (nothing to trade yet)"

    def check(counts, logits, mask, what):
        allowed = mask > 0.5
        assert counts[~allowed].sum() == 0, what + ": a masked entry was drawn"
        w = np.exp(logits[allowed].astype(np.float64) - logits[allowed].max())
        exp = N * w / w.sum()
        keep = exp >= 5
        chi2 = float((((counts[allowed] - exp) ** 2) / exp)[keep].sum())
        dof = int(keep.sum()) - 1
        # Wilson-Hilferty 99.9 % quantile of chi-square with `dof` degrees of freedom
        q = dof * (1 - 2 / (9 * dof) + 3.09 * math.sqrt(2 / (9 * dof))) ** 3
        assert chi2 < q, (what, chi2, q, dof)

    for i in range(4):
        check(ca[i], la[0, i], ma[i], "agent %d" % i)
    for b in range(7):
        check(cp[b], lp[0].reshape(7, -1)[b], mp[b], "bracket %d" % b)
