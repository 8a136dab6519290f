"""aie_sample_policy_actions, CPU side: the sampler's building blocks (csrc/aie_layout.h: aie_sampler_expf,
aie_sampler_entry_rng, the fixed prefix-sum order) against their Python transcription (tests/helpers.py) and libm, and --
through oracle/'s restatement, which the -m gpu tests hold the kernel to entry for entry -- that the sampler draws from
softmax(logits) restricted to the action mask (what a trainer expects of it: base_env.py:141-145,
tutorials/rllib/env_wrapper.py:50-211)."""
import ctypes as C
import math
import struct

import numpy as np
import pytest

from helpers import make_env, sampler_entry_rng, sampler_expf, sampler_pick_row, sampler_uniform

GTB = [["Build", {}], ["ContinuousDoubleAuction", {"max_num_orders": 5}], ["Gather", {}], ["PeriodicBracketTax", {}]]
C2 = dict(scenario_name="layout_from_file/simple_wood_and_stone", n_agents=4, world_size=[25, 25],
          episode_length=1000, components=GTB, starting_agent_coin=10,
          env_layout_file="quadrant_25x25_20each_30clump.txt")


def test_sampler_exp_is_the_same_bits_everywhere_and_close_to_libm():
    from oracle_lib import lib

    L = lib()
    rs = np.random.RandomState(0)
    worst = 0.0
    vals = list(-rs.uniform(0, 60, size=3000)) + list(-np.exp(rs.uniform(-30, 4.4, size=2000))) + [0.0, -0.0, -0.34657359, -0.346573591,
                                                                                                  -79.9, -80.0, -80.1, -1e30, -87.5]
    for y in vals:
        y = np.float32(y)
        got = L.aie_oracle_sampler_expf(float(y))
        assert struct.pack("<f", got) == struct.pack("<f", sampler_expf(y)), y
        if y > -80.0:
            worst = max(worst, abs(got - math.exp(float(y))) / math.exp(float(y)))
        else:
            assert got == 0.0
    assert worst < 4e-7, worst
    assert L.aie_oracle_sampler_expf(0.0) == 1.0
    for rnd in (0, 1, 511, 512, 0x7fffffff, 0xffffffff, 0x12345678):
        got = L.aie_oracle_sampler_uniform(rnd)
        assert got == float(sampler_uniform(rnd)) == ((rnd >> 9) + 0.5) / 2.0 ** 23 and 0.0 < got < 1.0


def test_fmaf_transcription_rounds_once():
    """helpers.fmaf (exact rational arithmetic) where rounding the float64 result a second time would go wrong."""
    from helpers import fmaf

    f32 = np.float32
    a, b = f32(1.0 + 2.0 ** -12), f32(1.0 + 2.0 ** -12)  # a b = 1 + 2^-11 + 2^-24: half a unit above 1 + 2^-11 ...
    assert fmaf(a, b, f32(2.0 ** -60)) == f32(1.0 + 2.0 ** -11 + 2.0 ** -23)  # ... plus a sliver float64 cannot hold: up
    assert fmaf(a, b, f32(-2.0 ** -60)) == f32(1.0 + 2.0 ** -11)  # minus the sliver: down
    assert fmaf(a, b, f32(0.0)) == f32(1.0 + 2.0 ** -11)  # the exact tie: to even
    rs = np.random.RandomState(1)
    for _ in range(300):  # where float64 holds the sum exactly, one more rounding is the answer
        x, y, z = f32(rs.randn()), f32(rs.randn()), f32(rs.randn())
        d = float(x) * float(y) + float(z)
        from fractions import Fraction
        if Fraction(float(x)) * Fraction(float(y)) + Fraction(float(z)) == Fraction(d):
            assert fmaf(x, y, z) == f32(d)


def test_sampler_row_equals_its_python_transcription():
    """oracle/'s row sampler (the kernel's twin) against the transcription: rows of 1 .. 150 entries (one, two and three
    64-entry chunks), masks, NaN and hopeless logits, fully masked rows."""
    from oracle_lib import lib

    L = lib()
    rs = np.random.RandomState(2)
    seen = set()
    for trial in range(400):
        n = int(rs.choice([1, 2, 7, 11, 22, 50, 63, 64, 65, 101, 128, 150]))
        lg = (rs.randn(n) * rs.choice([0.5, 3.0, 30.0])).astype(np.float32)
        mask = (rs.rand(n) < rs.choice([0.2, 0.7, 1.0])).astype(np.float32)
        if trial % 9 == 0:
            lg[rs.randint(n)] = np.nan
        if trial % 11 == 0:
            lg[:] = -1e30
        if trial % 13 == 0:
            mask[:] = 0.0
        if trial % 17 == 0:
            lg[:] = 0.25
        word = int(rs.randint(0, 2 ** 32, dtype=np.uint64))
        want = sampler_pick_row(lg, mask, word)
        got = L.aie_oracle_sample_row(lg.ctypes.data_as(C.c_void_p), mask.ctypes.data_as(C.c_void_p), 1, n, word)
        assert got == want, (trial, n, got, want)
        assert want == 0 or (mask[want] > 0.5 and lg[want] == lg[want])
        seen.add(want)
    assert len(seen) > 40


def test_sampler_entry_hash_is_uniform():
    from oracle_lib import lib

    L = lib()
    words = np.array([L.aie_oracle_sampler_entry_rng(0x1234567 + 977 * s, k) for s in range(256) for k in range(256)], np.uint32)
    assert int(words[5 * 256 + 9]) == sampler_entry_rng(0x1234567 + 977 * 5, 9)
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
