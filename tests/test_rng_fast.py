"""rng_mode="fast" (include/aie.h: AIE_RNG_FAST): a counter-based stream per replica (Philox2x32-10) in place of NumPy's
MT19937 -- a throughput mode the reference does not have (its trainers only ever call np.random.seed,
base_env.py:481-494).  The parity chain, so that nothing here is "statistical only":

    reference == oracle(MT19937) bit for bit         tests/test_oracle_vs_reference.py, tests/golden/ (unchanged)
    oracle(fast) differs from oracle(MT19937) in ONE function, rng_u32 (oracle/aie_oracle.c): the words drawn
    Philox2x32-10 itself                             Random123's known-answer vectors (below)
    how oracle(fast) consumes the stream             an independent Python transcription of one step's draws (below)
    HIP(fast) == oracle(fast) bit for bit            the -m gpu tests below: every state field, every observation
    oracle(fast) ~ oracle(MT19937) in distribution   chi-square / binomial checks below (regeneration rate, placement)
"""
import ctypes as C

import numpy as np
import pytest

from helpers import make_env

GTB = [["Build", {}], ["ContinuousDoubleAuction", {"max_num_orders": 5}], ["Gather", {}], ["PeriodicBracketTax", {}]]
C2 = dict(scenario_name="layout_from_file/simple_wood_and_stone", n_agents=4, world_size=[25, 25],
          episode_length=1000, components=GTB, starting_agent_coin=10,
          env_layout_file="quadrant_25x25_20each_30clump.txt")
C1 = dict(scenario_name="uniform/simple_wood_and_stone", n_agents=4, world_size=[15, 15], episode_length=1000,
          components=[["Build", {}], ["Gather", {}]], starting_agent_coin=10, starting_stone_coverage=0.10,
          starting_wood_coverage=0.10)


def _ose_cfg(n=12, T=3):
    rs = np.random.RandomState(4)
    return dict(scenario_name="one-step-economy", n_agents=n, world_size=[1, 1], episode_length=T,
                components=[["SimpleLabor", {"skills": [float(x) for x in np.sort(1 + rs.rand(n) * 2)]}],
                            ["PeriodicBracketTax", {"bracket_spacing": "us-federal", "period": 1,
                                                    "tax_model": "model_wrapper"}]])


def _philox(c0, c1, key):
    from oracle_lib import lib

    ctr = (C.c_uint32 * 2)(c0, c1)
    out = (C.c_uint32 * 2)()
    lib().aie_oracle_philox2x32_10(ctr, key, out)
    return int(out[0]), int(out[1])


def _oracle(cfg, E, seed, **extra):
    from oracle_lib import OracleEnv

    env = make_env(cfg, n_envs=E, **extra)
    o = OracleEnv(env.build_config(), env.layout_planes())
    o.seed(seed)
    o.reset()
    return env, o


# ---- the generator ---------------------------------------------------------------------------------------------------
def test_philox2x32_10_known_answers():
    """Random123 (Salmon et al., SC'11) kat_vectors, philox2x32 with 10 rounds: counter, key -> output."""
    assert _philox(0x00000000, 0x00000000, 0x00000000) == (0xff1dae59, 0x6cd10df2)
    assert _philox(0xffffffff, 0xffffffff, 0xffffffff) == (0x2c3f628b, 0xab4fd7ad)
    assert _philox(0x243f6a88, 0x85a308d3, 0x13198a2e) == (0xdd7ce038, 0xf62a4c12)


def test_fast_stream_is_the_documented_function_of_seed_replica_and_word_number():
    """include/aie.h: word g of replica e = element g & 1 of philox(counter = (lo32(g >> 1), hi32(g >> 1) | salt),
    key32), key32 = lo32(seed + e), salt = bits 32..47 of (seed + e) << 16; the first draw after seeding is word 624
    (block 1).  Checked through the oracle's own consumer: the agents' reset placement draws (randint via masked
    rejection on successive words, layout_from_file.py:360-370)."""
    seed = (0x1234 << 32) | 0xfffffffe  # replica 2's key wraps into the salt bits
    env, o = _oracle(C2, 4, seed, rng_mode="fast")
    st = o.t["mt"]
    assert st.shape == (4, 4) and o.t["mt_pos"].shape == (4,)
    for e in range(4):
        s = seed + e
        assert int(st[e, 0]) == s & 0xffffffff and int(st[e, 2]) == ((s >> 32) & 0xffff) << 16 and int(st[e, 3]) == 0
        assert int(st[e, 1]) == 1  # the reset drew from block 1
        # replay the placement of agent 0: row = first word & 31 that is <= 24, then the column likewise
        g, vals = 624, []
        while len(vals) < 2:
            pair = g >> 1
            w = _philox(pair & 0xffffffff, (pair >> 32) | int(st[e, 2]), int(st[e, 0]))[g & 1]
            g += 1
            if (w & 31) <= 24:
                vals.append(w & 31)
        # (the first candidate cell may be occupied / water in which case the reference redraws: accept either the
        # replayed cell or a cell reached after more draws, but the FIRST pair must have been looked at)
        r, c = int(o.t["loc_r"][e, 0]), int(o.t["loc_c"][e, 0])
        water = env.layout_planes()[2].reshape(25, 25)
        if not water[vals[0], vals[1]]:
            assert (r, c) == (vals[0], vals[1]), (e, r, c, vals)


def test_fast_mode_is_deterministic_and_keyed_by_global_replica():
    """Replica e of a batch seeded s == replica 0 of a batch seeded s + e (the sharding rule: env_offset adds to the
    seed), and two runs agree bit for bit."""
    rs = np.random.RandomState(0)
    _, big = _oracle(C2, 6, 11, rng_mode="fast")
    _, again = _oracle(C2, 6, 11, rng_mode="fast")
    _, shard = _oracle(C2, 2, 15, rng_mode="fast")
    for t in range(40):
        a = rs.randint(0, 50, size=(6, 4)).astype(np.int32)
        p = rs.randint(0, 22, size=(6, 7)).astype(np.int32)
        big.step(a, p)
        again.step(a, p)
        shard.step(a[4:], p[4:])
    for k in big.t:
        if big.t[k].shape[0] != 6:
            continue
        assert np.array_equal(big.t[k], again.t[k]), k
        assert np.array_equal(big.t[k][4:], shard.t[k]), k


def test_generated_layouts_have_a_stream_of_their_own_in_fast_mode():
    """uniform/ in the counter-stream mode: the source layout of a replica's k-th reset is a function of (its stream's key
    and salt, k) alone -- the same whatever the episode's steps drew in between (which is what lets the device draw it
    ahead of the reset) -- and a different one for every k; the fourth state word counts the resets.  In the parity mode
    the layout comes out of the replica's NumPy stream at the reset, so it DOES depend on the draws before it."""
    from oracle_lib import lib

    rs = np.random.RandomState(3)
    flags = {}
    for mode in ("fast", "numpy"):
        for steps in (0, 7):
            _, o = _oracle(dict(C1, episode_length=50), 5, 19, rng_mode=mode)
            first = o.t["cell_flags"].copy()
            for _ in range(steps):
                o.step(rs.randint(0, 6, size=(5, 4)).astype(np.int32), np.zeros((5, 1), np.int32))
            o.reset(np.ones(5, np.uint8))
            flags[mode, steps] = (first, o.t["cell_flags"].copy(), o.t["mt"].copy(), o.t["mt_pos"].copy())
    f0, f7 = flags["fast", 0], flags["fast", 7]
    assert np.array_equal(f0[0], f7[0]) and np.array_equal(f0[1], f7[1])      # layouts 0 and 1: the same with or without steps
    assert not np.array_equal(f0[0], f0[1])                                   # ... and different from each other
    assert (f0[2][:, 3] == 2).all() and (f7[2][:, 3] == 2).all()              # resets so far
    assert np.array_equal(f0[2][:, [0, 2]], f7[2][:, [0, 2]])                 # key and salt: the stream's identity
    assert not np.array_equal(f0[3], f7[3]) or not np.array_equal(f0[2][:, 1], f7[2][:, 1])  # the replica's own stream did move
    n0, n7 = flags["numpy", 0], flags["numpy", 7]
    assert np.array_equal(n0[0], n7[0]) and not np.array_equal(n0[1], n7[1])  # parity mode: the second layout follows the draws
    # the stream's definition (csrc/aie_layout.h: aie_layout_stream), word for word
    L = lib()
    for st in ([5, 9, 0x12340000, 0], [5, 9, 0x12340000, 1], [0xffffffff, 0, 0, 32767], [7, 3, 0xffff0000, 32768], [7, 3, 0, 98305]):
        buf = (C.c_uint32 * 4)(*st)
        out = (C.c_uint32 * 4)()
        L.aie_oracle_layout_stream(buf, out)
        k = st[3]
        assert out[0] == (st[0] + 0x9E3779B9 * (k >> 15)) & 0xffffffff
        assert out[1] == st[2] | 0x8000 | (k & 0x7fff)
        assert (out[3] << 32 | out[2]) == (out[0] << 32 | out[1]) != 0
    # the first words of layout stream 1 of replica 2 are Philox words of that key / counter (block 0, pairs 0, 1, ...)
    st = [int(v) for v in f0[2][2]]
    st[3] = 1
    out = (C.c_uint32 * 4)()
    L.aie_oracle_layout_stream((C.c_uint32 * 4)(*st), out)
    assert _philox(0, out[1], out[0]) != _philox(0, st[2], st[0])  # not the replica's own block 0


def test_fast_and_numpy_modes_differ_only_in_the_stream():
    """Same configuration: the two modes share every tensor's shape except the generator state, and where no draw
    decides anything (NO-OP actions, nothing to regenerate) their trajectories coincide."""
    _, of = _oracle(C2, 3, 5, rng_mode="fast")
    _, om = _oracle(C2, 3, 5)
    assert of.t["mt"].shape == (3, 4) and om.t["mt"].shape == (3, 624)
    for k in om.t:
        if k in ("regen_src_n", "regen_src_list"):  # (the record's own source list: the counter stream shares the batch's)
            continue
        if k != "mt":
            assert of.t[k].shape == om.t[k].shape, k
    z_a, z_p = np.zeros((3, 4), np.int32), np.zeros((3, 7), np.int32)
    pos0 = of.t["mt_pos"].copy(), om.t["mt_pos"].copy()
    blk0 = of.t["mt"][:, 1].copy()
    for t in range(5):
        of.step(z_a, z_p)
        om.step(z_a, z_p)
    for k in ("stone", "wood", "inv_coin", "inv_res", "timestep", "labor", "rewards_a", "done"):
        assert np.array_equal(of.t[k], om.t[k]), k
    # both consumed the same number of words: per step two agent-order permutations' worth of rejection draws differ,
    # the 4 H W regeneration words do not -- the fast stream moved on by about 5 x 2500 words = 20 blocks
    adv = (of.t["mt"][:, 1].astype(np.int64) - blk0) * 624 + of.t["mt_pos"] - pos0[0]
    assert ((adv >= 5 * 2500) & (adv < 5 * 2600)).all(), adv


def test_fast_mode_step_consumes_the_stream_exactly_as_documented():
    """An independent transcription of ONE step's draws in Python (Philox through the known-answer-checked function,
    NumPy's consumers written out): Build's and Gather's agent-order permutations (world.py:418-422: Fisher-Yates from
    the top, masked-rejection integers), then 2 H W doubles for the regeneration -- Wood's plane first, then Stone's,
    two words per double (layout_from_file.py:394-403) -- predicts which emptied source cells respawn and where the
    stream stands afterwards.  The device is compared with the oracle; this compares the oracle with the documentation."""
    E, p_regen = 3, 0.3
    env, o = _oracle(dict(C2, resource_regen_prob=p_regen), E, 21, rng_mode="fast")
    flags = o.t["cell_flags"].reshape(E, -1)
    o.t["stone"][...] = 0
    o.t["wood"][...] = 0
    st0, pos0 = o.t["mt"].copy(), o.t["mt_pos"].copy()
    o.step(np.zeros((E, 4), np.int32), np.zeros((E, 7), np.int32))
    HW = 625
    for e in range(E):
        key, blk, salt = int(st0[e, 0]), int(st0[e, 1]), int(st0[e, 2])
        g = blk * 624 + int(pos0[e])  # linear word number of the next draw (pos == 624: the first word of block + 1)

        def word(i):
            pair = i >> 1
            return _philox(pair & 0xffffffff, (pair >> 32) | salt, key)[i & 1]

        for _ in range(2):  # Build, then Gather: np.random.permutation(4)
            for i in (3, 2, 1):
                mask = 3 if i >= 2 else 1
                while True:
                    w = word(g) & mask
                    g += 1
                    if w <= i:
                        break
        wood = np.zeros(HW, np.int64)
        stone = np.zeros(HW, np.int64)
        for d in range(2 * HW):
            a, b = word(g) >> 5, word(g + 1) >> 6
            g += 2
            u = (a * 67108864.0 + b) / 9007199254740992.0
            cell = d % HW
            src = flags[e, cell] & (4 if d < HW else 2)
            if src and u < p_regen:
                (wood if d < HW else stone)[cell] = 1
        assert np.array_equal(o.t["wood"][e].reshape(-1), wood), e
        assert np.array_equal(o.t["stone"][e].reshape(-1), stone), e
        assert wood.sum() + stone.sum() > 5
        # where the stream stands: the oracle keeps (block, position in the block) with position in 1 .. 624
        blk1, pos1 = int(o.t["mt"][e, 1]), int(o.t["mt_pos"][e])
        assert blk1 * 624 + pos1 == g, (blk1, pos1, g)


# ---- distribution: oracle(fast) against oracle(MT19937) and against theory -------------------------------------------------
def _respawn_counts(mode, E, steps, seed):
    extra = dict(rng_mode="fast") if mode == "fast" else {}
    _, o = _oracle(dict(C2, resource_regen_prob=0.05), E, seed, **extra)
    src = (o.t["cell_flags"] & 2).astype(bool), (o.t["cell_flags"] & 4).astype(bool)  # stone / wood source blocks
    o.t["stone"][...] = 0
    o.t["wood"][...] = 0
    z_a, z_p = np.zeros((E, 4), np.int32), np.zeros((E, 7), np.int32)
    # (NO-OP agents do not gather -- move.py:126: only a move collects -- so respawned resources stay where they are)
    n_src = int(src[0].sum() + src[1].sum())
    filled = []
    for t in range(steps):
        o.step(z_a, z_p)
        filled.append(int((o.t["stone"][src[0]] > 0).sum() + (o.t["wood"][src[1]] > 0).sum()))
    return n_src, np.array(filled)


def test_regeneration_rate_matches_theory_in_both_modes():
    """Empty source blocks respawn with probability regen_weight per step (layout_from_file.py:394-403): after k steps
    a fraction 1 - (1 - p)^k is back.  ~10 000 Bernoulli chains per mode: both modes within 4 sigma of theory and of
    each other."""
    E, steps, p = 128, 20, 0.05
    res = {}
    for mode in ("numpy", "fast"):
        n_src, filled = _respawn_counts(mode, E, steps, seed=17)
        assert n_src > 5000
        for k in (1, 5, 20):
            q = 1 - (1 - p) ** k
            sigma = np.sqrt(n_src * q * (1 - q))
            assert abs(filled[k - 1] - n_src * q) < 4 * sigma, (mode, k, filled[k - 1], n_src * q, sigma)
        res[mode] = (n_src, filled)
    n_src = res["numpy"][0]
    for k in (1, 5, 20):
        q = 1 - (1 - p) ** k
        assert abs(res["numpy"][1][k - 1] - res["fast"][1][k - 1]) < 4 * np.sqrt(2 * n_src * q * (1 - q))


def test_reset_placement_has_the_same_distribution_in_both_modes():
    """Agent start cells are np.random.randint draws with rejection of occupied / water cells
    (layout_from_file.py:360-370).  Two-sample chi-square over the 25 rows and the 25 columns of agent 0's start cell,
    4096 replicas per mode (24 degrees of freedom: 99.9 % quantile 51.2)."""
    E = 4096
    _, of = _oracle(C2, E, 1, rng_mode="fast")
    _, om = _oracle(C2, E, 1)
    for key in ("loc_r", "loc_c"):
        hf = np.bincount(of.t[key][:, 0], minlength=25).astype(np.float64)
        hm = np.bincount(om.t[key][:, 0], minlength=25).astype(np.float64)
        ok = (hf + hm) > 0
        chi2 = float((((hf - hm) ** 2) / (hf + hm))[ok].sum())
        assert chi2 < 51.2, (key, chi2)
    # and the agent-order permutations behind Gather / Build resolve conflicts without a bias between the modes: the
    # four agents' total collected resources after 60 random steps agree within 4 sigma of the replica spread
    tot = {}
    for name, o in (("fast", of), ("numpy", om)):
        rs = np.random.RandomState(3)
        p = np.zeros((E, 7), np.int32)
        for t in range(60):
            o.step(rs.randint(0, 50, size=(E, 4)).astype(np.int32), p, nthreads=8)
        tot[name] = o.t["inv_res"].reshape(E, -1).sum(axis=1).astype(np.float64)
    d = tot["fast"].mean() - tot["numpy"].mean()
    se = np.sqrt(tot["fast"].var() / E + tot["numpy"].var() / E)
    assert abs(d) < 4 * se, (d, se)


def test_word_stream_bytes_are_uniform():
    """Chi-square of the byte values of 2^17 stream words of one replica (255 degrees of freedom: 99.9 % quantile
    330.5) and of the top bit between consecutive words (serial pairs)."""
    from oracle_lib import lib

    L = lib()
    n = 1 << 16
    ctr = (C.c_uint32 * 2)()
    out = (C.c_uint32 * 2)()
    words = np.empty(2 * n, np.uint32)
    for i in range(n):
        ctr[0], ctr[1] = i, 0
        L.aie_oracle_philox2x32_10(ctr, 12345, out)
        words[2 * i], words[2 * i + 1] = out[0], out[1]
    h = np.bincount(words.view(np.uint8), minlength=256).astype(np.float64)
    exp = words.size * 4 / 256.0
    assert float(((h - exp) ** 2 / exp).sum()) < 330.5
    top = (words >> 31).astype(np.int64)
    pairs = np.bincount(2 * top[:-1] + top[1:], minlength=4).astype(np.float64)
    exp = (words.size - 1) / 4.0
    assert float(((pairs - exp) ** 2 / exp).sum()) < 16.3  # 3 dof, 99.9 %


# ---- host API ------------------------------------------------------------------------------------------------------------
def test_rng_mode_is_validated_and_part_of_the_config():
    from ai_economist_amd import _cabi

    with pytest.raises(ValueError):
        make_env(C2, n_envs=1, rng_mode="philox")
    assert make_env(C2, n_envs=1).build_config().rng_mode == _cabi.RNG_NUMPY
    assert make_env(C2, n_envs=1, rng_mode="fast").build_config().rng_mode == _cabi.RNG_FAST
    with pytest.raises(NotImplementedError):  # host-drawn layouts continue the replica's NumPy stream
        make_env(dict(C1, world_size=[70, 70]), n_envs=1, rng_mode="fast").build_config()


def test_fast_instances_are_families_of_their_own():
    """The compile-time images of C2f / C3f differ from C2 / C3 (the generator is part of the record layout)."""
    from ai_economist_amd import _specs

    names = [s[0] for s in _specs.SPECS]
    assert any(n.startswith("C2f") for n in names) and any(n.startswith("C3f") for n in names)


# ---- the device against the oracle, bit for bit --------------------------------------------------------------------------
def _gpu_case(cfg, E, T, seed, generic=False, check_every=10, reset_at=(), nthreads=4):
    import torch
    from oracle_lib import OracleEnv
    from test_gpu_parity import _compare_all

    env = make_env(cfg, n_envs=E, device="cuda:0", rng_mode="fast")
    be = env.backend
    if generic:
        assert be.lib.aie_select_step_kernel(be.handle, 1) == 0
    env.seed(seed)
    env.reset()
    oracle = OracleEnv(env.build_config(), env.layout_planes())
    oracle.seed(seed)
    oracle.reset()
    assert be.tensors["mt"].shape == (E, 4)
    _compare_all(be, oracle, "after reset")
    for t in range(T):
        a, p = be.sample_random_actions(seed=99)
        env.step({"a": a, "p": p})
        torch.cuda.synchronize()
        oracle.step(a.cpu().numpy(), p.cpu().numpy(), nthreads=nthreads)
        if (t + 1) % check_every == 0 or t + 1 == T or (t + 1) in reset_at or t < 3:
            _compare_all(be, oracle, "step %d" % (t + 1))
        if (t + 1) in reset_at:
            done = be.tensors["done"].clone()
            assert bool(done.any())
            env.reset(done)
            oracle.reset(done.cpu().numpy())
            _compare_all(be, oracle, "reset after step %d" % (t + 1))
    return env, be, oracle


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["c2_instance", "c3_instance", "c2_generic", "c3_generic"])
def test_hip_fast_mode_equals_the_oracle_bit_for_bit(case):
    """C2f / C3f (compile-time instances) and the generic kernel: seeded uniform-random rollouts, every replica, every
    field incl. the generator state, across tax days, an episode end and the masked reset behind it."""
    n = 4 if case.startswith("c2") else 10
    cfg = dict(C2, n_agents=n, episode_length=120)
    env, be, _ = _gpu_case(cfg, 192 if n == 4 else 96, 150, seed=5, generic=case.endswith("generic"), reset_at=(120,))
    inst = be.lib.aie_step_kernel_instance(be.handle)
    assert (inst == -1) if case.endswith("generic") else (inst >= 0), inst


@pytest.mark.gpu
def test_hip_fast_mode_other_layouts_and_scalars():
    """Another layout file (more source blocks: the second chunk of the regeneration's source list), other scalars, a
    short episode; runs on the C2f instance's family or the generic kernel, whichever applies."""
    cfg = dict(C2, episode_length=60, starting_agent_coin=15, resource_regen_prob=0.05,
               env_layout_file="uniform_25x25_25each_65clump.txt")
    _gpu_case(cfg, 128, 130, seed=8, reset_at=(60, 120))


@pytest.mark.gpu
def test_hip_fast_mode_dynamic_layouts():
    """uniform/ (BASELINE configs[0]'s scenario): the reset draws the source layout from the replica's stream -- thousands
    of sequential words, through the block rows (mt_fast_rows) -- four wavefronts per replica."""
    _gpu_case(dict(C1, episode_length=40), 96, 85, seed=21, reset_at=(40, 80))


@pytest.mark.gpu
def test_hip_fast_mode_layouts_drawn_ahead_of_their_resets():
    """Generated layouts in the counter-stream mode come from a stream of their own, keyed by the replica's stream and its
    number of resets (csrc/aie_layout.h: aie_layout_stream), so the library draws them AHEAD of the reset once a quarter of
    the replicas have used theirs up.  A de-phased pattern of masked resets -- some replicas twice between two refills --
    takes both roads (installed from the staging area; drawn inside the reset) and equals the oracle, which draws every
    layout at its reset, field for field."""
    import torch
    from oracle_lib import OracleEnv
    from test_gpu_parity import _compare_all

    E = 96
    env = make_env(dict(C1, episode_length=1000), n_envs=E, device="cuda:0", rng_mode="fast")
    be = env.backend
    env.seed(33)
    env.reset()
    oracle = OracleEnv(env.build_config(), env.layout_planes())
    oracle.seed(33)
    oracle.reset()
    ctl = be.tensors["layout_stage_ctl"]
    torch.cuda.synchronize()
    base = ctl.cpu().reshape(-1).tolist()  # (layouts consumed since the last refill, the refill's go flag, installs, draws inside a reset)
    assert base[:2] == [0, 1] and base[2] + base[3] > 0 and (base[2] + base[3]) % E == 0, base  # the refill behind the reset ran
    _compare_all(be, oracle, "after reset")
    rs = np.random.RandomState(5)
    groups = [list(range(0, 8)), list(range(0, 8)), list(range(8, 40)), list(range(4, 12)), list(range(40, 96)), list(range(0, 96)),
              list(rs.choice(E, 20, replace=False)), list(rs.choice(E, 30, replace=False)), list(range(0, 8))]
    for k, grp in enumerate(groups):
        for _ in range(3):
            a, p = be.sample_random_actions(seed=99)
            env.step({"a": a, "p": p})
            oracle.step(a.cpu().numpy(), p.cpu().numpy(), nthreads=4)
        mask = np.zeros(E, np.uint8)
        mask[grp] = 1
        env.reset(torch.from_numpy(mask).to("cuda:0"))
        oracle.reset(mask)
        torch.cuda.synchronize()
        _compare_all(be, oracle, "masked reset %d" % k)
    n_empty, go, staged, inside = ctl.cpu().reshape(-1).tolist()
    staged, inside = staged - base[2], inside - base[3]
    assert staged + inside == sum(len(g) for g in groups)
    assert staged > 150 and inside >= 8, (staged, inside)  # (the second reset of replicas 0 - 7 found nothing staged)
    assert int(be.tensors["mt"][:, 3].min()) >= 2  # resets so far, the stream's fourth state word


@pytest.mark.gpu
def test_hip_fast_mode_staged_layout_follows_an_injected_stream():
    """A replica that receives another replica's whole state (tensor writes: the staging area knows nothing of it) is that
    replica's clone from then on -- its next reset must draw the CLONED stream's layout, although the staging area holds one
    for its old stream: the stage is keyed by the stream's identity and the reset count, so the clone draws inside the
    reset what the original installs from the staging area, and both get the same layout."""
    import torch

    E = 8
    env = make_env(dict(C1, episode_length=1000), n_envs=E, device="cuda:0", rng_mode="fast")
    be = env.backend
    env.seed(5)
    env.reset()
    for _ in range(3):
        a, p = be.sample_random_actions(seed=1)
        env.step({"a": a, "p": p})
    torch.cuda.synchronize()
    ctl0 = be.tensors["layout_stage_ctl"].cpu().reshape(-1).tolist()
    for k, t in be.tensors.items():  # replica 5 := replica 1, every per-replica tensor (incl. the generator's four words)
        if t.dim() >= 1 and t.shape[0] == E:
            t[5] = t[1]
    mask = torch.zeros(E, dtype=torch.uint8, device="cuda:0")
    mask[1] = mask[5] = 1
    env.reset(mask)
    torch.cuda.synchronize()
    ctl1 = be.tensors["layout_stage_ctl"].cpu().reshape(-1).tolist()
    assert ctl1[2] - ctl0[2] == 1 and ctl1[3] - ctl0[3] == 1  # one installed from the staging area, one drawn inside the reset
    for k in ("cell_flags", "stone", "wood", "loc_r", "loc_c", "mt", "mt_pos", "build_skill"):
        if k in be.tensors:
            assert torch.equal(be.tensors[k][5], be.tensors[k][1]), k
    assert not torch.equal(be.tensors["cell_flags"][5], be.tensors["cell_flags"][4])


@pytest.mark.gpu
def test_hip_fast_mode_one_step_economy():
    """one-step-economy: the agent-order permutation SimpleLabor draws and discards (the stream's position is state)."""
    _gpu_case(_ose_cfg(12, 3), 64, 20, seed=6, check_every=1, reset_at=(3, 6, 9, 12, 15, 18))
    _gpu_case(_ose_cfg(100, 2), 256, 12, seed=7, check_every=1, reset_at=(2, 4, 6, 8, 10, 12))


@pytest.mark.gpu
def test_hip_fast_mode_many_source_blocks_take_the_row_path():
    """More source doubles than the sparse list holds (AIE_SRC_CAP = 128; uniform layouts with 22 % coverage per resource
    on 20 x 20) regenerate through the block rows (scenario_step_regen_rows)."""
    cfg = dict(scenario_name="uniform/simple_wood_and_stone", n_agents=5, world_size=[20, 20], episode_length=25,
               components=[["Build", {}], ["Gather", {}]], starting_agent_coin=3, starting_stone_coverage=0.22,
               starting_wood_coverage=0.22, wood_regen_weight=0.3, stone_regen_weight=0.2)
    env, be, _ = _gpu_case(cfg, 16, 40, seed=4, check_every=6, reset_at=(25,))
    flags = be.tensors["cell_flags"].reshape(16, -1).cpu().numpy()
    assert ((((flags & 2) != 0).sum(axis=1) + ((flags & 4) != 0).sum(axis=1)) > 128).all()


@pytest.mark.gpu
@pytest.mark.parametrize("words", [8, 64])
def test_hip_fast_mode_draw_window_refills(words):
    """The components' draw window cut down to `words` (development hook; the environment then runs the full-featured
    kernel): every step of a 10-agent environment refills it several times from the counter stream -- block changes
    included -- and still equals the oracle, which knows no window."""
    import ctypes

    import torch
    from helpers import dev_library
    from oracle_lib import OracleEnv
    from test_gpu_parity import _compare_all

    cfg = dict(C2, n_agents=10, episode_length=40)
    with dev_library():
        env = make_env(cfg, n_envs=96, device="cuda:0", rng_mode="fast")
        env.seed(8)
        env.reset()
    be = env.backend
    be.lib.aie_dev_set_draw_window.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert be.lib.aie_dev_set_draw_window(be.handle, words) == 0
    oracle = OracleEnv(env.build_config(), env.layout_planes())
    oracle.seed(8)
    oracle.reset()
    for t in range(50):
        a, p = be.sample_random_actions(seed=9)
        be.step(a, p)
        torch.cuda.synchronize()
        oracle.step(a.cpu().numpy(), p.cpu().numpy(), nthreads=4)
        if t % 7 == 0 or t in (39, 49):
            _compare_all(be, oracle, "window %d, step %d" % (words, t + 1))
        if t == 39:
            env.reset(be.tensors["done"])
            oracle.reset(oracle.t["done"].copy())


@pytest.mark.gpu
def test_hip_fast_mode_many_agents_multi_zone_and_dense_log_replica():
    """30 agents (a draw window of 256 words, order books beyond a wavefront -> the full-featured kernel), multi_zone/ (the
    reset shuffles the zone grid with masked-rejection draws before it generates the layout) and an environment with a
    dense-log replica (the logged replica steps on aie_step_kernel_log): all in fast mode against the oracle."""
    _gpu_case(dict(C2, n_agents=30, episode_length=25, env_layout_file="uniform_25x25_25each_65clump.txt"), 24, 30, seed=3,
              check_every=5, reset_at=(25,))
    mz = dict(scenario_name="multi_zone/simple_wood_and_stone", n_agents=4, world_size=[16, 16], episode_length=20,
              components=[["Build", {}], ["Gather", {}]], starting_agent_coin=5)
    _gpu_case(mz, 48, 45, seed=12, check_every=5, reset_at=(20, 40))
    _gpu_case(dict(C2, episode_length=30, dense_log_frequency=1), 16, 35, seed=4, check_every=5, reset_at=(30,))


@pytest.mark.gpu
def test_hip_fast_mode_saez_random_rates_and_rng_state_injection():
    """tax_model "saez" draws its first periods' rates from the stream (np.random.uniform: sequential doubles inside the
    tax component); and aie_set_rng_state takes the counter stream's four words per replica."""
    import torch
    from oracle_lib import OracleEnv
    from test_gpu_parity import _compare_all

    comps = [["Build", {}], ["ContinuousDoubleAuction", {"max_num_orders": 5}], ["Gather", {}],
             ["PeriodicBracketTax", {"tax_model": "saez", "period": 5, "rate_min": 0.1, "rate_max": 0.9}]]
    _gpu_case(dict(C2, components=comps, episode_length=30), 32, 40, seed=9, check_every=5, reset_at=(30,))
    E = 8
    env = make_env(dict(C2, episode_length=20), n_envs=E, device="cuda:0", rng_mode="fast")
    env.seed(1)
    env.reset()
    be = env.backend
    keys = np.zeros((E, 4), np.uint32)
    keys[:, 0] = np.arange(E) + 4000
    keys[:, 1] = 17
    keys[:, 2] = 5 << 16
    pos = np.full(E, 123, np.int32)
    be.set_rng_state(keys, pos)
    oracle = OracleEnv(env.build_config(), env.layout_planes())
    oracle.seed(1)
    oracle.reset()
    oracle.t["mt"][...] = keys
    oracle.t["mt_pos"][...] = pos
    oracle.t["mt_has_gauss"][...] = 0
    oracle.t["mt_gauss"][...] = 0
    env.reset()
    oracle.reset()
    for t in range(6):
        a, p = be.sample_random_actions(seed=2)
        be.step(a, p)
        torch.cuda.synchronize()
        oracle.step(a.cpu().numpy(), p.cpu().numpy())
    _compare_all(be, oracle, "after state injection")
    assert (be.tensors["mt"].cpu().numpy().view(np.uint32)[:, 1] > 17).all()


@pytest.mark.gpu
def test_hip_fast_mode_is_shard_invariant():
    """Replica sharding: the counter stream is keyed by the GLOBAL replica id, so the second half of a
    64-replica batch equals a 32-replica batch created with env_offset=32 -- every tensor, after steps and a reset."""
    import torch

    cfg = dict(C2, episode_length=25)
    whole = make_env(cfg, n_envs=64, device="cuda:0", rng_mode="fast")
    half = make_env(cfg, n_envs=32, device="cuda:0", rng_mode="fast", env_offset=32)
    for env in (whole, half):
        env.seed(77)
        env.reset()
    bw, bh = whole.backend, half.backend
    for t in range(30):
        aw, pw = bw.sample_random_actions(seed=5)
        ah, ph = bh.sample_random_actions(seed=5, env_offset=32)
        assert torch.equal(aw[32:], ah) and torch.equal(pw[32:], ph)
        bw.step(aw, pw)
        bh.step(ah, ph)
        if t == 24:
            bw.reset(bw.tensors["done"])
            bh.reset(bh.tensors["done"])
    torch.cuda.synchronize()
    for k, v in bh.tensors.items():
        if v.shape[0] == 32:
            assert torch.equal(bw.tensors[k][32:], v), k


@pytest.mark.gpu
def test_aie_seed_fast_entry_point():
    import torch

    env = make_env(C2, n_envs=8, device="cuda:0", rng_mode="fast", env_offset=3)
    be = env.backend
    seed = (7 << 32) + 5
    be._check(be.lib.aie_seed_fast(be.handle, C.c_uint64(seed), C.c_int64(3), None))
    torch.cuda.synchronize()
    st = be.tensors["mt"].cpu().numpy().view(np.uint32)
    for e in range(8):
        s = seed + 3 + e
        assert list(st[e]) == [s & 0xffffffff, 0, ((s >> 32) & 0xffff) << 16, 0]
    assert (be.tensors["mt_pos"].cpu().numpy() == 624).all()
    plain = make_env(C2, n_envs=8, device="cuda:0").backend
    assert plain.lib.aie_seed_fast(plain.handle, C.c_uint64(1), C.c_int64(0), None) != 0
    assert b"rng_mode" in plain.lib.aie_last_error(plain.handle)
