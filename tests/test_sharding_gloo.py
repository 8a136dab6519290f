"""CPU tests of the N>1 path: world_size-2 gloo processes exercising the shard layout and
the single (reward, done) gather that bench.py / trainers use per step."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ai_economist_amd.sharding import RewardDoneGather, shard_range

    E_total, n = 16, 4
    lo, hi = shard_range(E_total, rank, world)
    E = hi - lo
    g = RewardDoneGather(E, n, device="cpu", dst=0)
    ok = True
    for t in range(3):
        gid = torch.arange(lo, hi, dtype=torch.float32)
        ra = gid[:, None] * 10 + torch.arange(n)[None, :] + 1000 * t
        rp = -gid - t
        done = ((torch.arange(lo, hi) + t) % 3 == 0).to(torch.uint8)
        res = g(ra, rp, done)
        if rank == 0:
            fa, fp, fd = res
            gid_all = torch.arange(E_total, dtype=torch.float32)
            ok &= torch.equal(fa, gid_all[:, None] * 10 + torch.arange(n)[None, :] + 1000 * t)
            ok &= torch.equal(fp, -gid_all - t)
            ok &= torch.equal(fd, (torch.arange(E_total) + t) % 3 == 0)
        else:
            ok &= res is None
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_reward_done_gather_world2_gloo():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def test_shard_ranges_cover_all_replicas():
    from ai_economist_amd.sharding import shard_range

    got = []
    for r in range(8):
        lo, hi = shard_range(32768, r, 8)
        got += list(range(lo, hi))
    assert got == list(range(32768))


class _FakeBackend:
    """What RewardLogGather needs from a DeviceBackend: set_reward_log + something that fills the slots (+ `tensors`
    with the current observations for the opt-in observation gather)."""

    def __init__(self, E, n):
        self.E, self.n, self.t, self.log = E, n, 0, None
        self.tensors = {"obs_a_flat": torch.zeros((E, n, 5)), "obs_a_action_mask": torch.zeros((E, n, 3))}

    def set_reward_log(self, n_slots):
        self.log = torch.zeros((n_slots, self.E, self.n + 2))
        return self.log

    def rewind_reward_log(self):
        self.slot0 = self.t

    def step(self, lo):
        slot = (self.t - getattr(self, "slot0", 0)) % self.log.shape[0]
        gid = torch.arange(lo, lo + self.E, dtype=torch.float32)
        self.log[slot, :, : self.n] = gid[:, None] * 10 + torch.arange(self.n)[None, :] + 1000 * self.t
        self.log[slot, :, self.n] = -gid - self.t
        self.log[slot, :, self.n + 1] = ((torch.arange(lo, lo + self.E) + self.t) % 3 == 0).float()
        # the observations the replicas hold after this step: a function of (global replica, agent, t)
        self.tensors["obs_a_flat"][...] = (gid[:, None, None] * 100 + torch.arange(self.n)[None, :, None] * 10
                                           + torch.arange(5)[None, None, :] + 0.5 * self.t)
        self.tensors["obs_a_action_mask"][...] = ((gid[:, None, None] + torch.arange(3)[None, None, :] + self.t) % 2)
        self.t += 1


def _log_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ai_economist_amd.sharding import RewardLogGather, shard_range

    E_total, n, K, T = 12, 3, 4, 22  # the rollout ends inside a block: finish() ships the two leftover steps
    lo, hi = shard_range(E_total, rank, world)
    be = _FakeBackend(hi - lo, n)
    g = RewardLogGather(be, steps_per_gather=K, dst=0, keep=True)
    launched = 0
    for t in range(T):
        be.step(lo)
        launched += int(g.after_step())
    g.finish()
    ok = launched == T // K
    if rank == 0:
        ok &= len(g.received) == T // K + 1 and g.received[-1].shape[1] == T % K
        gid = torch.arange(E_total, dtype=torch.float32)
        for blk, got in enumerate(g.received):  # [W, K (or the leftover), E, n + 2]
            steps = got.shape[1]
            full = got.permute(1, 0, 2, 3).reshape(steps, E_total, n + 2)  # rank-major replica order
            for k in range(steps):
                t = blk * K + k
                ok &= torch.equal(full[k, :, :n], gid[:, None] * 10 + torch.arange(n)[None, :] + 1000 * t)
                ok &= torch.equal(full[k, :, n], -gid - t)
                ok &= torch.equal(full[k, :, n + 1], ((torch.arange(E_total) + t) % 3 == 0).float())
    else:
        ok &= g.received == []
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_reward_log_gather_world2_gloo():
    """The batched exchange bench.py uses for N > 1: one gather per K steps, two alternating blocks."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_log_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def test_reward_log_gather_single_process_keeps_blocks():
    """Without a process group the blocks are simply handed back (keep=True): order and content."""
    from ai_economist_amd.sharding import RewardLogGather

    be = _FakeBackend(5, 2)
    g = RewardLogGather(be, steps_per_gather=3, keep=True)
    for t in range(10):
        be.step(0)
        g.after_step()
    g.finish()
    assert len(g.received) == 4 and tuple(g.received[3].shape) == (1, 1, 5, 4)  # steps 0-2, 3-5, 6-8 and the leftover 9
    assert torch.equal(g.received[3][0, 0, :, 2], -torch.arange(5, dtype=torch.float32) - 9)
    for blk, got in enumerate(g.received[:3]):
        assert tuple(got.shape) == (1, 3, 5, 4)
        for k in range(3):
            t = blk * 3 + k
            assert torch.equal(got[0, k, :, 2], -torch.arange(5, dtype=torch.float32) - t)


class _FakeTax:
    """Stands in for PeriodicBracketTax on a rank: E replicas with local Saez buffers of different lengths."""

    def __init__(self, rank):
        self.rank = rank
        self.lens = [3 + rank, 0, 5] if rank == 0 else [2, 4]
        self.got = None

    def local_saez_samples(self, env=None):
        rows = []
        for e, m in enumerate(self.lens):
            for k in range(m):
                rows.append([1000.0 * self.rank + 100.0 * e + k, 0.01 * k])
        return torch.tensor(rows, dtype=torch.float64).reshape(-1, 2)

    def set_global_saez_buffer(self, glob, env=None):
        self.got = glob.clone()


    _buffer_size = 5

    def pooled_saez_capacity(self, env=None):
        return 1000


class _FakeEnv:
    n_envs = 3

    def __init__(self, rank):
        self.tax = _FakeTax(rank)

    def get_component(self, name):
        assert name == "PeriodicBracketTax"
        return self.tax


def _saez_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ai_economist_amd.sharding import accumulate_and_broadcast_saez_buffers

    env = _FakeEnv(rank)
    glob = accumulate_and_broadcast_saez_buffers(env)
    want = torch.cat([_FakeTax(0).local_saez_samples(), _FakeTax(1).local_saez_samples()], dim=0)  # rank-major
    out[rank] = bool(torch.equal(glob, want) and torch.equal(env.tax.got, want) and glob.shape == (14, 2))
    dist.barrier()
    dist.destroy_process_group()


def test_saez_buffer_union_world2_gloo():
    """sharding.accumulate_and_broadcast_saez_buffers (reference: tutorials/rllib/utils/remote.py:56-73): ranks with
    different numbers of local samples end up with the same rank-major, replica-major concatenation."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_saez_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def _saez_capacity_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ai_economist_amd import foundation
    from ai_economist_amd.sharding import accumulate_and_broadcast_saez_buffers

    E, size = 3, 6

    def make(capacity=None):
        env = foundation.make_env_instance(
            "one-step-economy", n_agents=4, world_size=[1, 1], episode_length=2, n_envs=E,
            components=[("SimpleLabor", {"skills": [1.0, 2.0, 3.0, 4.0]}),
                        ("PeriodicBracketTax", {"bracket_spacing": "us-federal", "period": 1, "tax_model": "saez"})])
        tax = env.get_component("PeriodicBracketTax")
        tax._buffer_size = size
        tax._global_buffer_capacity = capacity
        # the device is not there on the CPU: FULL local buffers stand in for it, the capacity logic is the real one
        tax.local_saez_samples = lambda env=None: torch.arange(E * size * 2, dtype=torch.float64).reshape(-1, 2) + 1000 * rank
        got = {}

        def set_global(glob, env=None, tax=tax):
            assert glob.shape[0] <= tax.pooled_saez_capacity(env)
            got["g"] = glob.clone()

        tax.set_global_saez_buffer = set_global
        return env, tax, got

    env, tax, got = make()
    ok = env.build_config().saez_global_capacity == world * E * size  # the default covers every rank's full buffers
    glob = accumulate_and_broadcast_saez_buffers(env)
    ok = ok and glob.shape == (world * E * size, 2) and torch.equal(got["g"], glob)
    env_small, _, got_small = make(capacity=E * size)  # round 2's default: one rank's worth
    try:
        accumulate_and_broadcast_saez_buffers(env_small)
        ok = False
    except ValueError as exc:
        ok = ok and "_global_buffer_capacity" in str(exc) and "g" not in got_small
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_saez_pooled_capacity_covers_full_buffers_of_every_rank_world2_gloo():
    """with world_size > 1 the pooled buffer outgrew the default capacity (this rank's n_envs * _buffer_size)
    once the local buffers were full.  The default is now world_size * n_envs * _buffer_size, and a capacity that cannot
    hold the worst case is refused up front with the remedy in the message."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_saez_capacity_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def _obs_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ai_economist_amd.sharding import ObservationGather, RewardLogGather, shard_range

    E_total, n, K, T = 10, 2, 3, 7
    lo, hi = shard_range(E_total, rank, world)
    be = _FakeBackend(hi - lo, n)
    g = RewardLogGather(be, steps_per_gather=K, dst=0, keep=True, gather_obs=("obs_a_flat", "obs_a_action_mask"))
    for t in range(T):
        be.step(lo)
        g.after_step()
    g.finish()
    og = ObservationGather(be, keys=("obs_a_flat",))
    now = og()
    ok = True
    gid = torch.arange(E_total, dtype=torch.float32)

    def want_flat(t):
        return gid[:, None, None] * 100 + torch.arange(n)[None, :, None] * 10 + torch.arange(5)[None, None, :] + 0.5 * t

    if rank == 0:
        # one observation set per block, the partial one finish() ships included: steps 2, 5 and 6 -- in step with g.received
        ok &= len(g.received_obs) == T // K + 1 == len(g.received)
        for blk, obs in enumerate(g.received_obs):
            t = min((blk + 1) * K, T) - 1
            ok &= torch.equal(obs["obs_a_flat"], want_flat(t))
            ok &= torch.equal(obs["obs_a_action_mask"],
                              ((gid[:, None, None] + torch.arange(3)[None, None, :] + t) % 2).expand(E_total, n, 3))
        ok &= torch.equal(now["obs_a_flat"], want_flat(T - 1)) and tuple(now["obs_a_flat"].shape) == (E_total, n, 5)
    else:
        ok &= g.received_obs == [] and now is None
    try:
        ObservationGather(be, keys=("obs_a_nope",))
        ok = False
    except KeyError:
        pass
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_observation_gather_world2_gloo():
    """Opt-in gather of the observations (BASELINE north star: "(obs, reward, done)"): per block with the reward log,
    and on demand; rank-major replica order on the learner rank."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_obs_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}
