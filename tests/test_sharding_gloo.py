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
