"""Multi-GPU: env replicas are independent, so they shard across ranks with NO data-path
collective inside env.step(); the only exchange is one gather of (reward, done) per step
to the learner rank (RCCL over xGMI when the tensors are on GPUs, gloo in the CPU tests).

Reference: the reference has no distributed path at all; its
notion of data parallelism is WarpDrive's `num_envs` blocks on one GPU
(F/env_wrapper.py:202-211) and RLlib CPU rollout workers.
"""
import os


def shard_range(n_envs_total, rank, world_size):
    """Contiguous replica range [lo, hi) owned by `rank`; global replica id = lo + e."""
    assert n_envs_total % world_size == 0, "replicas must divide evenly across ranks"
    per = n_envs_total // world_size
    return rank * per, (rank + 1) * per


def dist_info():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


class RewardDoneGather:
    """Packs rewards_a [E, n], rewards_p [E], done [E] into one f32 [E, n + 2] buffer and
    gathers it to `dst` with a single collective per step."""

    def __init__(self, n_envs_local, n_agents, device, dst=0):
        import torch
        import torch.distributed as dist

        self.dist = dist
        self.dst = dst
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.E, self.n = n_envs_local, n_agents
        self.send = torch.empty((n_envs_local, n_agents + 2), dtype=torch.float32, device=device)
        self.recv = None
        if self.rank == dst:
            self.recv = [torch.empty_like(self.send) for _ in range(self.world)]

    def __call__(self, rewards_a, rewards_p, done):
        """Returns (rewards_a [W*E, n], rewards_p [W*E], done [W*E]) on dst, None elsewhere."""
        import torch

        self.send[:, : self.n] = rewards_a
        self.send[:, self.n] = rewards_p
        self.send[:, self.n + 1] = done.to(torch.float32)
        if self.world > 1:
            self.dist.gather(self.send, self.recv if self.rank == self.dst else None, dst=self.dst)
            if self.rank != self.dst:
                return None
            full = torch.cat(self.recv, dim=0)
        else:
            full = self.send
        return full[:, : self.n], full[:, self.n], full[:, self.n + 1] > 0.5
