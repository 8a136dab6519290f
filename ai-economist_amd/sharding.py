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


class ObservationGather:
    """Opt-in: the CURRENT observations of every rank's replicas on the learner rank -- BASELINE's north star names an
    "RCCL gather of (obs, reward, done)".  The default deployment keeps observations resident on the GPU that produced
    them (policy inference is data-parallel too; SURVEY.md 8(e): a full gather is 267 MB per GPU per step at C3);
    a learner that wants them anyway asks for the tensors by name:

        og = ObservationGather(backend, keys=("obs_a_flat", "obs_a_action_mask"))   # after env.reset()
        obs = og()      # dst: {key: tensor [W * E, ...]} (rank-major => global replica id = rank * E + e); others: None

    One collective per key and call (the tensors are contiguous [E, ...] blocks of the arena: no packing pass)."""

    def __init__(self, backend, keys=("obs_a_flat", "obs_a_action_mask", "obs_p_flat", "obs_p_action_mask"), dst=0):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.dst = torch, dist, dst
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.backend = backend
        self.keys = [k for k in keys if k in backend.tensors]
        missing = [k for k in keys if k not in backend.tensors]
        if missing:
            raise KeyError("ObservationGather: the environment has no tensor(s) %s" % missing)
        # two sets of snapshot / receive buffers: a gather started with start() runs while the steps go on (they
        # overwrite the live observation tensors, so what travels is a device-side snapshot taken in stream order)
        # (slot 2 belongs to the synchronous __call__: it must not reuse a slot a start()ed gather -- RewardLogGather's
        # block gathers use 0 and 1 -- may still have in flight)
        self.snap = [None, None, None]
        self.recv = [None, None, None]
        if self.world > 1:
            self.snap = [{k: torch.empty_like(backend.tensors[k].contiguous()) for k in self.keys} for _ in range(3)]
            if self.rank == dst:
                self.recv = [{k: [torch.empty_like(backend.tensors[k].contiguous()) for _ in range(self.world)]
                              for k in self.keys} for _ in range(3)]
        self.bytes_per_call = sum(int(backend.tensors[k].numel() * backend.tensors[k].element_size()) for k in self.keys)

    def start(self, slot=0):
        """Snapshots the current observations and starts their gather (asynchronous collectives, one per key); the
        steps that follow overlap it.  finish(slot) waits and returns what __call__ returns."""
        works = []
        if self.world > 1:
            for k in self.keys:
                self.snap[slot][k].copy_(self.backend.tensors[k])
                works.append(self.dist.gather(self.snap[slot][k], self.recv[slot][k] if self.rank == self.dst else None,
                                              dst=self.dst, async_op=True))
            return works
        # one rank: a copy, not a view of the live arena (the multi-rank path returns copies too)
        return {k: self.backend.tensors[k].clone() for k in self.keys}

    def finish(self, started, slot=0):
        if self.world > 1:
            for w in started:
                w.wait()
            if self.rank != self.dst:
                return None
            return {k: self.torch.cat(self.recv[slot][k], dim=0) for k in self.keys}
        return started if self.rank == self.dst else None

    def __call__(self):
        return self.finish(self.start(2), 2)


class RewardLogGather:
    """The same exchange, sized for xGMI: instead of one small collective per 40-microsecond step, the step
    kernel itself appends (rewards, done) of every step to a device-side log (aie_set_reward_log: slots of
    f32 [E, n + 2]) and ONE gather per `steps_per_gather` steps ships that many slots to the learner rank.
    The log holds two such blocks, so the (asynchronous) collective of one block overlaps the steps that fill
    the other.

        g = RewardLogGather(backend, steps_per_gather=64)   # after env.reset()
        for t in range(T):
            backend.step(...)            # or step_sample_next
            block = g.after_step()       # every 64th step: launches the gather of the block just completed
        g.finish()                       # waits for the outstanding collective

    On the destination rank `g.received` is the list of gathered blocks, each f32
    [W, steps_per_gather, E, n + 2] (rank-major => global replica id = rank * E + e), if `keep` is set.

    `gather_obs=(names...)` (opt-in): every completed block ALSO ships the observations the replicas hold at that
    moment -- what a learner on `dst` needs to act on / bootstrap from after it has consumed the block's rewards
    (`g.received_obs`: one {name: [W * E, ...]} dict per block on `dst`, ObservationGather).
    """

    def __init__(self, backend, steps_per_gather=64, dst=0, keep=False, force_collective=False, gather_obs=None,
                 host_staged=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.dst, self.K, self.keep = dst, int(steps_per_gather), keep
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.backend = backend
        self.log = backend.set_reward_log(2 * self.K)  # [2K, E, n + 2]
        # host_staged: the block travels through pinned host memory and the group's CPU collective (gloo gathers CPU
        # tensors only).  That is the exchange of several ranks that SHARE one device -- RCCL refuses two ranks on one
        # GPU -- i.e. of the two-ranks-on-one-GPU test and of `bench.py --oversubscribe-one-gpu`; default: whenever the
        # group's backend is gloo and the log lives on a GPU.
        if host_staged is None:
            host_staged = bool(dist.is_initialized() and dist.get_backend() == "gloo" and self.log.is_cuda)
        self.host_staged = bool(host_staged)
        self.stage = None
        if self.host_staged:
            self.stage = [torch.empty(self.log[: self.K].shape, dtype=self.log.dtype).pin_memory() for _ in range(2)]
        self.filled = 0   # steps written into the current block
        self.block = 0    # block being written (0 / 1)
        self.pending = [None, None]  # outstanding collective per block
        # force_collective: issue the gather even in a 1-rank group (exercises the RCCL path on one GPU)
        self.collective = self.world > 1 or (force_collective and dist.is_initialized())
        self.recv = None
        if self.rank == dst and self.collective:
            like = self.stage[0] if self.host_staged else self.log[: self.K]
            self.recv = [[torch.empty_like(like) for _ in range(self.world)] for _ in range(2)]
        self.received = []
        if gather_obs and self.host_staged:
            raise NotImplementedError("RewardLogGather: gather_obs travels over the device collective only (not host-staged)")
        self.obs_gather = ObservationGather(backend, gather_obs, dst) if gather_obs else None
        self.obs_pending = [None, None]  # observation gathers started with a block, waited for with it
        self.received_obs = []
        self.n_collectives = 0
        self.bytes_per_collective = int(self.log[: self.K].numel() * self.log.element_size())
        self.wait_seconds = 0.0  # host time spent waiting for a collective before its log block could be reused
        if self.collective:
            # the communicator is created by the first collective (hundreds of milliseconds): here, not inside
            # somebody's rollout
            warm = self.log[:1, :1].clone()
            if self.host_staged:
                warm = warm.cpu()
            self.dist.gather(warm, [torch.empty_like(warm) for _ in range(self.world)] if self.rank == dst else None,
                             dst=dst)

    def after_step(self):
        """Call once after every step; returns True when a block was handed to the collective."""
        self.filled += 1
        if self.filled < self.K:
            return False
        b = self.block
        view = self.log[b * self.K: (b + 1) * self.K]
        if self.collective:
            # async_op: the backend orders the collective behind the steps already enqueued (RCCL runs it on its
            # own stream), and wait() below orders later steps behind it -- the steps in between overlap it
            self.pending[b] = self.dist.gather(self._staged(view, b), self.recv[b] if self.rank == self.dst else None,
                                               dst=self.dst, async_op=True)
            self.n_collectives += 1
        elif self.keep:
            self.received.append(view.clone()[None])
        if self.obs_gather is not None:
            # asynchronous like the reward block: the collectives travel while the next block's steps run, and are
            # waited for where the block's own collective is
            self.obs_pending[b] = self.obs_gather.start(b)
        self.filled = 0
        self.block ^= 1
        self._wait(self.block)  # the block about to be overwritten must have left
        return True

    def _staged(self, view, b):
        """The block as the collective takes it: itself, or -- host_staged -- its copy in pinned host memory (a device ->
        host copy on the step stream, waited for here: the CPU collective reads it as soon as it is started)."""
        if not self.host_staged:
            return view
        dst = self.stage[b][: view.shape[0]]
        dst.copy_(view, non_blocking=True)
        self.torch.cuda.current_stream(view.device).synchronize()
        return dst

    def _wait(self, b):
        import time

        w = self.pending[b]
        if w is not None:
            t0 = time.perf_counter()
            w.wait()
            self.wait_seconds += time.perf_counter() - t0
            self.pending[b] = None
            if self.keep and self.rank == self.dst:
                self.received.append(self.torch.stack(self.recv[b]).clone())
        if self.obs_pending[b] is not None:
            t0 = time.perf_counter()
            obs = self.obs_gather.finish(self.obs_pending[b], b)
            self.wait_seconds += time.perf_counter() - t0
            self.obs_pending[b] = None
            if obs is not None:
                self.received_obs.append({k: v.clone() for k, v in obs.items()} if self.world > 1 else obs)

    def finish(self):
        """Ships what the current block holds (a rollout need not end on a block boundary) and waits for everything
        outstanding.  The partial block arrives as [W, filled, E, n + 2]."""
        self._wait(self.block ^ 1)
        self._wait(self.block)
        if self.filled:
            b, f = self.block, self.filled
            if self.obs_gather is not None:  # the partial block's observations travel too: received_obs stays in step with received
                self.obs_pending[b] = self.obs_gather.start(b)
            view = self.log[b * self.K: b * self.K + f]
            if self.collective:
                w = self.dist.gather(self._staged(view, b), [r[:f] for r in self.recv[b]] if self.rank == self.dst else None,
                                     dst=self.dst, async_op=True)
                self.n_collectives += 1
                import time

                t0 = time.perf_counter()
                w.wait()
                self.wait_seconds += time.perf_counter() - t0
                if self.keep and self.rank == self.dst:
                    self.received.append(self.torch.stack([r[:f] for r in self.recv[b]]).clone())
            elif self.keep:
                self.received.append(view.clone()[None])
            self._wait(b)  # (the partial block's observations)
            self.filled = 0
            # the next step writes slot 0 again (the writer's slot counter lives in the library)
            self.backend.rewind_reward_log()
            self.block = 0


def accumulate_and_broadcast_saez_buffers(envs, component_name="PeriodicBracketTax"):
    """The one algorithmic cross-replica exchange of the reference (tutorials/rllib/utils/remote.py:56-73
    `accumulate_and_broadcast_saez_buffers`): every environment replica's local Saez sample buffer is concatenated
    into one global buffer that all of them then use.  Here the replicas of a rank are a batch on one device and
    ranks are processes: local samples are concatenated in replica order on the device, all-gathered across the
    process group (RCCL on GPUs; variable lengths are padded to the longest), concatenated rank-major and handed to
    every environment of this rank (`PeriodicBracketTax.set_global_saez_buffer`).  `envs`: one batched environment or
    a list of them.  Returns the global buffer (float64 [G, 2])."""
    import torch
    import torch.distributed as dist

    if not isinstance(envs, (list, tuple)):
        envs = [envs]
    comps = [env.get_component(component_name) for env in envs]
    local = torch.cat([c.local_saez_samples(env) for c, env in zip(comps, envs)], dim=0)
    # capacity check up front: the pooled buffer can reach (ranks) x (replicas of every listed environment) x (local
    # buffer size) once the local buffers are full -- fail now, with the remedy, not in the middle of training
    world = dist.get_world_size() if dist.is_initialized() else 1
    worst = world * sum(int(env.n_envs) * int(c._buffer_size) for c, env in zip(comps, envs))
    for c, env in zip(comps, envs):
        if c.pooled_saez_capacity(env) < worst:
            raise ValueError("pooling %d rank(s) x %d environment(s) can reach %d Saez samples, but an environment's global "
                             "buffer holds %d: set PeriodicBracketTax._global_buffer_capacity = %d on the component before "
                             "the first reset()" % (world, len(envs), worst, c.pooled_saez_capacity(env), worst))
    if dist.is_initialized() and dist.get_world_size() > 1:
        world = dist.get_world_size()
        n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
        counts = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(counts, n)
        longest = int(max(int(c.item()) for c in counts))
        padded = torch.zeros((longest, 2), dtype=torch.float64, device=local.device)
        padded[: local.shape[0]] = local
        parts = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(parts, padded)
        glob = torch.cat([p[: int(c.item())] for p, c in zip(parts, counts)], dim=0)
    else:
        glob = local
    for c, env in zip(comps, envs):
        c.set_global_saez_buffer(glob, env)
    return glob
