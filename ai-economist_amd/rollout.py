"""Trainer-facing rollout loop: policy -> env.step with NOTHING on the host between the steps.

The reference's GPU trainer calls, per step, `sample actions from the policy` then `env_wrapper.step_all_envs()`
(ai_economist/training/training_script.py:88-133 builds that trainer; F/env_wrapper.py:355-377 is the step it drives),
each a handful of kernel launches issued from Python.  At 25 us per environment step the host is the bottleneck long
before the GPU is.  `GraphedStep` captures one iteration -- the caller's policy (any torch code that reads the
observation tensors and writes the action buffers), `aie_step`, and the masked reset of the replicas that finished --
into a hipGraph once and replays it: one host call per environment step, or per `unroll` steps.

Replay-safety is a property of the C ABI, not of this wrapper: aie_step takes nothing by value that changes from step
to step (the random-policy draw index and the reward-log slot are per-replica record fields the kernels advance
themselves, include/aie.h), and never loads code or synchronises.  tests/test_gpu_parity.py:
test_step_is_hipgraph_replayable replays a captured loop against its eager twin and the oracle.

torch is plumbing here (streams, graph capture, the policy network); the environment step is the HIP kernel.
"""


class GraphedStep:
    def __init__(self, env, policy, auto_reset=True, unroll=1, warmup=3):
        """env: a batched environment (foundation.make_env_instance(..., n_envs=E)), already reset.
        policy(tensors, actions_a, actions_p): reads observation tensors (env.backend.tensors: "obs_a_flat",
        "obs_a_action_mask", "obs_p_flat", ... zero-copy views of the arena) and writes int32 actions IN PLACE into
        actions_a [E, n, width] / actions_p [E, width_p].  It is captured: no host synchronisation, no data-dependent
        Python control flow, fixed shapes.
        auto_reset: replicas restart right behind the step that ends their episode (aie_set_auto_reset).
        unroll: environment steps per replay."""
        import torch

        self.torch = torch
        self.env, self.be = env, env.backend
        self.policy = policy
        self.unroll = int(unroll)
        be = self.be
        if auto_reset:
            be.set_auto_reset(True)
        # The capture records whichever step kernel is current and replays it forever, and auto-reset restarts replicas
        # without passing through aie_reset (where a finished background specialisation is adopted otherwise): wait for
        # the configuration's specialised kernels here, best effort (the generic kernel gives the same results)
        self.specialised = bool(be.specialize(required=False))
        self.actions_a, self.actions_p = be._action_buffers(0)
        self.graph = None
        self._capture(warmup)

    def _iteration(self):
        self.policy(self.be.tensors, self.actions_a, self.actions_p)
        self.be.step(self.actions_a, self.actions_p)

    def eager(self, iterations=1):
        """The same iteration issued call by call (what the capture recorded): `iterations` x unroll steps."""
        for _ in range(iterations * self.unroll):
            self._iteration()

    def _capture(self, warmup):
        torch = self.torch
        # torch's capture protocol: run the work a few times on a side stream first (lazy initialisations -- cuBLAS/
        # hipBLASLt handles, allocator pools -- must not happen inside the capture); the warm-up iterations ARE
        # environment steps (the caller resets afterwards if it wants a clean start)
        side = torch.cuda.Stream(device=self.be.device)
        side.wait_stream(torch.cuda.current_stream(self.be.device))
        with torch.cuda.stream(side):
            for _ in range(max(0, warmup)):
                self._iteration()
        torch.cuda.current_stream(self.be.device).wait_stream(side)
        torch.cuda.synchronize(self.be.device)
        self.warmup_steps = max(0, warmup)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            for _ in range(self.unroll):
                self._iteration()

    def replay(self, n=1):
        """n replays = n * unroll environment steps; asynchronous like every other call."""
        for _ in range(n):
            self.graph.replay()


class MaskedMLPPolicy:
    """A small policy network of the shape the reference's trainers use on the flat observations (fully connected
    trunk, one categorical head per action subspace, `action_mask` applied to the logits: base_env.py:141-145,
    tutorials/rllib/env_wrapper.py:50-211) with random-init weights: enough to put a real policy's launches between the
    environment steps.  The network is torch (three GEMMs with fused bias per actor class, two ReLUs); masking and
    sampling -- what every trainer does with its logits -- is ONE launch of the library (aie_sample_policy_actions:
    Gumbel-max under the action masks, float64 scores, the draw index a record field, so the loop can be captured).
    `sampler="torch"` keeps the round-4 formulation (Gumbel noise, masked_fill and argmax as ~15 elementwise launches per
    actor class) for comparison."""

    def __init__(self, be, hidden=128, seed=0, dtype=None, sampler="library", sample_seed=1234):
        import torch

        self.torch = torch
        self.be = be
        t = be.tensors
        g = torch.Generator(device="cpu").manual_seed(seed)
        dev = be.device
        dtype = dtype or torch.float32
        self.dtype = dtype
        FA = t["obs_a_flat"].shape[-1]
        MA = t["obs_a_action_mask"].shape[-1]
        FP = t["obs_p_flat"].shape[-1]
        MP = t["obs_p_action_mask"].shape[-1]

        def lin(i, o):
            return (torch.randn(i, o, generator=g) / i ** 0.5).to(dev, dtype), torch.zeros(o, device=dev, dtype=dtype)

        self.wa1, self.ba1 = lin(FA, hidden)
        self.wa2, self.ba2 = lin(hidden, hidden)
        self.wa3, self.ba3 = lin(hidden, MA)
        self.wp1, self.bp1 = lin(FP, hidden)
        self.wp2, self.bp2 = lin(hidden, hidden)
        self.wp3, self.bp3 = lin(hidden, MP)
        self.MA, self.MP = MA, MP
        cfg = be.cfg
        self.multi_a = bool(cfg.multi_action_mode_agents)
        self.multi_p = bool(cfg.multi_action_mode_planner)
        self.p_width = be._act_p_width()
        assert sampler in ("library", "torch")
        self.sampler = sampler
        self.sample_seed = int(sample_seed)
        if self.multi_a and sampler == "torch":
            raise NotImplementedError("MaskedMLPPolicy(sampler='torch'): single-action agents (the BASELINE configurations)")
        self._fused_relu = None  # decided at the first call (outside any capture: GraphedStep warms the policy up first)
        self.counter = torch.zeros((), dtype=torch.float32, device=dev)
        if sampler == "torch":
            self.idx_a = torch.arange(be.E * be.n * MA, device=dev, dtype=torch.float32).view(be.E, be.n, MA)
            self.idx_p = torch.arange(be.E * MP, device=dev, dtype=torch.float32).view(be.E, MP)

    def _gumbel(self, idx, salt):
        torch = self.torch
        u = torch.frac(torch.sin(idx * 12.9898 + (self.counter + salt) * 78.233) * 43758.5453).abs_()
        u = u.clamp_(1e-6, 1.0 - 1e-6)
        return -torch.log(-torch.log(u))

    def _layer(self, b, x, w):
        """relu(x @ w + b): one launch where the BLAS library fuses the activation into the GEMM's epilogue
        (torch._addmm_activation), else the GEMM and an in-place ReLU."""
        torch = self.torch
        if self._fused_relu is None:
            self._fused_relu = False
            fn = getattr(torch, "_addmm_activation", None)
            if fn is not None:
                try:
                    got = fn(b, x, w, use_gelu=False)
                    self._fused_relu = bool(torch.allclose(got, torch.addmm(b, x, w).relu_(), rtol=1e-4, atol=1e-4))
                except Exception:
                    self._fused_relu = False
        if self._fused_relu:
            return torch._addmm_activation(b, x, w, use_gelu=False)
        return torch.addmm(b, x, w).relu_()

    def logits(self, tensors):
        """(agents' logits [E * n, MA], planner's logits [E, MP]) in the layout of the flattened action masks."""
        torch = self.torch
        xa = tensors["obs_a_flat"].to(self.dtype).reshape(-1, self.wa1.shape[0])
        h = self._layer(self.ba1, xa, self.wa1)
        h = self._layer(self.ba2, h, self.wa2)
        la = torch.addmm(self.ba3, h, self.wa3).float()
        xp = tensors["obs_p_flat"].to(self.dtype)
        h = self._layer(self.bp1, xp, self.wp1)
        h = self._layer(self.bp2, h, self.wp2)
        lp = torch.addmm(self.bp3, h, self.wp3).float()
        return la, lp

    def __call__(self, tensors, actions_a, actions_p):
        torch = self.torch
        la, lp = self.logits(tensors)
        if self.sampler == "library":
            self.be.sample_policy_actions(la, lp, self.sample_seed, out=(actions_a, actions_p))
            return
        la = la.view(self.be.E, self.be.n, self.MA) + self._gumbel(self.idx_a, 0.0)
        la = la.masked_fill(tensors["obs_a_action_mask"] < 0.5, -1e30)
        actions_a.view(self.be.E, self.be.n).copy_(la.argmax(-1))
        lp = lp + self._gumbel(self.idx_p, 0.5)
        lp = lp.masked_fill(tensors["obs_p_action_mask"] < 0.5, -1e30)
        if self.multi_p and self.p_width > 1:  # one categorical head per bracket: [E, width, 1 + rates]
            actions_p.copy_(lp.view(self.be.E, self.p_width, -1).argmax(-1))
        else:
            actions_p.view(self.be.E).copy_(lp.argmax(-1))
        self.counter += 1.0
