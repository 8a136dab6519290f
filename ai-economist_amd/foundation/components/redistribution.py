"""`WealthRedistribution` (reference: F/components/redistribution.py:21-75; dynamics ->
wealth_component_step in csrc/aie_kernels.hip / aie_kernels_ose.hip) and
`PeriodicBracketTax` (reference: F/components/redistribution.py:78-346, 920-939;
dynamics -> tax_component_step / tax_enact in csrc/aie_kernels.hip).

Tax models: "model_wrapper" (planner picks discretised rates),
"us-federal-single-filer-2018-scaled", "fixed-bracket-rates" and "saez"
(redistribution.py:436-823: rates from the Saez formula over a per-replica buffer of
observed (income, marginal rate) pairs -> csrc/aie_kernels_saez.hip; the cross-replica
"global" buffer of the RLlib trainer, remote.py:56-73, is not part of the environment step).
"""
import numpy as np

from ... import _cabi
from .base import BaseComponent, component_registry


@component_registry.add
class WealthRedistribution(BaseComponent):
    """Passive: every step the mobile agents' total coin (inventory + escrow) is split evenly;
    no actions, state fields, observations, masks or metrics."""
    name = "WealthRedistribution"
    required_entities = ["Coin"]
    agent_subclasses = ["BasicMobileAgent"]
    comp_id = _cabi.COMP_WEALTH_REDISTRIBUTION

    def get_n_actions(self, agent_cls_name):
        return None

    def fill_config(self, cfg):
        pass


def _default_world_size():
    """Ranks whose replicas pool their Saez samples: the initialised process group, else the launcher's WORLD_SIZE."""
    import os

    try:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            return int(dist.get_world_size())
    except ImportError:
        pass
    return max(1, int(os.environ.get("WORLD_SIZE", "1")))


@component_registry.add
class PeriodicBracketTax(BaseComponent):
    name = "PeriodicBracketTax"
    component_type = "PeriodicTax"
    required_entities = ["Coin"]
    agent_subclasses = ["BasicMobileAgent", "BasicPlanner"]
    comp_id = _cabi.COMP_TAX

    US_FEDERAL_2018 = [0.1, 0.12, 0.22, 0.24, 0.32, 0.35, 0.37]

    def __init__(self, *base_args, disable_taxes=False, tax_model="model_wrapper",
                 period=100, rate_min=0.0, rate_max=1.0, rate_disc=0.05, n_brackets=5,
                 top_bracket_cutoff=100, usd_scaling=1000.0, bracket_spacing="us-federal",
                 fixed_bracket_rates=None, pareto_weight_type="inverse_income",
                 saez_fixed_elas=None, tax_annealing_schedule=None, **base_kwargs):
        super().__init__(*base_args, **base_kwargs)
        self.disable_taxes = bool(disable_taxes)
        self.tax_model = tax_model
        assert self.tax_model in ["model_wrapper", "us-federal-single-filer-2018-scaled",
                                  "saez", "fixed-bracket-rates"]
        self.pareto_weight_type = pareto_weight_type
        assert self.pareto_weight_type in ("inverse_income", "uniform")  # redistribution.py:636-643
        self._saez_fixed_elas = saez_fixed_elas
        if self._saez_fixed_elas is not None:
            self._saez_fixed_elas = float(self._saez_fixed_elas)
            assert self._saez_fixed_elas >= 0
        # samples a replica collects before the formula replaces random rates (redistribution.py:276)
        self._buffer_size = 500
        # pairs the pooled (global) buffer can hold: None = every replica of every rank with a full local buffer
        # (world_size * n_envs * _buffer_size, see pooled_saez_capacity); set it before the first reset() to pool
        # several environments per rank
        self._global_buffer_capacity = None
        self.tax_annealing_schedule = tax_annealing_schedule
        if tax_annealing_schedule is not None:  # redistribution.py:317-325
            assert isinstance(self.tax_annealing_schedule, (tuple, list))
            self._annealing_warmup = self.tax_annealing_schedule[0]
            self._annealing_slope = self.tax_annealing_schedule[1]
        self.period = int(period)
        assert self.period > 0
        self.rate_min = 0.0 if self.disable_taxes else float(rate_min)
        self.rate_max = 0.0 if self.disable_taxes else float(rate_max)
        assert 0 <= self.rate_min <= self.rate_max <= 1.0
        self.rate_disc = float(rate_disc)
        self.use_discretized_rates = self.tax_model == "model_wrapper"
        if self.use_discretized_rates:
            self.disc_rates = np.arange(self.rate_min, self.rate_max + self.rate_disc,
                                        self.rate_disc)
            self.disc_rates = self.disc_rates[self.disc_rates <= self.rate_max]
            assert len(self.disc_rates) > 1 or self.disable_taxes
            self.n_disc_rates = len(self.disc_rates)
        else:
            self.disc_rates = None
            self.n_disc_rates = 0

        self.n_brackets = int(n_brackets)
        assert self.n_brackets >= 2
        self.top_bracket_cutoff = float(top_bracket_cutoff)
        assert self.top_bracket_cutoff >= 10
        self.usd_scale = float(usd_scaling)
        assert self.usd_scale > 0
        self.bracket_spacing = bracket_spacing.lower()
        assert self.bracket_spacing in ["linear", "log", "us-federal"]
        if self.bracket_spacing == "linear":
            self.bracket_cutoffs = np.linspace(0, self.top_bracket_cutoff, self.n_brackets)
        elif self.bracket_spacing == "log":
            b0_max = self.top_bracket_cutoff / (2 ** (self.n_brackets - 2))
            self.bracket_cutoffs = np.concatenate(
                [[0], 2 ** np.linspace(np.log2(b0_max), np.log2(self.top_bracket_cutoff),
                                       n_brackets - 1)])
        else:
            self.bracket_cutoffs = (
                np.array([0, 9700, 39475, 84200, 160725, 204100, 510300]) / self.usd_scale)
            self.n_brackets = len(self.bracket_cutoffs)
            self.top_bracket_cutoff = float(self.bracket_cutoffs[-1])
        assert self.bracket_cutoffs[0] == 0
        if self.tax_model == "model_wrapper" and not self.disable_taxes and \
                len({int(c) for c in self.bracket_cutoffs}) < len(self.bracket_cutoffs):
            # The reference names the planner's action subspaces "TaxIndexBracket_%03d" % int(cutoff)
            # (redistribution.py:331-337): brackets whose cutoffs share an integer part would silently
            # share one action there.  Refuse instead of reproducing that.
            raise ValueError("bracket cutoffs {} collide after int(): choose a smaller usd_scaling / other "
                             "spacing".format([float(c) for c in self.bracket_cutoffs]))

        if self.tax_model == "us-federal-single-filer-2018-scaled":
            assert self.bracket_spacing == "us-federal"
        if self.tax_model == "fixed-bracket-rates":
            assert isinstance(fixed_bracket_rates, (tuple, list))
            assert np.min(fixed_bracket_rates) >= 0
            assert np.max(fixed_bracket_rates) <= 1
            assert len(fixed_bracket_rates) == self.n_brackets
            self._fixed_bracket_rates = np.array(fixed_bracket_rates, dtype=np.float64)
        else:
            self._fixed_bracket_rates = None

    # ---- Saez sample buffers (redistribution.py:515-546); they live on the device of the owning environment ----
    def reset_saez_buffers(self, env=None):
        """Empties every replica's sample buffer and the global one: random rates again until they refill
        (redistribution.py:546-550)."""
        env = env or self._env
        t = env.backend.tensors
        t["saez_buffer_len"].zero_()
        t["saez_reached_min_samples"].zero_()
        t["saez_additions"].zero_()
        t["saez_global_len"].zero_()

    def get_local_saez_buffer(self, env=None):
        """(buffer [E, capacity, 2] of (income, marginal rate) pairs, oldest first; filled lengths [E])."""
        t = (env or self._env).backend.tensors
        return t["saez_buffer"], t["saez_buffer_len"]

    def local_saez_samples(self, env=None):
        """Every replica's filled samples concatenated in replica order: float64 [sum(len), 2] on the device --
        what the reference's trainer collects per environment (tutorials/rllib/utils/remote.py:59-66)."""
        import torch

        buf, n = self.get_local_saez_buffer(env)
        keep = torch.arange(buf.shape[1], device=buf.device)[None, :] < n[:, None]
        return buf[keep]

    def set_global_saez_buffer(self, global_saez_buffer, env=None):
        """redistribution.py:530-533: from now on every replica's period start uses this buffer followed by its own
        samples added since the buffers were last reset.  `global_saez_buffer`: [G, 2] pairs (tensor / array / list),
        G <= the capacity fixed at construction (`_global_buffer_capacity`, default world_size * n_envs * _buffer_size:
        what every replica of every rank can pool)."""
        import torch

        env = env or self._env
        be = env.backend
        g = global_saez_buffer
        if not isinstance(g, torch.Tensor):
            g = torch.as_tensor(np.asarray(g, np.float64).reshape(-1, 2))
        g = g.to(device=be.device, dtype=torch.float64).contiguous()
        assert g.ndim == 2 and g.shape[1] == 2
        cap = self.pooled_saez_capacity(env)
        if g.shape[0] > cap:
            raise ValueError("global Saez buffer of %d pairs exceeds the capacity this environment was built with (%d = "
                             "_global_buffer_capacity, default world_size * n_envs * _buffer_size): set "
                             "PeriodicBracketTax._global_buffer_capacity before the first reset()" % (g.shape[0], cap))
        assert g.shape[0] == 0 or g.shape[0] >= int(be.tensors["saez_buffer_len"].max().item()), \
            "the global buffer must hold at least as many samples as a local one (redistribution.py:532)"
        be._check(be.lib.aie_set_global_saez_buffer(be.handle, g.data_ptr() if g.shape[0] else None, int(g.shape[0])))

    def pooled_saez_capacity(self, env=None):
        """Pairs the global buffer of `env` holds (fixed when its device arena is built)."""
        env = env or self._env
        cap = self._global_buffer_capacity
        if cap is not None:
            return int(cap)
        return _default_world_size() * int(env.n_envs) * int(self._buffer_size)

    def get_n_actions(self, agent_cls_name):
        if agent_cls_name == "BasicPlanner":
            if self.tax_model == "model_wrapper" and not self.disable_taxes:
                return [("TaxIndexBracket_{:03d}".format(int(r)), self.n_disc_rates)
                        for r in self.bracket_cutoffs]
        return 0

    def fill_config(self, cfg):
        if self.n_brackets > _cabi.MAX_BRACKETS:
            raise ValueError("n_brackets > {}".format(_cabi.MAX_BRACKETS))
        cfg.tax_disable = int(self.disable_taxes)
        cfg.tax_rate_max = float(self.rate_max)
        cfg.tax_rate_min = float(self.rate_min)
        cfg.saez_buffer_size = int(self._buffer_size)
        if self.tax_model == "saez":
            # every replica of every rank may pool a full local buffer (sharding.accumulate_and_broadcast_saez_buffers)
            cap = self._global_buffer_capacity
            cap = int(cap if cap is not None else _default_world_size() * cfg.n_envs * int(self._buffer_size))
            if not 0 <= cap < 2 ** 31:  # (an int32 field of the C ABI; 16 bytes per pair on the device)
                raise ValueError("saez global buffer capacity %d (ranks x replicas x buffer_size) does not fit: pass "
                                 "saez_global_capacity= explicitly (0 switches the cross-replica buffer off)" % cap)
            cfg.saez_global_capacity = cap
        cfg.saez_pareto_weight_uniform = int(self.pareto_weight_type == "uniform")
        cfg.saez_fixed_elas_given = int(self._saez_fixed_elas is not None)
        cfg.saez_fixed_elas = float(self._saez_fixed_elas or 0.0)
        if self.tax_annealing_schedule is not None:
            cfg.tax_annealing = 1
            cfg.tax_annealing_warmup = float(self._annealing_warmup)
            cfg.tax_annealing_slope = float(self._annealing_slope)
        cfg.tax_model = _cabi.TAX_MODEL[self.tax_model]
        cfg.tax_period = self.period
        cfg.tax_n_brackets = self.n_brackets
        for i, v in enumerate(self.bracket_cutoffs):
            cfg.tax_bracket_cutoffs[i] = float(v)
        if self.use_discretized_rates:
            if self.n_disc_rates > _cabi.MAX_RATES:
                raise ValueError("more than {} discretised rates".format(_cabi.MAX_RATES))
            cfg.tax_n_disc_rates = self.n_disc_rates
            for i, v in enumerate(self.disc_rates):
                cfg.tax_disc_rates[i] = float(v)
        else:
            cfg.tax_n_disc_rates = 0
            base = (self.US_FEDERAL_2018
                    if self.tax_model == "us-federal-single-filer-2018-scaled"
                    else np.zeros(self.n_brackets) if self.tax_model == "saez"
                    else self._fixed_bracket_rates)
            for i, v in enumerate(np.minimum(np.array(base, dtype=np.float64), self.rate_max)):
                cfg.tax_fixed_rates[i] = float(v)
