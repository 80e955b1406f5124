"""Preconditioned SGLD with burn-in (mirror of ``pysgmcmc/samplers/sgld.py``).

The per-step op chain ``sgld.py:149-211`` is kernel K2,
``sgmcmc_sgld_step_{f32,f64}``.
"""
import torch

from pysgmcmc_amd import kernels
from pysgmcmc_amd.samplers._fused_bnn import FusedBNNStepsMixin
from pysgmcmc_amd.samplers.base_classes import BurnInMCMCSampler
from pysgmcmc_amd.stepsize_schedules import ConstantStepsizeSchedule

__all__ = ("SGLDSampler",)


class SGLDSampler(FusedBNNStepsMixin, BurnInMCMCSampler):
    """Stochastic Gradient Langevin Dynamics with the RMSprop-like preconditioner
    adapted during burn-in (keywords/defaults as ``sgld.py:32-35``).

    ``strict_reference_quirks`` (per-instance attribute, default ``False``; not a constructor keyword so that
    ``get_sampler``'s keyword reflection stays the reference's): the reference never forwards
    ``stepsize_schedule`` to its base class (``sgld.py:96-100``, quirk Q1), so its SGLD always runs at the base
    default ``ConstantStepsizeSchedule(0.01)`` whatever the caller passes. That is a bug and is fixed by default;
    ``sampler.strict_reference_quirks = True`` reproduces it (the caller's schedule is then ignored, not even
    ``update``d), ``= False`` switches back. Two samplers in one process can differ."""

    _STATE_ROWS = ("tau", "g", "v_hat", "minv")
    _FUSED_ROWS = ("theta", "grad", "tau", "g", "v_hat", "minv")          # row order of the fused small-model kernel

    def __init__(self, params, cost_fun, batch_generator=None,
                 stepsize_schedule=ConstantStepsizeSchedule(0.01),
                 burn_in_steps=3000, A=1.0, scale_grad=1.0,
                 session=None, dtype=torch.float64, seed=None):
        self._caller_schedule = stepsize_schedule
        self._reference_schedule = ConstantStepsizeSchedule(0.01)      # base default, base_classes.py:323
        self._strict = False
        super().__init__(
            params=params, cost_fun=cost_fun, batch_generator=batch_generator,
            burn_in_steps=burn_in_steps, seed=seed,
            session=session, dtype=dtype, stepsize_schedule=stepsize_schedule
        )
        self.A = float(A)
        self.scale_grad = float(scale_grad)

    @property
    def strict_reference_quirks(self):
        return self._strict

    @strict_reference_quirks.setter
    def strict_reference_quirks(self, strict):
        self._strict = bool(strict)
        self.stepsize_schedule = self._reference_schedule if self._strict else self._caller_schedule
        self.epsilon = self.stepsize_schedule.initial_value

    def _bytes_per_element(self):
        return (10 if self._adapting else 4) * self.arena.row("theta").element_size()     # K2: 40 / 16 B per f32 parameter

    _SCALARS_KIND = "sgld"

    def _step_scalars(self, eps):
        return (eps, self.A, self.scale_grad)

    def _kernel_step(self, eps, xi, sl=None, opts=None):
        rows = self._sliced_rows(("theta", "grad", "tau", "g", "v_hat", "minv"), sl)
        r = self._r_row()
        kernels.sgld_step(
            *rows, r if (r is None or sl is None) else r[sl],
            eps, self.A, self.scale_grad, self._adapting,
            xi=xi, stats=self._step_stats(), grad_decay=self._grad_decay, launch=self._launch(), opts=opts, **self._noise_args())
        self._stats_written()

    # ------------------------------------------------------------------ fused small-model path (see _fused_bnn.py)
    def _fused_bnn_launch(self, starts, costs, eps, n_steps, n_chains=1, chain_stride=None, bases=None):
        gen, cost, a = self.batch_generator, self.cost_fun, self.arena
        rows = bases or [a.row(k) for k in self._FUSED_ROWS]
        kernels.bnn_fused_sgld_steps(
            *rows, self._bnn_layer_sizes(), gen.x_dev, gen.y_dev.reshape(-1), starts, gen.batch_size,
            cost.batch_size, cost.n_examples, cost.wdecay, cost.prior_mean, cost.prior_var,
            eps, self.scale_grad, self.A, self.n_iterations, n_steps, max(self.burn_in_steps, 0),
            self._philox_seed, costs, n_chains=n_chains, chain_stride=chain_stride)
