"""SGHMC with scale-adapted burn-in (mirror of ``pysgmcmc/samplers/sghmc.py``).

The whole per-step op chain of the reference (``sghmc.py:165-251``: r, tau, minv,
g, v_hat, noise scale, momentum, theta) is kernel K1,
``sgmcmc_sghmc_step_{f32,f64}`` -- one launch for all parameters.
"""
import torch

from pysgmcmc_amd import kernels
from pysgmcmc_amd.samplers._fused_bnn import FusedBNNStepsMixin
from pysgmcmc_amd.samplers.base_classes import BurnInMCMCSampler
from pysgmcmc_amd.stepsize_schedules import ConstantStepsizeSchedule

__all__ = ("SGHMCSampler",)


class SGHMCSampler(FusedBNNStepsMixin, BurnInMCMCSampler):
    """Stochastic Gradient Hamiltonian Monte-Carlo sampler with the burn-in
    adaptation of Springenberg et al. 2016 (same keywords and defaults as the
    reference constructor, ``sghmc.py:31-34``)."""

    _STATE_ROWS = ("V", "tau", "g", "v_hat", "minv")
    _FUSED_ROWS = ("theta", "V", "grad", "tau", "g", "v_hat", "minv")     # row order of the fused small-model kernel

    def __init__(self, params, cost_fun, batch_generator=None,
                 stepsize_schedule=ConstantStepsizeSchedule(0.01),
                 burn_in_steps=3000, mdecay=0.05, scale_grad=1.0,
                 session=None, dtype=torch.float64, seed=None):
        super().__init__(
            params=params, cost_fun=cost_fun, burn_in_steps=burn_in_steps,
            batch_generator=batch_generator,
            seed=seed, dtype=dtype, session=session,
            stepsize_schedule=stepsize_schedule
        )
        self.mdecay = float(mdecay)
        self.scale_grad = float(scale_grad)
        # momentum starts at zero (sghmc.py:152-155); tau = g = v_hat = minv = 1 set by the base

    def _bytes_per_element(self):
        return (12 if self._adapting else 6) * self.arena.row("theta").element_size()     # K1: 48 / 24 B per f32 parameter

    _SCALARS_KIND = "sghmc"

    def _step_scalars(self, eps):
        return (eps, self.scale_grad, self.mdecay)

    def _kernel_step(self, eps, xi, sl=None, opts=None):
        rows = self._sliced_rows(("theta", "V", "grad", "tau", "g", "v_hat", "minv"), sl)
        r = self._r_row()
        kernels.sghmc_step(
            *rows, r if (r is None or sl is None) else r[sl],
            eps, self.scale_grad, self.mdecay, self._adapting,
            xi=xi, stats=self._step_stats(), grad_decay=self._grad_decay, launch=self._launch(), opts=opts, **self._noise_args())
        self._stats_written()

    # ------------------------------------------------------------------ fused small-model path (see _fused_bnn.py)
    def _fused_bnn_launch(self, starts, costs, eps, n_steps, n_chains=1, chain_stride=None, bases=None):
        gen, cost, a = self.batch_generator, self.cost_fun, self.arena
        rows = bases or [a.row(k) for k in self._FUSED_ROWS]
        kernels.bnn_fused_sghmc_steps(
            *rows, self._bnn_layer_sizes(), gen.x_dev, gen.y_dev.reshape(-1), starts, gen.batch_size,
            cost.batch_size, cost.n_examples, cost.wdecay, cost.prior_mean, cost.prior_var,
            eps, self.scale_grad, self.mdecay, self.n_iterations, n_steps, max(self.burn_in_steps, 0),
            self._philox_seed, costs, n_chains=n_chains, chain_stride=chain_stride)
