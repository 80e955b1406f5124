"""SGHMC with scale-adapted burn-in (mirror of ``pysgmcmc/samplers/sghmc.py``).

The whole per-step op chain of the reference (``sghmc.py:165-251``: r, tau, minv,
g, v_hat, noise scale, momentum, theta) is kernel K1,
``sgmcmc_sghmc_step_{f32,f64}`` -- one launch for all parameters.
"""
import torch

from pysgmcmc_amd import kernels
from pysgmcmc_amd.samplers.base_classes import BurnInMCMCSampler
from pysgmcmc_amd.stepsize_schedules import ConstantStepsizeSchedule

__all__ = ("SGHMCSampler",)


class SGHMCSampler(BurnInMCMCSampler):
    """Stochastic Gradient Hamiltonian Monte-Carlo sampler with the burn-in
    adaptation of Springenberg et al. 2016 (same keywords and defaults as the
    reference constructor, ``sghmc.py:31-34``)."""

    _STATE_ROWS = ("V", "tau", "g", "v_hat", "minv")

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

    def _kernel_step(self, eps, xi):
        a = self.arena
        kernels.sghmc_step(
            a.row("theta"), a.row("V"), a.row("grad"),
            a.row("tau"), a.row("g"), a.row("v_hat"), a.row("minv"), self._r_row(),
            eps, self.scale_grad, self.mdecay, self._adapting,
            xi=xi, stats=self._step_stats(), grad_decay=self._grad_decay, **self._noise_args())
        if self._stats is not None:
            self._stats_valid = True          # the workspace now holds this step's per-block partials
            self._stats_out_valid = False     # K7 runs lazily (sampler.stats); the BNN head reads the partials

    # ------------------------------------------------------------------ fused small-model path
    def fused_bnn_available(self):
        """True when whole steps can run inside ONE kernel (``sgmcmc_bnn_fused_sghmc_steps``): the cost is
        the library's MLP-BNN cost (``BNNCost``, weight prior folded), fed by a ``WindowBatches`` generator,
        the net has one output unit, at most 8 layers, and its activations fit the LDS."""
        cost, gen = self.cost_fun, self.batch_generator
        if self.device.type != "cuda" or self.noise_source is not None:
            return False
        if not (hasattr(cost, "fold_prior") and cost.fold_prior and hasattr(gen, "next_starts")):
            return False
        if gen.x_placeholder is not cost.x_placeholder or gen.y_placeholder is not cost.y_placeholder:
            return False
        sizes = self._bnn_layer_sizes()
        if sizes is None or sizes[-1] != 1 or len(sizes) - 1 > 8:
            return False
        lds = 160 + ((2 * sum(sizes) + 1) * gen.batch_size + self.arena.n + 4) * self.arena.row("theta").element_size()
        return lds <= 160 * 1024 and gen.x_dev.dtype == self._torch_dtype and gen.x_dev.is_contiguous()

    def _bnn_layer_sizes(self):
        shapes = self.arena.shapes
        if len(shapes) < 3 or len(shapes) % 2 == 0 or shapes[-1] not in ((1, 1), (1,), ()):
            return None
        sizes = [shapes[0][0]]
        for l in range((len(shapes) - 1) // 2):
            w, b = shapes[2 * l], shapes[2 * l + 1]
            if len(w) != 2 or w[0] != sizes[-1] or b != (w[1],):
                return None
            sizes.append(w[1])
        return sizes

    def fused_bnn_steps(self, n_steps):
        """Advance the chain by ``n_steps`` complete steps in one launch (one workgroup; see
        ``csrc/sgmcmc_bnn_fused.hip``). Same chain as ``n_steps`` calls of ``next()`` up to the rounding of
        the matrix products (same windows, same Philox stream, same update operator). Returns the
        device tensor of the ``n_steps`` costs. Needs a stepsize that is constant over the chunk."""
        if not self.fused_bnn_available():
            raise ValueError("fused_bnn_steps: this sampler/cost/batch generator does not fit the fused small-model kernel")
        n_steps = int(n_steps)
        eps = [next(self.stepsize_schedule) for _ in range(n_steps)]
        if any(e != eps[0] for e in eps):
            raise ValueError("fused_bnn_steps needs a constant stepsize over the chunk")
        self.epsilon = eps[0]
        gen, cost, a = self.batch_generator, self.cost_fun, self.arena
        starts = torch.as_tensor(gen.next_starts(n_steps), dtype=torch.int32).to(self.device)
        costs = torch.empty(n_steps, dtype=self._torch_dtype, device=self.device)
        kernels.bnn_fused_sghmc_steps(
            a.row("theta"), a.row("V"), a.row("grad"), a.row("tau"), a.row("g"), a.row("v_hat"), a.row("minv"),
            self._bnn_layer_sizes(), gen.x_dev, gen.y_dev.reshape(-1), starts, gen.batch_size,
            cost.batch_size, cost.n_examples, cost.wdecay, cost.prior_mean, cost.prior_var,
            eps[0], self.scale_grad, self.mdecay, self.n_iterations, n_steps, max(self.burn_in_steps, 0),
            self._philox_seed, costs)
        self.n_iterations += n_steps
        self._stats_valid = False                 # theta moved without the statistics workspace
        self._grad_decay = float(cost.wdecay / ((a.n + 3e-16) * cost.n_examples))
        self.cost = costs[-1]
        return costs
