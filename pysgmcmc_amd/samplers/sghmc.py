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

    # ------------------------------------------------------------------ weight-gradient GEMM with the update as epilogue
    def _fused_gemm_plan(self):
        """Slices of the arena for ``fuse_update_into_gemm``: one per hidden dense layer of an MLP cost function,
        ``[W_l | everything up to W_{l+1}]`` (the top one runs to the end of the arena), or None when the model / dtype /
        alignment does not fit ``sgmcmc_gemm_tn_sghmc_f32`` (the sampler then steps as usual)."""
        a, params = self.arena, self.params
        if self._torch_dtype != torch.float32 or not hasattr(self.cost_fun, "cost_and_grad") or (len(params) - 1) % 2:
            return None
        n_layers = (len(params) - 1) // 2
        L = n_layers - 1
        if n_layers < 2 or params[2 * L].dim() != 2 or params[2 * L].shape[1] != 1:
            return None
        plan, offs = [], [int(a.offsets[2 * l]) for l in range(L)] + [a.n]
        for l in range(L):
            W = params[2 * l]
            M, N = (int(W.shape[0]), int(W.shape[1])) if W.dim() == 2 else (0, 0)
            lo, hi = offs[l], offs[l + 1]
            if W.dim() != 2 or N % 128 or M % 4 or lo % 4 or lo != int(a.offsets[2 * l]) or hi - lo < M * N or (l == 0 and lo != 0):
                return None
            plan.append(dict(layer=l, lo=lo, hi=hi, M=M, N=N, n_tail=hi - lo - M * N,
                             blocks=kernels.gemm_tn_sghmc_blocks(M, N, hi - lo - M * N)))
        base = 0
        for p in plan:
            p["rec_base"], base = base, base + p["blocks"]
        return plan, base

    def _fused_weight_update(self, plan, total, eps):
        """The hook handed to the cost pipeline while the fused graph is captured."""
        a = self.arena
        by_layer = {p["layer"]: p for p in plan}

        def hook(l, h_in, delta):
            p = by_layer.get(l)
            if p is None or h_in.shape[0] % 16:
                return False
            sl = slice(p["lo"], p["hi"])
            kernels.gemm_tn_sghmc(h_in, delta, a.row("theta")[sl], a.row("V")[sl], a.row("minv")[sl],
                                  a.row("grad")[p["lo"] + p["M"] * p["N"]:p["hi"]] if p["n_tail"] else None,
                                  eps, self.scale_grad, self.mdecay, grad_decay=self._fused_grad_decay, seed=self._philox_seed,
                                  step=0, step_dev=self._step_ctr, first_element=p["lo"], stats=self._step_stats(),
                                  stats_base=p["rec_base"], stats_total=total)
            return True
        return hook

    # ------------------------------------------------------------------ fused small-model path (see _fused_bnn.py)
    def _fused_bnn_launch(self, starts, costs, eps, n_steps, n_chains=1, chain_stride=None, bases=None):
        gen, cost, a = self.batch_generator, self.cost_fun, self.arena
        rows = bases or [a.row(k) for k in self._FUSED_ROWS]
        kernels.bnn_fused_sghmc_steps(
            *rows, self._bnn_layer_sizes(), gen.x_dev, gen.y_dev.reshape(-1), starts, gen.batch_size,
            cost.batch_size, cost.n_examples, cost.wdecay, cost.prior_mean, cost.prior_var,
            eps, self.scale_grad, self.mdecay, self.n_iterations, n_steps, max(self.burn_in_steps, 0),
            self._philox_seed, costs, n_chains=n_chains, chain_stride=chain_stride)
