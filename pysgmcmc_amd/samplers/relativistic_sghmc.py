"""Relativistic SGHMC (Lu et al., AISTATS 2017; mirror of
``pysgmcmc/samplers/relativistic_sghmc.py``).

The per-step op chain ``relativistic_sghmc.py:120-140`` is kernel K3,
``sgmcmc_rsghmc_step_{f32,f64}``, applied per element. The reference keeps ONE
scalar momentum per parameter *tensor* (``:108-113``) and therefore only works
for 1-element parameters; for those this class is step-for-step identical, and
for larger tensors it is the natural element-wise generalisation.

Initial momenta. The reference draws them by adaptive rejection sampling from
p(p) ~ exp(-m c^2 sqrt(p^2/(m^2 c^2) + 1)) with the un-vendored ``arspy``
package (``:208-223``; parity unpinned -- neither arspy nor its outputs are in the
reference). ARS is an exact sampler, so any correct sampler of that law is
distribution-identical. Here the draws are made by inverse-CDF on a tabulated
CDF (65 537 knots, tail mass < 1e-13), driven by the chain's Philox stream
(reserved step index 2^63), so a seed fixes them; tests/test_relativistic_momentum.py
KS-tests them against a restatement of Gilks & Wild's sampler (oracle/ars_oracle.py)
and against the law's quadrature CDF.

``strict_reference_quirks`` (per-instance attribute, default ``False``; quirk Q5): the
reference draws ONE momentum per parameter *tensor* on the host
(``n_params=len(self.params)``, ``:108-113``) and reshapes every update to that scalar's
shape, so it is only defined for 1-element parameters. ``sampler.strict_reference_quirks =
True`` (before the first step) redraws the momenta that way -- ``len(params)`` draws from the
host ``RandomState(seed)``, one per tensor -- and refuses parameters with more than one
element like the reference's arithmetic does.
"""
import numpy as np
import torch

from pysgmcmc_amd import kernels
from pysgmcmc_amd.samplers.base_classes import MCMCSampler
from pysgmcmc_amd.stepsize_schedules import ConstantStepsizeSchedule

__all__ = ("RelativisticSGHMCSampler",)

_INIT_STREAM_STEP = 1 << 63


def _relativistic_quantile_table(m, c, knots=65537):
    """(cdf, p) table of the 1-D relativistic momentum law on a uniform p grid."""
    p_max = 40.0 / c + 10.0 * m * c            # log-density < -40 relative to the mode beyond this
    p = np.linspace(-p_max, p_max, knots)
    logpdf = -m * c ** 2 * np.sqrt(p ** 2 / (m ** 2 * c ** 2) + 1.0)
    pdf = np.exp(logpdf - logpdf.max())
    cdf = np.concatenate([[0.0], np.cumsum(0.5 * (pdf[1:] + pdf[:-1]))])
    cdf /= cdf[-1]
    return cdf, p


def _sample_relativistic_momentum(m, c, n_params, bounds=(float("-inf"), float("inf")), seed=None,
                                  device="cpu", dtype=torch.float64):
    """``n_params`` draws from the relativistic momentum law (flat tensor).

    On a GPU the uniforms come from the K5 Philox kernel; on the CPU (host-logic
    tests, tiny n) from ``numpy.random.RandomState(seed)``.

    >>> len(_sample_relativistic_momentum(m=1.0, c=1.0, n_params=10)) == 10
    True
    """
    assert isinstance(m, float)
    assert isinstance(c, float)
    cdf, p = _relativistic_quantile_table(m, c)
    device = torch.device(device)
    if device.type == "cuda":
        z = torch.empty(n_params, dtype=torch.float64, device=device)
        kernels.philox_normal(z, 0 if seed is None else seed, _INIT_STREAM_STEP)
        u = torch.special.ndtr(z)
    else:
        u = torch.as_tensor(np.random.RandomState(seed).uniform(size=n_params))
    cdf_t = torch.as_tensor(cdf, device=device)
    p_t = torch.as_tensor(p, device=device)
    idx = torch.searchsorted(cdf_t, u.to(cdf_t.dtype)).clamp(1, cdf_t.numel() - 1)
    c0, c1 = cdf_t[idx - 1], cdf_t[idx]
    w = ((u - c0) / (c1 - c0).clamp_min(1e-300)).clamp(0.0, 1.0)
    out = p_t[idx - 1] + w * (p_t[idx] - p_t[idx - 1])
    lo, hi = bounds
    return out.clamp(lo, hi).to(dtype)


class RelativisticSGHMCSampler(MCMCSampler):
    """Relativistic SGHMC (keywords/defaults as ``relativistic_sghmc.py:24-27``)."""

    _STATE_ROWS = ("p",)

    def __init__(self, params, cost_fun, batch_generator=None,
                 stepsize_schedule=ConstantStepsizeSchedule(0.001),
                 mass=1.0, speed_of_light=1.0, D=1.0, Bhat=0.0,
                 session=None, dtype=torch.float64, seed=None):
        super().__init__(
            params=params, cost_fun=cost_fun, batch_generator=batch_generator,
            stepsize_schedule=stepsize_schedule,
            seed=seed, dtype=dtype, session=session
        )
        self.mass = float(mass)
        self.speed_of_light = float(speed_of_light)
        self.D = float(D)
        self.Bhat = float(Bhat)
        self._strict = False
        self._draw_initial_momenta()

    def _draw_initial_momenta(self):
        if self._strict:
            # the reference: one scalar per parameter tensor, drawn on the host (relativistic_sghmc.py:108-113)
            sizes = self.arena.sizes
            assert all(int(sz) == 1 for sz in sizes), (
                "strict_reference_quirks: the reference keeps one scalar momentum per parameter tensor and reshapes the "
                "update to its shape (relativistic_sghmc.py:108-113,126-129); only 1-element parameters are defined")
            per_tensor = _sample_relativistic_momentum(m=self.mass, c=self.speed_of_light, n_params=len(self.params),
                                                      seed=self.seed, device="cpu", dtype=self._torch_dtype)
            row = self.arena.row("p")
            for off, sz, val in zip(self.arena.offsets, sizes, per_tensor):
                row[int(off):int(off) + int(sz)] = val.to(row.device)
            return
        p0 = _sample_relativistic_momentum(
            m=self.mass, c=self.speed_of_light, n_params=self.arena.n,
            seed=self._philox_seed if self.device.type == "cuda" else self.seed,
            device=self.device, dtype=self._torch_dtype)
        self.arena.row("p").copy_(p0)

    @property
    def strict_reference_quirks(self):
        return self._strict

    @strict_reference_quirks.setter
    def strict_reference_quirks(self, strict):
        assert self.n_iterations == 0, "strict_reference_quirks selects how the INITIAL momenta are drawn: set it before stepping"
        self._strict = bool(strict)
        self._draw_initial_momenta()

    @property
    def momentum(self):
        return self.arena.views("p")

    def _bytes_per_element(self):
        return 5 * self.arena.row("theta").element_size()                                  # K3: 20 B per f32 parameter

    _SCALARS_KIND = "rsghmc"

    def _step_scalars(self, eps):
        return (eps, self.mass, self.speed_of_light, self.D, self.Bhat)

    def _kernel_step(self, eps, xi, sl=None, opts=None):
        rows = self._sliced_rows(("theta", "p", "grad"), sl)
        kernels.rsghmc_step(
            *rows, eps, self.mass, self.speed_of_light, self.D, self.Bhat,
            xi=xi, stats=self._step_stats(), grad_decay=self._grad_decay, launch=self._launch(), opts=opts, **self._noise_args())
        self._stats_written()
