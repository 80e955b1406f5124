"""Toy target densities used as sampler test inputs (mirror of the sampler-relevant
part of ``pysgmcmc/diagnostics/objective_functions.py``: ``to_negative_log_likelihood``
:8-46, ``banana_log_likelihood`` :49-59, Gaussian mixtures :62-98, ``sinc`` :101-102).
They accept torch tensors (differentiable) or plain numbers/ndarrays.
"""
import functools
import math

import numpy as np
import torch

__all__ = [
    "to_negative_log_likelihood", "banana_log_likelihood",
    "gaussian_mixture_model_log_likelihood", "gmm1_log_likelihood",
    "gmm2_log_likelihood", "gmm3_log_likelihood", "gmm2d_log_likelihood", "sinc",
]


def to_negative_log_likelihood(log_likelihood_function):
    """Decorator: log likelihood -> negative log likelihood, name preserved.

    >>> ll = lambda a, b: math.log(a + b)
    >>> nll = to_negative_log_likelihood(ll)
    >>> nll(4, 5) == -ll(4, 5), nll.__name__ == ll.__name__
    (True, True)
    """
    @functools.wraps(log_likelihood_function)
    def negative_log_likelihood(*args, **kwargs):
        return -log_likelihood_function(*args, **kwargs)
    return negative_log_likelihood


def banana_log_likelihood(x):
    """Banana density of the relativistic Monte Carlo paper.

    >>> float(banana_log_likelihood((0, 10)))
    -0.0
    """
    return -0.5 * (0.01 * x[0] ** 2 + (x[1] + 0.1 * x[0] ** 2 - 10) ** 2)


def gaussian_mixture_model_log_likelihood(x, mu=(-5, 0, 5), var=(1., 1., 1.),
                                          weights=(1. / 3., 1. / 3., 1. / 3.)):
    """1-D Gaussian mixture log density, via logsumexp."""
    assert len(mu) == len(var) == len(weights)
    if isinstance(x, (list, tuple)):
        assert len(x) == 1
        x = x[0]
    if isinstance(x, torch.Tensor):
        terms = [math.log(w) - 0.5 * math.log(2.0 * math.pi * v) - 0.5 * ((x - m) ** 2) / v
                 for m, v, w in zip(mu, var, weights)]
        return torch.logsumexp(torch.stack([t.reshape(()) for t in terms]), dim=0)
    x = float(np.asarray(x).ravel()[0])
    terms = np.array([math.log(w) - 0.5 * math.log(2.0 * math.pi * v) - 0.5 * ((x - m) ** 2) / v
                      for m, v, w in zip(mu, var, weights)])
    mx = terms.max()
    return mx + math.log(np.exp(terms - mx).sum())


def gmm1_log_likelihood(x):
    return gaussian_mixture_model_log_likelihood(x)


def gmm2_log_likelihood(x):
    return gaussian_mixture_model_log_likelihood(x, var=[1. / 0.5, 0.5, 1. / 0.5])


def gmm3_log_likelihood(x):
    return gaussian_mixture_model_log_likelihood(x, var=[1. / 0.3, 0.3, 1. / 0.3])


def gmm2d_log_likelihood(x, centers=((-5., 0.), (0., 0.), (5., 0.))):
    """2-D equal-weight isotropic unit-variance mixture (BASELINE.json configs[0];
    the reference's mixtures are 1-D, this is their 2-D analogue). ``x`` = one
    2-element tensor or a list of two scalars."""
    if isinstance(x, (list, tuple)):
        x = torch.stack([xi.reshape(()) for xi in x]) if len(x) == 2 else x[0]
    x = x.reshape(-1)
    c = torch.as_tensor(centers, dtype=x.dtype, device=x.device)
    sq = ((x[None, :] - c) ** 2).sum(dim=1)
    return torch.logsumexp(-0.5 * sq - math.log(2.0 * math.pi) - math.log(len(centers)), dim=0)


def sinc(x):
    return np.sinc(x * 10 - 5).sum(axis=1)
