"""Samplers (same exports as ``pysgmcmc/samplers/__init__.py:1-12``, minus SVGD,
which is not part of the SG-MCMC update path -- see DESIGN.md "Out of scope")."""
from pysgmcmc_amd.samplers.sghmc import SGHMCSampler
from pysgmcmc_amd.samplers.sgld import SGLDSampler
from pysgmcmc_amd.samplers.relativistic_sghmc import RelativisticSGHMCSampler

__all__ = (
    "SGHMCSampler",
    "SGLDSampler",
    "RelativisticSGHMCSampler",
)
