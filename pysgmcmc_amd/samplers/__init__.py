"""Samplers (same exports as ``pysgmcmc/samplers/__init__.py:1-12``)."""
from pysgmcmc_amd.samplers.sghmc import SGHMCSampler
from pysgmcmc_amd.samplers.sgld import SGLDSampler
from pysgmcmc_amd.samplers.relativistic_sghmc import RelativisticSGHMCSampler
from pysgmcmc_amd.samplers.svgd import SVGDSampler
from pysgmcmc_amd.samplers.concurrent_chains import ConcurrentChains

__all__ = (
    "SGHMCSampler",
    "SGLDSampler",
    "RelativisticSGHMCSampler",
    "SVGDSampler",
    "ConcurrentChains",
)
