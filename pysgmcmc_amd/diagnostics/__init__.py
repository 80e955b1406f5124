"""Diagnostics: toy targets, trace containers and cross-chain statistics (same exports as
``pysgmcmc/diagnostics/__init__.py:1-9``)."""
from pysgmcmc_amd.diagnostics.sample_chains import PYSGMCMCTrace, pymc3_multitrace
from pysgmcmc_amd.diagnostics.sampler_diagnostics import effective_sample_sizes, gelman_rubin

__all__ = ("PYSGMCMCTrace", "pymc3_multitrace", "effective_sample_sizes", "gelman_rubin")
