"""Models driven by the SG-MCMC update path (same export as ``pysgmcmc/models/__init__.py``)."""
from pysgmcmc_amd.models.bayesian_neural_network import BayesianNeuralNetwork

__all__ = ("BayesianNeuralNetwork",)
