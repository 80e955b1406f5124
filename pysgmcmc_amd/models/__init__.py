"""Models driven by the SG-MCMC update path. Exports what ``pysgmcmc/models/__init__.py:2-13`` exports except ``BaseModel`` (the
abstract model interface of ``models/base_model.py`` is outside the sampler path; its two normalisation
helpers live in ``bayesian_neural_network``)."""
from pysgmcmc_amd.models.bayesian_neural_network import (
    BayesianNeuralNetwork,
    log_variance_prior_log_like,
    weight_prior_log_like,
)

__all__ = ("BayesianNeuralNetwork", "log_variance_prior_log_like", "weight_prior_log_like")
