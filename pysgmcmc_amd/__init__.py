"""pysgmcmc_amd -- MI355X-native SG-MCMC update path behind the pysgmcmc sampler API.

Drop-in for ``pysgmcmc.samplers`` / ``pysgmcmc.sampling`` (SGLD, SGHMC,
relativistic SGHMC): same class names, constructor keywords, ``next(sampler)``
iterator protocol. The per-parameter update runs as one fused HIP kernel per
step (``csrc/sgmcmc_kernels.hip``) reached through the C ABI in
``include/sgmcmc_hip.h``; there is no CPU fallback.

    from pysgmcmc_amd.samplers import SGHMCSampler, SGLDSampler, RelativisticSGHMCSampler
    from pysgmcmc_amd.sampling import Sampler
    from pysgmcmc_amd.models import BayesianNeuralNetwork
"""
__version__ = "0.1.0"
