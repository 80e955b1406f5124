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
import os as _os

# hipGraph replay is how the launch-bound part of a step runs (samplers/base_classes.py). The HIP runtime's graph "packet
# capture" path (pre-recorded AQL packets, on by default in ROCm 7) costs this workload 3.5 us of device time per graph launch on
# MI355X -- 181.0 -> 177.5 us per step of the 10 M-parameter chain, 778 -> 759 at 49.8 M -- so the
# package asks for the plain path unless the process says otherwise. The runtime reads the flag when it initialises (the first
# HIP call, not `import torch`): import this package before touching the GPU, or export the variable yourself.
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

__version__ = "0.1.0"


def runtime_env():
    """The HIP-runtime settings this package asked for, as the process has them now (see above)."""
    return {k: _os.environ.get(k) for k in ("DEBUG_CLR_GRAPH_PACKET_CAPTURE",)}
