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

__version__ = "0.1.0"

# what the process was started with, recorded before prefer_plain_graph_launch() can write the variable itself
_PLAIN_LAUNCH_EXPORTED_AT_IMPORT = _os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") == "0"
_PLAIN_LAUNCH_SET_IN_TIME = False


def prefer_plain_graph_launch():
    """Ask the HIP runtime for its plain hipGraph launch path instead of the pre-recorded AQL packets ("packet capture", the default
    in ROCm 7): ``DEBUG_CLR_GRAPH_PACKET_CAPTURE=0``. For chains whose step is DEVICE-bound -- the hipGraph stepping modes of
    ``samplers/base_classes.py`` on models of millions of parameters -- it takes 3.5 us of idle device time off every graph launch on
    MI355X (10 M parameters: 181.0 -> 177.5 us per step; 49.8 M: 778 -> 759; profiles/HISTORY.md). It costs host time per launch
    (45 -> 65-87 us per step there), so chains whose step is HOST-bound lose: the 3 x 50 BNN of BASELINE configs[1] replays its
    graph at 9.2 k instead of 14.9 k steps/s. Hence a call, not a default. The runtime reads the variable when it initialises (its
    first HIP call, not ``import torch``): call this before anything touches the GPU. Returns False when that is already too late (HIP initialised, or a
    profiler's tool library preloaded -- unless the variable was already exported by the caller's environment)."""
    global _PLAIN_LAUNCH_SET_IN_TIME
    import torch
    _os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
    if _PLAIN_LAUNCH_EXPORTED_AT_IMPORT or _PLAIN_LAUNCH_SET_IN_TIME:
        return True               # the caller's environment had it, or an earlier call here set it before the runtime came up
    # a profiler's preloaded tool library (rocprofv3) initialises the HIP runtime before Python runs: too late, export it outside
    preloaded = ("rocprof" in _os.environ.get("LD_PRELOAD", "") or "ROCP_TOOL_LIBRARIES" in _os.environ
                 or "HSA_TOOLS_LIB" in _os.environ)
    # (a call that came too late leaves the variable set all the same; later calls must not read that as "exported by the caller")
    _PLAIN_LAUNCH_SET_IN_TIME = not torch.cuda.is_initialized() and not preloaded
    return _PLAIN_LAUNCH_SET_IN_TIME


def configure_for_device_bound_chains(gemm_tuning=True, plain_graph_launch=True, tuning_ms=30, tuning_iters=20):
    """ONE switch for chains whose step is DEVICE-bound -- hipGraph stepping (the default of the samplers' ``use_hip_graph`` and of
    ``BayesianNeuralNetwork.train``) on models of millions of parameters. It applies what ``bench.py`` runs its BNN workloads
    with, so a chain built through the public API after this call steps at the rate the benchmark line reports as ``value``.
    Without it a chain gets ``value_product_defaults`` of the same line: 0.98-1.0 of ``value`` at 10 M parameters, 0.98 at 49.8 M
    (round 6, ``profiles/r06_bench_*.json``; two processes on one box differ by +-4 % on their own) -- since round 6 ``BNNCost``
    picks the GEMM solutions of a device-bound plan by itself (``auto_gemm_tuning``), so all this call still adds is the graph
    launch path:

    * ``gemm_tuning``: PyTorch's TunableOp stays in tuning mode for every GEMM shape the process meets (not only the plan's
      evaluation that ``BNNCost`` tunes on its own; ``models.bayesian_neural_network.enable_gemm_tuning``; process-wide PyTorch
      setting, same fp32 arithmetic);
    * ``plain_graph_launch``: :func:`prefer_plain_graph_launch` -- only effective BEFORE the first HIP call of the process
      (~2 % of the step at 10 M and at 49.8 M parameters).

    Host-bound chains (the 3 x 50 BNN of the reference's tests) lose with the second one: do not call this for them.
    Returns what took effect: ``{"gemm_tuning": bool, "plain_graph_launch": bool}``."""
    took = {"gemm_tuning": False, "plain_graph_launch": False}
    if plain_graph_launch:
        took["plain_graph_launch"] = bool(prefer_plain_graph_launch())
    if gemm_tuning:
        from pysgmcmc_amd.models.bayesian_neural_network import enable_gemm_tuning
        try:
            enable_gemm_tuning(True, max_duration_ms=int(tuning_ms), max_iterations=int(tuning_iters))
            took["gemm_tuning"] = True
        except Exception as exc:                              # tuning is an optimisation, never a requirement
            import logging
            logging.warning("pysgmcmc_amd: GEMM tuning unavailable (%s); the BLAS heuristics stay", exc)
    return took


def runtime_env():
    """The HIP-runtime settings this package asked for, as the process has them now (see above)."""
    return {k: _os.environ.get(k) for k in ("DEBUG_CLR_GRAPH_PACKET_CAPTURE",)}
