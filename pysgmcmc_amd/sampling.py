"""Sampler enum and keyword-reflecting factory (mirror of ``pysgmcmc/sampling.py``).

``Sampler.get_sampler(method, **kwargs)`` resolves the sampler class, validates
``kwargs`` against the constructor signature and fills in defaults. The
``ValueError`` texts are part of the reference's tested behaviour
(``sampling.py:134-171`` doctests) and are reproduced character for character.
"""
from enum import Enum
from inspect import Parameter, signature

__all__ = ["Sampler"]


def _sampler_class(method):
    if method == Sampler.SGHMC:
        from pysgmcmc_amd.samplers.sghmc import SGHMCSampler
        return SGHMCSampler
    if method == Sampler.SGLD:
        from pysgmcmc_amd.samplers.sgld import SGLDSampler
        return SGLDSampler
    if method == Sampler.RelativisticSGHMC:
        from pysgmcmc_amd.samplers.relativistic_sghmc import RelativisticSGHMCSampler
        return RelativisticSGHMCSampler
    if method == Sampler.SVGD:
        from pysgmcmc_amd.samplers.svgd import SVGDSampler
        return SVGDSampler
    raise ValueError(
        "Sampling method {sampler} is supported, but function "
        "'pysgmcmc.sampling.get_sampler' is missing an `import` "
        "statement for the corresponding sampler object. "
        "Please add an import in the appropriate location."
    )


class Sampler(Enum):
    """Enumeration of sampling methods (values as ``sampling.py:8-11``)."""

    SGHMC = "SGHMC"
    RelativisticSGHMC = "RelativisticSGHMC"
    SGLD = "SGLD"
    SVGD = "SVGD"

    @staticmethod
    def is_burn_in_mcmc(sampling_method):
        """True for samplers with a burn-in phase.

        >>> Sampler.is_burn_in_mcmc(Sampler.SGHMC), Sampler.is_burn_in_mcmc(Sampler.SGLD)
        (True, True)
        >>> Sampler.is_burn_in_mcmc(Sampler.RelativisticSGHMC)
        False
        >>> Sampler.is_burn_in_mcmc(0), Sampler.is_burn_in_mcmc("test")
        (False, False)
        """
        return sampling_method in (Sampler.SGHMC, Sampler.SGLD)

    @staticmethod
    def is_supported(sampling_method):
        """True for methods the BNN model accepts (``sampling.py:64``).

        >>> Sampler.is_supported(Sampler.SGHMC)
        True
        >>> Sampler.is_supported(0), Sampler.is_supported("test")
        (False, False)
        """
        return sampling_method in (Sampler.SGHMC, Sampler.SGLD)

    @classmethod
    def get_sampler(cls, sampling_method, **sampler_args):
        """Construct the sampler for ``sampling_method``; keywords not given fall
        back to the constructor defaults.

        >>> import torch
        >>> params = [torch.tensor(0.)]
        >>> cost_fun = lambda params: sum(p.sum() for p in params)
        >>> sampler = Sampler.get_sampler(Sampler.SGHMC, params=params, cost_fun=cost_fun, dtype=torch.float32)
        >>> type(sampler).__name__, sampler.dtype
        ('SGHMCSampler', torch.float32)
        >>> type(Sampler.get_sampler(Sampler.SGLD, params=params, cost_fun=cost_fun)).__name__
        'SGLDSampler'
        >>> Sampler.get_sampler(Sampler.SGHMC, dtype=torch.float32)
        Traceback (most recent call last):
          ...
        ValueError: sampling.Sampler.get_sampler: params was not overwritten as sampler argument in `sampler_args` and does not have any default value in SGHMCSampler.__init__Please pass an explicit value for this parameter.
        >>> Sampler.get_sampler(Sampler.SGHMC, unknown_argument=None, params=params, cost_fun=cost_fun)
        Traceback (most recent call last):
          ...
        ValueError: sampling.Sampler.get_sampler: 'SGHMCSampler' does not take any parameter with name 'unknown_argument' which was specified as argument to this sampler. Please ensure, that you only specify sampler arguments that fit the corresponding sampling method.
        For your choice of sampling method ('Sampler.SGHMC'), supported parameters are:
        -params
        -cost_fun
        -batch_generator
        -stepsize_schedule
        -burn_in_steps
        -mdecay
        -scale_grad
        -session
        -dtype
        -seed
        """
        sampler_cls = _sampler_class(sampling_method)
        accepted = [name for name in signature(sampler_cls.__init__).parameters if name != "self"]
        formal = signature(sampler_cls.__init__).parameters

        for name in sampler_args:
            if name not in formal:
                raise ValueError(
                    "sampling.Sampler.get_sampler: '{sampler_name}' "
                    "does not take any parameter with name '{parameter}' "
                    "which was specified as argument to this sampler. "
                    "Please ensure, that you only specify sampler arguments "
                    "that fit the corresponding sampling method.\n"
                    "For your choice of sampling method ('{sampler}'), supported parameters are:\n"
                    "{valid_parameters}".format(
                        sampler_name=sampler_cls.__name__, sampler=sampling_method, parameter=name,
                        valid_parameters="\n".join("-{}".format(a) for a in accepted)))

        resolved = {}
        for name in accepted:
            if name in sampler_args:
                resolved[name] = sampler_args[name]
            elif formal[name].default is not Parameter.empty:
                resolved[name] = formal[name].default
            else:
                raise ValueError(
                    "sampling.Sampler.get_sampler: "
                    "{param_name} was not overwritten as sampler argument "
                    "in `sampler_args` and does not have any default value "
                    "in {sampler}.__init__"
                    "Please pass an explicit value for this parameter.".format(
                        param_name=name, sampler=sampler_cls.__name__))
        return sampler_cls(**resolved)
