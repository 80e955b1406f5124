"""Many steps of many chains on the reference's built-in toy targets in one kernel launch.

The reference steps a toy chain with one ``session.run`` per sample -- its sampler tests
(``pysgmcmc/tests/samplers/sampler_testing.py:14-59``) and its ESS experiment
(``docs/source/experiments/compute_ess.py:176-246``: 2e6 steps per stepsize) both do. With a hipGraph a toy
``next(sampler)`` still costs ~50 us (a handful of dependent launches). :class:`BuiltinTargetChains` takes ordinary
sampler objects -- built through the public API, so seeds, initial momenta (incl. ``strict_reference_quirks``),
hyper-parameters and burn-in are exactly theirs -- whose ``cost_fun`` is one of the targets of
``pysgmcmc_amd.diagnostics.objective_functions`` and advances ALL of them ``n_steps`` steps in ONE launch of
``sgmcmc_toy_chains_*`` (one lane per chain, state in registers, analytic gradient, the update operators and Philox
stream of K1-K3). Afterwards every sampler is where ``n_steps`` calls of ``next(sampler)`` would have left it (up to the
rounding of the analytic vs. autograd gradient) and can be stepped normally again.
"""
import torch

from pysgmcmc_amd import kernels
from pysgmcmc_amd.diagnostics import objective_functions as targets
from pysgmcmc_amd.stepsize_schedules import ConstantStepsizeSchedule

__all__ = ("BuiltinTargetChains", "builtin_target_of")

_THIRD = 1.0 / 3.0
# log-likelihood function -> (target id of sgmcmc_toy_chains_*, its parameters, dimension)
_TARGETS = {
    targets.gmm1_log_likelihood: (0, [-5, 0, 5, 1.0, 1.0, 1.0, _THIRD, _THIRD, _THIRD], 1),
    targets.gmm2_log_likelihood: (0, [-5, 0, 5, 1.0 / 0.5, 0.5, 1.0 / 0.5, _THIRD, _THIRD, _THIRD], 1),
    targets.gmm3_log_likelihood: (0, [-5, 0, 5, 1.0 / 0.3, 0.3, 1.0 / 0.3, _THIRD, _THIRD, _THIRD], 1),
    targets.banana_log_likelihood: (1, [], 2),
    targets.gmm2d_log_likelihood: (2, [-5.0, 0.0, 0.0, 0.0, 5.0, 0.0], 2),
}


def builtin_target_of(cost_fun):
    """``(target id, parameters, dim)`` of a sampler cost function that is ``to_negative_log_likelihood(f)`` of a
    built-in target ``f``; raises ``ValueError`` otherwise."""
    inner = getattr(cost_fun, "__wrapped__", None)
    if inner in _TARGETS:
        return _TARGETS[inner]
    raise ValueError("cost function %r is not the negative log likelihood of a built-in toy target (%s)" % (
        getattr(cost_fun, "__name__", cost_fun), ", ".join(sorted(f.__name__ for f in _TARGETS))))


class BuiltinTargetChains(object):
    """``BuiltinTargetChains(samplers).run(n_steps, keep_every)`` -> kept samples ``[n_kept, n_chains, dim]`` (device
    tensor): ``kept[j, c]`` = chain c's parameters after its step ``j * keep_every`` of this run, what
    ``itertools.islice(sampler, 0, n_steps, keep_every)`` yields."""

    _KIND = {"SGHMCSampler": 0, "SGLDSampler": 1, "RelativisticSGHMCSampler": 2}

    def __init__(self, samplers):
        samplers = list(samplers)
        assert samplers, "at least one sampler"
        s0 = samplers[0]
        self.kind = self._KIND.get(type(s0).__name__)
        if self.kind is None:
            raise ValueError("BuiltinTargetChains steps SGHMC, SGLD and relativistic SGHMC samplers, not %s" % type(s0).__name__)
        self.target, self.target_params, self.dim = builtin_target_of(s0.cost_fun)
        for s in samplers:
            if type(s) is not type(s0) or builtin_target_of(s.cost_fun)[0:2] != (self.target, self.target_params):
                raise ValueError("all chains must share the sampler class and the target")
            if s.arena.n != self.dim or s._torch_dtype != s0._torch_dtype or s.device != s0.device:
                raise ValueError("every chain needs %d scalar parameter(s) of one dtype on one device" % self.dim)
            if not isinstance(s.stepsize_schedule, ConstantStepsizeSchedule) or s.batch_generator is not None:
                raise ValueError("BuiltinTargetChains needs a ConstantStepsizeSchedule and no batch generator")
            if self._scalars(s) != self._scalars(s0) or s.n_iterations != s0.n_iterations \
                    or getattr(s, "burn_in_steps", 0) != getattr(s0, "burn_in_steps", 0):
                raise ValueError("all chains must share hyper-parameters, burn-in and step count")
        self.samplers = samplers
        self.device, self.dtype = s0.device, s0._torch_dtype
        # Philox keys are unsigned 64-bit; the device array is int64 (same bits)
        keys = [s._philox_seed - (1 << 64) if s._philox_seed >= (1 << 63) else s._philox_seed for s in samplers]
        self.seeds = torch.tensor(keys, dtype=torch.int64, device=self.device)
        self._rows = {0: ("theta", "V", "tau", "g", "v_hat", "minv"), 1: ("theta", "tau", "g", "v_hat", "minv"),
                      2: ("theta", "p")}[self.kind]

    @staticmethod
    def _scalars(s):
        return s._step_scalars(s.stepsize_schedule.initial_value)

    def run(self, n_steps, keep_every=1, keep=True):
        n_steps, keep_every = int(n_steps), int(keep_every)
        s0 = self.samplers[0]
        m = len(self.samplers)
        state = {name: torch.stack([s.arena.row(name) for s in self.samplers]).contiguous() for name in self._rows}
        kept = None
        if keep:
            kept = torch.empty((n_steps + keep_every - 1) // keep_every, m, self.dim, dtype=self.dtype, device=self.device)
        kernels.toy_chains(self.kind, self.target, self.target_params, state["theta"], state.get("V", state.get("p")),
                           state.get("tau"), state.get("g"), state.get("v_hat"), state.get("minv"), self._scalars(s0),
                           self.seeds, s0.n_iterations, n_steps, getattr(s0, "burn_in_steps", 0), keep_every, kept)
        for c, s in enumerate(self.samplers):
            for name in self._rows:
                s.arena.row(name).copy_(state[name][c])
            s.n_iterations += n_steps
            s._stats_valid = False
        return kept
