"""Chain/trace containers (mirror of ``pysgmcmc/diagnostics/sample_chains.py``; the reference
adapts its chains to ``pymc3.backends.base.MultiTrace``, which is not vendored -- here the same
interface is provided without pymc3).

``PYSGMCMCTrace``: one chain; ``samples[i][j]`` = value of variable ``j`` at iteration ``i``.
``MultiTrace``: the chains of one experiment, ``get_values(varname, combine=False)`` like pymc3's.
"""
import logging
from itertools import islice

import numpy as np

__all__ = ["PYSGMCMCTrace", "MultiTrace", "multitrace", "pymc3_multitrace"]


class PYSGMCMCTrace(object):
    """A single chain of samples with named variables.

    >>> trace = PYSGMCMCTrace(chain_id=0, samples=[[0., 0.], [0.2, -0.2], [0.3, -0.5], [0.1, 0.]], varnames=["x", "y"])
    >>> len(trace), trace.n_vars, trace.varnames
    (4, 2, ['x', 'y'])
    >>> trace.get_values("x").tolist(), trace[1].tolist()
    ([0.0, 0.2, 0.3, 0.1], [0.0, -0.2, -0.5, 0.0])
    >>> trace.get_values("y", burn=1, thin=2).tolist()
    [-0.2, 0.0]
    >>> trace.point(1)
    {'x': 0.2, 'y': -0.2}
    >>> trace.get_values(varname="FANTASYVARNAME")
    Traceback (most recent call last):
      ...
    ValueError: Queried `PYSGMCMCTrace` for values of parameter with name 'FANTASYVARNAME' but the trace does not contain any parameter of that name. Known variable names were: '['x', 'y']'
    """

    def __init__(self, chain_id, samples, varnames=None):
        self.chain = chain_id
        self.samples = samples
        first_sample = self.samples[0]
        try:
            self.n_vars = len(first_sample)
        except TypeError:                       # a single scalar parameter
            self.n_vars = 1
            self.samples = [[s] for s in self.samples]
        assert self.n_vars >= 1, "The first sample needs to have at least one variable."
        if varnames is None:
            logging.warning(
                "Variables in a trace were not named when instantiating "
                "a `pysgmcmc.diagnostics.sample_chain.PYSGMCMCTrace` "
                "from that trace. We will give them anonymous names "
                "by enumerating all target parameter dimensions."
            )
            self.varnames = [str(i) for i in range(self.n_vars)]
        else:
            self.varnames = list(varnames)
        assert len(self.varnames) == self.n_vars

    @classmethod
    def from_sampler(cls, chain_id, sampler, n_samples, keep_every=1, varnames=None):
        """Draw ``n_samples`` from ``sampler``. (``keep_every`` is accepted and, like in the
        reference ``sample_chains.py:166-169``, not applied; use ``get_values(thin=...)``.)"""
        samples = []
        for sample, _ in islice(sampler, n_samples):
            if not isinstance(sample, (list, tuple)):
                sample = [sample]                     # single parameter: base_classes.py:302-304
            samples.append([np.asarray(v.detach().cpu() if hasattr(v, "detach") else v) for v in sample])
        if varnames is None:
            varnames = list(getattr(sampler, "param_names", None) or [str(i) for i in range(len(samples[0]))])
        return PYSGMCMCTrace(chain_id, samples, varnames)

    def __getitem__(self, index):
        if isinstance(index, slice):
            return self._slice(index)
        assert isinstance(index, int)
        assert 0 <= index < len(self.varnames)
        return self.get_values(self.varnames[index])

    def _slice(self, slice_):
        return PYSGMCMCTrace(chain_id=self.chain, samples=self.samples[slice_], varnames=self.varnames)

    def point(self, index):
        sample = self.samples[index]
        return {name: sample[k] for k, name in enumerate(self.varnames)}

    def __len__(self):
        return len(self.samples)

    def get_values(self, varname, burn=0, thin=1):
        if varname not in self.varnames:
            raise ValueError(
                "Queried `PYSGMCMCTrace` for values of parameter with "
                "name '{name}' but the trace does not contain any "
                "parameter of that name. "
                "Known variable names were: '{varnames}'"
                .format(name=varname, varnames=self.varnames)
            )
        k = self.varnames.index(varname)
        return np.asarray([sample[k] for sample in self.samples[burn::thin]])


class MultiTrace(object):
    """The chains of one experiment (the part of pymc3's MultiTrace the diagnostics use)."""

    def __init__(self, traces):
        self._straces = {t.chain: t for t in traces}
        assert len(self._straces) == len(traces), "chain ids must be unique"

    @property
    def nchains(self):
        return len(self._straces)

    @property
    def chains(self):
        return sorted(self._straces)

    @property
    def varnames(self):
        return self._straces[self.chains[0]].varnames

    def __len__(self):
        return len(self._straces[self.chains[0]])

    def get_values(self, varname, burn=0, thin=1, combine=True):
        vals = [self._straces[c].get_values(varname, burn=burn, thin=thin) for c in self.chains]
        return np.concatenate(vals) if combine else vals


def multitrace(get_sampler, n_chains=2, samples_per_chain=100, keep_every=1, parameter_names=None):
    """Run ``n_chains`` fresh chains one after another on this device (the reference's
    ``pymc3_multitrace``, ``sample_chains.py:338-384``; ``get_sampler(session=...)`` is called
    with ``session=None``). For chains in parallel, one per GPU, see ``sampler_diagnostics``."""
    traces = []
    for chain_id in range(n_chains):
        sampler = get_sampler(session=None)
        traces.append(PYSGMCMCTrace.from_sampler(chain_id=chain_id, sampler=sampler, n_samples=samples_per_chain,
                                                 keep_every=keep_every, varnames=parameter_names))
    return MultiTrace(traces)


pymc3_multitrace = multitrace
