"""Many independent SGHMC chains of one small BNN advanced together by the fused-step kernel.

The reference runs chains one after the other, each in a fresh TF graph
(``pysgmcmc/diagnostics/sample_chains.py:369-382``). ``sgmcmc_bnn_fused_sghmc_steps_*`` runs one
workgroup per chain, so up to one chain per CU (256 on MI355X) advance in the time of one: the chains'
states are re-homed back to back in ONE allocation (chain ``c`` at ``+ c * chain_stride`` in every state
row) and every launch covers ``n_steps`` steps of all of them. Each chain stays a normal
:class:`~pysgmcmc_amd.samplers.sghmc.SGHMCSampler` (``next()``, ``minv``, ``state_dict`` ... keep working on
the shared memory); chain ``c`` uses Philox seed ``seed_0 + c`` and its own window stream.
"""
import numpy as np
import torch

from pysgmcmc_amd.samplers.sghmc import SGHMCSampler

__all__ = ("FusedBNNChains",)


class FusedBNNChains(object):
    """Group of SGHMC (or of SGLD) chains that fit the fused small-model kernel.

    Parameters
    ----------
    samplers : list of SGHMCSampler (or list of SGLDSampler)
        Chains over the SAME dataset and network shape, built with ``seed = s, s + 1, s + 2, ...``, equal
        hyper-parameters and ``fused_bnn_available()``; all at the same iteration.
    """

    def __init__(self, samplers):
        samplers = list(samplers)
        assert samplers, "FusedBNNChains needs at least one chain"
        first = samplers[0]
        for c, s in enumerate(samplers):
            if type(s) is not type(first) or not hasattr(s, "_fused_bnn_launch") or not s.fused_bnn_available():
                raise ValueError("chain %d does not fit the fused small-model kernel" % c)
            same = (s._bnn_layer_sizes() == first._bnn_layer_sizes()
                    and s._torch_dtype == first._torch_dtype and s.device == first.device
                    and s.batch_generator.x_dev.data_ptr() == first.batch_generator.x_dev.data_ptr()
                    and s.batch_generator.y_dev.data_ptr() == first.batch_generator.y_dev.data_ptr()
                    and s.batch_generator.batch_size == first.batch_generator.batch_size
                    and s.n_iterations == first.n_iterations and s.burn_in_steps == first.burn_in_steps
                    and s.scale_grad == first.scale_grad
                    and getattr(s, "mdecay", None) == getattr(first, "mdecay", None)
                    and getattr(s, "A", None) == getattr(first, "A", None)
                    and all(getattr(s.cost_fun, k) == getattr(first.cost_fun, k)
                            for k in ("batch_size", "n_examples", "wdecay", "prior_mean", "prior_var")))
            if not same:
                raise ValueError("chain %d differs from chain 0 in data, network or hyper-parameters" % c)
            if s._philox_seed != ((first._philox_seed + c) & 0xFFFFFFFFFFFFFFFF):
                raise ValueError("chain seeds must be consecutive (seed_0 + chain index); chain %d is not" % c)
        self.samplers = samplers
        self.n_chains = len(samplers)
        self.chain_stride = int(first.arena.storage.numel())
        # one allocation, the chains' arenas back to back
        self.storage = torch.empty(self.n_chains * self.chain_stride, dtype=first._torch_dtype, device=first.device)
        for c, s in enumerate(samplers):
            s._rebind_arena(self.storage[c * self.chain_stride:(c + 1) * self.chain_stride])

    @property
    def n_iterations(self):
        return self.samplers[0].n_iterations

    def theta(self):
        """``[n_chains, n_params]`` view of every chain's current parameters (no copy)."""
        a = self.samplers[0].arena
        return torch.as_strided(self.storage, (self.n_chains, a.n), (self.chain_stride, 1),
                                a.row("theta").storage_offset() - self.storage.storage_offset())

    def steps(self, n_steps):
        """Advance every chain by ``n_steps`` steps in one launch; returns the ``[n_chains, n_steps]`` costs
        (cost at the parameters before each step). Needs a stepsize that is constant over the chunk."""
        n_steps = int(n_steps)
        first = self.samplers[0]
        if any(s.n_iterations != first.n_iterations for s in self.samplers):
            raise ValueError("FusedBNNChains.steps: the chains are no longer at the same iteration "
                             "(a member was stepped on its own)")
        eps = None
        for s in self.samplers:
            e = [next(s.stepsize_schedule) for _ in range(n_steps)]
            if any(v != e[0] for v in e) or (eps is not None and e[0] != eps):
                raise ValueError("FusedBNNChains.steps needs one constant stepsize for all chains over the chunk")
            eps = e[0]
            s.epsilon = eps
        gen, cost, a = first.batch_generator, first.cost_fun, first.arena
        starts = np.stack([s.batch_generator.next_starts(n_steps) for s in self.samplers]).astype(np.int32)
        starts = torch.as_tensor(starts).to(first.device).reshape(-1)
        costs = torch.empty(self.n_chains * n_steps, dtype=first._torch_dtype, device=first.device)
        rows = [a.row(k) for k in first._FUSED_ROWS]
        # chain 0's rows are the bases; the kernel adds chain * chain_stride. Hand it views that span all chains.
        span = (self.n_chains - 1) * self.chain_stride + a.n
        bases = [torch.as_strided(self.storage, (span,), (1,), r.storage_offset() - self.storage.storage_offset())
                 for r in rows]
        first._fused_bnn_launch(starts, costs, eps, n_steps, n_chains=self.n_chains, chain_stride=self.chain_stride,
                                bases=bases)
        costs = costs.view(self.n_chains, n_steps)
        for c, s in enumerate(self.samplers):
            s.n_iterations += n_steps
            s._stats_valid = False
            s._grad_decay = float(cost.wdecay / ((a.n + 3e-16) * cost.n_examples))
            s.cost = costs[c, -1]
        return costs

    def collect(self, n_samples, every=100):
        """``n_samples`` thinned snapshots of all chains: a ``[n_chains, n_samples, n_params]`` device tensor (one
        launch of ``every`` steps per snapshot), ready for
        ``diagnostics.sampler_diagnostics.gelman_rubin_from_chains`` / ``effective_n``."""
        a = self.samplers[0].arena
        out = torch.empty(self.n_chains, int(n_samples), a.n, dtype=self.storage.dtype, device=self.storage.device)
        for k in range(int(n_samples)):
            self.steps(every)
            out[:, k].copy_(self.theta())
        return out

    @classmethod
    def for_dataset(cls, X, y, n_chains, hidden=(50, 50, 50), batch_size=20, seed=0, dtype=torch.float32,
                    device="cuda:0", stepsize=0.01, burn_in_steps=1000, mdecay=0.05, init_seed=None):
        """``n_chains`` chains of the reference's default BNN set-up
        (``pysgmcmc/models/bayesian_neural_network.py:151-155,451-457``: ``scale_grad = N``) over one dataset."""
        from pysgmcmc_amd.data_batches import Placeholder, generate_batches
        from pysgmcmc_amd.models.bayesian_neural_network import BNNCost, init_mlp_params
        from pysgmcmc_amd.stepsize_schedules import ConstantStepsizeSchedule
        X = np.asarray(X)
        X = X.reshape(X.shape[0], -1)
        xp, yp = Placeholder(dtype=dtype, device=device), Placeholder(dtype=dtype, device=device)
        shared = generate_batches(X, y, xp, yp, batch_size, seed=seed)       # ONE resident copy of the dataset
        chains = []
        for c in range(int(n_chains)):
            gen = type(shared)(shared.x_dev, shared.y_dev, xp, yp, shared.batch_size,
                               np.random.RandomState(seed + c))
            params = init_mlp_params(X.shape[1], hidden=hidden, seed=(seed if init_seed is None else init_seed) + c,
                                     dtype=dtype, device=device)
            chains.append(SGHMCSampler(
                params=params, cost_fun=BNNCost(xp, yp, batch_size=shared.batch_size, n_examples=X.shape[0]),
                batch_generator=gen, stepsize_schedule=ConstantStepsizeSchedule(stepsize),
                burn_in_steps=burn_in_steps, mdecay=mdecay, scale_grad=float(X.shape[0]),
                session=device, dtype=dtype, seed=seed + c))
        return cls(chains)
