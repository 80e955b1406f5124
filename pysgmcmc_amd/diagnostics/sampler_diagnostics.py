"""Cross-chain convergence diagnostics on the GPU: Gelman-Rubin R-hat and effective
sample size (the function of ``pysgmcmc/diagnostics/sampler_diagnostics.py:47-194``,
which delegates to the un-vendored ``pymc3.diagnostics``; formulas restated from
pymc3 3.1 -- parity unpinned).

The reference runs chains one after another in fresh TF graphs
(``pysgmcmc/diagnostics/sample_chains.py:369-382``). Here chains are independent
processes, one per GPU, and the ONLY communication on the whole path is the
R-hat exchange below:

  per chain  : Welford running mean / M2 of theta          (K4, 20 B/param/update)
  exchange   : pack [mean, mean^2, var] -> ONE all-reduce(SUM) of 3P floats over
               RCCL/xGMI -> every rank finishes R-hat locally (+ K6 summary)
  ESS        : needs lagged history, so it is computed on thinned, low-dimensional
               traces (cost + a few coordinates): all_gather of (n_kept x K) floats.

``torch.distributed`` is the transport (backend "nccl" = RCCL on ROCm, "gloo" in
the CPU tests); with no process group the functions work on a single chain or on
explicitly passed per-chain moments.
"""
import math

import numpy as np
import torch

from pysgmcmc_amd import kernels

__all__ = ["ChainMoments", "cross_chain_rhat", "RhatExchange", "RhatSummary", "ShardedRhatSummary", "gelman_rubin_from_chains", "ess_across_ranks", "effective_n",
           "effective_sample_sizes", "gelman_rubin"]


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


class ChainMoments(object):
    """Running per-parameter mean and M2 of one chain (Welford, kernel K4)."""

    def __init__(self, n, device, dtype=torch.float32):
        self.n = int(n)
        self.mean = torch.zeros(self.n, dtype=dtype, device=device)
        self.m2 = torch.zeros(self.n, dtype=dtype, device=device)
        self.count = 0

    def update(self, theta_flat):
        self.count += 1
        kernels.moments_update(theta_flat, self.mean, self.m2, self.count)

    def reset(self):
        self.count = 0
        self.mean.zero_()
        self.m2.zero_()

    def variance(self):
        return self.m2 / max(self.count - 1, 1)


class RhatSummary(object):
    """Device-resident K6 summary {sum, sum^2, min, max} of an R-hat vector. Reading it (``mean``, ``max``,
    ``as_dict``) is the only host synchronisation; the exchange itself never waits for the host."""

    def __init__(self, n, device):
        self.n = int(n)
        self.out4 = torch.zeros(4, dtype=torch.float64, device=device)
        self.workspace = kernels.summary_workspace(device)

    def as_dict(self):
        s = self.out4.cpu().numpy()
        return {"mean": float(s[0] / self.n), "max": float(s[3])}

    @property
    def mean(self):
        return self.as_dict()["mean"]

    @property
    def max(self):
        return self.as_dict()["max"]


def cross_chain_rhat(moments, group=None, pack=None, rhat=None, with_summary=True):
    """R-hat of every parameter across the chains of the process group.

    One all-reduce of ``3 * n`` elements in the moments' dtype. All ranks must call with the same
    ``moments.count``. Returns ``(rhat, summary)`` with ``summary = {"mean", "max"}``
    (python floats; forces a sync) or ``None`` when ``with_summary`` is False.
    """
    dist = _dist()
    if dist is None:
        raise RuntimeError("cross_chain_rhat needs an initialised torch.distributed process group (>= 2 chains)")
    world = dist.get_world_size(group)
    if world < 2:
        raise RuntimeError("R-hat needs at least 2 chains")
    n = moments.n
    dt, dev = moments.mean.dtype, moments.mean.device
    if pack is None:
        pack = torch.empty(3 * n, dtype=dt, device=dev)
    if rhat is None:
        rhat = torch.empty(n, dtype=dt, device=dev)
    kernels.rhat_pack(moments.mean, moments.m2, moments.count, pack)
    dist.all_reduce(pack, group=group)
    if not with_summary:
        kernels.rhat_finish(pack, n, world, moments.count, rhat)
        return rhat, None
    summ = RhatSummary(n, dev)
    kernels.rhat_finish(pack, n, world, moments.count, rhat, summ.out4, summ.workspace)
    return rhat, summ.as_dict()


class _NoCollective(object):
    """Stands in for the async work handle when the chains are all local (nothing to wait for)."""

    def wait(self):
        return True


class RhatExchange(object):
    """Non-blocking form of :func:`cross_chain_rhat`: ``start`` packs a snapshot of the moments and
    issues the collective asynchronously on RCCL's own stream, sampling continues, ``finish`` waits
    (stream-level) for it and computes R-hat. The snapshot buffer is private, so the chain may
    keep updating its moments in between (overlaps the only collective of the path with compute).

    ``mode``:
      * ``"allreduce"``: ONE all-reduce(SUM) of ``3 n`` elements; every rank finishes all ``n`` parameters and holds
        the full R-hat vector (``exchange.rhat``). Per rank ``6 n (m-1)/m`` elements cross xGMI.
      * ``"reduce_scatter"`` (reduce-scatter + all-gather): the pack is laid out in ``m`` parameter
        shards, ONE reduce-scatter(SUM) leaves on rank ``r`` the summed rows of shard ``r`` (``3 n (m-1)/m`` elements
        per rank: half the traffic), every rank finishes only ``n/m`` parameters (``exchange.rhat`` = its shard), and the
        K6 summaries of the shards are combined with an all-gather of 4 doubles. ``gather()`` all-gathers the R-hat
        shards into the full vector when it is wanted (then ``4 n (m-1)/m`` in total, still 2/3 of the all-reduce).

    ``finish()`` never synchronises with the host: the R-hat summary stays in device memory
    (``exchange.summary``) and is read when the caller asks for it.
    ``finish(with_summary=True)`` reads it immediately (a ``.cpu()`` sync; end-of-run reporting).

    Several chains per rank (chains that share a GPU, ``samplers.ConcurrentChains``): ``start([moments_a, moments_b, ...])``
    adds the local chains' packs before the collective -- the pack is a sum over chains -- and R-hat is over
    ``world x len(list)`` chains (every rank must pass equally many, with equal counts). With ``mode="allreduce"`` and no
    process group (or one rank) the same call gives the R-hat of the local chains alone, no collective at all. The caller
    orders the chains' streams before ``start`` (``ConcurrentChains.join()``)."""

    def __init__(self, n, device, group=None, dtype=torch.float32, mode="allreduce"):
        assert mode in ("allreduce", "reduce_scatter")
        self.n = int(n)
        self.group = group
        self.mode = mode
        self.exchanges = 0
        self._work = None
        self._count = 0
        self._local_chains = 1
        self._pack_more = None
        if mode == "allreduce":
            self.n_shards, self.shard_len, self.n_valid, self.rank = 1, self.n, self.n, 0
            self.pack = torch.empty(3 * self.n, dtype=dtype, device=device)
            self.shard_sum = self.pack
            self.rhat = torch.empty(self.n, dtype=dtype, device=device)
            self.summary = RhatSummary(self.n, device)
            return
        dist = _dist()
        if dist is None or dist.get_world_size(group) < 2:
            raise RuntimeError("RhatExchange(mode='reduce_scatter') needs an initialised process group with >= 2 chains")
        m, self.rank = dist.get_world_size(group), dist.get_rank(group)
        L = (self.n + m - 1) // m
        L4 = (L + 3) // 4 * 4                                   # 16-byte shard boundaries when no shard becomes empty
        if (m - 1) * L4 < self.n:
            L = L4
        if (m - 1) * L >= self.n:
            raise ValueError("RhatExchange: fewer parameters (%d) than chains (%d)" % (self.n, m))
        self.n_shards, self.shard_len = m, L
        self.n_valid = max(0, min(L, self.n - self.rank * L))
        self.pack = torch.empty(3 * m * L, dtype=dtype, device=device)
        self.shard_sum = torch.empty(3 * L, dtype=dtype, device=device)
        self.rhat = torch.empty(L, dtype=dtype, device=device)     # this rank's shard of R-hat (first n_valid entries)
        self.summary = ShardedRhatSummary(self.n, m, device, group)
        # gloo (CPU tests, single-GPU rehearsals) has no reduce-scatter: all-reduce the pack, keep the own chunk
        self._native_rs = str(dist.get_backend(group)).lower() == "nccl"

    @property
    def pending(self):
        return self._work is not None

    def start(self, moments):
        local = list(moments) if isinstance(moments, (list, tuple)) else [moments]
        if not local:
            raise ValueError("RhatExchange.start: no moments")
        for mom in local:
            if mom.mean.dtype != self.pack.dtype:
                # the pack kernel is chosen from the moments' dtype: an f64 pack into this f32 buffer would overrun it
                raise TypeError("RhatExchange was built for %s but the chain's moments are %s: pass dtype=%s" % (
                    self.pack.dtype, mom.mean.dtype, mom.mean.dtype))
        if any(mom.count != local[0].count for mom in local):
            raise ValueError("RhatExchange.start: the local chains' moments hold different sample counts")
        dist = _dist()
        world = 1 if dist is None else dist.get_world_size(self.group)
        if world * len(local) < 2 or (world < 2 and self.mode != "allreduce"):
            raise RuntimeError("RhatExchange needs >= 2 chains: an initialised process group with >= 2 ranks, or (mode="
                               "'allreduce') several local chains")
        assert not self.pending, "finish() the previous exchange first"
        kernels.rhat_pack(local[0].mean, local[0].m2, local[0].count, self.pack, self.n_shards, self.shard_len)
        for mom in local[1:]:                 # the pack is additive over chains: local chains are summed before the collective
            if self._pack_more is None:
                self._pack_more = torch.empty_like(self.pack)
            kernels.rhat_pack(mom.mean, mom.m2, mom.count, self._pack_more, self.n_shards, self.shard_len)
            self.pack.add_(self._pack_more)
        self._count = local[0].count
        self._local_chains = len(local)
        if world < 2:
            self._work = _NoCollective()
        elif self.mode == "reduce_scatter" and self._native_rs:
            self._work = dist.reduce_scatter_tensor(self.shard_sum, self.pack, group=self.group, async_op=True)
        else:
            self._work = dist.all_reduce(self.pack, group=self.group, async_op=True)

    def finish(self, with_summary=False):
        dist = _dist()
        self._work.wait()                     # stream-level wait: the current stream now depends on the collective
        local_only = isinstance(self._work, _NoCollective)
        self._work = None
        m = (1 if local_only else dist.get_world_size(self.group)) * self._local_chains
        if self.mode == "reduce_scatter" and not self._native_rs:
            L3 = 3 * self.shard_len
            self.shard_sum.copy_(self.pack[self.rank * L3:(self.rank + 1) * L3])
        kernels.rhat_finish(self.shard_sum, self.n_valid, m, self._count, self.rhat,
                            self.summary.out4, self.summary.workspace, ld=self.shard_len)
        if self.mode == "reduce_scatter":
            self.summary.combine()            # all-gather of 4 doubles per rank, still no host synchronisation
        self.exchanges += 1
        return self.rhat, (self.summary.as_dict() if with_summary else None)

    def gather(self):
        """The full R-hat vector on every rank (``mode="reduce_scatter"``: one all-gather of the shards)."""
        if self.mode == "allreduce":
            return self.rhat
        dist = _dist()
        m = dist.get_world_size(self.group)
        full = torch.empty(m * self.shard_len, dtype=self.rhat.dtype, device=self.rhat.device)
        if self._native_rs:
            dist.all_gather_into_tensor(full, self.rhat, group=self.group)
        else:
            parts = [torch.empty_like(self.rhat) for _ in range(m)]
            dist.all_gather(parts, self.rhat, group=self.group)
            full = torch.cat(parts)
        return full[:self.n]


class ShardedRhatSummary(RhatSummary):
    """Summary of an R-hat vector that is spread over the ranks: every rank reduces its shard (K6), ``combine``
    all-gathers the 4 doubles of each rank; ``as_dict`` (the host read) merges them."""

    def __init__(self, n, m, device, group=None):
        super().__init__(n, device)
        self.group = group
        self.all4 = torch.zeros(m, 4, dtype=torch.float64, device=device)

    def combine(self):
        dist = _dist()
        parts = list(self.all4.unbind(0))
        dist.all_gather(parts, self.out4, group=self.group)

    def as_dict(self):
        a = self.all4.cpu().numpy()
        return {"mean": float(a[:, 0].sum() / self.n), "max": float(a[:, 3].max())}


def gelman_rubin_from_chains(chains):
    """R-hat from explicit chains ``(m, n_samples, P)`` on any device (torch ops; small inputs).

    B = n var_c(mean_c), W = mean_c(var_c), Vhat = W (n-1)/n + B/n, Rhat = sqrt(Vhat / W)."""
    x = torch.as_tensor(chains).double()
    m, n = x.shape[0], x.shape[1]
    B = n * x.mean(dim=1).var(dim=0, unbiased=True)
    W = x.var(dim=1, unbiased=True).mean(dim=0)
    return torch.sqrt((W * (n - 1) / n + B / n) / W)


def effective_n(traces):
    """Effective sample size of a scalar quantity from ``(m, n)`` traces (variogram estimate:
    n_eff = m n / (1 + 2 sum_t rho_t), truncated at the first odd t with rho_{t-1} + rho_t < 0;
    ``pysgmcmc/diagnostics/sampler_diagnostics.py:76-82``). All lags are evaluated at once
    with an FFT autocovariance on the traces' device."""
    x = torch.as_tensor(traces).double()
    m, n = x.shape
    chain_mean = x.mean(dim=1, keepdim=True)
    chain_var = x.var(dim=1, unbiased=True)
    B = n * chain_mean.squeeze(1).var(unbiased=True) if m > 1 else torch.zeros((), dtype=x.dtype, device=x.device)
    W = chain_var.mean()
    Vhat = W * (n - 1) / n + B / n
    # variogram V_t = mean_c 1/(n-t) sum_i (x_{i+t} - x_i)^2, from prefix sums and the FFT autocorrelation
    xc = x - chain_mean
    nfft = 1 << int(math.ceil(math.log2(2 * n)))
    f = torch.fft.rfft(xc, n=nfft, dim=1)
    acov = torch.fft.irfft(f * f.conj(), n=nfft, dim=1)[:, :n]          # sum_i xc_i xc_{i+t}
    sq = xc * xc
    csum = torch.cumsum(sq, dim=1)
    total = csum[:, -1:]
    lags = torch.arange(n, device=x.device)
    head = torch.cat([torch.zeros(m, 1, dtype=x.dtype, device=x.device), csum[:, :-1]], dim=1)
    # sum_{i<n-t} xc_i^2 = csum[n-t-1];  sum_{i>=t} xc_i^2 = total - csum[t-1]
    s_first = csum.flip(1)                                             # index t -> csum[n-1-t]
    s_last = total - head                                              # index t -> total - csum[t-1]
    vario = ((s_first + s_last - 2.0 * acov) / (n - lags).clamp_min(1)).mean(dim=0)
    rho = (1.0 - vario / (2.0 * Vhat)).cpu()
    rho[0] = 1.0
    t = 1
    negative = False
    while not negative and t < n:
        if t % 2 == 0:
            negative = bool(rho[t - 1] + rho[t] < 0)
        t += 1
    return int(m * n / (1.0 + 2.0 * float(rho[1:t].sum())))


def ess_across_ranks(local_traces, group=None):
    """ESS of each of K scalar summaries. ``local_traces``: this chain's ``(n_kept, K)`` thinned
    trace (device tensor). With a process group the traces of all chains are all-gathered
    (``n_kept * K`` floats per rank); returns a list of K ints on every rank."""
    x = torch.as_tensor(local_traces)
    if x.dim() == 1:
        x = x[:, None]
    dist = _dist()
    if dist is not None and dist.get_world_size(group) > 1:
        bufs = [torch.empty_like(x) for _ in range(dist.get_world_size(group))]
        dist.all_gather(bufs, x.contiguous(), group=group)
        allx = torch.stack(bufs)                                       # (m, n, K)
    else:
        allx = x[None]
    return [effective_n(allx[:, :, k]) for k in range(allx.shape[2])]


# ---------------------------------------------------------------------------
# The reference's entry points (pysgmcmc/diagnostics/sampler_diagnostics.py:47-194):
# sequential chains from a sampler factory, one value per parameter dimension.
# ---------------------------------------------------------------------------

def _per_variable(get_sampler, n_chains, samples_per_chain, fun):
    from pysgmcmc_amd.diagnostics.sample_chains import multitrace
    mt = multitrace(get_sampler, n_chains=n_chains, samples_per_chain=samples_per_chain)
    out = {}
    for name in mt.varnames:
        vals = np.stack([np.asarray(v, dtype=np.float64) for v in mt.get_values(name, combine=False)])   # (m, n, ...)
        flat = vals.reshape(vals.shape[0], vals.shape[1], -1)
        res = np.asarray([fun(flat[:, :, k]) for k in range(flat.shape[2])])
        out[name] = res.reshape(vals.shape[2:]) if vals.ndim > 2 else res[0]
    return out


def effective_sample_sizes(get_sampler, n_chains=2, samples_per_chain=100):
    """ESS per parameter dimension: ``{varname: array}`` (signature of ``sampler_diagnostics.py:47``).
    ``get_sampler(session=...)`` returns a fresh (possibly burnt-in) sampler."""
    return _per_variable(get_sampler, n_chains, samples_per_chain, lambda x: effective_n(torch.as_tensor(x)))


def gelman_rubin(get_sampler, n_chains=2, samples_per_chain=100):
    """R-hat per parameter dimension: ``{varname: array}`` (signature of ``sampler_diagnostics.py:118``)."""
    return _per_variable(get_sampler, n_chains, samples_per_chain,
                         lambda x: float(gelman_rubin_from_chains(torch.as_tensor(x)[:, :, None])[0]))
