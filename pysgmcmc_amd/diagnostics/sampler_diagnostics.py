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

__all__ = ["ChainMoments", "cross_chain_rhat", "RhatExchange", "RhatSummary", "gelman_rubin_from_chains", "ess_across_ranks", "effective_n",
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


class RhatExchange(object):
    """Non-blocking form of :func:`cross_chain_rhat`: ``start`` packs a snapshot of the moments and
    issues the all-reduce asynchronously on RCCL's own stream, sampling continues, ``finish`` waits
    (stream-level) for the collective and computes R-hat. The snapshot buffer is private, so the chain may
    keep updating its moments in between (overlaps the only collective of the path with compute).

    ``finish()`` never synchronises with the host: the R-hat summary stays in device memory
    (``exchange.summary``, a :class:`RhatSummary`) and is read when the caller asks for it.
    ``finish(with_summary=True)`` reads it immediately (a ``.cpu()`` sync; end-of-run reporting)."""

    def __init__(self, n, device, group=None, dtype=torch.float32):
        self.n = int(n)
        self.group = group
        self.pack = torch.empty(3 * self.n, dtype=dtype, device=device)
        self.rhat = torch.empty(self.n, dtype=dtype, device=device)
        self.summary = RhatSummary(self.n, device)
        self.exchanges = 0
        self._work = None
        self._count = 0

    @property
    def pending(self):
        return self._work is not None

    def start(self, moments):
        dist = _dist()
        if dist is None or dist.get_world_size(self.group) < 2:
            raise RuntimeError("RhatExchange needs an initialised process group with >= 2 chains")
        assert not self.pending, "finish() the previous exchange first"
        kernels.rhat_pack(moments.mean, moments.m2, moments.count, self.pack)
        self._count = moments.count
        self._work = dist.all_reduce(self.pack, group=self.group, async_op=True)

    def finish(self, with_summary=False):
        dist = _dist()
        self._work.wait()                     # stream-level wait: the current stream now depends on the collective
        self._work = None
        kernels.rhat_finish(self.pack, self.n, dist.get_world_size(self.group), self._count, self.rhat,
                            self.summary.out4, self.summary.workspace)
        self.exchanges += 1
        return self.rhat, (self.summary.as_dict() if with_summary else None)


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
