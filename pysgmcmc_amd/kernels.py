"""Thin torch-tensor front end of the C ABI (``include/sgmcmc_hip.h``).

Every function takes flat, contiguous device tensors, passes their raw
pointers to ``libsgmcmc_hip.so`` and launches on torch's *current* HIP stream
(so ``torch.cuda.Event`` timing, stream ordering with the autograd kernels and
``torch.cuda.graph`` capture all see the launch). PyTorch is plumbing here:
device memory and streams only.

No CPU path exists: a CPU tensor raises ``SgmcmcLibraryError``.
"""
import torch

from pysgmcmc_amd._lib import SgmcmcLibraryError, check, lib

__all__ = [
    "sghmc_step", "sgld_step", "rsghmc_step", "philox_normal", "philox_bits",
    "moments_update", "rhat_pack", "rhat_finish", "summary",
    "LaunchConfig", "KernelEvents", "StepOpts", "step_stats_records", "step_scalars", "toy_chains", "set_launch_config", "get_launch_config", "summary_workspace", "counter_add", "StepStats", "bnn_head", "bnn_dense_tanh_backward", "bnn_dense_tanh_backward_fits", "colsum_finish", "tanh_backward", "tanh_backward_colsum", "bnn_last_layer_backward", "bnn_fused_sghmc_steps", "step_stats_finish",
    "bnn_fused_sgld_steps", "window_gather", "tanh_rowdot", "bias_tanh", "bnn_dense_tanh", "bnn_dense_tanh_fits", "bnn_head_last_layer_backward", "svgd_workspace", "svgd_step", "svgd_kernel", "svgd_max_particles",
]

_SFX = {torch.float32: "f32", torch.float64: "f64"}


def _sfx(t):
    try:
        return _SFX[t.dtype]
    except KeyError:
        raise TypeError("pysgmcmc_amd kernels support float32/float64, got %s" % t.dtype)


def _ptr(t, like=None):
    """Raw device pointer of a flat contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise SgmcmcLibraryError(
            "pysgmcmc_amd: tensor lives on %s; the SG-MCMC update path runs only as HIP kernels on an "
            "AMD GPU (no CPU fallback)." % t.device)
    if not t.is_contiguous():
        raise ValueError("pysgmcmc_amd: kernel arguments must be contiguous")
    if like is not None:
        if t.dtype != like.dtype:
            raise TypeError("pysgmcmc_amd: dtype mismatch %s vs %s" % (t.dtype, like.dtype))
        if t.numel() != like.numel():
            raise ValueError("pysgmcmc_amd: length mismatch %d vs %d" % (t.numel(), like.numel()))
        if t.device != like.device:
            raise ValueError("pysgmcmc_amd: device mismatch %s vs %s" % (t.device, like.device))
    return t.data_ptr()


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _ctr(step_dev):
    """Device step counter (int64 tensor with one element) -> pointer, or NULL."""
    if step_dev is None:
        return None
    if step_dev.dtype != torch.int64 or step_dev.numel() != 1:
        raise TypeError("step_dev must be a one-element int64 device tensor")
    return _ptr(step_dev)


class StepStats(object):
    """Device buffers for the statistics a step kernel reduces on the fly:
    ``out`` = float64[4] {sum theta'^2, sum V'^2 (p'^2), sum minv, sum minv^2}."""

    def __init__(self, n, device):
        self.out = torch.zeros(4, dtype=torch.float64, device=device)
        self.workspace = torch.empty(max(lib().sgmcmc_step_stats_workspace_bytes(int(n)), 32),
                                     dtype=torch.uint8, device=device)


def _stats(stats):
    return (None,) if stats is None else (_ptr(stats.workspace),)


def step_stats_finish(stats):
    """K7: reduce the per-block partials the last step kernel left in ``stats.workspace`` into ``stats.out``."""
    with _on(stats.out):
        rc = lib().sgmcmc_step_stats_finish(_ptr(stats.workspace), _ptr(stats.out), _stream(stats.out))
    check(rc, "sgmcmc_step_stats_finish")
    return stats.out


def counter_add(counter, inc=1):
    """counter += inc on the current stream (graph-capturable)."""
    with _on(counter):
        rc = lib().sgmcmc_counter_add_u64(_ctr(counter), int(inc), _stream(counter))
    check(rc, "sgmcmc_counter_add_u64")


def _on(t):
    """Device guard for the launch; refuses CPU tensors (there is no CPU update path)."""
    _ptr(t)
    return torch.cuda.device(t.device)


class LaunchConfig(object):
    """Launch geometry of the streaming kernels (``sgmcmc_launch_t``; performance only, results never depend
    on it). Pass one as ``launch=`` to a kernel call, or hang it on a sampler (``sampler.launch``). ``None``
    everywhere = the library's measured defaults. There is no process-wide setting in the C ABI; the
    module-level default below exists for the sweep tools and is plain Python state of the caller."""

    __slots__ = ("_c", "events")

    def __init__(self, block_threads=0, quads_per_thread=0, max_blocks=0, nontemporal=-1, events=None):
        from pysgmcmc_amd._lib import LaunchStruct
        # events: a KernelEvents pair that receives the kernel's own start/stop timestamps (hipExtLaunchKernel)
        self.events = events
        self._c = LaunchStruct(int(block_threads), int(quads_per_thread), int(max_blocks), int(nontemporal),
                               None if events is None else events.start, None if events is None else events.stop)

    def as_dict(self):
        c = self._c
        return {"block_threads": c.block_threads, "quads_per_thread": c.quads_per_thread,
                "max_blocks": c.max_blocks, "nontemporal": c.nontemporal}



class KernelEvents(object):
    """A pair of HIP events that a launch fills with the KERNEL's own start and stop timestamps (the duration
    rocprofv3 reports for the kernel), via ``LaunchConfig(events=...)``. ``elapsed_us()`` after a synchronise."""

    def __init__(self, device=None):
        """``device``: the device whose launches the events will time (default: the current device)."""
        import ctypes
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
            check(lib().sgmcmc_event_create(ctypes.byref(a)), "sgmcmc_event_create")
            check(lib().sgmcmc_event_create(ctypes.byref(b)), "sgmcmc_event_create")
        self.start, self.stop = a.value, b.value

    def elapsed_us(self):
        import ctypes
        ms = ctypes.c_float()
        check(lib().sgmcmc_event_elapsed_ms(self.start, self.stop, ctypes.byref(ms)), "sgmcmc_event_elapsed_ms")
        return float(ms.value) * 1e3

    def us_until(self, later):
        """Microseconds from the end of this kernel to the end of the kernel ``later`` timed."""
        import ctypes
        ms = ctypes.c_float()
        check(lib().sgmcmc_event_elapsed_ms(self.stop, later.stop, ctypes.byref(ms)), "sgmcmc_event_elapsed_ms")
        return float(ms.value) * 1e3

    def synchronize(self):
        """Block the host until the kernel has finished."""
        check(lib().sgmcmc_event_synchronize(self.stop), "sgmcmc_event_synchronize")

    def __del__(self):
        try:
            lib().sgmcmc_event_destroy(self.start)
            lib().sgmcmc_event_destroy(self.stop)
        except Exception:                     # interpreter shutdown
            pass


_default_launch = None


def set_launch_config(block_threads=0, quads_per_thread=0, max_blocks=0, nontemporal=-1):
    """Python-side default ``LaunchConfig`` for calls that pass no ``launch=`` (sweep tools). Validated by the
    library at the next launch. ``set_launch_config()`` with no arguments restores the library defaults."""
    global _default_launch
    cfg = LaunchConfig(block_threads, quads_per_thread, max_blocks, nontemporal)
    _default_launch = None if cfg.as_dict() == LaunchConfig().as_dict() else cfg


def get_launch_config():
    """The effective defaults: the library's (block_threads -1 = auto, 1 quad per lane, uncapped grid,
    nontemporal 2 = auto) overlaid with the Python-side default."""
    out = {"block_threads": -1, "quads_per_thread": 1, "max_blocks": 1 << 20, "nontemporal": 2}
    if _default_launch is not None:
        d = _default_launch.as_dict()
        out.update({k: v for k, v in d.items() if v != (-1 if k == "nontemporal" else 0)})
    return out


def _launch(launch):
    import ctypes
    cfg = launch if launch is not None else _default_launch
    return None if cfg is None else ctypes.byref(cfg._c)


class StepOpts(object):
    """Optional extras of one step call (``sgmcmc_step_opts_t``; see ``include/sgmcmc_hip.h``):

    ``first_element``  this launch covers the slice of the chain's parameter vector that starts there (multiple of 4);
    ``stats_base`` / ``stats_total``  statistics record slot of block 0 / record count of the whole step;
    ``theta_sq_only``  reduce only sum theta'^2;  ``hbm_resident``  geometry hint for slices of a large arena;
    ``skip_minv_store``  burn-in step that does not write minv;
    ``moments`` = (mean, m2, count)  fold theta' into the Welford moments in the same pass (K4 fused);
    ``scalars_dev``  device block from :func:`step_scalars` that overrides the by-value scalars;
    ``gather`` = (X, y, start, x_out, y_out)  the NEXT step's minibatch window as a side job of the launch (what
    :func:`window_gather` does, without a launch of its own; see :func:`gather_fits_step_launch`)."""

    __slots__ = ("_c", "_keep")

    def __init__(self, first_element=0, stats_base=0, stats_total=0, theta_sq_only=False, hbm_resident=False,
                 skip_minv_store=False, moments=None, scalars_dev=None, gather=None, like=None):
        from pysgmcmc_amd import _lib
        flags = (_lib.STEP_HBM_RESIDENT if hbm_resident else 0) | (_lib.STEP_SKIP_MINV_STORE if skip_minv_store else 0)
        mean = m2 = None
        count = 0
        if moments is not None:
            mean, m2, count = moments
            if like is not None:
                for t in (mean, m2):
                    if t.dtype != like.dtype or t.numel() != like.numel() or t.device != like.device:
                        raise TypeError("pysgmcmc_amd: fused moments must match theta in dtype, length and device")
        if scalars_dev is not None and like is not None and (scalars_dev.dtype != like.dtype or scalars_dev.numel() < 5):
            raise TypeError("pysgmcmc_amd: scalars_dev must hold 5 elements of the step's dtype")
        gx = gy = gxo = gyo = None
        gstart = gbatch = gdim = gld = 0
        if gather is not None:
            gx, gy, gstart, gxo, gyo = gather
            if not gather_fits_step_launch(gx, gy, gstart, gxo, gyo) or (like is not None and (gx.dtype != like.dtype or gx.device != like.device)):
                raise ValueError("pysgmcmc_amd: this window cannot ride in a step launch (see kernels.gather_fits_step_launch)")
            gbatch, gdim, gld = int(gxo.shape[0]), int(gx.shape[1]), int(gxo.stride(0))
        self._keep = (mean, m2, scalars_dev, gx, gy, gxo, gyo)
        self._c = _lib.StepOptsStruct(int(first_element), int(stats_base), int(stats_total),
                                      _lib.STATS_THETA_SQ if theta_sq_only else 0, flags, _ptr(mean), _ptr(m2), int(count),
                                      _ptr(scalars_dev), *[None if t is None else t.data_ptr() for t in (gx, gy, gxo, gyo)],     # (x_out is a pitched view)
                                      int(gstart), gbatch, gdim, gld, 0)


def gather_fits_step_launch(X, y, start, x_out, y_out):
    """True when the window ``X[start:start + B]``, ``y[start:start + B]`` -> ``x_out [B, D]`` (row pitch ``x_out.stride(0)``),
    ``y_out [B]`` can be gathered as the side job of a step launch (``StepOpts(gather=...)``): one dtype and device, contiguous
    dataset, rows of a multiple of 16 bytes, 16-byte aligned source window and destination."""
    if not (isinstance(X, torch.Tensor) and X.is_cuda and X.dim() == 2 and X.is_contiguous() and y.is_contiguous() and y.dim() == 1):
        return False
    if not (X.dtype == y.dtype == x_out.dtype == y_out.dtype and X.device == y.device == x_out.device == y_out.device):
        return False
    if X.dtype not in (torch.float32, torch.float64) or x_out.dim() != 2 or x_out.stride(1) != 1 or not y_out.is_contiguous():
        return False
    B, D, es = int(x_out.shape[0]), int(X.shape[1]), X.element_size()
    if x_out.shape[1] != D or y_out.numel() != B or B == 0 or start < 0 or start + B > X.shape[0] or y.numel() != X.shape[0]:
        return False
    return ((D * es) % 16 == 0 and (x_out.stride(0) * es) % 16 == 0 and (X.data_ptr() + start * D * es) % 16 == 0
            and x_out.data_ptr() % 16 == 0 and y_out.data_ptr() % 4 == 0)


def _opts(opts, like):
    import ctypes
    if opts is None:
        return None
    if not isinstance(opts, StepOpts):
        opts = StepOpts(like=like, **opts)
    return ctypes.byref(opts._c)


def step_stats_records(n, launch):
    """Number of blocks (= statistics records) a vector-path step launch of ``n`` elements uses under ``launch``
    (a :class:`LaunchConfig` with an explicit ``block_threads``)."""
    import ctypes
    blocks = int(lib().sgmcmc_step_stats_records(int(n), ctypes.byref(launch._c)))
    if blocks == 0:
        check(-1, "sgmcmc_step_stats_records")
    return blocks


def step_scalars(kind, out, *scalars):
    """Fill the device block ``out`` (>= 5 elements of the step's dtype) with the derived scalars of a step, for
    ``StepOpts(scalars_dev=out)``: kind "sghmc" (eps, scale_grad, mdecay), "sgld" (eps, A, scale_grad) or "rsghmc"
    (eps, mass, c, D, b_hat)."""
    f = getattr(lib(), "sgmcmc_%s_scalars_%s" % (kind, _sfx(out)))
    if out.numel() < 5:
        raise ValueError("pysgmcmc_amd: the scalars block needs 5 elements")
    with _on(out):
        rc = f(*[float(v) for v in scalars], _ptr(out), _stream(out))
    check(rc, "sgmcmc_%s_scalars" % kind)
    return out


def sghmc_step(theta, V, grad, tau, g, v_hat, minv, r, eps, scale_grad, mdecay, adapt, xi=None, seed=0, step=0, step_dev=None,
               stats=None, grad_decay=0.0, launch=None, opts=None):
    """K1, one fused SGHMC step in place (pysgmcmc/samplers/sghmc.py:165-251)."""
    f = getattr(lib(), "sgmcmc_sghmc_step_" + _sfx(theta))
    with _on(theta):
        rc = f(_ptr(theta), _ptr(V, theta), _ptr(grad, theta), _ptr(tau, theta), _ptr(g, theta),
               _ptr(v_hat, theta), _ptr(minv, theta), _ptr(r, theta), theta.numel(),
               float(eps), float(scale_grad), float(mdecay), float(grad_decay), int(bool(adapt)), _ptr(xi, theta),
               int(seed), int(step), _ctr(step_dev), *_stats(stats), _opts(opts, theta), _launch(launch), _stream(theta))
    check(rc, "sgmcmc_sghmc_step")


def sgld_step(theta, grad, tau, g, v_hat, minv, r, eps, A, scale_grad, adapt, xi=None, seed=0, step=0, step_dev=None,
              stats=None, grad_decay=0.0, launch=None, opts=None):
    """K2, one fused SGLD step in place (pysgmcmc/samplers/sgld.py:149-211)."""
    f = getattr(lib(), "sgmcmc_sgld_step_" + _sfx(theta))
    with _on(theta):
        rc = f(_ptr(theta), _ptr(grad, theta), _ptr(tau, theta), _ptr(g, theta), _ptr(v_hat, theta),
               _ptr(minv, theta), _ptr(r, theta), theta.numel(), float(eps), float(A), float(scale_grad),
               float(grad_decay), int(bool(adapt)), _ptr(xi, theta), int(seed), int(step), _ctr(step_dev), *_stats(stats),
               _opts(opts, theta), _launch(launch), _stream(theta))
    check(rc, "sgmcmc_sgld_step")


def rsghmc_step(theta, p, grad_cost, eps, mass, c, D, b_hat, xi=None, seed=0, step=0, step_dev=None,
                stats=None, grad_decay=0.0, launch=None, opts=None):
    """K3, one fused relativistic SGHMC step (pysgmcmc/samplers/relativistic_sghmc.py:120-140)."""
    f = getattr(lib(), "sgmcmc_rsghmc_step_" + _sfx(theta))
    with _on(theta):
        rc = f(_ptr(theta), _ptr(p, theta), _ptr(grad_cost, theta), theta.numel(), float(eps), float(mass),
               float(c), float(D), float(b_hat), float(grad_decay), _ptr(xi, theta), int(seed), int(step), _ctr(step_dev), *_stats(stats),
               _opts(opts, theta), _launch(launch), _stream(theta))
    check(rc, "sgmcmc_rsghmc_step")


def toy_chains(sampler, target, target_params, theta, mom, tau, g, v_hat, minv, scalars, seeds, first_step, n_steps,
               burn_in_steps, keep_every=1, kept=None):
    """``n_steps`` steps of ``theta.shape[0]`` independent chains on a built-in toy target in one launch
    (``sgmcmc_toy_chains_*``, see include/sgmcmc_hip.h). ``theta`` etc.: ``[n_chains, dim]`` device tensors;
    ``seeds``: int64 device tensor ``[n_chains]``; ``kept``: ``[ceil(n_steps / keep_every), n_chains, dim]`` or None."""
    import ctypes
    f = getattr(lib(), "sgmcmc_toy_chains_" + _sfx(theta))
    n_chains, dim = int(theta.shape[0]), int(theta.shape[1])
    tp = [float(v) for v in target_params]
    k = {0: len(tp) // 3, 1: 0, 2: len(tp) // 2}[int(target)]
    if seeds.dtype != torch.int64 or seeds.numel() != n_chains:
        raise TypeError("seeds must be an int64 device tensor with one entry per chain")
    n_kept = (int(n_steps) + int(keep_every) - 1) // int(keep_every)
    if kept is not None and (kept.numel() != n_kept * n_chains * dim or kept.dtype != theta.dtype):
        raise ValueError("kept must hold ceil(n_steps / keep_every) x n_chains x dim elements of theta's dtype")
    arr = (ctypes.c_double * max(len(tp), 1))(*tp)
    sc = (ctypes.c_double * 5)(*([float(v) for v in scalars] + [0.0] * (5 - len(scalars))))
    with _on(theta):
        rc = f(int(sampler), int(target), arr, k, _ptr(theta), _ptr(mom, theta), _ptr(tau, theta), _ptr(g, theta),
               _ptr(v_hat, theta), _ptr(minv, theta), n_chains, dim, sc, _ptr(seeds), int(first_step), int(n_steps),
               int(burn_in_steps), int(keep_every), _ptr(kept), _stream(theta))
    check(rc, "sgmcmc_toy_chains")
    return kept


def philox_normal(out, seed, step, step_dev=None, launch=None):
    """K5, out[i] = xi(seed, step, i)."""
    f = getattr(lib(), "sgmcmc_philox_normal_" + _sfx(out))
    with _on(out):
        rc = f(_ptr(out), out.numel(), int(seed), int(step), _ctr(step_dev), _launch(launch), _stream(out))
    check(rc, "sgmcmc_philox_normal")
    return out


def philox_bits(out, seed, step, step_dev=None):
    if out.dtype not in (torch.int32, torch.uint32):
        raise TypeError("philox_bits wants a 32-bit integer tensor")
    with _on(out):
        rc = lib().sgmcmc_philox_bits_u32(_ptr(out), out.numel(), int(seed), int(step), _ctr(step_dev), _stream(out))
    check(rc, "sgmcmc_philox_bits_u32")
    return out


def moments_update(theta, mean, m2, count, launch=None):
    """K4, Welford update of (mean, m2) with the sample theta; count includes it."""
    f = getattr(lib(), "sgmcmc_moments_update_" + _sfx(theta))
    with _on(theta):
        rc = f(_ptr(theta), _ptr(mean, theta), _ptr(m2, theta), theta.numel(), int(count), _launch(launch), _stream(theta))
    check(rc, "sgmcmc_moments_update")


def rhat_pack(mean, m2, count, out3, n_shards=1, shard_len=None):
    """This chain's contribution to the R-hat exchange: ``[mean | mean^2 | m2 / (count - 1)]``, either as one
    ``3 n`` buffer (``n_shards = 1``: all-reduce) or as ``n_shards`` chunks of ``[3][shard_len]`` (reduce-scatter:
    chunk ``s`` = the rows of parameters ``[s * shard_len, (s + 1) * shard_len)``, zero beyond ``n``)."""
    n = mean.numel()
    shard_len = n if shard_len is None else int(shard_len)
    if out3.numel() != 3 * int(n_shards) * shard_len:
        raise ValueError("out3 must hold 3 * n_shards * shard_len elements")
    if out3.dtype != mean.dtype or m2.dtype != mean.dtype:
        # the kernel is picked from mean.dtype: an f64 pack into an f32 buffer of equal length would write past its end
        raise TypeError("rhat_pack: mean, m2 and out3 must share a dtype (got %s, %s, %s)" % (mean.dtype, m2.dtype, out3.dtype))
    if out3.device != mean.device:
        raise ValueError("rhat_pack: out3 lives on %s, the moments on %s" % (out3.device, mean.device))
    f = getattr(lib(), "sgmcmc_rhat_pack_" + _sfx(mean))
    with _on(mean):
        rc = f(_ptr(mean), _ptr(m2, mean), n, int(count), int(n_shards), shard_len, _ptr(out3), _stream(mean))
    check(rc, "sgmcmc_rhat_pack")


def rhat_finish(sum3, n, m_chains, count, rhat, summary_out4=None, summary_workspace=None, ld=None):
    """R-hat of ``n`` parameters from the chain-summed rows ``sum3 = [S_mean | S_sq | S_var]`` (row pitch ``ld``,
    default ``n``). With ``summary_out4`` (float64[4] device tensor) and ``summary_workspace`` the K6 summary
    {sum, sum^2, min, max} of R-hat is left on the device (no host sync)."""
    f = getattr(lib(), "sgmcmc_rhat_finish_" + _sfx(sum3))
    if rhat.dtype != sum3.dtype:
        raise TypeError("rhat and sum3 must share a dtype")
    if summary_out4 is not None and summary_out4.dtype != torch.float64:
        raise TypeError("summary_out4 must be float64")
    ld = int(n if ld is None else ld)
    if sum3.numel() < 2 * ld + int(n) or rhat.numel() < int(n):
        raise ValueError("rhat_finish: buffers shorter than n / ld say")
    with _on(sum3):
        rc = f(_ptr(sum3), int(n), ld, int(m_chains), int(count), _ptr(rhat), _ptr(summary_out4), _ptr(summary_workspace),
               _stream(sum3))
    check(rc, "sgmcmc_rhat_finish")


def summary_workspace(device):
    return torch.empty(lib().sgmcmc_summary_workspace_bytes(), dtype=torch.uint8, device=device)


def summary(x, out4=None, workspace=None):
    """K6, deterministic {sum, sum of squares, min, max} of a flat array -> float64[4] device tensor."""
    f = getattr(lib(), "sgmcmc_summary_" + _sfx(x))
    if out4 is None:
        out4 = torch.empty(4, dtype=torch.float64, device=x.device)
    if workspace is None:
        workspace = torch.empty(lib().sgmcmc_summary_workspace_bytes(), dtype=torch.uint8, device=x.device)
    with _on(x):
        rc = f(_ptr(x), x.numel(), _ptr(out4), _ptr(workspace), _stream(x))
    check(rc, "sgmcmc_summary")
    return out4


def bnn_head(mean, y, log_var, theta_sumsq, batch_size, n_examples, n_params, wdecay, prior_mean, prior_var,
             delta, cost_out, grad_log_var_out, mse_out, fold_prior_grad=False, stats_workspace=None,
             last_bias=None, grad_last_bias_out=None, add_last_bias=False):
    """Loss head of the BNN cost path in one launch (see include/sgmcmc_hip.h). ``theta_sumsq`` is a
    float64 device scalar, or None when ``stats_workspace`` (a StepStats.workspace) is given."""
    f = getattr(lib(), "sgmcmc_bnn_head_" + _sfx(mean))
    if theta_sumsq is not None and theta_sumsq.dtype != torch.float64:
        raise TypeError("theta_sumsq must be a float64 device scalar")
    with _on(mean):
        rc = f(_ptr(mean), _ptr(y, mean), _ptr(log_var), _ptr(theta_sumsq), _ptr(stats_workspace), _ptr(last_bias),
               mean.numel(), float(batch_size), float(n_examples), float(n_params), float(wdecay), float(prior_mean),
               float(prior_var), int(bool(fold_prior_grad)) | (2 if add_last_bias else 0), _ptr(delta, mean), _ptr(cost_out),
               _ptr(grad_log_var_out), _ptr(grad_last_bias_out), _ptr(mse_out), _stream(mean))
    check(rc, "sgmcmc_bnn_head")


def tanh_backward(delta, h):
    """delta *= 1 - h^2 in place."""
    f = getattr(lib(), "sgmcmc_tanh_backward_" + _sfx(delta))
    with _on(delta):
        rc = f(_ptr(delta), _ptr(h, delta), delta.numel(), _stream(delta))
    check(rc, "sgmcmc_tanh_backward")


def tanh_backward_colsum(delta, h, colsum, bias=None, beta=0.0):
    """delta *= 1 - h^2 in place on a row-major (rows, cols) matrix and colsum[c] = sum_r delta[r, c]
    (+ beta * bias[c]): tanh backward fused with the bias gradient of that layer."""
    f = getattr(lib(), "sgmcmc_tanh_backward_colsum_" + _sfx(delta))
    rows, cols = delta.shape
    with _on(delta):
        rc = f(_ptr(delta), _ptr(h, delta), rows, cols, _ptr(bias), float(beta), _ptr(colsum), _stream(delta))
    check(rc, "sgmcmc_tanh_backward_colsum")


def bnn_last_layer_backward(dvec, w, h, delta_prev, colsum, gw, bias_prev=None, beta=0.0):
    """Backward of a single-output last layer fused with the tanh backward of the layer below
    (see include/sgmcmc_hip.h): fills delta_prev (rows, cols), colsum (cols,), gw (cols,)."""
    f = getattr(lib(), "sgmcmc_bnn_last_layer_backward_" + _sfx(h))
    rows, cols = h.shape
    with _on(h):
        rc = f(_ptr(dvec), _ptr(w), _ptr(h), rows, cols, _ptr(bias_prev), float(beta), _ptr(delta_prev, h),
               _ptr(colsum), _ptr(gw), _stream(h))
    check(rc, "sgmcmc_bnn_last_layer_backward")


def bnn_fused_sghmc_steps(theta, V, grad, tau, g, v_hat, minv, layer_sizes, X, y, window_starts, batch,
                          batch_size, n_examples, wdecay, prior_mean, prior_var, eps, scale_grad, mdecay,
                          first_step, n_steps, burn_in_steps, seed_base, cost_out, xi=None, n_chains=1,
                          chain_stride=None):
    """``n_steps`` complete SGHMC steps of a small tanh-MLP BNN per chain in one launch (see
    include/sgmcmc_hip.h). Rows are ``(n_chains * chain_stride,)`` or, for one chain, ``(n_params,)``."""
    import ctypes
    f = getattr(lib(), "sgmcmc_bnn_fused_sghmc_steps_" + _sfx(theta))
    sizes = [int(v) for v in layer_sizes]
    n_layers = len(sizes) - 1
    n_params = sum(sizes[l] * sizes[l + 1] + sizes[l + 1] for l in range(n_layers)) + 1
    if chain_stride is None:
        chain_stride = theta.numel() if n_chains == 1 else theta.numel() // n_chains
    if window_starts.dtype != torch.int32 or window_starts.numel() != n_chains * n_steps:
        raise TypeError("window_starts must be an int32 device tensor of n_chains * n_steps entries")
    arr = (ctypes.c_int * len(sizes))(*sizes)
    with _on(theta):
        rc = f(_ptr(theta), _ptr(V), _ptr(grad), _ptr(tau), _ptr(g), _ptr(v_hat), _ptr(minv), n_params,
               int(chain_stride), int(n_chains), arr, n_layers, _ptr(X), _ptr(y), int(X.shape[0]),
               _ptr(window_starts), int(batch), float(batch_size), float(n_examples), float(wdecay),
               float(prior_mean), float(prior_var), float(eps), float(scale_grad), float(mdecay),
               int(first_step), int(n_steps), int(burn_in_steps), int(seed_base), _ptr(xi), _ptr(cost_out),
               _stream(theta))
    check(rc, "sgmcmc_bnn_fused_sghmc_steps")
    return cost_out


def bnn_fused_sgld_steps(theta, grad, tau, g, v_hat, minv, layer_sizes, X, y, window_starts, batch,
                         batch_size, n_examples, wdecay, prior_mean, prior_var, eps, scale_grad, A,
                         first_step, n_steps, burn_in_steps, seed_base, cost_out, xi=None, n_chains=1,
                         chain_stride=None):
    """The fused small-model kernel with the preconditioned-SGLD update (``sgmcmc_bnn_fused_sgld_steps_*``)."""
    import ctypes
    f = getattr(lib(), "sgmcmc_bnn_fused_sgld_steps_" + _sfx(theta))
    sizes = [int(v) for v in layer_sizes]
    n_layers = len(sizes) - 1
    n_params = sum(sizes[l] * sizes[l + 1] + sizes[l + 1] for l in range(n_layers)) + 1
    if chain_stride is None:
        chain_stride = theta.numel() if n_chains == 1 else theta.numel() // n_chains
    if window_starts.dtype != torch.int32 or window_starts.numel() != n_chains * n_steps:
        raise TypeError("window_starts must be an int32 device tensor of n_chains * n_steps entries")
    arr = (ctypes.c_int * len(sizes))(*sizes)
    with _on(theta):
        rc = f(_ptr(theta), _ptr(grad), _ptr(tau), _ptr(g), _ptr(v_hat), _ptr(minv), n_params,
               int(chain_stride), int(n_chains), arr, n_layers, _ptr(X), _ptr(y), int(X.shape[0]),
               _ptr(window_starts), int(batch), float(batch_size), float(n_examples), float(wdecay),
               float(prior_mean), float(prior_var), float(eps), float(scale_grad), float(A),
               int(first_step), int(n_steps), int(burn_in_steps), int(seed_base), _ptr(xi), _ptr(cost_out),
               _stream(theta))
    check(rc, "sgmcmc_bnn_fused_sgld_steps")
    return cost_out


def bias_tanh(a, bias):
    """``a = tanh(a + bias)`` in place (``a``: contiguous ``[rows, cols]``, ``bias``: ``[cols]``): a hidden layer's activation
    after a plain forward GEMM."""
    rows, cols = int(a.shape[0]), int(a.shape[1])
    if bias.numel() != cols or bias.dtype != a.dtype or not a.is_contiguous() or not bias.is_contiguous():
        raise ValueError("pysgmcmc_amd: bias_tanh shapes / dtypes do not match")
    with _on(a):
        rc = getattr(lib(), "sgmcmc_bias_tanh_" + _sfx(a))(_ptr(a), _ptr(bias), rows, cols, _stream(a))
    check(rc, "sgmcmc_bias_tanh")
    return a


def bnn_dense_tanh_fits(h, W, out):
    """Can :func:`bnn_dense_tanh` take this layer? f32 device tensors, batch a multiple of 32, fan-out a multiple of 64, fan-in a
    multiple of 16 and >= 64, 16-byte aligned rows. Any number of 32 x 64 output tiles: more than one per compute unit run in
    rounds, a thin last round as half tiles (``include/sgmcmc_hip.h``)."""
    if not (h.is_cuda and h.dtype == W.dtype == out.dtype == torch.float32 and h.dim() == W.dim() == out.dim() == 2):
        return False
    M, K, N = int(h.shape[0]), int(h.shape[1]), int(W.shape[1])
    if W.shape[0] != K or tuple(out.shape) != (M, N) or M % 32 or N % 64 or K % 16 or K < 64:
        return False
    for t in (h, W, out):
        if t.stride(1) != 1 or t.stride(0) % 4 or t.data_ptr() % 16:
            return False
    return True


def bnn_dense_tanh_dot_parts(M, N, device):
    """Rows of the ``dot_parts`` buffer :func:`bnn_dense_tanh` fills for an ``M x N`` layer on ``device`` (one per column tile:
    64 columns, or 32 where the launch uses half tiles)."""
    with torch.cuda.device(device):
        n = int(lib().sgmcmc_bnn_dense_tanh_dot_parts(int(M), int(N)))
    if n <= 0:
        raise ValueError("pysgmcmc_amd: bnn_dense_tanh_dot_parts: M %% 32 == 0 and N %% 64 == 0 required, got %d x %d" % (M, N))
    return n


def bnn_dense_tanh(h, W, bias, out, w_next=None, dot_parts=None, stats_workspace=None, tsq_parts=None):
    """``out = tanh(h @ W + bias)`` in ONE launch on the fp32 matrix cores (``sgmcmc_bnn_dense_tanh_f32``; shapes as
    :func:`bnn_dense_tanh_fits` demands). With ``w_next [N]`` and ``dot_parts [bnn_dense_tanh_dot_parts(M, N), M]`` the launch also
    leaves the per-column-tile partial dot products of ``out`` with ``w_next`` (the single output unit); with ``stats_workspace`` and
    ``tsq_parts`` it adds up the sum(theta^2) records like :func:`tanh_rowdot`."""
    M, K = int(h.shape[0]), int(h.shape[1])
    N = int(W.shape[1])
    if not bnn_dense_tanh_fits(h, W, out) or bias.numel() != N or bias.dtype != h.dtype:
        raise ValueError("pysgmcmc_amd: bnn_dense_tanh: shapes / dtypes / alignment do not fit the kernel (see bnn_dense_tanh_fits)")
    if dot_parts is not None and (w_next is None or w_next.numel() != N or not dot_parts.is_contiguous()
                                  or dot_parts.numel() != bnn_dense_tanh_dot_parts(M, N, h.device) * M):
        raise ValueError("pysgmcmc_amd: bnn_dense_tanh needs w_next [N] and dot_parts [bnn_dense_tanh_dot_parts(M, N), M]")
    if tsq_parts is not None and (tsq_parts.dtype != torch.float64 or tsq_parts.numel() < 16 or (M // 32) * (N // 64) < 16):
        raise ValueError("pysgmcmc_amd: bnn_dense_tanh: tsq_parts must be float64[16] and the launch needs >= 16 output tiles")
    with torch.cuda.device(h.device):                          # (h may be a pitched view: rows are what must be contiguous)
        rc = lib().sgmcmc_bnn_dense_tanh_f32(h.data_ptr(), W.data_ptr(), _ptr(bias), out.data_ptr(), M, N, K, h.stride(0),
                                             W.stride(0), out.stride(0), _ptr(w_next), _ptr(dot_parts), _ptr(stats_workspace),
                                             _ptr(tsq_parts), _stream(h))
    check(rc, "sgmcmc_bnn_dense_tanh_f32")
    return out


def bnn_dense_tanh_backward_fits(delta, W, act, out):
    """Can :func:`bnn_dense_tanh_backward` take this layer? ``delta [M, K]``, ``W [N, K]`` (the weights of the layer above, fan-in
    N), ``act`` / ``out [M, N]``: the limits of :func:`bnn_dense_tanh_fits` with the roles of W's two axes swapped."""
    if not (delta.is_cuda and delta.dtype == W.dtype == act.dtype == out.dtype == torch.float32
            and delta.dim() == W.dim() == act.dim() == out.dim() == 2):
        return False
    M, K, N = int(delta.shape[0]), int(delta.shape[1]), int(W.shape[0])
    if W.shape[1] != K or tuple(out.shape) != (M, N) or tuple(act.shape) != (M, N) or M % 32 or N % 64 or K % 16 or K < 64:
        return False
    for t in (delta, W, act, out):
        if t.stride(1) != 1 or t.stride(0) % 4 or t.data_ptr() % 16:
            return False
    return True


def bnn_dense_tanh_backward(delta, W, act, out, colsum_parts=None, finish=None):
    """``out = (delta @ W.T) * (1 - act ** 2)`` in ONE launch on the fp32 matrix cores (``sgmcmc_bnn_dense_tanh_backward_f32``):
    the backward step through a hidden tanh layer, d cost / d pre-activation of the layer below from the one above.
    ``colsum_parts [M // 32, N]`` (optional) receives the column sums of ``out`` per 32-row tile -- the bias gradient once its
    rows are added up, which the NEXT launch does on the side when handed ``finish = (parts, colsum, bias, beta)``
    (``colsum[:] = parts.sum(0) + beta * bias``, rows added in order), or :func:`colsum_finish`. Deterministic."""
    M, K = int(delta.shape[0]), int(delta.shape[1])
    N = int(W.shape[0])
    if not bnn_dense_tanh_backward_fits(delta, W, act, out):
        raise ValueError("pysgmcmc_amd: bnn_dense_tanh_backward: shapes / dtypes / alignment do not fit the kernel "
                         "(see bnn_dense_tanh_backward_fits)")
    if colsum_parts is not None and (colsum_parts.dtype != delta.dtype or colsum_parts.numel() != (M // 32) * N
                                     or not colsum_parts.is_contiguous() or colsum_parts.device != delta.device):
        raise ValueError("pysgmcmc_amd: bnn_dense_tanh_backward: colsum_parts must be a contiguous [M // 32, N] tensor")
    fin = _colsum_finish_args(finish, delta)
    if fin[0] is not None and colsum_parts is not None and fin[0] == colsum_parts.data_ptr():
        raise ValueError("pysgmcmc_amd: bnn_dense_tanh_backward: the finish job needs partial sums other than this launch's own")
    with torch.cuda.device(delta.device):
        rc = lib().sgmcmc_bnn_dense_tanh_backward_f32(
            delta.data_ptr(), W.data_ptr(), act.data_ptr(), out.data_ptr(), _ptr(colsum_parts), M, N, K, delta.stride(0),
            W.stride(0), act.stride(0), out.stride(0), *fin, _stream(delta))
    check(rc, "sgmcmc_bnn_dense_tanh_backward_f32")
    return out


def _colsum_finish_args(finish, like):
    """(parts ptr, rows, n, bias ptr, beta, colsum ptr) of a ``finish = (parts [rows, n], colsum [n], bias [n] | None, beta)`` job."""
    if finish is None:
        return (None, 0, 0, None, 0.0, None)
    parts, colsum, bias, beta = finish
    n = int(colsum.numel())
    if (parts.dim() != 2 or parts.shape[1] != n or not parts.is_contiguous() or not colsum.is_contiguous()
            or parts.dtype != like.dtype or colsum.dtype != like.dtype or parts.device != like.device or colsum.device != like.device
            or (beta != 0.0 and (bias is None or bias.numel() != n or bias.dtype != like.dtype or not bias.is_contiguous()))):
        raise ValueError("pysgmcmc_amd: column-sum finish job: parts [rows, n], colsum [n] (and bias [n] with beta != 0) of the launch's dtype / device")
    return (parts.data_ptr(), int(parts.shape[0]), n, bias.data_ptr() if beta != 0.0 else None, float(beta), colsum.data_ptr())


def colsum_finish(parts, colsum, bias=None, beta=0.0):
    """``colsum[:] = parts.sum(0) (+ beta * bias)``, the rows of ``parts`` added in order: the second half of
    :func:`bnn_dense_tanh_backward`'s bias gradient as a launch of its own (``sgmcmc_colsum_finish_f32``)."""
    if parts.dtype != torch.float32 or not parts.is_cuda:
        raise TypeError("pysgmcmc_amd: colsum_finish takes float32 device tensors")
    fin = _colsum_finish_args((parts, colsum, bias, beta), parts)
    with torch.cuda.device(parts.device):
        rc = lib().sgmcmc_colsum_finish_f32(*fin, _stream(parts))
    check(rc, "sgmcmc_colsum_finish_f32")
    return colsum


def bnn_planes_bytes(M, N):
    """Bytes of one set of three bf16 planes of an ``[M, N]`` float32 matrix (``sgmcmc_bnn_planes_bytes``; 0: invalid shape)."""
    return int(lib().sgmcmc_bnn_planes_bytes(int(M), int(N)))


def _stack_stride(ts):
    """Elements between consecutive tensors of ``ts`` (equal shapes, row-major with one row pitch, equally spaced), or raise."""
    t0 = ts[0]
    if any(t.shape != t0.shape or t.dtype != torch.float32 or not t.is_cuda or t.dim() != 2 or t.stride(1) != 1
           or t.stride(0) != t0.stride(0) for t in ts):
        raise ValueError("pysgmcmc_amd: a batch of equally shaped float32 [rows, cols] device matrices with one row pitch is expected")
    gaps = {ts[k + 1].data_ptr() - ts[k].data_ptr() for k in range(len(ts) - 1)}
    if len(gaps) > 1 or any(g <= 0 or g % 4 for g in gaps):
        raise ValueError("pysgmcmc_amd: the matrices of a batch must lie at one constant, positive stride")
    return gaps.pop() // 4 if gaps else 0


def bnn_split_planes(mats, planes):
    """Write every ``[M, N]`` float32 matrix of ``mats`` (equally spaced) as three exact bf16 planes into the uint8 buffer
    ``planes`` (``len(mats) * bnn_planes_bytes(M, N)`` bytes; layout: include/sgmcmc_hip.h) -- the operands of
    :func:`bnn_gw_planes`. One launch (``sgmcmc_bnn_split_planes_f32``)."""
    stride = _stack_stride(mats)
    M, N = int(mats[0].shape[0]), int(mats[0].shape[1])
    one = bnn_planes_bytes(M, N)
    if one == 0 or planes.dtype != torch.uint8 or not planes.is_cuda or planes.numel() < one * len(mats) or planes.data_ptr() % 16:
        raise ValueError("pysgmcmc_amd: bnn_split_planes needs M %% 16 == 0 and a 16-byte aligned uint8 device buffer of "
                         "len(mats) * bnn_planes_bytes(M, N) bytes")
    with torch.cuda.device(planes.device):
        rc = lib().sgmcmc_bnn_split_planes_f32(mats[0].data_ptr(), len(mats), stride, M, N, int(mats[0].stride(0)), planes.data_ptr(),
                                               one, _stream(planes))
    check(rc, "sgmcmc_bnn_split_planes_f32")
    return planes


def bnn_gw_planes(a_planes, b_planes, outs, M):
    """``outs[z][nA, nB] = A_z^T B_z`` for the plane sets written by :func:`bnn_split_planes` (``a_planes``: ``len(outs)`` sets of
    ``[M, nA]`` matrices, ``b_planes``: of ``[M, nB]``) -- the weight gradients ``h^T delta`` of equally shaped layers as ONE launch
    on the bf16 matrix pipe at fp32 accuracy (``sgmcmc_bnn_gw_planes_f32``). ``outs``: equally spaced float32 device views."""
    stride = _stack_stride(outs)
    nA, nB = int(outs[0].shape[0]), int(outs[0].shape[1])
    sa, sb = bnn_planes_bytes(M, nA), bnn_planes_bytes(M, nB)
    if (sa == 0 or a_planes.dtype != torch.uint8 or b_planes.dtype != torch.uint8 or a_planes.numel() < sa * len(outs)
            or b_planes.numel() < sb * len(outs) or a_planes.device != outs[0].device or b_planes.device != outs[0].device):
        raise ValueError("pysgmcmc_amd: bnn_gw_planes needs M %% 16 == 0 and plane buffers of len(outs) sets each on the outputs' device")
    with torch.cuda.device(outs[0].device):
        rc = lib().sgmcmc_bnn_gw_planes_f32(a_planes.data_ptr(), sa, b_planes.data_ptr(), sb, outs[0].data_ptr(), stride, len(outs),
                                            int(M), nA, nB, int(outs[0].stride(0)), _stream(outs[0]))
    check(rc, "sgmcmc_bnn_gw_planes_f32")
    return outs


def tanh_rowdot(a, w, out, stats_workspace=None, tsq_parts=None, bias=None):
    """``a = tanh(a [+ bias])`` in place (``[rows, cols]``) and ``out[r] = a[r] . w`` in one launch. With ``stats_workspace``
    (a ``StepStats.workspace``) and ``tsq_parts`` (float64[16] device tensor) the launch also adds up the
    sum(theta^2) partials into 16 slices for :func:`bnn_head_last_layer_backward`."""
    f = getattr(lib(), "sgmcmc_bias_tanh_rowdot_" + _sfx(a))
    rows, cols = int(a.shape[0]), int(a.shape[1])
    if w.numel() != cols or out.numel() != rows or (bias is not None and (bias.numel() != cols or bias.dtype != a.dtype)):
        raise ValueError("pysgmcmc_amd: tanh_rowdot shapes do not match")
    if tsq_parts is not None and (tsq_parts.dtype != torch.float64 or tsq_parts.numel() < 16):
        raise TypeError("tsq_parts must be a float64 device tensor of 16 elements")
    with _on(a):
        rc = f(_ptr(a), _ptr(bias), _ptr(w), rows, cols, _ptr(out), _ptr(stats_workspace), _ptr(tsq_parts), _stream(a))
    check(rc, "sgmcmc_tanh_rowdot")


def bnn_head_last_layer_backward(mean, y, log_var, tsq_parts, last_bias, batch_size, n_examples, n_params, wdecay,
                                 prior_mean, prior_var, w, h, bias_prev, beta, cost_out, grad_log_var_out,
                                 grad_last_bias_out, mse_out, delta_prev, colsum, gw, fold_prior_grad=False,
                                 add_last_bias=True):
    """Loss head + backward of the single-output last layer (+ tanh backward and bias gradient of the layer below) in
    ONE launch; see ``include/sgmcmc_hip.h``. ``mean`` = the pre-bias outputs of :func:`tanh_rowdot` (``[rows]``), or the
    per-column-tile partial dot products of :func:`bnn_dense_tanh` (``[n_parts, rows]``, added in the launch)."""
    f = getattr(lib(), "sgmcmc_bnn_head_last_layer_backward_" + _sfx(h))
    rows, cols = h.shape
    n_parts = mean.numel() // rows
    if (n_parts * rows != mean.numel() or y.numel() != rows or not mean.is_contiguous()
            or not (mean.dim() == 1 or (mean.dim() == 2 and mean.shape[1] == rows))):
        raise ValueError("pysgmcmc_amd: bnn_head_last_layer_backward: mean must be a contiguous [rows] or [n_parts, rows] tensor, y [rows]")
    if y.dtype != h.dtype or y.device != h.device or mean.dtype != h.dtype or mean.device != h.device:
        raise TypeError("pysgmcmc_amd: bnn_head_last_layer_backward: mean and y must have the dtype and device of h")
    with _on(h):
        rc = f(_ptr(mean), n_parts, _ptr(y), _ptr(log_var), _ptr(tsq_parts), _ptr(last_bias), rows, cols, float(batch_size),
               float(n_examples), float(n_params), float(wdecay), float(prior_mean), float(prior_var),
               int(bool(fold_prior_grad)) | (2 if add_last_bias else 0), _ptr(w), _ptr(h), _ptr(bias_prev), float(beta),
               _ptr(cost_out), _ptr(grad_log_var_out), _ptr(grad_last_bias_out), _ptr(mse_out), _ptr(delta_prev, h),
               _ptr(colsum), _ptr(gw), _stream(h))
    check(rc, "sgmcmc_bnn_head_last_layer_backward")


def window_gather(X, y, start, x_out, y_out):
    """``x_out[:] = X[start:start + B]``, ``y_out[:] = y[start:start + B]`` in one launch (B = rows of x_out; ``x_out`` may be a
    pitched view -- rows contiguous, any row stride)."""
    f = getattr(lib(), "sgmcmc_window_gather_" + _sfx(X))
    batch = int(x_out.shape[0])
    dim = int(X.numel() // X.shape[0])
    if x_out.numel() != batch * dim or y_out.numel() != batch or x_out.dim() != 2 or x_out.stride(1) != 1 or x_out.stride(0) < dim:
        raise ValueError("pysgmcmc_amd: window_gather output shapes do not match the window")
    if x_out.dtype != X.dtype or y_out.dtype != X.dtype or x_out.device != X.device or not y_out.is_contiguous():
        raise TypeError("pysgmcmc_amd: window_gather outputs must have the dataset's dtype and device")
    with torch.cuda.device(X.device):
        rc = f(_ptr(X), _ptr(y), int(X.shape[0]), int(start), batch, dim, x_out.data_ptr(), int(x_out.stride(0)), _ptr(y_out), _stream(X))
    check(rc, "sgmcmc_window_gather")


def svgd_max_particles():
    return int(lib().sgmcmc_svgd_max_particles())


def svgd_workspace(n_particles, like):
    """Device workspace of the SVGD kernels for ``n_particles`` particles in ``like``'s dtype/device."""
    nbytes = int(lib().sgmcmc_svgd_workspace_bytes(int(n_particles), like.element_size()))
    if nbytes == 0:
        raise ValueError("pysgmcmc_amd: SVGD supports 1..%d particles, got %d" % (svgd_max_particles(), n_particles))
    if not like.is_cuda:
        _ptr(like)                       # raises the no-CPU-path error
    return torch.empty(nbytes // like.element_size(), dtype=like.dtype, device=like.device)


def svgd_step(particles, grad, hist_grad, n_particles, dim, eps, alpha, fudge_factor, workspace, ld=None,
              repulsion_sign=1):
    """One SVGD step in place on the ``[n_particles x ld]`` matrices (``sgmcmc_svgd_step_*``)."""
    f = getattr(lib(), "sgmcmc_svgd_step_" + _sfx(particles))
    ld = int(dim if ld is None else ld)
    for t in (particles, grad, hist_grad):
        if t.numel() < (n_particles - 1) * ld + dim:
            raise ValueError("pysgmcmc_amd: SVGD matrix shorter than (n_particles - 1) * ld + dim")
    with _on(particles):
        rc = f(_ptr(particles), _ptr(grad), _ptr(hist_grad), int(n_particles), int(dim), ld, float(eps), float(alpha),
               float(fudge_factor), int(repulsion_sign), _ptr(workspace), _stream(particles))
    check(rc, "sgmcmc_svgd_step")


def svgd_kernel(particles, n_particles, dim, workspace, ld=None, kernel_gradients=True):
    """``svgd_kernel(particles)`` of the reference: returns (kernel_matrix [n, n], kernel_gradients [n, dim]
    or None, bandwidth tensor {median, h, h^2}) as fresh device tensors."""
    f = getattr(lib(), "sgmcmc_svgd_kernel_" + _sfx(particles))
    ld = int(dim if ld is None else ld)
    n = int(n_particles)
    K = torch.empty((n, n), dtype=particles.dtype, device=particles.device)
    kg = torch.empty((n, int(dim)), dtype=particles.dtype, device=particles.device) if kernel_gradients else None
    bw = torch.empty(3, dtype=particles.dtype, device=particles.device)
    with _on(particles):
        rc = f(_ptr(particles), n, int(dim), ld, _ptr(workspace), _ptr(K), _ptr(kg), int(dim), _ptr(bw),
               _stream(particles))
    check(rc, "sgmcmc_svgd_kernel")
    return K, kg, bw
