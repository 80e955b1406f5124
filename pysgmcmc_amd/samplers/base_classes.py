"""Sampler base classes: the per-step driver behind ``next(sampler)``.

Mirror of ``pysgmcmc/samplers/base_classes.py`` (``MCMCSampler`` :17-310,
``BurnInMCMCSampler`` :313-456): same constructor keywords, same iterator
protocol, same return convention -- ``next(sampler) -> (sample, cost)`` where
``sample`` is theta AFTER the step (a list of arrays in the parameters' shapes, or
the bare array for a single parameter, :302-304) and ``cost`` is the cost at the
parameters BEFORE the step (the same ``cost`` tensor feeds ``tf.gradients`` and
is fetched next to the assign ops, :298-300).

What differs, by design:
  * ``params`` are torch tensors; the sampler re-points them to views of one flat
    device arena (``pysgmcmc_amd.arena.FlatArena``) and the whole update of all
    parameters is ONE fused HIP kernel launch (``pysgmcmc_amd.kernels``) instead
    of ~25 TF ops per tensor.
  * ``cost_fun(params)`` is evaluated every step with PyTorch-ROCm autograd (or,
    if it provides ``cost_and_grad(params, grad_views)``, writes gradients
    straight into the arena).
  * ``session`` is accepted for signature compatibility; a ``torch.device`` (or
    device string) there selects the GPU, anything else is ignored.
  * ``use_hip_graph`` (attribute): ``True`` captures the cost/gradient pipeline (the many small
    launches) into one hipGraph replayed every step, followed by a direct launch of the fused
    update (so stepsize, phase and Philox step stay by-value arguments); ``"full"`` captures the
    update kernel too (Philox step read from a device counter). See ``_step_graph``.
  * ``collect_stats`` (attribute, default True): the step kernel also reduces
    {sum theta^2, sum V^2, sum minv, sum minv^2} from registers (``sampler.stats``); a cost
    function that sets ``accepts_theta_sumsq`` gets sum theta^2 for free (BNN weight prior).
    ``"theta_sq"`` reduces only sum theta^2 (all a BNN pipeline consumes); ``False`` nothing.
  * ``attach_moments``: the chain's Welford moments ride in the update launch.
  * ``sample_format`` (attribute, not a constructor keyword so that the
    ``get_sampler`` keyword reflection stays identical to the reference):
    ``"numpy"`` (default, reference behaviour: one D2H copy of theta per step),
    ``"device"`` (cloned device tensors) or ``"view"`` (zero-copy views of the
    live arena; they change at the next step).
"""
import logging
import os

import numpy as np
import torch

from pysgmcmc_amd import kernels
from pysgmcmc_amd.arena import FlatArena
from pysgmcmc_amd.stepsize_schedules import ConstantStepsizeSchedule

__all__ = ("MCMCSampler", "BurnInMCMCSampler")

_DTYPES = {
    torch.float32: torch.float32, torch.float64: torch.float64,
    "float32": torch.float32, "float64": torch.float64,
    np.float32: torch.float32, np.float64: torch.float64,
    np.dtype("float32"): torch.float32, np.dtype("float64"): torch.float64,
    float: torch.float64,
}


def as_torch_dtype(dtype):
    try:
        return _DTYPES[dtype]
    except (KeyError, TypeError):
        pass
    name = getattr(dtype, "name", None)          # e.g. a tf.DType
    if name in ("float32", "float64"):
        return _DTYPES[name]
    raise AssertionError("unsupported sampler dtype: %r (float32 / float64)" % (dtype,))


def _pick_device(session, params):
    if isinstance(session, (torch.device, str, int)):
        return torch.device(session) if not isinstance(session, int) else torch.device("cuda", session)
    for p in params:
        if isinstance(p, torch.Tensor) and p.is_cuda:
            return p.device
    if torch.cuda.is_available():
        return torch.device("cuda", torch.cuda.current_device())
    # No GPU: construction still works (host logic is testable); stepping raises
    # SgmcmcLibraryError in pysgmcmc_amd.kernels -- there is no CPU update path.
    return torch.device("cpu")


class MCMCSampler(object):
    """Generic base class of all MCMC samplers (iterator protocol + step driver)."""

    # state rows (besides theta and grad) the subclass's kernel needs
    _STATE_ROWS = ()
    # "sghmc" / "sgld" / "rsghmc": the kernel can read its stepsize-derived scalars from a device block
    # (kernels.step_scalars); None: by value only
    _SCALARS_KIND = None
    MAX_STEPSIZE_GRAPHS = 4               # use_hip_graph='full' on a by-value stepsize kernel: graphs kept (least recently used one evicted)
    # every parameter starts at a multiple of this many elements in the arena rows (1 = dense)
    _PARAM_ALIGN = 1

    def __init__(self, params, cost_fun, batch_generator=None,
                 stepsize_schedule=ConstantStepsizeSchedule(0.01),
                 session=None, dtype=torch.float64, seed=None):
        # same checks as pysgmcmc/samplers/base_classes.py:73-89
        assert batch_generator is None or hasattr(batch_generator, "__next__")
        assert seed is None or isinstance(seed, int)
        assert callable(cost_fun)
        assert hasattr(stepsize_schedule, "update")
        assert hasattr(stepsize_schedule, "__next__")
        assert hasattr(stepsize_schedule, "initial_value")

        self.dtype = dtype
        self._torch_dtype = as_torch_dtype(dtype)
        self.n_iterations = 0
        self.seed = seed
        # Philox key: the user's seed, or fresh entropy when seed is None
        self._philox_seed = (int(seed) if seed is not None
                             else int.from_bytes(os.urandom(8), "little")) & 0xFFFFFFFFFFFFFFFF
        self.stepsize_schedule = stepsize_schedule
        self.batch_generator = batch_generator
        self.session = session
        self.sample_format = "numpy"
        # launch geometry of this chain's update kernel (kernels.LaunchConfig; None = library defaults)
        self.launch = None
        # pysgmcmc_amd.profiling.UpdateKernelTimer (or None): per-launch kernel timestamps of the update kernel
        self.kernel_timer = None

        params = list(params)
        for p in params:
            assert isinstance(p, torch.Tensor), "params must be torch tensors"
        self.params = params
        self.device = _pick_device(session, params)
        self.arena = FlatArena(params, self._STATE_ROWS, self._torch_dtype, self.device,
                               param_align=self._PARAM_ALIGN)
        for p in self.params:
            p.requires_grad_(True)

        # names of the parameters (the reference reads `param.name` of tf.Variables for trace variable
        # names, diagnostics/sample_chains.py:174-180); enumerated strings by default (:80-90)
        self.param_names = [str(i) for i in range(len(params))]
        self.cost_fun = cost_fun
        self.cost = None                      # last evaluated cost (device tensor)
        self.epsilon = self.stepsize_schedule.initial_value
        # flat (n_i, 1) views of theta, the role of `vectorize` (tensor_utils.py:92-95)
        self.vectorized_params = [v.view(-1, 1) for v in self.arena.views("theta")]
        self.theta_t = self.arena.views("theta")
        # injected-noise hook (tests / reproducing a trajectory): callable(step, n) -> flat tensor or None
        self.noise_source = None
        # fused step statistics (sum theta^2, sum V^2, sum minv, sum minv^2), reduced inside the kernel
        self.collect_stats = True
        self._stats = None
        self._stats_valid = False
        self._stats_out_valid = False
        # Welford moments folded into the update launch (attach_moments)
        self._moments, self._moments_every = None, 1
        # hipGraph mode
        self.use_hip_graph = False
        self._scalars_dev = None
        self._scalars_value = None
        self._graphs = {}
        self._full_graph_misses, self._full_graph_disabled = 0, False
        self._static_feeds = {}
        # The NEXT step's minibatch window gathered by THIS step's update launch (`sgmcmc_step_opts_t.gather_*`): a graph-stepped
        # chain then has no gather launch of its own (4.9 us of the 10 M-parameter step's 178). `prefetch_windows = False`
        # switches it off. `_pending_window`: (start, in_the_buffers) of a window drawn from the generator but not consumed yet.
        self.prefetch_windows = True
        self._pending_window = None
        self._fed_from_generator = False         # the current step's window came from the generator through the static buffers
        self._step_ctr = None
        self._ctr_value = -1
        self._capturing = False
        self._grad_decay = 0.0
        self._view_cache = None

    def _rebind_arena(self, storage):
        """Re-home this chain's state in ``storage`` (see ``FlatArena.rebind``) and refresh the cached views."""
        self.arena.rebind(storage)
        self.vectorized_params = [v.view(-1, 1) for v in self.arena.views("theta")]
        self.theta_t = self.arena.views("theta")
        self._graphs.clear()
        self._static_feeds.clear()
        self._stats_valid = False
        self._view_cache = None

    # ------------------------------------------------------------------ feeds
    def _next_batch(self):
        """Next ``{placeholder: value}`` dict, or ``{}`` without a generator. A window the previous step's update launch
        drew ahead (``_window_to_prefetch``) IS the generator's next batch: it is handed out here before any fresh draw."""
        if self._pending_window is not None:
            gen, (start, _) = self.batch_generator, self._pending_window
            self._pending_window = None
            B = gen.batch_size
            return {gen.x_placeholder: gen.x_dev[start:start + B], gen.y_placeholder: gen.y_dev[start:start + B].reshape(-1, 1)}
        if self.batch_generator is not None:
            return next(self.batch_generator)
        return dict()

    def _next_stepsize(self):
        self.epsilon = next(self.stepsize_schedule)
        return self.epsilon

    @staticmethod
    def _feed(feed_dict):
        for placeholder, value in feed_dict.items():
            if hasattr(placeholder, "feed"):
                placeholder.feed(value)

    # ------------------------------------------------------------------ cost
    def _step_stats(self):
        """StepStats buffers for the kernel (None on CPU state or when disabled)."""
        if not self.collect_stats or self.device.type != "cuda":
            return None
        if self._stats is None:
            self._stats = kernels.StepStats(self.arena.n, self.device)
        return self._stats

    def _ensure_stats(self):
        """Make the statistics workspace valid before the first step: sum theta^2 by K6, stored as a
        one-partial workspace (header nparts = 1), so consumers always read the same buffer the step
        kernels refresh afterwards."""
        st = self._step_stats()
        if st is not None and not self._stats_valid:
            s = kernels.summary(self.arena.row("theta"))
            st.workspace.view(torch.int64)[0:1].fill_(1)
            ws = st.workspace.view(torch.float64)
            ws[4:8].zero_()
            ws[4:5].copy_(s[1:2])
            self._stats_valid = True
            self._stats_out_valid = False
        return st

    @property
    def stats(self):
        """``{"theta_sq", "momentum_sq", "minv_sum", "minv_sq"}`` after the last step (host floats; K7)."""
        st = self._ensure_stats()
        if st is None:
            return None
        if not self._stats_out_valid:
            kernels.step_stats_finish(st)
            self._stats_out_valid = True
        v = st.out.cpu().numpy()
        return {"theta_sq": float(v[0]), "momentum_sq": float(v[1]), "minv_sum": float(v[2]), "minv_sq": float(v[3])}

    def _noise_args(self):
        """(seed, step, step_dev) of the in-register Philox stream; graph capture uses the device counter."""
        if self._capturing:
            return dict(seed=self._philox_seed, step=0, step_dev=self._step_ctr)
        return dict(seed=self._philox_seed, step=self.n_iterations, step_dev=None)

    def _cost_and_grad(self):
        """Evaluate cost at the current theta and leave d cost/d theta in the arena's grad row."""
        fused = getattr(self.cost_fun, "cost_and_grad", None)
        if fused is not None:
            cost = fused(self.params, self.arena.grad_views, **self._cost_kwargs())
            # a cost function may leave a term coef * theta of its gradient to the update kernel
            self._grad_decay = float(getattr(self.cost_fun, "grad_theta_coef", 0.0))
            return cost.detach() if isinstance(cost, torch.Tensor) else torch.as_tensor(cost)
        self._grad_decay = 0.0
        with torch.enable_grad():
            cost = self.cost_fun(self.params)
            if not isinstance(cost, torch.Tensor) or not cost.requires_grad:
                raise ValueError("cost_fun(params) must return a torch tensor that depends on params")
            grads = torch.autograd.grad(cost, self.params, grad_outputs=torch.ones_like(cost),
                                        allow_unused=True)
        with torch.no_grad():
            dst, src = [], []
            for view, g in zip(self.arena.grad_views, grads):
                if g is None:
                    view.zero_()
                else:
                    dst.append(view)
                    src.append(g.reshape(view.shape).to(view.dtype))
            if dst:
                torch._foreach_copy_(dst, src)
        return cost.detach()

    def _draw_noise_sample(self, sigma, shape):
        """``sigma * N(0, 1)`` of the given shape (``base_classes.py:199-220``), materialised by the K5
        Philox kernel from this chain's stream. The step kernels never call this: they draw the
        same stream in registers. Uses the reserved step index 2^62 + n_iterations."""
        n = 1
        for d in tuple(shape):
            n *= int(d)
        out = torch.empty(n, dtype=self._torch_dtype, device=self.device)
        kernels.philox_normal(out, self._philox_seed, (1 << 62) + self.n_iterations)
        return torch.as_tensor(sigma, dtype=self._torch_dtype, device=self.device) * out.reshape(tuple(shape))

    def _draw_noise(self):
        """Injected xi for this step, or None for the in-register Philox stream."""
        if self.noise_source is None:
            return None
        xi = self.noise_source(self.n_iterations, self.arena.n)
        if xi is None:
            return None
        xi = torch.as_tensor(xi, dtype=self._torch_dtype, device=self.device).reshape(-1).contiguous()
        assert xi.numel() == self.arena.n
        return xi

    # ------------------------------------------------------------------ output
    def _format_sample(self):
        fmt = self.sample_format
        if fmt == "numpy":
            # ONE D2H copy of the flat theta row into a fresh host array; the per-parameter arrays are
            # views of it (every step gets its own buffer, so earlier samples are never overwritten)
            theta = self.arena.row("theta").detach()
            if theta.is_cuda and theta.numel() * theta.element_size() >= (1 << 20):
                # large chains: D2H into page-locked memory (PyTorch caches and recycles these blocks;
                # 3-4x the rate of a pageable copy), the ndarray keeps the block alive
                host = torch.empty(theta.shape, dtype=theta.dtype, pin_memory=True)
                host.copy_(theta, non_blocking=True)
                torch.cuda.current_stream(theta.device).synchronize()
                flat = host.numpy()
            else:
                # .cpu() of a host tensor is the tensor itself: clone so the sample never aliases the arena
                flat = (theta.cpu() if theta.is_cuda else theta.clone()).numpy()
            out = [flat[o:o + s].reshape(shp)
                   for o, s, shp in zip(self.arena.offsets, self.arena.sizes, self.arena.shapes)]
        elif fmt == "device":
            out = [v.detach().clone() for v in self.arena.views("theta")]
        elif fmt == "view":
            # the views alias the arena and never change: build them once (50 particles = 50 tensor objects per step)
            if self._view_cache is None:
                self._view_cache = [v.detach() for v in self.arena.views("theta")]
            out = list(self._view_cache)
        else:
            raise ValueError("sample_format must be 'numpy', 'device' or 'view'")
        if len(out) == 1:
            out = out[0]                     # base_classes.py:302-304
        return out

    def _format_cost(self, cost):
        if self.sample_format == "numpy":
            return cost.detach().cpu().numpy()
        return cost

    # ------------------------------------------------------------------ step
    def _kernel_step(self, eps, xi, sl=None, opts=None):
        """Launch the update of the elements ``sl`` (a slice of the arena rows; None = all) with the step extras
        ``opts`` (a dict of ``kernels.StepOpts`` keywords or None)."""
        raise NotImplementedError

    def _bytes_per_element(self):
        """Bytes one update launch touches per parameter (mirrors the library's auto-geometry rule)."""
        return 6 * self.arena.row("theta").element_size()

    def _sliced_rows(self, names, sl):
        rows = [self.arena.row(k) for k in names]
        return rows if sl is None else [r[sl] for r in rows]

    def _stats_written(self):
        if self._stats is not None:
            self._stats_valid = True          # the workspace now holds this step's per-block records
            self._stats_out_valid = False     # K7 runs lazily (sampler.stats); the BNN head reads the records

    def _launch(self):
        """Launch configuration of the update kernel for THIS launch: the chain's geometry plus the timestamp events of an
        attached, enabled kernel timer (never inside a graph capture)."""
        base = self.launch
        t = self.kernel_timer
        if t is not None and t._current is not None:
            return t.launch_config(base)
        return base

    def _timed_kernel_step(self, eps, xi, sl=None, opts=None, tag=None):
        t = self.kernel_timer
        kw = {} if (sl is None and opts is None) else dict(sl=sl, opts=opts)
        if t is None or not t.enabled or self._capturing or not t.due(tag):
            return self._kernel_step(eps, xi, **kw)
        if t.device is None:
            t.device = self.device
        t.begin(tag)
        ok = False
        try:
            self._kernel_step(eps, xi, **kw)
            ok = True
        finally:
            t.end(launched=ok)

    # -- what one update launch carries besides the update --
    def attach_moments(self, moments, every=1):
        """Fold theta' into ``moments`` (a ``ChainMoments`` of this chain's dtype) on every ``every``-th step INSIDE the
        update launch (K4 fused into K1-K3: +16 B/param on those launches instead of a separate 20 B/param pass);
        ``moments.count`` advances with it. ``attach_moments(None)`` detaches."""
        if moments is not None:
            assert moments.n == self.arena.n and moments.mean.dtype == self._torch_dtype, \
                "moments must match the chain in length and dtype"
        self._moments, self._moments_every = moments, max(int(every), 1)

    def _moments_due(self):
        return self._moments is not None and (self.n_iterations + 1) % self._moments_every == 0

    def _update_opts(self, lo, hi, rec_base=0, rec_total=0, moments=None, sliced=False):
        o = {}
        if lo:
            o["first_element"] = int(lo)
        if rec_base or rec_total:
            o["stats_base"], o["stats_total"] = int(rec_base), int(rec_total)
        if self.collect_stats == "theta_sq":
            o["theta_sq_only"] = True
        if sliced and self._arena_is_hbm_resident():
            o["hbm_resident"] = True
        if self._skip_minv_store_now():
            o["skip_minv_store"] = True
        if moments is not None:
            o["moments"] = (moments.mean[lo:hi], moments.m2[lo:hi], moments.count)
        if self._capturing and self._scalars_dev is not None:
            o["scalars_dev"] = self._scalars_dev
        return o or None

    def _skip_minv_store_now(self):
        return False

    def _arena_is_hbm_resident(self):
        return self.arena.n * self._bytes_per_element() > (640 << 20)

    def _update(self, eps, xi):
        """The whole update of this step as ONE launch (eager stepping, the plain graph modes)."""
        moments = None
        if self._moments_due() and not self._capturing:
            moments = self._moments
        if self._SCALARS_KIND is None:
            # a sampler whose kernel takes no step extras (SVGD): plain launch, the Welford pass on its own (K4)
            self._timed_kernel_step(eps, xi, tag=(self.n_iterations, 0, self.arena.n))
            if moments is not None:
                moments.update(self.arena.row("theta"))
            return
        if moments is not None:
            moments.count += 1
        # (a captured launch replays its arguments and cannot carry the by-value Welford count: see _step_graph_full)
        opts = self._update_opts(0, self.arena.n, moments=moments)
        gather = self._window_to_prefetch()
        if gather is not None:
            opts = dict(opts or {}, gather=gather)
        self._timed_kernel_step(eps, xi, opts=opts, tag=(self.n_iterations, 0, self.arena.n))

    def _window_to_prefetch(self):
        """(X, y, start, x_out, y_out) of the NEXT step's minibatch window if this step's update launch should gather it: the
        chain is fed by a window generator through static feed buffers (the hipGraph stepping modes, ``BNNCost``'s eager mode),
        this step took its window from the generator, and the launch takes per-step arguments (not being captured). The draw
        is the generator's next one, made one step early; ``state_dict`` keeps it."""
        gen = self.batch_generator
        if (not self.prefetch_windows or self._capturing or self.device.type != "cuda" or self._pending_window is not None
                or not self._fed_from_generator or not hasattr(gen, "next_starts")):
            return None
        bx, by = self._static_feeds.get(gen.x_placeholder), self._static_feeds.get(gen.y_placeholder)
        if bx is None or by is None or gen.x_placeholder.value is not bx:
            return None
        X, y = gen.x_dev, gen.y_dev.reshape(-1)
        if X.dtype != self._torch_dtype or not kernels.gather_fits_step_launch(X, y, 0, bx, by.reshape(-1)):
            return None
        start = int(gen.next_starts(1)[0])
        if not kernels.gather_fits_step_launch(X, y, start, bx, by.reshape(-1)):      # (a misaligned source window)
            self._pending_window = (start, False)
            return None
        self._pending_window = (start, True)
        return (X, y, start, bx, by.reshape(-1))

    def _step(self, feed_dict):
        assert (feed_dict is None or hasattr(feed_dict, "update"))
        if feed_dict is None:
            feed_dict = dict()
        if self.use_hip_graph and self.noise_source is None and self.device.type == "cuda":
            return self._step_graph(feed_dict)
        eps = None
        if self.device.type == "cuda" and getattr(self.cost_fun, "wants_static_feeds", False):
            # a cost function whose arithmetic depends on being fed through its own buffers (BNNCost: the first layer's bias
            # gradient from the [x | 1]^T delta product) gets them in eager stepping too: eager == hipGraph stepping, bit for bit
            self._feed_static(feed_dict)
            eps = self._next_stepsize()
        else:
            feed_dict.update(self._next_batch())
            eps = self._next_stepsize()
            self._feed(feed_dict)
        cost = self._cost_and_grad()          # U(theta_{t-1}) and its gradient
        self.cost = cost
        with torch.no_grad():
            self._update(eps, self._draw_noise())
        return self._finish_step(cost)

    def _finish_step(self, cost):
        sample = self._format_sample()        # theta_t
        cost_out = self._format_cost(cost)
        self.stepsize_schedule.update(sample, cost_out)
        self.n_iterations += 1
        return sample, cost_out

    # ------------------------------------------------------------------ hipGraph mode
    def _graph_key(self):
        return ("full",)

    def _feed_static(self, feed_dict):
        """Next minibatch into the STATIC feed buffers a captured cost pipeline reads."""
        gen = self.batch_generator
        if (not feed_dict and hasattr(gen, "next_starts") and gen.x_dev.is_cuda
                and gen.x_placeholder in self._static_feeds and gen.y_placeholder in self._static_feeds
                and gen.x_dev.dtype == gen.y_dev.dtype == self._static_feeds[gen.x_placeholder].dtype):
            # window generator with static feed buffers in place: the next window goes into them with ONE launch
            # (same RandomState draw as next(generator)); otherwise two slice copies below
            bx, by = self._static_feeds[gen.x_placeholder], self._static_feeds[gen.y_placeholder]
            self._fed_from_generator = True
            pending, self._pending_window = self._pending_window, None
            if pending is None:
                kernels.window_gather(gen.x_dev, gen.y_dev.reshape(-1), int(gen.next_starts(1)[0]), bx, by)
            elif not pending[1]:                  # drawn already, but the buffers were used for something else since
                kernels.window_gather(gen.x_dev, gen.y_dev.reshape(-1), pending[0], bx, by)
            # (else: the previous step's update launch put this window into the buffers)
            gen.x_placeholder.value, gen.y_placeholder.value = bx, by
        else:
            self._fed_from_generator = False
            feed_dict.update(self._next_batch())      # (a window drawn one step ahead is handed out first)
        for placeholder, value in feed_dict.items():
            if not hasattr(placeholder, "feed"):
                continue
            value = placeholder.feed(value).value
            buf = self._static_feeds.get(placeholder)
            if buf is None or buf.shape != value.shape or buf.dtype != value.dtype:
                # the cost function may bring its own buffer (BNNCost: x pitched, with a column of ones behind the data)
                make = getattr(self.cost_fun, "static_feed_buffer", None)
                buf = make(placeholder, value) if make is not None else None
                if buf is None:
                    buf = value.clone()
                else:
                    buf.copy_(value)
                self._static_feeds[placeholder] = buf
                self._graphs.clear()
            else:
                buf.copy_(value)                  # (one torch._foreach_copy_ for all feeds measured slower: 246 vs 236 us/step)
            placeholder.value = buf

    def _step_graph(self, feed_dict):
        """One step with the launch-bound part replayed from hipGraphs.

        ``use_hip_graph = True``: the graph holds cost + gradient into the arena; the fused update is launched
        directly (stepsize / phase / Philox step by value, HIP events can time it).
        ``use_hip_graph = "full"``: ONE graph per phase (burn-in / frozen) also holds the update: the Philox step comes
        from a device counter and the stepsize-derived scalars from a device block refreshed by a tiny direct launch
        whenever the schedule moves (``StepOpts.scalars_dev``), so a SCHEDULED stepsize replays the same graph.
        Feeds are copied into static buffers first. Requirements: static feed shapes; a cost function without host
        synchronisation. (Round 3's two further modes -- the update on a side stream under the backward GEMMs, and the
        update as the epilogue of a hand-written weight-gradient GEMM -- were measured slower / no faster
        (``profiles/r03_overlap_probe.txt``, ``r03_gemm_fusion_probe.txt``) and were removed in round 5.)"""
        self._feed_static(feed_dict)
        eps = self._next_stepsize()
        self._ensure_stats()
        if self.use_hip_graph == "full":
            return self._step_graph_full(eps)
        entry = self._graphs.get(("cost",))
        if entry is None:
            entry = self._graphs[("cost",)] = self._capture_cost()
        graph, cost = entry
        with torch.no_grad():
            graph.replay()
            self._update(eps, None)
        self.cost = cost
        return self._finish_step(cost)

    def _step_graph_full(self, eps):
        if self._step_ctr is None:
            self._step_ctr = torch.zeros(1, dtype=torch.int64, device=self.device)
        if self._scalars_dev is None and self._SCALARS_KIND is not None:
            self._scalars_dev = torch.zeros(8, dtype=self._torch_dtype, device=self.device)
            self._scalars_value = None
        if self._ctr_value != self.n_iterations:
            self._step_ctr.fill_(self.n_iterations)
            self._ctr_value = self.n_iterations
        key = self._graph_key()
        if self._SCALARS_KIND is None:
            # a sampler whose kernel takes its stepsize by value only (SVGD): one graph per stepsize, the MAX_STEPSIZE_GRAPHS most
            # recently used ones are kept (a cyclic schedule of a few values keeps replaying them). A schedule that keeps MOVING
            # would capture a graph per step: after 2 * MAX_STEPSIZE_GRAPHS captures in a row without one replay of a kept graph
            # the sampler steps with the cost graph + direct update from then on. The caller's `use_hip_graph` stays what it
            # was set to (state_dict / introspection agree with the configuration); `_full_graph_disabled` holds the decision.
            self._scalars_dev = None
            key = key + (float(eps),)
            if not self._full_graph_disabled and key not in self._graphs:
                self._full_graph_misses += 1
                if self._full_graph_misses > 2 * self.MAX_STEPSIZE_GRAPHS:
                    logging.warning("pysgmcmc_amd: use_hip_graph='full' met %d new stepsizes in a row on a sampler whose kernel takes "
                                    "the stepsize by value; stepping with the cost graph + direct update from here on",
                                    self._full_graph_misses)
                    self._full_graph_disabled = True
                    for k in [k for k in self._graphs if k[:1] == ("full",)]:
                        del self._graphs[k]
                else:
                    kept = [k for k in self._graphs if k[:1] == ("full",)]      # dict order = least recently used first
                    for k in kept[:max(len(kept) - self.MAX_STEPSIZE_GRAPHS + 1, 0)]:
                        del self._graphs[k]
            elif not self._full_graph_disabled:
                self._full_graph_misses = 0
                self._graphs[key] = self._graphs.pop(key)            # most recently used last
            if self._full_graph_disabled:
                entry = self._graphs.get(("cost",))
                if entry is None:
                    entry = self._graphs[("cost",)] = self._capture_cost()
                graph, cost = entry
                with torch.no_grad():
                    graph.replay()
                    self._update(eps, None)
                self.cost = cost
                return self._finish_step(cost)
        else:
            scal = self._step_scalars(eps)
            if scal != self._scalars_value:       # the schedule moved: refresh the device block (1-thread launch)
                kernels.step_scalars(self._SCALARS_KIND, self._scalars_dev, *scal)
                self._scalars_value = scal
        entry = self._graphs.get(key)
        if entry is None:
            entry = self._graphs[key] = self._capture_full(eps)
        graph, cost = entry
        graph.replay()
        self._ctr_value += 1
        if self._moments_due():
            # a captured launch replays its arguments, the Welford count cannot ride in it: separate K4 launch
            self._moments.update(self.arena.row("theta"))
        self.cost = cost
        return self._finish_step(cost)

    def _warm_cost(self):
        cur = torch.cuda.current_stream(self.device)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):                 # warm-up: cost only, no state change
            self._cost_and_grad()
        cur.wait_stream(side)

    def _cost_kwargs(self):
        kw = {}
        if getattr(self.cost_fun, "accepts_theta_sumsq", False):
            st = self._ensure_stats()
            if st is not None:
                kw["theta_sumsq_partials"] = st.workspace
        return kw

    def _capture_cost(self):
        """Capture the cost/gradient pipeline into one graph. Returns (graph, cost)."""
        self._warm_cost()
        graph = torch.cuda.CUDAGraph()
        # thread_local: other threads (e.g. the RCCL watchdog of a multi-chain job) may keep calling
        # HIP while this thread captures
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            cost = self._cost_and_grad()
        return graph, cost

    def _capture_full(self, eps):
        self._warm_cost()
        graph = torch.cuda.CUDAGraph()
        self._capturing = True
        try:
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                cost = self._cost_and_grad()
                with torch.no_grad():
                    self._update(eps, None)
                kernels.counter_add(self._step_ctr, 1)
        finally:
            self._capturing = False
        return graph, cost

    # iterator protocol, base_classes.py:226-310
    def __iter__(self):
        return self

    def __next__(self, feed_dict=None):
        return self._step(feed_dict)

    # ------------------------------------------------------------------ state
    def state_dict(self):
        """Everything needed to resume the chain bit-exactly (the reference cannot checkpoint): the arena, the step
        counter and Philox key, and -- when they can report it -- the position of the minibatch generator's window
        stream and of the stepsize schedule."""
        state = {"arena": self.arena.state_dict(), "n_iterations": self.n_iterations,
                 "philox_seed": self._philox_seed, "epsilon": self.epsilon}
        for key, obj in (("batch_generator", self.batch_generator), ("stepsize_schedule", self.stepsize_schedule)):
            if hasattr(obj, "state_dict"):
                state[key] = obj.state_dict()
        if self._pending_window is not None:      # the generator stands one draw ahead: the window the next step will take
            state["pending_window"] = int(self._pending_window[0])
        return state

    def load_state_dict(self, state):
        self.arena.load_state_dict(state["arena"])
        self._stats_valid = False
        self.n_iterations = int(state["n_iterations"])
        self._philox_seed = int(state["philox_seed"])
        self.epsilon = state["epsilon"]
        for key, obj in (("batch_generator", self.batch_generator), ("stepsize_schedule", self.stepsize_schedule)):
            if key in state and hasattr(obj, "load_state_dict"):
                obj.load_state_dict(state[key])
        self._pending_window = (int(state["pending_window"]), False) if state.get("pending_window") is not None else None


class BurnInMCMCSampler(MCMCSampler):
    """Base class of samplers that adapt a diagonal preconditioner (``minv``)
    during the first ``burn_in_steps`` steps and freeze it afterwards.

    Burn-in switch, pysgmcmc/samplers/base_classes.py:432-456: steps
    ``0 .. burn_in_steps-1`` adapt; from step ``burn_in_steps`` on the captured
    ``minv`` is used unchanged. ``burn_in_steps <= 0`` never freezes (nothing is
    ever captured to feed, :449) => perpetual adaptation, replicated here.
    """

    _STATE_ROWS = ("tau", "g", "v_hat", "minv")

    def __init__(self, params, cost_fun, batch_generator=None,
                 stepsize_schedule=ConstantStepsizeSchedule(0.01),
                 burn_in_steps=3000,
                 session=None, dtype=torch.float64, seed=None):
        assert isinstance(burn_in_steps, int)
        super().__init__(params=params, cost_fun=cost_fun,
                         stepsize_schedule=stepsize_schedule,
                         batch_generator=batch_generator,
                         seed=seed, dtype=dtype, session=session)
        self.burn_in_steps = burn_in_steps
        # initial statistics, sghmc.py:126-149 / sgld.py:117-141
        for name in ("tau", "g", "v_hat", "minv"):
            self.arena.fill(name, 1.0)
        self.minv_t = self.arena.views("minv")
        # Set True to also materialise r = 1/(tau+1) like the reference's R_i variable
        # (+4 B/param of traffic; r is derivable from tau).
        self.materialize_r = False
        # The reference captures minv on every burn-in step (base_classes.py:438-441), yet only the LAST burn-in step's
        # value is ever consumed (it is what the frozen steps are fed, :449-454). False: burn-in steps other than the
        # last do not write minv (44 instead of 48 B/param for SGHMC, 36 instead of 40 for SGLD: -8 % per burn-in step at
        # 49.8 M parameters, profiles/r03_adapt_sweep.txt); the chain is bit-identical, and `.minv` read DURING burn-in
        # is then computed from the current v_hat (the preconditioner the next step will use) instead of read back.
        self.store_minv_every_step = True
        self._minv_summary = None

    @property
    def is_burning_in(self):
        return self.n_iterations < self.burn_in_steps

    @property
    def _adapting(self):
        return self.is_burning_in or self.burn_in_steps <= 0

    def _graph_key(self):
        return ("full", bool(self._adapting), self._skip_minv_store_now())

    def _skip_minv_store_now(self):
        """This step adapts but need not write minv: not the last burn-in step (perpetual adaptation never freezes)."""
        if self.store_minv_every_step or not self._adapting:
            return False
        return self.burn_in_steps <= 0 or self.n_iterations != self.burn_in_steps - 1

    @property
    def minv(self):
        """Adapted inverse mass, one ``(n_i, 1)`` ndarray per parameter (base_classes.py:438-441)."""
        if not self.store_minv_every_step and self._adapting and self.n_iterations > 0:
            # burn-in steps did not write minv: form it from the statistics (tensor_utils.py:269,319-323)
            from pysgmcmc_amd.tensor_utils import safe_divide, safe_sqrt
            vh = self.arena.row("v_hat")
            flat = safe_divide(torch.ones_like(vh), safe_sqrt(vh))
            return [flat[o:o + n].reshape(-1, 1).cpu().numpy() for o, n in zip(self.arena.offsets, self.arena.sizes)]
        return [v.detach().reshape(-1, 1).cpu().numpy() for v in self.arena.views("minv")]

    @property
    def minv_summary(self):
        """``{"mean","std","min","max"}`` of the preconditioner, by the K6 reduction kernel."""
        s = kernels.summary(self.arena.row("minv")).cpu().numpy()
        n = float(self.arena.n)
        mean = s[0] / n
        var = max(s[1] / n - mean * mean, 0.0)
        return {"mean": mean, "std": var ** 0.5, "min": s[2], "max": s[3]}

    def _r_row(self):
        if not self.materialize_r:
            return None
        if "r" not in self.arena._rows:
            self.arena._rows["r"] = torch.full((self.arena.n,), 0.5, dtype=self._torch_dtype, device=self.device)
        return self.arena._rows["r"]
