"""The two stepping modes of round 3 that were measured and lost, kept runnable OUTSIDE the product
(``pysgmcmc_amd.samplers`` has four modes: eager, cost graph, full graph, fused small-model steps):

  * ``overlap_update``: the update of every finished slice of the arena on a side stream under the remaining backward GEMMs
    -- bit-identical chain, 239-246 vs 199 us per step at 10 M parameters (``profiles/r03_overlap_probe.txt``);
  * ``fuse_update_into_gemm``: the frozen SGHMC update as the epilogue of a hand-written fp32 matrix-core weight-gradient GEMM
    (``csrc/sgmcmc_gemm.hip``) -- bit-exact against K1 on the gradient it formed, a draw in the pipeline
    (``profiles/r03_gemm_fusion_probe.txt``).

Use ``experimental(SGHMCSampler)`` (etc.) with a ``HookedBNNCost``; ``tools/experiments/test_experiments_gpu.py`` keeps the
bit-equality tests. Needs ``libsgmcmc_hip_experiments.so`` (``make -C tools/experiments``) for the GEMM mode."""
import torch

from pysgmcmc_amd import kernels
from pysgmcmc_amd.models.bayesian_neural_network import BNNCost
from tools.experiments import gemm_kernels


class HookedBNNCost(BNNCost):
    """``BNNCost`` whose HIP pipeline announces finished layer gradients (``cost_and_grad_iter``) and lets the caller take over
    a layer's weight-gradient product (``weight_update``) or run it on the experiment GEMM (``gw_gemm = "mfma"``)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        # weight-gradient products gW = h^T delta: "blas" (rocBLAS / hipBLASLt through torch) or "mfma" (gemm_kernels.gemm_tn,
        # the product the fused GEMM + update kernel forms: same bits as that kernel's gradient)
        self.gw_gemm = "blas"
        # the round-3 pipeline as it was measured: library / experiment products on dense operands (no pitched feed buffer with
        # the bias gradient's column of ones, no fused backward products -- both came after these experiments)
        self.bias_gradient_from_product = False
        self.fused_dense_backward = False

    @torch.no_grad()
    def cost_and_grad(self, params, grad_views, theta_sumsq=None, theta_sumsq_partials=None, weight_update=None):
        assert self.use_hip_kernels, "the hooks belong to the HIP path"
        gen = self._hip_pipeline(params, grad_views, theta_sumsq, theta_sumsq_partials, weight_update)
        while True:
            try:
                next(gen)
            except StopIteration as stop:
                return stop.value

    # layers with at least this many weights announce their finished gradient (grad_ready_points): the sampler may
    # then update that slice of the arena on a side stream under the rest of the backward pass
    OVERLAP_MIN_WEIGHTS = 1 << 18

    def grad_ready_points(self, params):
        """Parameter indices p, in the order the backward pass reaches them, at which the gradients of ``params[p:]``
        are complete and their values are no longer read: after the weight-gradient GEMM of hidden layer l (p = 2 l),
        for every layer but the first (its gradient is the last thing the pipeline computes)."""
        n_layers = (len(params) - 1) // 2
        L = n_layers - 1
        single_out = params[2 * L].shape[1] == 1 and n_layers >= 2
        top = L - 1 if single_out else L
        return [2 * l for l in range(top, 0, -1) if params[2 * l].numel() >= self.OVERLAP_MIN_WEIGHTS]

    def cost_and_grad_iter(self, params, grad_views, theta_sumsq=None, theta_sumsq_partials=None):
        """``cost_and_grad`` as a generator: yields p at every point of ``grad_ready_points`` (in that order) and
        returns the cost (``StopIteration.value``). HIP path only; advance it under ``torch.no_grad()`` (a generator
        cannot hold a grad-mode context across its yields)."""
        return self._hip_pipeline(params, grad_views, theta_sumsq, theta_sumsq_partials)

    def _hip_pipeline(self, params, grad_views, theta_sumsq, theta_sumsq_partials=None, weight_update=None):
        """``weight_update(l, h_in, delta_l)`` (optional): called where the weight-gradient product of hidden layer ``l``
        would be issued -- every gradient that follows W_l in the arena up to the next layer's weights is complete and W_l is
        no longer read. If it returns True the caller has consumed the product itself (the sampler's GEMM + update kernel,
        ``kernels.gemm_tn_sghmc``) and ``grad_views[2 l]`` is NOT written."""
        from pysgmcmc_amd import kernels
        X, Y = self.x_placeholder.value, self.y_placeholder.value
        B = X.shape[0]
        n_layers = (len(params) - 1) // 2
        ws = self._buffers(params, B)
        hs, ds = ws["h"], ws["d"]
        L = n_layers - 1
        single_out = params[2 * L].shape[1] == 1 and n_layers >= 2
        ready = set(self.grad_ready_points(params))
        # forward; a single-output last layer is a plain GEMV whose bias the loss head adds
        h = X
        fuse_top = single_out and self.fuse_tanh_rowdot
        # loss head folded into the last layer's backward (one launch less): needs the sum(theta^2) records of the
        # previous step kernel, which the rowdot launch reduces to 16 slices on the side
        fuse_head = fuse_top and self.fuse_head and theta_sumsq_partials is not None
        for l in range(n_layers):
            W, b = params[2 * l], params[2 * l + 1]
            if l == L and single_out:
                if not fuse_top:
                    torch.mv(h, W.view(-1), out=hs[l].view(-1))
            elif l < L:
                # hidden layer: plain product, the bias rides in the activation launch (the library's plain GEMM is 1.4-2.1 us
                # faster than its bias-epilogue one at batch 256: tools/fwd_gemm_probe.py)
                torch.mm(h, W, out=hs[l])
            else:
                torch.addmm(b, h, W, out=hs[l])
            if l == L - 1 and fuse_top:
                # bias + tanh of the last hidden layer and the output unit's dot product in one launch
                kernels.tanh_rowdot(hs[l], params[2 * L].view(-1), hs[L].view(-1),
                                    stats_workspace=theta_sumsq_partials if fuse_head else None,
                                    tsq_parts=ws["tsq_parts"] if fuse_head else None, bias=b.view(-1))
            elif l < L:
                kernels.bias_tanh(hs[l], b.view(-1))
            h = hs[l]
        n_params = float(sum(p.numel() for p in params))
        if theta_sumsq is None and theta_sumsq_partials is None:
            theta_sumsq = torch.zeros((), dtype=torch.float64, device=X.device)
            for p in params:
                theta_sumsq = theta_sumsq + (p.double() ** 2).sum()
        prior_coef = self.wdecay / ((n_params + 3e-16) * self.n_examples)
        self.grad_theta_coef = prior_coef if self.fold_prior else 0.0
        beta = 0.0 if self.fold_prior else prior_coef
        if fuse_head:
            # loss head + gW_L + delta_{L-1} (incl. tanh') + gb_{L-1} + gb_L + d/d log_var in ONE launch
            kernels.bnn_head_last_layer_backward(
                hs[L].view(-1), Y.reshape(-1), params[-1], ws["tsq_parts"], params[2 * L + 1], self.batch_size,
                self.n_examples, n_params, self.wdecay, self.prior_mean, self.prior_var, params[2 * L].view(-1), hs[L - 1],
                params[2 * (L - 1) + 1], beta, ws["cost"], grad_views[-1], grad_views[2 * L + 1], ws["mse"], ds[L - 1],
                grad_views[2 * (L - 1) + 1], grad_views[2 * L].view(-1), fold_prior_grad=self.fold_prior, add_last_bias=True)
        else:
            # loss head: delta_L, cost, d/d log_var, mse and (single-output net) the last bias gradient
            kernels.bnn_head(hs[L].view(-1), Y.reshape(-1), params[-1], theta_sumsq, self.batch_size, self.n_examples,
                             n_params, self.wdecay, self.prior_mean, self.prior_var,
                             ds[L].view(-1), ws["cost"], grad_views[-1], ws["mse"], fold_prior_grad=self.fold_prior,
                             stats_workspace=theta_sumsq_partials,
                             last_bias=params[2 * L + 1] if single_out else None,
                             grad_last_bias_out=grad_views[2 * L + 1] if single_out else None,
                             add_last_bias=single_out)
        self.last_mse = ws["mse"]
        for l in range(L, -1, -1):
            h_in = X if l == 0 else hs[l - 1]
            W, b = params[2 * l], params[2 * l + 1]
            if l == L and fuse_head:
                continue
            if l == L and single_out:
                # gW_L, delta_{L-1} (incl. tanh') and gb_{L-1} in one launch
                kernels.bnn_last_layer_backward(ds[l].view(-1), W.view(-1), hs[l - 1], ds[l - 1],
                                                grad_views[2 * (l - 1) + 1], grad_views[2 * l].view(-1),
                                                bias_prev=params[2 * (l - 1) + 1], beta=beta)
                continue
            # delta_{l-1} = delta_l W_l^T FIRST: it is the last reader of W_l, so once gW_l exists (next GEMM) the layer's
            # slice of the arena may be updated while the rest of the backward pass runs
            if l > 0:
                torch.mm(ds[l], W.t(), out=ds[l - 1])
            # gW_l = h_{l-1}^T delta_l written directly into the gradient arena. The weight-prior term
            # coef * theta is added by the update kernel (fold_prior) or rides in the GEMM epilogue (beta).
            if weight_update is not None and weight_update(l, h_in, ds[l]):
                pass                                          # product + update of this layer's slice done by the sampler's kernel
            elif (self.gw_gemm == "mfma" and self.fold_prior and W.dtype == torch.float32 and W.shape[1] % 128 == 0
                  and W.shape[0] % 4 == 0 and h_in.shape[0] % 16 == 0 and grad_views[2 * l].data_ptr() % 16 == 0):
                gemm_kernels.gemm_tn(h_in, ds[l], grad_views[2 * l])   # the library's own fp32 matrix-core product (k-ordered fmaf chain)
            elif self.fold_prior:
                torch.mm(h_in.t(), ds[l], out=grad_views[2 * l])
            else:
                torch.addmm(W, h_in.t(), ds[l], beta=prior_coef, alpha=1.0, out=grad_views[2 * l])
            if l == L:
                # bias gradient of a multi-output last layer (hidden layers get theirs from the fused kernel below)
                if self.fold_prior:
                    torch.mv(ds[l].t(), ws["ones"], out=grad_views[2 * l + 1])
                else:
                    torch.addmv(b, ds[l].t(), ws["ones"], beta=prior_coef, alpha=1.0, out=grad_views[2 * l + 1])
            if 2 * l in ready:
                yield 2 * l                                   # gradients of params[2 l:] complete, values no longer read
            if l > 0:
                # delta_{l-1} *= 1 - h_{l-1}^2, and gb_{l-1} = column sums of the result (+ beta * b_{l-1})
                kernels.tanh_backward_colsum(ds[l - 1], hs[l - 1], grad_views[2 * (l - 1) + 1],
                                             bias=params[2 * (l - 1) + 1], beta=beta)
        return ws["cost"].reshape(())



class ExperimentalStepping(object):
    """Mixin in front of a product sampler class: adds ``overlap_update`` and (SGHMC) ``fuse_update_into_gemm``."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.overlap_update = False
        self.fuse_update_into_gemm = False
        self._fused_plan = None
        self._slice_plan = None
        self._slice_launch = None
        self._side_stream = None

    def _rebind_arena(self, storage):
        super()._rebind_arena(storage)
        self._slice_plan = None
        self._fused_plan = None

    def _launch(self):
        base = self._slice_launch if self._slice_launch is not None else self.launch
        t = self.kernel_timer
        if t is not None and t._current is not None:
            return t.launch_config(base)
        return base

    def _step_graph(self, feed_dict):
        if self.use_hip_graph == "full" or not (self.overlap_update or self.fuse_update_into_gemm):
            return super()._step_graph(feed_dict)
        self._feed_static(feed_dict)
        eps = self._next_stepsize()
        self._ensure_stats()
        if self.fuse_update_into_gemm and self._fused_plan is not False and not getattr(self, "_adapting", False) \
                and not self._moments_due() and hasattr(self, "_fused_gemm_plan"):
            done = self._step_graph_fused_gemm(eps)
            if done is not None:
                return done
        entry = self._graphs.get(("cost_segments",))
        if entry is None:
            entry = self._graphs[("cost_segments",)] = self._capture_cost_segments()
        segments, cost = entry
        with torch.no_grad():
            if len(segments) == 1:
                segments[0][0].replay()
                self._update(eps, None)
            else:
                self._replay_overlapped(segments, eps)
        self.cost = cost
        return self._finish_step(cost)

    def _step_graph_fused_gemm(self, eps):
        """Frozen step whose update rides in the weight-gradient GEMMs (see ``fuse_update_into_gemm``): ONE graph holds the
        cost pipeline, the fused GEMM + update launches (Philox step from the device counter) and the counter increment.
        Returns None when the model does not fit the fused kernel (the caller then steps the usual way)."""
        if self._fused_plan is None:
            self._fused_plan = self._fused_gemm_plan() or False
            if self._fused_plan is False:
                return None
        plan, total = self._fused_plan
        if self._step_ctr is None:
            self._step_ctr = torch.zeros(1, dtype=torch.int64, device=self.device)
        if self._ctr_value != self.n_iterations:
            self._step_ctr.fill_(self.n_iterations)
            self._ctr_value = self.n_iterations
        key = ("fused_gemm", float(eps))
        entry = self._graphs.get(key)
        if entry is None:
            self._warm_cost()
            graph = torch.cuda.CUDAGraph()
            self._fused_grad_decay = float(getattr(self.cost_fun, "grad_theta_coef", 0.0))    # set by the warm-up evaluation
            with torch.cuda.graph(graph, capture_error_mode="thread_local"), torch.no_grad():
                cost = self.cost_fun.cost_and_grad(self.params, self.arena.grad_views,
                                                   weight_update=self._fused_weight_update(plan, total, eps), **self._cost_kwargs())
                kernels.counter_add(self._step_ctr, 1)
            cost = cost.detach() if isinstance(cost, torch.Tensor) else torch.as_tensor(cost)
            entry = self._graphs[key] = (graph, cost)
        graph, cost = entry
        graph.replay()
        self._ctr_value += 1
        self._stats_written()
        self.cost = cost
        return self._finish_step(cost)

    def _plan_slices(self, ready_points):
        """Arena slices of an overlapped step. ``ready_points`` = parameter indices p (descending): when the cost
        pipeline reaches that point, the gradients of params[p:] are complete and their values no longer read.
        Returns [(lo, hi, record_base)] in launch order plus the record total, or None when a boundary is not
        quad-aligned (slices must start on a Philox quad)."""
        a = self.arena
        bt = self._slice_block_threads()
        cfg = kernels.LaunchConfig(block_threads=bt,
                                   **{k: v for k, v in (self.launch.as_dict() if self.launch is not None else {}).items()
                                      if k != "block_threads"})
        bounds = [a.n] + [int(a.offsets[p]) for p in ready_points] + [0]
        if any(b % 4 for b in bounds[1:]) or sorted(set(bounds), reverse=True) != bounds:
            return None
        spans = [(bounds[i + 1], bounds[i]) for i in range(len(bounds) - 1)]      # launch order: high addresses first
        blocks = [kernels.step_stats_records(hi - lo, cfg) for lo, hi in spans]
        total = sum(blocks)
        plan, base = [], total
        for (lo, hi), nb in zip(spans, blocks):
            base -= nb                                                           # records in memory order
            plan.append((lo, hi, base))
        return plan, total, cfg

    def _slice_block_threads(self):
        if self.launch is not None and self.launch.as_dict()["block_threads"] > 0:
            return self.launch.as_dict()["block_threads"]
        return 128 if self._arena_is_hbm_resident() else 256

    def _capture_cost_segments(self):
        """Capture the cost/gradient pipeline: one graph, or -- overlapped update -- one graph per segment between
        the points where a slice of the gradient is complete. Returns ([(graph, slice or None)], cost)."""
        self._warm_cost()
        iter_fn = getattr(self.cost_fun, "cost_and_grad_iter", None) if self.overlap_update else None
        plan = None
        if iter_fn is not None:
            points = list(self.cost_fun.grad_ready_points(self.params))
            plan = self._plan_slices(points) if points else None
        if plan is None:
            graph = torch.cuda.CUDAGraph()
            # thread_local: other threads (e.g. the RCCL watchdog of a multi-chain job) may keep calling
            # HIP while this thread captures
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                cost = self._cost_and_grad()
            return [(graph, None)], cost
        spans, total, cfg = plan
        self._slice_plan = (spans, total, cfg)
        gen = iter_fn(self.params, self.arena.grad_views, **self._cost_kwargs())
        segments, cost, pool = [], None, None
        for k in range(len(spans)):
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, pool=pool, capture_error_mode="thread_local"), torch.no_grad():
                try:
                    next(gen)
                except StopIteration as stop:
                    cost = stop.value
            pool = graph.pool()
            segments.append((graph, spans[k]))
        assert cost is not None, "cost_and_grad_iter must yield exactly once per grad_ready_points() entry"
        self._grad_decay = float(getattr(self.cost_fun, "grad_theta_coef", 0.0))
        cost = cost.detach() if isinstance(cost, torch.Tensor) else torch.as_tensor(cost)
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=self.device)
            self._fork_events = [torch.cuda.Event() for _ in range(len(spans))]
            self._join_event = torch.cuda.Event()
        return segments, cost

    def _replay_overlapped(self, segments, eps):
        """graph segment -> update of the slice whose gradient it completed, on the side stream -> next segment ...;
        the last slice (nothing left to hide under) runs on the main stream, which then waits for the side stream."""
        spans, total, cfg = self._slice_plan
        moments = None
        if self._moments_due():
            moments = self._moments
            moments.count += 1
        main = torch.cuda.current_stream(self.device)
        side = self._side_stream
        self._slice_launch = cfg
        try:
            last = len(segments) - 1
            for k, (graph, (lo, hi, base)) in enumerate(segments):
                graph.replay()
                opts = self._update_opts(lo, hi, base, total, moments=moments, sliced=True)
                if k < last:
                    self._fork_events[k].record(main)
                    side.wait_event(self._fork_events[k])
                    with torch.cuda.stream(side):
                        self._timed_kernel_step(eps, None, sl=slice(lo, hi), opts=opts, tag=(self.n_iterations, lo, hi))
                else:
                    self._timed_kernel_step(eps, None, sl=slice(lo, hi), opts=opts, tag=(self.n_iterations, lo, hi))
            self._join_event.record(side)
            main.wait_event(self._join_event)
        finally:
            self._slice_launch = None

    # ------------------------------------------------------------------ weight-gradient GEMM with the update as epilogue
    def _fused_gemm_plan(self):
        """Slices of the arena for ``fuse_update_into_gemm``: one per hidden dense layer of an MLP cost function,
        ``[W_l | everything up to W_{l+1}]`` (the top one runs to the end of the arena), or None when the model / dtype /
        alignment does not fit ``sgmcmc_gemm_tn_sghmc_f32`` (the sampler then steps as usual)."""
        a, params = self.arena, self.params
        if "V" not in self._STATE_ROWS or not hasattr(self, "mdecay"):
            return None                                       # the fused kernel carries the SGHMC update only
        if self._torch_dtype != torch.float32 or not hasattr(self.cost_fun, "cost_and_grad") or (len(params) - 1) % 2:
            return None
        # the hook would decline (and leave a slice without any update) for a batch that is not a multiple of 16, and it adds
        # the weight-prior term only through grad_decay: decide both before anything is captured
        batch = getattr(getattr(self.cost_fun, "x_placeholder", None), "value", None)
        if batch is None or batch.shape[0] % 16 or not getattr(self.cost_fun, "fold_prior", False):
            return None
        n_layers = (len(params) - 1) // 2
        L = n_layers - 1
        if n_layers < 2 or params[2 * L].dim() != 2 or params[2 * L].shape[1] != 1:
            return None
        plan, offs = [], [int(a.offsets[2 * l]) for l in range(L)] + [a.n]
        for l in range(L):
            W = params[2 * l]
            M, N = (int(W.shape[0]), int(W.shape[1])) if W.dim() == 2 else (0, 0)
            lo, hi = offs[l], offs[l + 1]
            if W.dim() != 2 or N % 128 or M % 4 or lo % 4 or lo != int(a.offsets[2 * l]) or hi - lo < M * N or (l == 0 and lo != 0):
                return None
            plan.append(dict(layer=l, lo=lo, hi=hi, M=M, N=N, n_tail=hi - lo - M * N,
                             blocks=gemm_kernels.gemm_tn_sghmc_blocks(M, N, hi - lo - M * N)))
        base = 0
        for p in plan:
            p["rec_base"], base = base, base + p["blocks"]
        return plan, base

    def _fused_weight_update(self, plan, total, eps):
        """The hook handed to the cost pipeline while the fused graph is captured."""
        a = self.arena
        by_layer = {p["layer"]: p for p in plan}

        def hook(l, h_in, delta):
            p = by_layer.get(l)
            if p is None or h_in.shape[0] % 16:
                return False
            sl = slice(p["lo"], p["hi"])
            gemm_kernels.gemm_tn_sghmc(h_in, delta, a.row("theta")[sl], a.row("V")[sl], a.row("minv")[sl],
                                  a.row("grad")[p["lo"] + p["M"] * p["N"]:p["hi"]] if p["n_tail"] else None,
                                  eps, self.scale_grad, self.mdecay, grad_decay=self._fused_grad_decay, seed=self._philox_seed,
                                  step=0, step_dev=self._step_ctr, first_element=p["lo"], stats=self._step_stats(),
                                  stats_base=p["rec_base"], stats_total=total)
            return True
        return hook



def experimental(sampler_class):
    """``sampler_class`` with the two experimental stepping modes mixed in (same constructor)."""
    return type("Experimental" + sampler_class.__name__, (ExperimentalStepping, sampler_class), {})


def upgrade(sampler):
    """Give an already built product sampler (and its ``BNNCost``) the experimental modes in place."""
    sampler.__class__ = experimental(type(sampler))
    sampler.overlap_update = sampler.fuse_update_into_gemm = False
    sampler._fused_plan = sampler._slice_plan = sampler._slice_launch = sampler._side_stream = None
    if type(sampler.cost_fun) is BNNCost:
        sampler.cost_fun.__class__ = HookedBNNCost
        sampler.cost_fun.gw_gemm = "blas"
    return sampler
