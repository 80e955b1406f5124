"""Bayesian neural network on the SG-MCMC update path (mirror of
``pysgmcmc/models/bayesian_neural_network.py``): the cost path that produces the
gradients the fused kernels consume, and the ``train`` / ``predict`` driver.

Cost (``bayesian_neural_network.py:365-388``), for a tanh MLP with a learnable scalar
log-variance ``output_bias`` (``:28-69``):

    NLL = -[ sum_i( -(y_i-mu_i)^2 * 0.5/(exp(s)+1e-16) - 0.5 s ) / batch_size
             + LVP(s)/N + WP(theta)/N ]
    LVP = mean_rows sum_cols[ sdiv(-(s - ln 1e-6)^2, 2*0.01) - 0.5 ln 0.01 ]   (:102-107)
    WP  = sdiv( sum_tensors sum -0.5*wdecay*w^2, n_params )                    (:131-141)

``batch_size`` is the CONFIGURED size, ``N`` the dataset size; WP runs over every
trainable tensor including biases and ``output_bias`` (``:386``).

:class:`BNNCost` evaluates this two ways:
  * ``cost_fun(params)`` -- plain torch ops, differentiable by autograd (any
    ``get_net``-style network works this way);
  * ``cost_and_grad(params, grad_views)`` -- the MI355X path for MLPs: analytic
    backward whose GEMMs (rocBLAS/hipBLASLt via ``torch.addmm(out=...)``) write
    d cost/d W straight into the sampler's flat gradient arena with the weight-prior
    term folded into the GEMM epilogue (``beta * W``), so no gradient tensor is ever
    allocated, gathered or copied between the backward pass and the fused update.
"""
import collections
import copy
import logging
import math
import os
import weakref
from collections import deque
from time import time

import numpy as np
import torch

from pysgmcmc_amd.data_batches import Placeholder, generate_batches
from pysgmcmc_amd.sampling import Sampler
from pysgmcmc_amd.stepsize_schedules import ConstantStepsizeSchedule
from pysgmcmc_amd.tensor_utils import safe_divide

__all__ = [
    "get_default_net", "init_mlp_params", "mlp_forward", "log_variance_prior_log_like",
    "weight_prior_log_like", "BNNCost", "BayesianNeuralNetwork",
    "zero_mean_unit_var_normalization", "zero_mean_unit_var_unnormalization", "enable_gemm_tuning",
]


# ------------------------------------------------------------------ GEMM selection

def enable_gemm_tuning(enable=True, results_file=None, max_duration_ms=30, max_iterations=20):
    """Let PyTorch's TunableOp pick the fastest rocBLAS / hipBLASLt solution for every GEMM shape of the
    cost path the first time it is seen (a few ms per shape, done in the warm-up step before hipGraph
    capture). Measured on MI355X for the 10 M-parameter BNN at batch 256: 4 185 vs 3 915 samples/s.
    Process-wide PyTorch setting, therefore opt-in; same fp32 arithmetic, possibly another summation order."""
    import os
    import tempfile
    import torch.cuda.tunable as tunable
    tunable.enable(bool(enable))
    tunable.tuning_enable(bool(enable))
    if enable:
        tunable.set_max_tuning_duration(int(max_duration_ms))
        tunable.set_max_tuning_iterations(int(max_iterations))
        tunable.set_filename(results_file or os.path.join(
            tempfile.gettempdir(), "pysgmcmc_amd_tunableop_%d.csv" % os.getpid()))


# Plans of at least this many parameters are DEVICE-bound (their step is matrix products on the GPU, not launch overhead): their
# first evaluation picks the library GEMM solutions by measurement (BNNCost.auto_gemm_tuning). Environment opt-out for the
# whole process: PYSGMCMC_AMD_AUTO_GEMM_TUNING=0.
AUTO_GEMM_TUNING_MIN_PARAMS = 1000000


def _auto_gemm_tuning_default():
    import os
    return os.environ.get("PYSGMCMC_AMD_AUTO_GEMM_TUNING", "1").strip().lower() not in ("0", "false", "off", "no")


class _GemmTuningScope(object):
    """TunableOp tuning switched on for ONE evaluation of a cost plan, and off again afterwards. The selections it made stay in
    use (``tunable.enable`` stays on: with tuning off TunableOp only looks solutions up); nothing is touched when the caller has
    tuning on already (``enable_gemm_tuning`` / ``pysgmcmc_amd.configure_for_device_bound_chains``) or a stream is capturing."""

    def __init__(self, max_duration_ms=30, max_iterations=20):
        self.max_duration_ms, self.max_iterations, self.prev, self.active = max_duration_ms, max_iterations, None, False

    def __enter__(self):
        import torch.cuda.tunable as tunable
        if torch.cuda.is_current_stream_capturing() or (tunable.is_enabled() and tunable.tuning_is_enabled()):
            return self
        self.prev = (tunable.is_enabled(), tunable.tuning_is_enabled(), tunable.get_max_tuning_duration(),
                     tunable.get_max_tuning_iterations())
        enable_gemm_tuning(True, max_duration_ms=self.max_duration_ms, max_iterations=self.max_iterations)
        self.active = True
        return self

    def __exit__(self, *exc):
        if self.active:
            import torch.cuda.tunable as tunable
            tunable.tuning_enable(False)                          # look-ups only from here on; the picks stay in use
            tunable.set_max_tuning_duration(self.prev[2])
            tunable.set_max_tuning_iterations(self.prev[3])
        return False


# ------------------------------------------------------------------ normalisation
# pysgmcmc/models/base_model.py:125-137

def zero_mean_unit_var_normalization(X, mean=None, std=None):
    if mean is None:
        mean = np.mean(X, axis=0)
    if std is None:
        std = np.std(X, axis=0)
    return (X - mean) / std, mean, std


def zero_mean_unit_var_unnormalization(X_normalized, mean, std):
    return X_normalized * std + mean


# ------------------------------------------------------------------ network

def init_mlp_params(n_inputs, hidden=(50, 50, 50), seed=None, dtype=torch.float64, device="cpu"):
    """Parameters of the default net in TF ``trainable_variables`` order:
    ``[W1, b1, ..., W_L, b_L, output_bias]`` (``bayesian_neural_network.py:28-69``).

    Kernels: ``variance_scaling_initializer(factor=1.0)`` = truncated normal with
    variance 1/fan_in (:29-55); here N(0, 1/fan_in) truncated at 2 sd, drawn from
    ``torch.Generator(seed)`` (TF's RNG stream itself is not reproducible here).
    Biases zero; ``output_bias = log(1e-3)`` as a (1, 1) tensor (:58-61)."""
    gen = torch.Generator(device="cpu")
    if seed is not None:
        gen.manual_seed(int(seed))
    sizes = [int(n_inputs)] + [int(h) for h in hidden] + [1]
    params = []
    for fan_in, fan_out in zip(sizes[:-1], sizes[1:]):
        w = torch.empty(fan_in, fan_out, dtype=torch.float64)
        torch.nn.init.trunc_normal_(w, mean=0.0, std=1.0, a=-2.0, b=2.0, generator=gen)
        # rescale so the truncated draw has variance 1/fan_in (tf.contrib's initializer widens the
        # pre-truncation stddev to sqrt(1.3/fan_in) for the same purpose)
        w.mul_(math.sqrt(1.0 / fan_in) / 0.87962566103423978)
        params.append(w.to(dtype=dtype, device=device))
        params.append(torch.zeros(fan_out, dtype=dtype, device=device))
    params.append(torch.full((1, 1), math.log(1e-3), dtype=dtype, device=device))
    return params


def mlp_forward(params, X):
    """Network output ``(B, 2)`` = [mean, log_var] (``:63-67``)."""
    h = X
    n_layers = (len(params) - 1) // 2
    for l in range(n_layers):
        a = torch.addmm(params[2 * l + 1], h, params[2 * l])
        h = torch.tanh(a) if l < n_layers - 1 else a
    return torch.cat([h, torch.ones_like(h) * params[-1]], dim=1)


def get_default_net(inputs, seed=None, dtype=torch.float64, params=None):
    """Functional stand-in for the reference's ``get_default_net``: with ``params`` it
    evaluates the 3x50 tanh net; without, it creates them first (returns (output, params))."""
    if params is None:
        params = init_mlp_params(inputs.shape[1], seed=seed, dtype=dtype, device=inputs.device)
        return mlp_forward(params, inputs), params
    return mlp_forward(params, inputs)


# ------------------------------------------------------------------ priors

def log_variance_prior_log_like(log_var, mean=1e-6, var=0.01, dtype=torch.float64):
    """Prior on the predicted log variance (``:77-107``)."""
    log_var = torch.as_tensor(log_var, dtype=dtype)
    mean_t = torch.tensor(mean, dtype=dtype, device=log_var.device)
    var_t = torch.tensor(var, dtype=dtype, device=log_var.device)
    inner = safe_divide(-torch.square(log_var - torch.log(mean_t)), 2.0 * var_t) - 0.5 * torch.log(var_t)
    return torch.mean(torch.sum(inner, dim=1))


def weight_prior_log_like(parameters, wdecay=1.0, dtype=torch.float64):
    """Gaussian prior on all parameters, normalised by their count (``:110-141``)."""
    first = torch.as_tensor(parameters[0])
    log_like = torch.zeros((), dtype=dtype, device=first.device)
    n_params = 0.0
    for p in parameters:
        p = torch.as_tensor(p).to(dtype)
        log_like = log_like + torch.sum(-wdecay * 0.5 * torch.square(p))
        n_params += float(p.numel())
    return safe_divide(log_like, torch.tensor(n_params, dtype=dtype, device=first.device))


# ------------------------------------------------------------------ cost path

class _CostPlan(object):
    """The launch sequence one configuration of the MLP cost path runs (built by ``BNNCost._plan``, walked by every step)."""
    __slots__ = ("forward", "head", "backward", "ones_row", "x_ones", "single_out", "gw_batch", "gw_planes", "fresh")

    def __init__(self, forward, head, backward, ones_row, x_ones, single_out, gw_batch=None, gw_planes=False):
        self.forward, self.head, self.backward = tuple(forward), head, dict(backward)
        self.ones_row, self.x_ones, self.single_out, self.gw_batch = ones_row, x_ones, single_out, gw_batch
        self.gw_planes = bool(gw_planes)                          # the batched weight gradients on the bf16 matrix pipe (3 exact planes)
        self.fresh = True                                         # not evaluated yet (BNNCost.auto_gemm_tuning acts on the first evaluation)

    def as_dict(self):
        d = {"forward": list(self.forward), "head": self.head, "backward": dict(self.backward),
             "first_layer_bias_gradient": "from the [x | 1]^T delta product" if self.ones_row else "column sums"}
        if self.gw_batch is not None:
            d["weight_gradients_in_one_batched_product"] = list(range(self.gw_batch[0], self.gw_batch[1] + 1))
            d["batched_weight_gradient_arithmetic"] = ("3 exact bf16 planes per operand, 6 bf16 MFMA products, fp32 accumulation"
                                                       if self.gw_planes else "library fp32 product")
        return d


class BNNCost(object):
    """Negative log likelihood of the MLP BNN as a sampler ``cost_fun``.

    ``x_placeholder`` / ``y_placeholder`` are fed by the batch generator; ``batch_size``
    and ``n_examples`` are the configured constants of ``:377,380``.
    """

    __name__ = "negative_log_likelihood"
    MAX_CACHED_PLANS = 16
    AUTO_GEMM_TUNING_WARM_EVALUATIONS = 8                    # untuned evaluations of a plan before its tuning evaluation
    # the sampler may pass sum(theta^2) reduced by the previous update kernel (weight prior value)
    accepts_theta_sumsq = True

    def __init__(self, x_placeholder, y_placeholder, batch_size, n_examples, wdecay=1.0,
                 prior_mean=1e-6, prior_var=0.01, fold_prior=True):
        self.x_placeholder = x_placeholder
        self.y_placeholder = y_placeholder
        self.batch_size = int(batch_size)
        self.n_examples = int(n_examples)
        self.wdecay = float(wdecay)
        self.prior_mean = float(prior_mean)
        self.prior_var = float(prior_var)
        self.last_mse = None
        self._ws = {}
        # fold_prior: on the GPU cost_and_grad path the weight-prior gradient coef * theta is NOT
        # written into the gradient arena; the sampler passes `grad_theta_coef` to the update kernel
        # (grad_decay), which adds it in registers. Saves a read of theta and a 40 MB epilogue copy
        # per step at 10 M parameters. The autograd path (__call__) always includes the term.
        self.fold_prior = bool(fold_prior)
        self.grad_theta_coef = 0.0
        # True: loss head / tanh backward run as library kernels (device tensors required, raises
        # otherwise). False: the same algebra in device-agnostic torch ops -- an explicit opt-in used
        # to cross-check the kernels, never selected automatically.
        self.use_hip_kernels = True
        # True: every hidden layer whose shapes fit (f32, batch % 32 == 0, widths % 64 == 0, fan-in % 16 == 0) runs as ONE launch
        # per direction -- fp32 matrix-core product with bias + tanh (forward; for the last hidden layer also the output unit's dot
        # product) or tanh' + bias-gradient column sums (backward) as its epilogue, csrc/sgmcmc_bnn_gemm.hip -- and the first
        # layer's bias gradient comes out of its weight-gradient product as the extra row of [x | 1]^T delta. False: library
        # products + the small activation / tanh' launches for every layer. Which launches a configuration gets is decided once,
        # in _plan(); the steps only walk the plan (see plan_summary()).
        # True takes a layer to the fused launch where that PAYS: at most one 32 x 64 output tile per compute unit (batch 256 x
        # 2048 columns). The kernels also run larger layers (rounds of workgroups, a thin last round as half tiles) and "all"
        # forces them there, but on configs[4]'s 256 x 4864 x 4864 layers the library's stream-K product is within 3 % of the
        # matrix pipe and the step LOSES 25 us of 783 with the fused launches (profiles/r05_dense_rounds.txt).
        self.fused_layers = True
        # True: the FIRST evaluation of a plan of >= AUTO_GEMM_TUNING_MIN_PARAMS parameters runs with PyTorch's TunableOp tuning on
        # (a few ms per GEMM shape; the samplers' warm-up step, before any hipGraph capture) and tuning is switched off again
        # right after it: the library products of the plan then run the solutions that measured fastest instead of the heuristic
        # picks -- 5 % of the step at 10 M parameters, 22 % at 49.8 M (profiles/r06_product_defaults.txt). Same fp32 arithmetic,
        # possibly another summation order. Process-wide side effect: TunableOp stays ENABLED (look-ups only). False, or
        # PYSGMCMC_AMD_AUTO_GEMM_TUNING=0 in the environment: the library's heuristics, nothing touched.
        # The batched weight gradients gW = h^T delta of equally shaped hidden layers on the bf16 matrix pipe at fp32 accuracy
        # (csrc/sgmcmc_bnn_gw.hip: both operands written as 3 exact bf16 planes, 6 MFMA products): "auto" takes them there when the
        # gradients have at least 4 tiles of 128 x 128 per compute unit -- configs[4]'s 4864 x 4864 layers (131 + 2 x 5 us against
        # 188 for the library's fp32 products); the 2048 x 2048 layers of the 10 M-parameter net stay on the library (26 + 2 x 3 us
        # against 33: nothing once the planes are written). True: whenever the shapes fit; False: never. Needs fold_prior (the
        # kernel overwrites the gradient slices) and float32.
        self.gw_on_bf16_planes = "auto"
        self.auto_gemm_tuning = _auto_gemm_tuning_default()
        self.gemm_tuning_applied = None                           # None: no plan evaluated yet; "auto" | "caller" | "off"
        # pitched feed buffers handed out by static_feed_buffer(), held WEAKLY: a buffer lives as long as a sampler (or a cached
        # plan) still uses the view it was given; the plan cache is bounded (a ragged last batch, alternating batch sizes or a
        # second sampler each add a few entries, a long run must not accumulate them)
        self._x_ext = weakref.WeakValueDictionary()
        self._plans = collections.OrderedDict()
        # (Forking the weight-gradient GEMMs onto a second stream inside the captured graph was measured
        # on MI355X at batch 256: 291 us/step vs 275 us on one stream -- not kept.)

    # -- autograd path (any differentiable network) --
    def __call__(self, params, *_):
        X, Y = self.x_placeholder.value, self.y_placeholder.value
        nll, mse = self.negative_log_likelihood(params, X, Y)
        self.last_mse = mse.detach()
        return nll

    def negative_log_likelihood(self, params, X, Y):
        dtype = params[0].dtype
        out = mlp_forward(params, X)
        f_mean = out[:, 0].reshape(-1, 1)
        f_log_var = out[:, 1].reshape(-1, 1)
        f_var_inv = 1.0 / (torch.exp(f_log_var) + 1e-16)
        mse = torch.square(Y - f_mean)
        log_like = torch.sum(torch.sum(-mse * (0.5 * f_var_inv) - 0.5 * f_log_var, dim=1))
        log_like = log_like / self.batch_size
        log_like = log_like + log_variance_prior_log_like(
            f_log_var, mean=self.prior_mean, var=self.prior_var, dtype=dtype) / self.n_examples
        log_like = log_like + weight_prior_log_like(params, wdecay=self.wdecay, dtype=dtype) / self.n_examples
        return -log_like, torch.mean(mse)

    # -- fused analytic path (MLP): gradients land in the arena --
    @property
    def wants_static_feeds(self):
        """The sampler should feed this cost function through static buffers (see :meth:`static_feed_buffer`) in every stepping
        mode, so that eager and hipGraph stepping run the same arithmetic. Eager steps then COPY the caller's feeds into those
        buffers and rebind ``placeholder.value`` to them (the reference's ``feed_dict`` semantics: the fed value is read, not kept)."""
        return bool(self.use_hip_kernels and self.fused_layers)

    def static_feed_buffer(self, placeholder, value):
        """Buffer the sampler's hipGraph modes should keep feeding ``placeholder`` through (``None``: no preference). For x on
        the HIP path: a ``[batch, dim]`` view of a ``[batch, dim + 4]`` buffer whose column ``dim`` holds ones, so that
        ``[x | 1]^T delta`` is the first layer's weight AND bias gradient in one product. Every buffer handed out stays known
        (keyed by its address): a second sampler or another batch shape does not take the product path away from the first."""
        if (placeholder is not self.x_placeholder or not self.use_hip_kernels or not self.fused_layers
                or not value.is_cuda or value.dtype != torch.float32 or value.dim() != 2 or value.shape[1] % 4):
            return None
        ext = torch.zeros(value.shape[0], value.shape[1] + 4, dtype=value.dtype, device=value.device)
        ext[:, value.shape[1]] = 1.0
        self._x_ext[ext.data_ptr()] = ext
        return ext[:, :value.shape[1]]

    def _buffers(self, params, B):
        n_layers = (len(params) - 1) // 2
        widths = [params[2 * l].shape[1] for l in range(n_layers)]
        key = (B, params[0].dtype, params[0].device, tuple(widths))
        ws = self._ws.get(key)
        if ws is None:
            dt, dev = params[0].dtype, params[0].device

            def rows_of(widths):
                # layers of equal width share ONE block [layers][B][w] (their weight-gradient products can then run as one strided
                # batched product, see _plan); others get their own buffer
                out, i = [], 0
                while i < len(widths):
                    j = i
                    while j + 1 < len(widths) and widths[j + 1] == widths[i]:
                        j += 1
                    block = torch.empty(j - i + 1, B, widths[i], dtype=dt, device=dev)
                    out.extend(block[k] for k in range(j - i + 1))
                    i = j + 1
                return out
            ws = {"h": rows_of(widths), "d": rows_of(widths),
                  "ones": torch.ones(B, dtype=dt, device=dev),
                  # per-column-tile partial dot products of the last hidden layer with the output unit's weights (bnn_dense_tanh)
                  "dot_parts": None,
                  "tsq_parts": torch.zeros(16, dtype=torch.float64, device=dev),
                  "cost": torch.zeros(1, dtype=dt, device=dev), "mse": torch.zeros(1, dtype=dt, device=dev)}
            self._ws = {key: ws}
            self._plans = collections.OrderedDict()
        return ws

    @torch.no_grad()
    def cost_and_grad(self, params, grad_views, theta_sumsq=None, theta_sumsq_partials=None):
        """NLL at ``params`` with d NLL/d params written into ``grad_views`` (views of the sampler's
        gradient arena): rocBLAS GEMMs + the library's loss-head and fused tanh-backward/bias-gradient
        kernels (10 launches per step for the 4-layer net at batch 256: gather, 3 fused forward layers, loss head + last layer's
        backward, 2 fused backward products, the hidden layers' weight gradients as one batched product, the first layer's, update)."""
        if self.use_hip_kernels:
            # no silent fallback: the HIP path needs device tensors (kernels.* raises on CPU tensors)
            return self._cost_and_grad_hip(params, grad_views, theta_sumsq, theta_sumsq_partials)
        return self._cost_and_grad_torch(params, grad_views, theta_sumsq)

    def _forward(self, params, X, hs):
        n_layers = len(hs)
        h = X
        for l in range(n_layers):
            W, b = params[2 * l], params[2 * l + 1]
            if W.shape[1] == 1:
                # single output unit: a GEMV (rocBLAS picks a 45 us GEMM kernel for N = 1 at K = 2048)
                torch.addmv(b.expand(h.shape[0]), h, W.view(-1), out=hs[l].view(-1))
            else:
                torch.addmm(b, h, W, out=hs[l])
            if l < n_layers - 1:
                torch.tanh_(hs[l])
            h = hs[l]

    # ---- the plan: which launch sequence this configuration runs, decided once
    def _plan(self, params, grad_views, X, ws, have_partials):
        """Launch sequence for (layer shapes, dtype, batch, feed buffer, prior mode, statistics partials available), cached.

        forward[l]:  "dense_tanh" | "dense_tanh+dot" (last hidden layer: + the output unit's partial dot products and the
                     sum(theta^2) slices) | "mm+bias_tanh" | "mm+bias_tanh_rowdot" | "by rowdot" (the single output unit: formed
                     in the launch before) | "addmm" (a multi-output last layer)
        head:        "head+last_layer_backward" (one launch; needs the statistics partials) | "head"
        backward[l]: how delta_{l-1} is formed from delta_l, l = L .. 1: "in head launch" | "last_layer_backward" |
                     "dense_tanh_backward" | "mm+tanh_backward_colsum" | "mm+tanh_backward" (bias gradient from the product)
        ones_row:    the first layer's [gW_0 ; gb_0] = [x | 1]^T delta as one product."""
        from pysgmcmc_amd import kernels
        key = (X.data_ptr(), X.stride(0), tuple(X.shape), bool(have_partials), self.fold_prior, self.fused_layers,
               params[0].data_ptr(), grad_views[0].data_ptr(), self.gw_on_bf16_planes)
        plan = self._plans.get(key)
        if plan is not None:
            self._plans.move_to_end(key)
            return plan
        B, n_layers = int(X.shape[0]), (len(params) - 1) // 2
        L = n_layers - 1
        hs, ds = ws["h"], ws["d"]
        single_out = params[2 * L].shape[1] == 1 and n_layers >= 2
        fused_head = single_out and have_partials
        cus = torch.cuda.get_device_properties(X.device).multi_processor_count if X.is_cuda else 0
        # the fused launch pays while its workgroups run as ONE round (see __init__); "all": wherever the kernel takes the shape
        pays = lambda rows, cols: self.fused_layers == "all" or (rows // 32) * (cols // 64) <= cus
        forward, h = [], X
        for l in range(n_layers):
            W = params[2 * l]
            fits = l < L and self.fused_layers and kernels.bnn_dense_tanh_fits(h, W, hs[l]) and pays(B, int(W.shape[1]))
            if l == L:
                forward.append("by rowdot" if single_out else "addmm")
            elif l == L - 1 and single_out:
                # with the output unit's dot product in the launch, the loss head must add the partials: the fused head only
                top_fits = fits and fused_head and (B // 32) * (int(W.shape[1]) // 64) >= 16 and B <= 1024
                forward.append("dense_tanh+dot" if top_fits else "mm+bias_tanh_rowdot")
                if top_fits and ws["dot_parts"] is None:
                    ws["dot_parts"] = torch.zeros(kernels.bnn_dense_tanh_dot_parts(B, int(W.shape[1]), X.device), B, dtype=X.dtype, device=X.device)
            else:
                forward.append("dense_tanh" if fits else "mm+bias_tanh")
            h = hs[l]
        ext = self._x_ext.get(X.data_ptr())
        D_in = int(X.shape[1])
        adjacent = lambda a, b: (a.is_contiguous() and b.is_contiguous()
                                 and b.data_ptr() == a.data_ptr() + a.numel() * a.element_size())
        # [W_0 ; b_0] must be one contiguous matrix in the gradient arena (and in theta's when the prior rides in the product)
        ones_row = (self.fused_layers and ext is not None and L >= 2 and X.shape[0] == ext.shape[0]
                    and X.stride(0) == ext.stride(0) and ext.shape[1] > D_in and adjacent(grad_views[0], grad_views[1])
                    and (self.fold_prior or adjacent(params[0], params[1])))
        backward = {}
        for l in range(L, 0, -1):
            if l == L and fused_head:
                backward[l] = "in head launch"
            elif l == L and single_out:
                backward[l] = "last_layer_backward"
            elif (self.fused_layers and kernels.bnn_dense_tanh_backward_fits(ds[l], params[2 * l], hs[l - 1], ds[l - 1])
                  and pays(B, int(params[2 * l].shape[0]))):
                backward[l] = "dense_tanh_backward"
            else:
                backward[l] = "mm+tanh_backward" if (l == 1 and ones_row) else "mm+tanh_backward_colsum"
        if "dense_tanh_backward" in backward.values() and "colsum_parts" not in ws:
            widest = max(int(p.shape[0]) for p in params[0:-1:2])
            ws["colsum_parts"] = [torch.zeros((B // 32) * widest, dtype=X.dtype, device=X.device) for _ in range(2)]
        # weight gradients of consecutive hidden layers of one shape as ONE strided batched product (2 x [2048 x 2048 x 256]: 34.9 us
        # against 41.5 for two products in a row): layers lo .. hi whose inputs h_{l-1}, deltas, gradient slices (and, with the prior
        # in the product, weights) lie at one constant stride each
        gw_batch = None

        def stride_of(ts):
            """Elements between consecutive tensors of ``ts`` if they are contiguous, equally spaced views of ONE allocation, else None."""
            base = ts[0].untyped_storage().data_ptr()
            if not all(t.is_contiguous() and t.untyped_storage().data_ptr() == base for t in ts):
                return None                                       # (one strided view must be able to span them: one allocation)
            gaps = {ts[k + 1].data_ptr() - ts[k].data_ptr() for k in range(len(ts) - 1)}
            gap = gaps.pop() if len(gaps) == 1 else 0
            return gap // ts[0].element_size() if gap > 0 and gap % ts[0].element_size() == 0 else None
        hi = L - 1 if single_out else L
        for lo in range(1, hi):
            run = range(lo, hi + 1)
            if not all(params[2 * l].shape == params[2 * lo].shape and hs[l - 1].shape == hs[lo - 1].shape for l in run):
                continue
            strides = [stride_of([hs[l - 1] for l in run]), stride_of([ds[l] for l in run]), stride_of([grad_views[2 * l] for l in run]),
                       0 if self.fold_prior else stride_of([params[2 * l] for l in run])]
            if None not in strides and hs[0].is_cuda:
                gw_batch = (lo, hi) + tuple(strides)
                break
        gw_planes = False
        if gw_batch is not None and self.gw_on_bf16_planes and self.fold_prior and X.dtype == torch.float32 and B % 16 == 0:
            lo, hi = gw_batch[0], gw_batch[1]
            fan_in, fan_out = (int(v) for v in params[2 * lo].shape)
            tiles = (hi - lo + 1) * ((fan_in + 127) // 128) * ((fan_out + 127) // 128)
            if self.gw_on_bf16_planes is True or tiles >= 4 * max(cus, 1):
                gw_planes = True
                n = hi - lo + 1
                ws["planes_h"] = torch.empty(n * kernels.bnn_planes_bytes(B, fan_in), dtype=torch.uint8, device=X.device)
                ws["planes_d"] = torch.empty(n * kernels.bnn_planes_bytes(B, fan_out), dtype=torch.uint8, device=X.device)
        plan = _CostPlan(forward, "head+last_layer_backward" if fused_head else "head", backward, bool(ones_row),
                         ext[:, :D_in + 1] if ones_row else None, single_out, gw_batch, gw_planes)
        self._plans[key] = plan
        while len(self._plans) > self.MAX_CACHED_PLANS:
            self._plans.popitem(last=False)                      # least recently used; its feed buffer goes with its last user
        return plan

    def plan_summary(self, params, grad_views, theta_sumsq_partials=None):
        """The launch sequence :meth:`cost_and_grad` runs for the current feeds, as a dict (see :meth:`_plan`)."""
        X = self.x_placeholder.value
        return self._plan(params, grad_views, X, self._buffers(params, X.shape[0]), theta_sumsq_partials is not None).as_dict()

    def _cost_and_grad_hip(self, params, grad_views, theta_sumsq, theta_sumsq_partials=None):
        X, Y = self.x_placeholder.value, self.y_placeholder.value
        ws = self._buffers(params, X.shape[0])
        plan = self._plan(params, grad_views, X, ws, theta_sumsq_partials is not None)
        if plan.fresh and not X.is_cuda:
            plan.fresh = False                                    # (host tensors: the kernels below refuse them, loudly)
        if plan.fresh:
            plan.fresh = False
            import torch.cuda.tunable as tunable
            n_total = sum(int(p.numel()) for p in params)
            walk = lambda: self._walk_plan(plan, params, grad_views, theta_sumsq, theta_sumsq_partials, X, Y, ws)
            by_caller = tunable.is_enabled() and tunable.tuning_is_enabled()
            device_bound = n_total >= AUTO_GEMM_TUNING_MIN_PARAMS and not torch.cuda.is_current_stream_capturing()
            if by_caller or (self.auto_gemm_tuning and device_bound):
                self.gemm_tuning_applied = "caller" if by_caller else "auto"
                if device_bound:
                    # the library's own picks first: candidates should not be timed on a device that has just woken up (clocks
                    # still ramping, cold caches)
                    if by_caller:
                        tunable.tuning_enable(False)
                    try:
                        for _ in range(self.AUTO_GEMM_TUNING_WARM_EVALUATIONS):
                            walk()
                    finally:
                        if by_caller:
                            tunable.tuning_enable(True)
                if by_caller:
                    return walk()                                 # tuned under the caller's own TunableOp settings
                with _GemmTuningScope():
                    return walk()
            self.gemm_tuning_applied = "off"
        return self._walk_plan(plan, params, grad_views, theta_sumsq, theta_sumsq_partials, X, Y, ws)

    def _walk_plan(self, plan, params, grad_views, theta_sumsq, theta_sumsq_partials, X, Y, ws):
        from pysgmcmc_amd import kernels
        L = (len(params) - 1) // 2 - 1
        hs, ds = ws["h"], ws["d"]
        fused_head = plan.head == "head+last_layer_backward"
        # ---- forward
        h, mean = X, hs[L].view(-1)                               # mean: the output unit's pre-bias mean, as the loss head reads it
        for l, op in enumerate(plan.forward):
            W, b = params[2 * l], params[2 * l + 1]
            if op == "dense_tanh":
                kernels.bnn_dense_tanh(h, W, b.view(-1), hs[l])
            elif op == "dense_tanh+dot":
                kernels.bnn_dense_tanh(h, W, b.view(-1), hs[l], w_next=params[2 * L].view(-1), dot_parts=ws["dot_parts"],
                                       stats_workspace=theta_sumsq_partials, tsq_parts=ws["tsq_parts"])
                mean = ws["dot_parts"]
            elif op == "mm+bias_tanh":
                # plain product, the bias rides in the activation launch (the library's plain GEMM is 1.4-2.1 us faster than its
                # bias-epilogue one at batch 256, round 3)
                torch.mm(h, W, out=hs[l])
                kernels.bias_tanh(hs[l], b.view(-1))
            elif op == "mm+bias_tanh_rowdot":
                # bias + tanh of the last hidden layer and the output unit's dot product in one launch (+ the 16 slices of the
                # previous step kernel's sum(theta^2) records when the fused head follows)
                torch.mm(h, W, out=hs[l])
                kernels.tanh_rowdot(hs[l], params[2 * L].view(-1), hs[L].view(-1),
                                    stats_workspace=theta_sumsq_partials if fused_head else None,
                                    tsq_parts=ws["tsq_parts"] if fused_head else None, bias=b.view(-1))
            elif op == "addmm":
                torch.addmm(b, h, W, out=hs[l])
            h = hs[l]
        # ---- loss head
        n_params = float(sum(p.numel() for p in params))
        if theta_sumsq is None and theta_sumsq_partials is None:
            theta_sumsq = torch.zeros((), dtype=torch.float64, device=X.device)
            for p in params:
                theta_sumsq = theta_sumsq + (p.double() ** 2).sum()
        prior_coef = self.wdecay / ((n_params + 3e-16) * self.n_examples)
        self.grad_theta_coef = prior_coef if self.fold_prior else 0.0
        beta = 0.0 if self.fold_prior else prior_coef
        if fused_head:
            # loss head + gW_L + delta_{L-1} (incl. tanh') + gb_{L-1} + gb_L + d/d log_var in ONE launch
            kernels.bnn_head_last_layer_backward(
                mean, Y.reshape(-1), params[-1], ws["tsq_parts"], params[2 * L + 1], self.batch_size,
                self.n_examples, n_params, self.wdecay, self.prior_mean, self.prior_var, params[2 * L].view(-1), hs[L - 1],
                params[2 * (L - 1) + 1], beta, ws["cost"], grad_views[-1], grad_views[2 * L + 1], ws["mse"], ds[L - 1],
                grad_views[2 * (L - 1) + 1], grad_views[2 * L].view(-1), fold_prior_grad=self.fold_prior, add_last_bias=True)
        else:
            # loss head: delta_L, cost, d/d log_var, mse and (single-output net) the last bias gradient
            so = plan.single_out
            kernels.bnn_head(hs[L].view(-1), Y.reshape(-1), params[-1], theta_sumsq, self.batch_size, self.n_examples,
                             n_params, self.wdecay, self.prior_mean, self.prior_var,
                             ds[L].view(-1), ws["cost"], grad_views[-1], ws["mse"], fold_prior_grad=self.fold_prior,
                             stats_workspace=theta_sumsq_partials,
                             last_bias=params[2 * L + 1] if so else None,
                             grad_last_bias_out=grad_views[2 * L + 1] if so else None, add_last_bias=so)
        self.last_mse = ws["mse"]
        self._backward_hip(plan, params, grad_views, X, ws, prior_coef, beta)
        return ws["cost"].reshape(())

    def _backward_hip(self, plan, params, grad_views, X, ws, prior_coef, beta):
        """delta_{l-1} = (delta_l W_l^T) tanh'(h_{l-1}) first (the last reader of W_l), then gW_l = h_{l-1}^T delta_l written
        directly into the gradient arena; the weight-prior term coef * theta is added by the update kernel (fold_prior) or rides
        in the GEMM epilogue (beta). Bias gradients: column sums of delta, from whichever launch the plan names."""
        from pysgmcmc_amd import kernels
        hs, ds = ws["h"], ws["d"]
        L, B = len(hs) - 1, int(X.shape[0])
        pending, parts_turn = None, 0                             # column sums a fused backward launch left to be added up
        for l in range(L, -1, -1):
            W, b = params[2 * l], params[2 * l + 1]
            op = plan.backward.get(l)
            if op == "in head launch":
                continue
            if op == "last_layer_backward":
                # gW_L, delta_{L-1} (incl. tanh') and gb_{L-1} in one launch
                kernels.bnn_last_layer_backward(ds[l].view(-1), W.view(-1), hs[l - 1], ds[l - 1], grad_views[2 * (l - 1) + 1],
                                                grad_views[2 * l].view(-1), bias_prev=params[2 * (l - 1) + 1], beta=beta)
                continue
            if op == "dense_tanh_backward":
                # product and tanh' of the layer below in ONE launch; its bias gradient as column sums per 32-row tile, added up
                # on the side by the NEXT such launch (or by a small launch after the loop) -- which is also how this launch
                # finishes the sums of the one before it. (Layer 0's bias gradient comes out of its weight-gradient product.)
                parts = None
                if not (l == 1 and plan.ones_row):
                    n_tiles, width = B // 32, int(W.shape[0])
                    parts = ws["colsum_parts"][parts_turn][:n_tiles * width].view(n_tiles, width)
                    parts_turn ^= 1
                kernels.bnn_dense_tanh_backward(ds[l], W, hs[l - 1], ds[l - 1], colsum_parts=parts, finish=pending)
                pending = None if parts is None else (parts, grad_views[2 * (l - 1) + 1], params[2 * (l - 1) + 1].view(-1), beta)
            elif op is not None:
                torch.mm(ds[l], W.t(), out=ds[l - 1])
            # weight gradient
            if plan.gw_batch is not None and plan.gw_batch[0] <= l <= plan.gw_batch[1]:
                lo, hi, s_h, s_d, s_g, s_w = plan.gw_batch
                if l == lo and plan.gw_planes:                    # ... on the bf16 matrix pipe: both operands as 3 exact planes first
                    run = range(lo, hi + 1)
                    kernels.bnn_split_planes([hs[k - 1] for k in run], ws["planes_h"])
                    kernels.bnn_split_planes([ds[k] for k in run], ws["planes_d"])
                    kernels.bnn_gw_planes(ws["planes_h"], ws["planes_d"], [grad_views[2 * k] for k in run], B)
                elif l == lo:                                     # the last delta of the group exists now: ONE strided batched product
                    n = hi - lo + 1
                    stack = lambda t, st: torch.as_strided(t, (n,) + tuple(t.shape), (st,) + tuple(t.stride()))
                    A, D, G = stack(hs[lo - 1], s_h).transpose(1, 2), stack(ds[lo], s_d), stack(grad_views[2 * lo], s_g)
                    if self.fold_prior:
                        torch.bmm(A, D, out=G)
                    else:
                        torch.baddbmm(stack(W, s_w), A, D, beta=prior_coef, alpha=1.0, out=G)
            elif l == 0 and plan.ones_row:
                # [x | 1]^T delta = [gW_0 ; gb_0] onto the arena's [W_0 ; b_0] slice: 785 rows cost the library what 784 do
                rows, width = int(X.shape[1]) + 1, int(W.shape[1])
                gWb = torch.as_strided(grad_views[0], (rows, width), (width, 1))
                if self.fold_prior:
                    torch.mm(plan.x_ones.t(), ds[0], out=gWb)
                else:
                    torch.addmm(torch.as_strided(W, (rows, width), (width, 1)), plan.x_ones.t(), ds[0], beta=prior_coef, alpha=1.0, out=gWb)
            elif self.fold_prior:
                torch.mm((X if l == 0 else hs[l - 1]).t(), ds[l], out=grad_views[2 * l])
            else:
                torch.addmm(W, (X if l == 0 else hs[l - 1]).t(), ds[l], beta=prior_coef, alpha=1.0, out=grad_views[2 * l])
            if l == L:
                # bias gradient of a multi-output last layer (hidden layers get theirs from the launches above / below)
                if self.fold_prior:
                    torch.mv(ds[l].t(), ws["ones"], out=grad_views[2 * l + 1])
                else:
                    torch.addmv(b, ds[l].t(), ws["ones"], beta=prior_coef, alpha=1.0, out=grad_views[2 * l + 1])
            if op in ("mm+tanh_backward", "mm+tanh_backward_colsum"):
                if pending is not None:
                    kernels.colsum_finish(*pending)
                    pending = None
                if op == "mm+tanh_backward":
                    kernels.tanh_backward(ds[l - 1], hs[l - 1])
                else:
                    # delta_{l-1} *= 1 - h_{l-1}^2, and gb_{l-1} = column sums of the result (+ beta * b_{l-1})
                    kernels.tanh_backward_colsum(ds[l - 1], hs[l - 1], grad_views[2 * (l - 1) + 1],
                                                 bias=params[2 * (l - 1) + 1], beta=beta)
        if pending is not None:
            kernels.colsum_finish(*pending)

    def _cost_and_grad_torch(self, params, grad_views, theta_sumsq):
        self.grad_theta_coef = 0.0                                # the torch path always writes the full gradient
        X, Y = self.x_placeholder.value, self.y_placeholder.value
        B = X.shape[0]
        n_layers = (len(params) - 1) // 2
        ws = self._buffers(params, B)
        hs, ds = ws["h"], ws["d"]
        self._forward(params, X, hs)
        mean = hs[-1]
        s = params[-1].reshape(())
        es = torch.exp(s)
        inv = 1.0 / (es + 1e-16)
        resid = Y - mean
        sq = resid * resid
        sse = sq.sum()
        n_params = float(sum(p.numel() for p in params))
        wp_den = n_params + (2e-16 + 1e-16)                      # safe_divide, n_params > 0
        lvp_den = 2.0 * self.prior_var + (2e-16 + 1e-16)
        ln_mean = math.log(self.prior_mean)
        if theta_sumsq is not None:
            sumsq = theta_sumsq.to(params[0].dtype)
        else:
            sumsq = torch.zeros((), dtype=params[0].dtype, device=params[0].device)
            for p in params:
                sumsq = sumsq + (p * p).sum()
        log_like = (-(sse * (0.5 * inv)) - 0.5 * s * B) / self.batch_size
        lvp = -(s - ln_mean) ** 2 / lvp_den - 0.5 * math.log(self.prior_var)
        wp = (-0.5 * self.wdecay) * sumsq / wp_den
        cost = -(log_like + lvp / self.n_examples + wp / self.n_examples)
        self.last_mse = sse / sq.numel()
        prior_coef = self.wdecay / (wp_den * self.n_examples)    # d cost/d theta_j of the weight prior = coef * theta_j
        ds_ = -((sse * (0.5 * es * inv * inv) - 0.5 * B) / self.batch_size
                + (-2.0 * (s - ln_mean) / lvp_den) / self.n_examples) + prior_coef * s
        grad_views[-1].copy_(ds_.reshape(grad_views[-1].shape))
        torch.mul(resid, -(inv / self.batch_size), out=ds[-1])
        for l in range(n_layers - 1, -1, -1):
            h_in = X if l == 0 else hs[l - 1]
            W, b = params[2 * l], params[2 * l + 1]
            torch.addmm(W, h_in.t(), ds[l], beta=prior_coef, alpha=1.0, out=grad_views[2 * l])
            torch.sum(ds[l], dim=0, out=grad_views[2 * l + 1])
            grad_views[2 * l + 1].add_(b, alpha=prior_coef)
            if l > 0:
                torch.mm(ds[l], W.t(), out=ds[l - 1])
                ds[l - 1].mul_(1.0 - hs[l - 1] * hs[l - 1])
        return cost


# ------------------------------------------------------------------ model driver

class BayesianNeuralNetwork(object):
    """BNN whose weights are sampled with SG-MCMC (constructor keywords and
    defaults as ``bayesian_neural_network.py:147-156``; ``session`` selects the
    device or is ignored; ``get_net`` is replaced by ``hidden`` widths of a tanh MLP,
    default 3x50)."""

    def __init__(self, session=None, sampling_method=Sampler.SGHMC,
                 get_net=get_default_net,
                 batch_generator=generate_batches,
                 batch_size=20,
                 stepsize_schedule=ConstantStepsizeSchedule(np.sqrt(1e-4)),
                 n_nets=100, n_iters=50000,
                 burn_in_steps=1000, sample_steps=100,
                 normalize_input=True, normalize_output=True,
                 seed=None, dtype=torch.float64, hidden=(50, 50, 50), n_chains=1, **sampler_kwargs):
        # same sanity checks as :241-262
        assert isinstance(n_chains, int) and n_chains >= 1
        assert isinstance(n_nets, int)
        assert isinstance(n_iters, int)
        assert isinstance(burn_in_steps, int)
        assert isinstance(sample_steps, int)
        assert isinstance(batch_size, int)
        from pysgmcmc_amd.samplers.base_classes import as_torch_dtype
        self._torch_dtype = as_torch_dtype(dtype)
        assert n_nets > 0
        assert n_iters > 0
        assert burn_in_steps >= 0
        assert sample_steps > 0
        assert batch_size > 0
        assert callable(get_net)
        assert callable(batch_generator)
        assert hasattr(stepsize_schedule, "update")
        assert hasattr(stepsize_schedule, "__next__")
        if not Sampler.is_supported(sampling_method):
            raise ValueError(
                "'BayesianNeuralNetwork.__init__' received unsupported input "
                "for parameter 'sampling_method'. Input was: {input}.\n"
                "Supported sampling methods are enumerated in "
                "'Sampler' enum type.".format(input=sampling_method)
            )
        self.sampling_method = sampling_method
        self.stepsize_schedule = stepsize_schedule
        self.get_net = get_net
        self.batch_generator = batch_generator
        self.normalize_input = normalize_input
        self.normalize_output = normalize_output
        self.n_nets = n_nets
        self.n_iters = n_iters
        self.batch_size = batch_size
        self.sampler_kwargs = sampler_kwargs
        self.burn_in_steps = burn_in_steps
        self.sample_steps = sample_steps
        self.samples = deque(maxlen=n_nets)
        self.seed = seed
        self.dtype = dtype
        self.session = session
        self.hidden = tuple(hidden)
        # n_chains > 1 (extension): that many independent chains advance together -- in the fused small-model kernel (one
        # launch per chunk for all chains), or, for a net that kernel does not take, each chain through its own cost pipeline
        # on its own stream (samplers.ConcurrentChains) -- and every collection point contributes one network per chain, so
        # `n_nets` networks need 1/n_chains of the sampling iterations (the reference would run the chains one after the other)
        self.n_chains = n_chains
        self.chains = None
        self.is_trained = False
        # use the analytic-backward cost path (gradients straight into the arena)
        self.fused_cost = True
        # replay the ~25 launches of the cost/gradient pipeline from one hipGraph per step; the 3x50
        # default net is launch-bound (50x50 GEMMs), SURVEY.md 8(f).1
        self.use_hip_graph = True
        # run whole steps of small nets inside one kernel (csrc/sgmcmc_bnn_fused.hip) when the model fits:
        # the chain advances in chunks between the logging / sample-collection points of the loop below
        self.use_fused_steps = True

    def _device(self):
        if isinstance(self.session, (torch.device, str)):
            return torch.device(self.session)
        return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")

    # -- training: an event-driven loop ------------------------------------------------------------------
    # The reference visits the host once per sample (bayesian_neural_network.py:508-531) although only a few
    # iterations do anything there: every 512th burn-in iteration logs, every `sample_steps`-th sampling iteration
    # logs and keeps a network. Here the chain is ADVANCED FROM EVENT TO EVENT -- by the fused whole-step kernel
    # (many steps per launch, many chains per launch) or by next(sampler) -- and the host only looks at the events.

    BURN_IN_LOG_EVERY = 512                      # bayesian_neural_network.py:505

    def _is_burn_in_iteration(self, i):
        return i <= self.burn_in_steps           # `<=`: the reference's off-by-one against the sampler's `<` (:514, quirk Q8)

    def _host_events(self):
        """Ascending iteration indices at which the loop must look at the chain (log and/or collect), ending with the
        last iteration of the budget."""
        last = self.n_iters - 1
        for i in range(self.n_iters):
            every = self.BURN_IN_LOG_EVERY if self._is_burn_in_iteration(i) else self.sample_steps
            if i % every == 0 or i == last:
                yield i

    def _log_full_training_error(self, i, clock):
        with torch.no_grad():
            nll, mse = self.cost.negative_log_likelihood(self.network_params, self._X_full, self._Y_full)
        if self._is_burn_in_iteration(i):        # the reference's two log lines (:492-503)
            logging.info("Iter {:8d} : NLL = {:.4e} MSE = {:.4e} Samples = {} Time = {:5.2f}".format(
                i, float(nll), float(mse), len(self.samples), time() - clock))
        else:
            logging.info("Iter {:8d} : NLL = {:.4e} MSE = {:.4e} Time = {:5.2f}".format(i, float(nll), float(mse), time() - clock))

    def _visit(self, i, clock):
        """Host work of event iteration ``i``; True once ``n_nets`` networks are kept."""
        every = self.BURN_IN_LOG_EVERY if self._is_burn_in_iteration(i) else self.sample_steps
        if i % every:
            return False                         # the closing iteration of the budget: nothing to do
        self._log_full_training_error(i, clock)
        if self._is_burn_in_iteration(i):
            return False
        chains = self.chains.samplers if self.chains is not None else [self.sampler]
        for chain in chains:                     # one network per chain and collection point
            self.samples.append([v.clone() for v in chain.arena.views("theta")])
            if len(self.samples) == self.n_nets:
                return True
        return False

    def _adopt_data(self, X, y, device):
        self.X, self.y = X, y
        if self.normalize_input:
            self.X, self.x_mean, self.x_std = zero_mean_unit_var_normalization(self.X)
        if self.normalize_output:
            self.y, self.y_mean, self.y_std = zero_mean_unit_var_normalization(self.y)
        self._X_full = torch.as_tensor(self.X, dtype=self._torch_dtype, device=device)
        self._Y_full = torch.as_tensor(self.y, dtype=self._torch_dtype, device=device).reshape(-1, 1)

    def _build_sampler(self, n_datapoints, n_inputs, device):
        self.X_Minibatch = Placeholder(dtype=self._torch_dtype, shape=(None, n_inputs), name="X_Minibatch", device=device)
        self.Y_Minibatch = Placeholder(dtype=self._torch_dtype, name="Y_Minibatch", device=device)
        self.network_params = init_mlp_params(n_inputs, hidden=self.hidden, seed=self.seed,
                                              dtype=self._torch_dtype, device=device)
        # torch tensors cannot carry a `.name` like tf.Variables: the names TF would give the default net's variables
        self.network_param_names = [n for l in range(1, len(self.hidden) + 2)
                                    for n in ("fc_layer_%d/kernel:0" % l, "fc_layer_%d/bias:0" % l)] + ["output_bias:0"]
        self.cost = BNNCost(self.X_Minibatch, self.Y_Minibatch, self.batch_size, n_datapoints)
        # fused_cost = False hides cost_and_grad, so the sampler differentiates the cost with autograd
        cost_fun = self.cost if self.fused_cost else (lambda params, *_: self.cost(params))
        kw = self.sampler_kwargs
        kw.update(params=self.network_params, cost_fun=cost_fun, session=device, seed=self.seed, dtype=self.dtype,
                  stepsize_schedule=self.stepsize_schedule,
                  batch_generator=self.batch_generator(x=self.X, x_placeholder=self.X_Minibatch, y=self.y,
                                                       y_placeholder=self.Y_Minibatch, batch_size=self.batch_size,
                                                       seed=self.seed))
        if Sampler.is_burn_in_mcmc(self.sampling_method):
            kw.update(scale_grad=n_datapoints, burn_in_steps=self.burn_in_steps)        # :451-457
        self.sampler = Sampler.get_sampler(self.sampling_method, **kw)
        self.sampler.sample_format = "view"      # samples stay on the device; no per-step D2H copy of all parameters
        self.sampler.param_names = list(self.network_param_names)
        self.sampler.collect_stats = "theta_sq"  # the loss head is the only consumer of the fused statistics
        self.sampler.use_hip_graph = bool(self.use_hip_graph and self.fused_cost and device.type == "cuda")

    def train(self, X, y, *args, **kwargs):
        """Sample ``n_nets`` networks from the posterior given data ``X (N, D)``, ``y (N,)``: the sample sequence,
        logging cadence and collection rule of ``bayesian_neural_network.py:391-533`` (including the
        ``iteration_index <= burn_in_steps`` off-by-one at :514), driven from event to event."""
        assert X.ndim == 2 and X.shape[0] == y.shape[0]
        clock = time()
        device = self._device()
        self._adopt_data(X, y, device)
        self._build_sampler(X.shape[0], X.shape[1], device)
        self.samples.clear()
        fused = bool(self.use_fused_steps and self.fused_cost and hasattr(self.sampler, "fused_bnn_available")
                     and self.sampler.fused_bnn_available())
        self.used_fused_steps = fused
        self.chains = None
        if self.n_chains > 1:
            if not fused and device.type != "cuda":
                raise ValueError("BayesianNeuralNetwork(n_chains > 1) needs an AMD GPU: the chains advance together, in the "
                                 "fused small-model kernel or as concurrent streams; got device %s" % device)
            self.chains = self._build_chain_group(X.shape[0], X.shape[1], device, fused)
        if self.chains is not None:
            advance = self.chains.steps          # every chain, n steps, one launch
        elif fused:
            advance = self.sampler.fused_bnn_steps
        else:
            advance = lambda n: [next(self.sampler) for _ in range(n)]
        logging.info("Starting sampling")
        done = 0                                 # iterations produced so far
        for event in self._host_events():
            advance(event + 1 - done)            # the chain now stands after iteration `event`
            done = event + 1
            if self._visit(event, clock):
                break
        self.is_trained = True

    def _build_chain_group(self, n_datapoints, n_inputs, device, fused=True):
        """Chains 1 .. n_chains-1 next to ``self.sampler`` (chain 0): own initial weights (init seed + c), own window
        stream (RandomState(seed + c)) and Philox seed ``seed_0 + c``; one resident copy of the dataset. ``fused``: all
        chains advance in the fused small-model kernel (one launch per chunk); otherwise -- a model that kernel does not
        fit -- every chain steps through its own cost pipeline on its own stream (``ConcurrentChains``)."""
        from pysgmcmc_amd.samplers.concurrent_chains import ConcurrentChains
        from pysgmcmc_amd.samplers.fused_chains import FusedBNNChains
        first = self.sampler
        gen0 = first.batch_generator
        members = [first]
        for c in range(1, self.n_chains):
            params = init_mlp_params(n_inputs, hidden=self.hidden, seed=None if self.seed is None else self.seed + c,
                                     dtype=self._torch_dtype, device=device)
            if fused:
                xp, yp = self.X_Minibatch, self.Y_Minibatch
            else:                                 # a captured cost pipeline reads the chain's own static feed buffers
                xp = Placeholder(dtype=self._torch_dtype, shape=(None, n_inputs), name="X_Minibatch_%d" % c, device=device)
                yp = Placeholder(dtype=self._torch_dtype, name="Y_Minibatch_%d" % c, device=device)
            cost = BNNCost(xp, yp, self.batch_size, n_datapoints)
            kw = dict(self.sampler_kwargs)
            kw.update({
                "params": params,
                "cost_fun": cost if self.fused_cost else (lambda ps, *_, cost=cost: cost(ps)),
                "batch_generator": type(gen0)(gen0.x_dev, gen0.y_dev, xp, yp, gen0.batch_size,
                                              np.random.RandomState(None if self.seed is None else self.seed + c)),
                "seed": int((first._philox_seed + c) & 0xFFFFFFFFFFFFFFFF),
                "stepsize_schedule": copy.deepcopy(self.stepsize_schedule),
            })
            member = Sampler.get_sampler(self.sampling_method, **kw)
            member.sample_format = "view"
            member.collect_stats = first.collect_stats
            member.use_hip_graph = first.use_hip_graph
            members.append(member)
        return FusedBNNChains(members) if fused else ConcurrentChains(members)

    def compute_network_output(self, params, input_data):
        """Network output ``(N, 2)`` for one set of sampled weights (``:535-557``)."""
        dev = params[0].device if isinstance(params[0], torch.Tensor) else self._device()
        x = torch.as_tensor(input_data, dtype=self._torch_dtype, device=dev)
        ps = [torch.as_tensor(p, dtype=self._torch_dtype, device=dev) for p in params]
        with torch.no_grad():
            return mlp_forward(ps, x).cpu().numpy()

    PREDICT_ACTIVATION_BYTES = 1 << 30           # device memory one batched pass of predict() may spend on activations

    def _network_outputs(self, x):
        """Outputs ``(n_nets, N, 2)`` of EVERY kept network at ``x``: the kept weights are stacked per layer and the layers run as
        batched products on the device, with ONE copy to the host at the end -- the reference (and :meth:`compute_network_output`)
        evaluates the networks one by one (``:599-607``), 100 small launches and 100 copies. The rows of ``x`` go through in chunks
        so that the (n_nets, rows, widest layer) activations of a pass stay below ``PREDICT_ACTIVATION_BYTES`` (a large test set
        must not need n_nets times the memory the network-by-network loop did); kept samples given as numpy arrays are accepted
        like ``compute_network_output`` accepts them."""
        nets = list(self.samples)
        first = nets[0][0]
        dev = first.device if isinstance(first, torch.Tensor) else self._device()
        as_t = lambda p: torch.as_tensor(p, dtype=self._torch_dtype, device=dev)
        n_layers = (len(nets[0]) - 1) // 2
        Ws = [torch.stack([as_t(net[2 * l]) for net in nets]) for l in range(n_layers)]             # (nets, fan_in, fan_out)
        bs = [torch.stack([as_t(net[2 * l + 1]).reshape(-1) for net in nets]).unsqueeze(1) for l in range(n_layers)]
        log_var = torch.stack([as_t(net[-1]).reshape(()) for net in nets]).reshape(-1, 1, 1)
        x = torch.as_tensor(x, dtype=self._torch_dtype, device=dev)
        widest = max([int(x.shape[1])] + [int(W.shape[2]) for W in Ws])
        per_row = 2 * len(nets) * widest * x.element_size()                  # input and output of the widest layer
        rows = max(1, min(int(x.shape[0]), self.PREDICT_ACTIVATION_BYTES // max(per_row, 1)))
        out = torch.empty(len(nets), int(x.shape[0]), 2, dtype=self._torch_dtype, device=dev)
        with torch.no_grad():
            for lo in range(0, int(x.shape[0]), rows):
                h = x[lo:lo + rows].unsqueeze(0).expand(len(nets), -1, -1)
                for l in range(n_layers):
                    h = torch.baddbmm(bs[l], h, Ws[l])
                    if l < n_layers - 1:
                        h = torch.tanh_(h)
                out[:, lo:lo + rows, 0:1] = h
            out[:, :, 1:2] = log_var
        return out.cpu().numpy()

    def predict(self, X_test, return_individual_predictions=False, *args, **kwargs):
        """Predictive mean and variance at ``X_test (N, D)`` (``:560-630``): the kept networks' means and noise
        variances ``(n_nets, N)`` if ``return_individual_predictions``, else the ensemble mean and the variance of the
        networks' means."""
        assert X_test.ndim == 2
        if not self.is_trained:
            raise ValueError(
                "Calling `bnn.predict()` on an untrained "
                "Bayesian Neural Network 'bnn' is not supported! "
                "Please call `bnn.train()` before calling `bnn.predict()`"
            )
        x = zero_mean_unit_var_normalization(X_test, self.x_mean, self.x_std)[0] if self.normalize_input else X_test
        outputs = self._network_outputs(x)                                   # (nets, N, 2)
        means, noise_var = outputs[:, :, 0], np.exp(outputs[:, :, 1])
        if return_individual_predictions:
            if self.normalize_output:
                means = zero_mean_unit_var_unnormalization(means, self.y_mean, self.y_std)
                noise_var = noise_var * self.y_std ** 2
            return means, noise_var
        ensemble_mean = means.mean(axis=0)
        ensemble_var = ((means - ensemble_mean) ** 2).mean(axis=0)
        if self.normalize_output:
            ensemble_mean = zero_mean_unit_var_unnormalization(ensemble_mean, self.y_mean, self.y_std)
            ensemble_var = ensemble_var * self.y_std ** 2
        return ensemble_mean, ensemble_var
