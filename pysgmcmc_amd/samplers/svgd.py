"""Stein variational gradient descent (Liu & Wang, NIPS 2016; mirror of
``pysgmcmc/samplers/svgd.py``).

The reference stacks the particles into an ``(n, d)`` tensor, builds one TF op per
particle PAIR for the distances (``tensor_utils.py:397-408``), sorts all ``n*n``
distances for the median bandwidth and runs two dense matmuls per step
(``svgd.py:118-181``). Here the particles are the rows of the arena's theta row
(``[n x d]``, contiguous), their cost gradients the rows of the grad row, and one
step is ``sgmcmc_svgd_step_{f32,f64}``: four launches that read the particles twice
and the gradients once (csrc/sgmcmc_svgd.hip).

Sign of the kernel-gradient term (quirk Q10). The reference adds
``kernel_gradients`` to ``K @ grad(cost)`` and SUBTRACTS the sum from the particles
(``svgd.py:124-127,139-143``). The driving term is right (``-K grad(cost) = K grad log p``)
but the kernel-gradient term enters with the wrong sign: it pulls particles together
instead of pushing them apart, and the particle cloud collapses onto the mode. That
is a bug; it is fixed here by default (Liu & Wang's update). Set the per-instance attribute
``sampler.strict_reference_quirks = True`` to reproduce the reference's arithmetic exactly
(``repulsion_sign = +1``); two samplers in one process can differ.
"""
import torch

from pysgmcmc_amd import kernels
from pysgmcmc_amd.samplers.base_classes import MCMCSampler
from pysgmcmc_amd.stepsize_schedules import ConstantStepsizeSchedule

__all__ = ("SVGDSampler",)


class SVGDSampler(MCMCSampler):
    """Stein Variational Gradient Descent sampler (keywords/defaults as ``svgd.py:24-27``).

    Parameters
    ----------
    particles : list of torch.Tensor
        The particles, each a (different) guess of the target parameters; all of one size.
    cost_fun : callable
        Takes ONE particle and returns its scalar cost (``svgd.py:37-40``). A cost function
        with the attribute ``batched = True`` is instead called once with the ``[n, d]``
        matrix of all particles and must return the ``n`` costs (one backward pass for all
        particles instead of ``n``). A plain per-particle cost is batched automatically with
        ``torch.func.vmap`` when that reproduces the particle-by-particle costs and gradients.
    alpha, fudge_factor : float
        Decay of the running mean of squared updates and the constant added to its square
        root (``svgd.py:129-137``; AdaGrad with momentum, as in Liu & Wang's code).

    ``next(sampler)`` returns ``(particles, costs)``: the list of updated particles and the
    vector of the ``n`` costs at the particles before the step.
    """

    _STATE_ROWS = ("historical_grad",)
    # particle rows start 64 elements (256 B) aligned: the kernels then use 16-byte accesses throughout
    _PARAM_ALIGN = 64

    def __init__(self, particles, cost_fun, batch_generator=None,
                 stepsize_schedule=ConstantStepsizeSchedule(0.1),
                 alpha=0.9, fudge_factor=1e-6, session=None,
                 dtype=torch.float64, seed=None):
        assert isinstance(alpha, (int, float))
        assert isinstance(fudge_factor, (int, float))
        assert callable(cost_fun)

        particles = list(particles)
        assert len(particles) >= 1, "SVGDSampler needs at least one particle"
        sizes = {int(p.numel()) for p in particles}
        assert len(sizes) == 1, "all particles must have the same number of elements"

        self._particle_cost_fun = cost_fun
        batched = bool(getattr(cost_fun, "batched", False))

        def cost_fun_wrapper(params):
            # svgd.py:86-87: the cost of every particle (tf.map_fn over the stacked particles)
            if batched:
                return cost_fun(self.particles)
            return torch.stack([cost_fun(p).reshape(()) for p in params])

        cost_fun_wrapper.__name__ = getattr(cost_fun, "__name__", "cost_fun")

        super().__init__(
            params=particles, cost_fun=cost_fun_wrapper, batch_generator=batch_generator,
            session=session, seed=seed, dtype=dtype, stepsize_schedule=stepsize_schedule
        )
        self.alpha = float(alpha)
        self.fudge_factor = float(fudge_factor)
        self.n_particles = len(particles)
        self.particle_dim = sizes.pop()
        self.repulsion_sign = -1              # Liu & Wang; +1 = the reference as written (strict_reference_quirks)
        # the particles as one [n, d] matrix: the theta row of the arena (svgd.py:84 tf.stack)
        self.particle_pitch = ((self.particle_dim + 63) // 64) * 64
        self.particles = self._matrix("theta")
        if batched:
            self.particles.requires_grad_(True)
        self._batched = batched
        self._vmapped = None
        self._workspace = None
        self.collect_stats = False

    @property
    def strict_reference_quirks(self):
        """True = the reference's update as written (attracting kernel-gradient term, quirk Q10)."""
        return self.repulsion_sign == 1

    @strict_reference_quirks.setter
    def strict_reference_quirks(self, strict):
        self.repulsion_sign = 1 if strict else -1
        self._graphs.clear()                  # a fully captured step has the sign baked in

    # the base class differentiates `cost_fun(self.params)` particle by particle (n backward passes); a batched
    # cost differentiates the [n, d] matrix in one pass, and a plain per-particle cost is batched with
    # torch.func.vmap when that reproduces the loop's costs and gradients on the first step
    def _cost_and_grad(self):
        if self._batched:
            return self._cost_and_grad_matrix(self.cost_fun)
        if self._vmapped is None:
            self._vmapped = self._try_vmap()
        if self._vmapped:
            return self._cost_and_grad_matrix(self._vmapped)
        return super()._cost_and_grad()

    def _cost_and_grad_matrix(self, fun):
        self._grad_decay = 0.0
        with torch.enable_grad():
            cost = fun(self.params)
            if not isinstance(cost, torch.Tensor) or not cost.requires_grad:
                raise ValueError("cost_fun(particles) must return a torch tensor that depends on the particles")
            grad, = torch.autograd.grad(cost, [self.particles], grad_outputs=torch.ones_like(cost))
        with torch.no_grad():
            self._matrix("grad").copy_(grad)
        return cost.detach()

    def _try_vmap(self):
        """Batched form of the per-particle cost, or False. Only for particles that share one shape; checked
        against the particle-by-particle evaluation before it is trusted."""
        if len({tuple(p.shape) for p in self.params}) != 1 or not hasattr(torch, "func"):
            return False
        shape = tuple(self.params[0].shape)
        cost_fun = self._particle_cost_fun
        try:
            batched = torch.func.vmap(lambda row: cost_fun(row.reshape(shape)).reshape(()))
            self.particles.requires_grad_(True)
            fun = lambda params: batched(self.particles)
            with torch.enable_grad():
                c_b = fun(self.params)
                g_b, = torch.autograd.grad(c_b, [self.particles], grad_outputs=torch.ones_like(c_b))
                c_l = self.cost_fun(self.params)
                g_l = torch.autograd.grad(c_l, self.params, grad_outputs=torch.ones_like(c_l), allow_unused=True)
            g_l = torch.stack([torch.zeros(shape, dtype=c_l.dtype, device=c_l.device) if g is None else g
                               for g in g_l]).reshape(self.n_particles, -1)
            tol = 1e-5 if self._torch_dtype == torch.float32 else 1e-11
            ok = (torch.allclose(c_b, c_l, rtol=tol, atol=tol)
                  and torch.allclose(g_b, g_l.to(g_b.dtype), rtol=tol, atol=tol * (1.0 + float(g_l.abs().max()))))
            return fun if ok else False
        except Exception:                          # data-dependent control flow, in-place ops, feeds ... : keep the loop
            return False

    def _matrix(self, row):
        """``[n, d]`` view of an arena row (row pitch ``particle_pitch`` elements)."""
        flat = self.arena.row(row)
        return torch.as_strided(flat, (self.n_particles, self.particle_dim), (self.particle_pitch, 1))

    def _ws(self):
        if self._workspace is None:
            self._workspace = kernels.svgd_workspace(self.n_particles, self.arena.row("theta"))
        return self._workspace

    def _kernel_step(self, eps, xi):
        a = self.arena
        kernels.svgd_step(a.row("theta"), a.row("grad"), a.row("historical_grad"),
                          self.n_particles, self.particle_dim, eps, self.alpha, self.fudge_factor,
                          self._ws(), ld=self.particle_pitch, repulsion_sign=self.repulsion_sign)

    def svgd_kernel(self, particles=None):
        """Kernel matrix and summed kernel gradients of the current particles (``svgd.py:149-181``):
        ``(kernel_matrix [n, n], kernel_gradients [n, d])`` as device tensors. ``particles`` may be
        an ``[n, d]`` device tensor to evaluate instead."""
        ld = None
        if particles is None:
            x, n, d, ld = self.arena.row("theta"), self.n_particles, self.particle_dim, self.particle_pitch
            ws = self._ws()
        else:
            x = torch.as_tensor(particles).to(device=self.device, dtype=self._torch_dtype).contiguous()
            assert x.dim() == 2, "svgd_kernel: a 2-d tensor must be passed."
            n, d = int(x.shape[0]), int(x.shape[1])
            ws = kernels.svgd_workspace(n, x)
        K, kg, bw = kernels.svgd_kernel(x.reshape(-1), n, d, ws, ld=ld)
        self.bandwidth = bw
        return K, kg
