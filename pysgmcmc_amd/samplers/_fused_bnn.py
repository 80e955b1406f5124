"""Whole steps of a small BNN in one kernel (csrc/sgmcmc_bnn_fused.hip), shared by the two samplers the
reference's BNN accepts (``pysgmcmc/sampling.py:40,64``: SGHMC and SGLD)."""
import numpy as np
import torch

from pysgmcmc_amd import kernels

__all__ = ("FusedBNNStepsMixin",)


class FusedBNNStepsMixin(object):
    """``fused_bnn_available()`` / ``fused_bnn_steps(n)`` for burn-in samplers; the sampler supplies
    ``_fused_bnn_launch(starts, costs, eps, n_steps)``."""

    def fused_bnn_available(self):
        """True when whole steps can run inside ONE kernel (``sgmcmc_bnn_fused_{sghmc,sgld}_steps``): the cost is
        the library's MLP-BNN cost (``BNNCost``, weight prior folded), fed by a ``WindowBatches`` generator,
        the net has one output unit, at most 8 layers, and its activations fit the LDS."""
        cost, gen = self.cost_fun, self.batch_generator
        if self.device.type != "cuda" or self.noise_source is not None:
            return False
        if not (hasattr(cost, "fold_prior") and cost.fold_prior and hasattr(gen, "next_starts")):
            return False
        if gen.x_placeholder is not cost.x_placeholder or gen.y_placeholder is not cost.y_placeholder:
            return False
        sizes = self._bnn_layer_sizes()
        if sizes is None or sizes[-1] != 1 or len(sizes) - 1 > 8:
            return False
        lds = 160 + ((2 * sum(sizes) + 1) * gen.batch_size + self.arena.n + 4) * self.arena.row("theta").element_size()
        return lds <= 160 * 1024 and gen.x_dev.dtype == self._torch_dtype and gen.x_dev.is_contiguous()

    def _bnn_layer_sizes(self):
        shapes = self.arena.shapes
        if len(shapes) < 3 or len(shapes) % 2 == 0 or shapes[-1] not in ((1, 1), (1,), ()):
            return None
        sizes = [shapes[0][0]]
        for l in range((len(shapes) - 1) // 2):
            w, b = shapes[2 * l], shapes[2 * l + 1]
            if len(w) != 2 or w[0] != sizes[-1] or b != (w[1],):
                return None
            sizes.append(w[1])
        return sizes

    def fused_bnn_steps(self, n_steps):
        """Advance the chain by ``n_steps`` complete steps in one launch (one workgroup; see
        ``csrc/sgmcmc_bnn_fused.hip``). Same chain as ``n_steps`` calls of ``next()`` up to the rounding of
        the matrix products (same windows, same Philox stream, same update operator). Returns the
        device tensor of the ``n_steps`` costs. Needs a stepsize that is constant over the chunk."""
        if not self.fused_bnn_available():
            raise ValueError("fused_bnn_steps: this sampler/cost/batch generator does not fit the fused small-model kernel")
        n_steps = int(n_steps)
        eps = [next(self.stepsize_schedule) for _ in range(n_steps)]
        if any(e != eps[0] for e in eps):
            raise ValueError("fused_bnn_steps needs a constant stepsize over the chunk")
        self.epsilon = eps[0]
        gen, cost, a = self.batch_generator, self.cost_fun, self.arena
        pending, self._pending_window = getattr(self, "_pending_window", None), None
        if pending is None:
            first = []
        else:                                     # a window next(sampler) drew one step ahead (base_classes._window_to_prefetch)
            first = [int(pending[0])]
        starts_host = np.concatenate([np.asarray(first, dtype=np.int32), gen.next_starts(n_steps - len(first))]) if first else gen.next_starts(n_steps)
        starts = torch.as_tensor(starts_host, dtype=torch.int32).to(self.device)
        costs = torch.empty(n_steps, dtype=self._torch_dtype, device=self.device)
        self._fused_bnn_launch(starts, costs, eps[0], n_steps)
        self.n_iterations += n_steps
        self._stats_valid = False                 # theta moved without the statistics workspace
        self._grad_decay = float(cost.wdecay / ((a.n + 3e-16) * cost.n_examples))
        self.cost = costs[-1]
        return costs
