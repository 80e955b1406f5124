"""Flat per-chain state ("arena") for the fused update kernels.

The reference keeps, per parameter *tensor*, a shadow ``(n_i, 1)`` column copy of
theta plus one column per state array (``pysgmcmc/tensor_utils.py:87-98``,
``pysgmcmc/samplers/sghmc.py:126-155``) and loops over tensors in Python
(``sghmc.py:165``). Here one chain owns ONE device allocation of ``rows x stride``
elements: every state array (theta, grad, V, tau, g, v_hat, minv, ...) is a
contiguous row of all ``n`` parameters, densely packed in declaration order, each
row starting 256-B aligned. The user's parameter tensors are re-pointed to views
into the theta row, so the model reads what the kernel writes with no
gather/scatter, and the whole chain is updated by one kernel launch per step.

HBM layout (f32, P parameters): rows * 4P bytes; 10 M params, SGHMC = 7 rows =
280 MB; 50 M = 1.4 GB (of 288 GB).
"""
import torch

__all__ = ["FlatArena"]

_ROW_ALIGN_ELEMS = 64      # 256 B for f32, 512 B for f64


class FlatArena(object):
    """Densely packed flat state for one chain.

    Parameters
    ----------
    params : list of torch.Tensor
        Target parameters. Their values are copied into the ``theta`` row and
        ``p.data`` is re-pointed to a view of it (shape preserved).
    rows : iterable of str
        Names of the state rows to allocate besides ``theta`` and ``grad``.
    dtype, device : torch dtype / device of the arena.
    param_align : int
        Start every parameter at a multiple of this many elements (default 1 = dense). The SVGD
        sampler uses 64 so that its particles form an ``[n x ld]`` matrix with 256-byte aligned rows.
    """

    def __init__(self, params, rows, dtype, device, param_align=1):
        self.dtype = dtype
        self.device = torch.device(device)
        self.shapes = [tuple(p.shape) for p in params]
        self.sizes = [int(p.numel()) for p in params]
        self.offsets = []
        off = 0
        for s in self.sizes:
            self.offsets.append(off)
            off += ((s + param_align - 1) // param_align) * param_align   # param_align = 1: densely packed
        self.n = off
        names = ["theta", "grad"] + [r for r in rows if r not in ("theta", "grad")]
        self.row_names = names
        stride = ((max(self.n, 1) + _ROW_ALIGN_ELEMS - 1) // _ROW_ALIGN_ELEMS) * _ROW_ALIGN_ELEMS
        self.stride = stride
        self.storage = torch.zeros(len(names) * stride, dtype=dtype, device=self.device)
        self._rows = {}
        for k, name in enumerate(names):
            self._rows[name] = self.storage[k * stride:k * stride + self.n]
        self._params = list(params)
        # adopt the parameters: copy values in, alias p.data to the arena
        theta = self._rows["theta"]
        with torch.no_grad():
            for p, o, s, shp in zip(params, self.offsets, self.sizes, self.shapes):
                seg = theta[o:o + s].view(shp)
                seg.copy_(p.detach().to(device=self.device, dtype=dtype))
                p.data = seg
        self.grad_views = [self._rows["grad"][o:o + s].view(shp)
                           for o, s, shp in zip(self.offsets, self.sizes, self.shapes)]

    def rebind(self, storage):
        """Move the arena into ``storage`` (same length, dtype and device; e.g. a slice of one allocation that
        holds several chains back to back): values are copied, the rows, gradient views and the parameters'
        ``.data`` then alias the new memory."""
        assert storage.numel() == self.storage.numel() and storage.dtype == self.storage.dtype
        assert storage.device == self.storage.device and storage.is_contiguous()
        with torch.no_grad():
            storage.copy_(self.storage)
            self.storage = storage
            for k, name in enumerate(self.row_names):
                self._rows[name] = storage[k * self.stride:k * self.stride + self.n]
            theta = self._rows["theta"]
            for p, o, s, shp in zip(self._params, self.offsets, self.sizes, self.shapes):
                p.data = theta[o:o + s].view(shp)
        self.grad_views = [self._rows["grad"][o:o + s].view(shp)
                           for o, s, shp in zip(self.offsets, self.sizes, self.shapes)]

    def row(self, name):
        """Flat length-n view of a state row."""
        return self._rows[name]

    def views(self, name):
        """Per-parameter views (original shapes) of a state row."""
        flat = self._rows[name]
        return [flat[o:o + s].view(shp) for o, s, shp in zip(self.offsets, self.sizes, self.shapes)]

    def fill(self, name, value):
        self._rows[name].fill_(value)

    def nbytes(self):
        return self.storage.numel() * self.storage.element_size()

    def state_dict(self):
        """Checkpointable copy of every row (flat buffers; the reference has no checkpointing)."""
        return {name: self._rows[name].detach().clone() for name in self.row_names}

    def load_state_dict(self, state):
        for name, t in state.items():
            self._rows[name].copy_(t.to(device=self.device, dtype=self.dtype))
