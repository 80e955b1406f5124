"""Tensor helpers on the sampler hot path (mirror of ``pysgmcmc/tensor_utils.py``,
hot subset only: vectorize :17-104, unvectorize :107-153, safe_divide :211-269,
safe_sqrt :272-323, uninitialized_params :579-605).

In the kernels ``safe_divide`` / ``safe_sqrt`` are device functions
(``csrc/sgmcmc_kernels.hip``: ``sdiv`` / ``ssqrt``); the torch versions here serve
host-side code such as the BNN priors. ``vectorize`` needs no shadow variable:
the flat arena already *is* the vectorized parameter.
"""
import torch

__all__ = ["vectorize", "unvectorize", "safe_divide", "safe_sqrt", "uninitialized_params"]


def _as_tensor(x, like=None):
    if isinstance(x, torch.Tensor):
        return x
    if like is not None:
        return torch.as_tensor(x, dtype=like.dtype, device=like.device)
    return torch.as_tensor(x)


def vectorize(tensor):
    """Column view ``(n, 1)`` of ``tensor`` (no copy for contiguous input).

    >>> import torch
    >>> vectorize(torch.zeros(2, 3)).shape
    torch.Size([6, 1])
    >>> vectorize(torch.tensor(1.0)).shape
    torch.Size([1, 1])
    """
    return _as_tensor(tensor).reshape(-1, 1)


def unvectorize(tensor, original_shape):
    """Inverse of :func:`vectorize`.

    >>> import torch
    >>> t = torch.arange(6.0).reshape(2, 3)
    >>> torch.equal(unvectorize(vectorize(t), t.shape), t)
    True
    """
    return _as_tensor(tensor).reshape(tuple(original_shape))


def safe_divide(x, y, small_constant=1e-16):
    """``x / (y + (2 * sign(y) * small_constant + small_constant))``.

    >>> import torch
    >>> bool(torch.isinf(safe_divide(torch.tensor(1.0), torch.tensor(0.0))))
    False
    >>> bool(torch.isinf(torch.tensor(1.0) / (torch.tensor(-1e-16) + 1e-16)))
    True
    >>> bool(torch.isinf(safe_divide(torch.tensor(1.0), torch.tensor(-1e-16))))
    False
    """
    y = _as_tensor(y, like=x if isinstance(x, torch.Tensor) else None)
    x = _as_tensor(x, like=y)
    return x / (y + (2.0 * torch.sign(y) * small_constant + small_constant))


def safe_sqrt(x, clip_value_min=0.0, clip_value_max=float("inf")):
    """``sqrt(clamp(x, clip_value_min, clip_value_max))``.

    >>> import torch
    >>> bool(torch.isnan(torch.sqrt(torch.tensor(-1e-16))))
    True
    >>> float(safe_sqrt(torch.tensor(-1e-16)))
    0.0
    """
    return torch.sqrt(torch.clamp(_as_tensor(x), min=clip_value_min, max=clip_value_max))


def uninitialized_params(params, session=None):
    """Torch tensors are always initialised; kept for API compatibility
    (the reference initialises TF variables lazily, tensor_utils.py:579-605)."""
    return []
