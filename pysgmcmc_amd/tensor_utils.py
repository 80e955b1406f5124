"""Tensor helpers on the sampler hot path (mirror of ``pysgmcmc/tensor_utils.py``,
hot subset only: vectorize :17-104, unvectorize :107-153, safe_divide :211-269,
safe_sqrt :272-323, uninitialized_params :579-605).

In the kernels ``safe_divide`` / ``safe_sqrt`` are device functions
(``csrc/sgmcmc_kernels.hip``: ``sdiv`` / ``ssqrt``); the torch versions here serve
host-side code such as the BNN priors. ``vectorize`` needs no shadow variable:
the flat arena already *is* the vectorized parameter.
"""
import math

import torch

__all__ = ["vectorize", "unvectorize", "median", "safe_divide", "safe_sqrt", "pdist", "squareform", "uninitialized_params"]


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


def median(tensor):
    """Median of all elements; an even count averages the two middle values (``tensor_utils.py:159-204``; ``torch.median`` would
    return the lower one).

    >>> import torch
    >>> float(median(torch.tensor([1, 3, 5], dtype=torch.float64)))
    3.0
    >>> float(median(torch.tensor([1, 3, 5, 7], dtype=torch.float64)))
    4.0
    """
    flat = torch.sort(_as_tensor(tensor).reshape(-1)).values
    n = flat.numel()
    mid = n // 2
    if n % 2 == 1:
        return flat[mid]
    return (flat[mid - 1] + flat[mid]) / 2


def pdist(tensor, metric="euclidean"):
    """Condensed pairwise distances of the rows of a 2-d tensor, scipy's order (``tensor_utils.py:326-419``); euclidean only.

    >>> import torch
    >>> from scipy.spatial.distance import pdist as pdist_scipy
    >>> x = torch.tensor([[0.77228064, 0.09543156], [0.3918973, 0.96806584], [0.66008144, 0.22163063]], dtype=torch.float64)
    >>> bool(torch.allclose(pdist(x), torch.from_numpy(pdist_scipy(x.numpy()))))
    True
    >>> pdist(x, metric="lengthy_metric")
    Traceback (most recent call last):
     ...
    NotImplementedError: tensor_utils.pdist: Metric 'lengthy_metric' currently not supported!
    >>> pdist(torch.rand(2, 2, 1))
    Traceback (most recent call last):
     ...
    ValueError: tensor_utils.pdist: A 2-d tensor must be passed.
    """
    tensor = _as_tensor(tensor)
    if tensor.dim() != 2:
        raise ValueError("tensor_utils.pdist: A 2-d tensor must be passed.")
    if metric != "euclidean":
        raise NotImplementedError("tensor_utils.pdist: Metric '{metric}' currently not supported!".format(metric=metric))
    i, j = torch.triu_indices(tensor.shape[0], tensor.shape[0], offset=1, device=tensor.device)
    return torch.linalg.vector_norm(tensor[i] - tensor[j], dim=1)


def squareform(tensor):
    """Condensed distance vector -> symmetric distance matrix with a zero diagonal (``tensor_utils.py:422-576``); 1-d input only.

    >>> import torch
    >>> from scipy.spatial.distance import pdist as scipy_pdist, squareform as scipy_squareform
    >>> x = torch.rand(4, 2, dtype=torch.float64)
    >>> bool(torch.allclose(squareform(pdist(x)), torch.from_numpy(scipy_squareform(scipy_pdist(x.numpy())))))
    True
    >>> squareform(torch.rand(4, 4))
    Traceback (most recent call last):
     ...
    NotImplementedError: tensor_utils.squareform: Only 1-d (vector) input is supported!
    >>> squareform(torch.rand(4))
    Traceback (most recent call last):
     ...
    ValueError: Incompatible vector size. It must be a binomial coefficient n choose 2 for some integer n >=2.
    """
    tensor = _as_tensor(tensor)
    if tensor.dim() != 1:
        raise NotImplementedError("tensor_utils.squareform: Only 1-d (vector) input is supported!")
    n = tensor.shape[0]
    if n == 0:
        return torch.zeros((1, 1), dtype=tensor.dtype, device=tensor.device)
    dimension = int(math.ceil(math.sqrt(n * 2)))
    if dimension * (dimension - 1) != n * 2:
        raise ValueError("Incompatible vector size. It must be a binomial coefficient n choose 2 for some integer n >=2.")
    out = torch.zeros((dimension, dimension), dtype=tensor.dtype, device=tensor.device)
    i, j = torch.triu_indices(dimension, dimension, offset=1, device=tensor.device)
    out[i, j] = tensor
    return out + out.t()
