"""Minibatch generators (mirror of ``pysgmcmc/data_batches.py``).

The reference yields ``{tf.placeholder: ndarray}`` feed dicts that the sampler
passes to ``session.run``. Here a :class:`Placeholder` is a tiny mutable slot
holding a device tensor; a cost function closes over its placeholders exactly
like a TF graph does, and the sampler "feeds" a batch by calling
``placeholder.feed(value)`` for every item of the yielded dict.

The dataset is moved to the device ONCE; a batch is a contiguous window
``[start, start+B)`` (``data_batches.py:118-123``), so feeding is a zero-copy
slice of the resident dataset. ``start`` comes from the same host RNG stream as
the reference (``numpy.random.RandomState(seed).randint(0, N-B+1)``), so batch
sequences are seed-matched with the reference.
"""
import logging

import numpy as np
import torch

__all__ = ["Placeholder", "WindowBatches", "generate_batches", "generate_shuffled_batches"]


class Placeholder(object):
    """Feedable slot, the torch stand-in for ``tf.placeholder``."""

    def __init__(self, dtype=None, shape=None, name=None, device=None):
        self.dtype = dtype
        self.shape = shape
        self.name = name
        self.device = device
        self.value = None

    def feed(self, value):
        if not isinstance(value, torch.Tensor):
            value = torch.as_tensor(np.asarray(value))
        if self.dtype is not None and value.dtype != self.dtype:
            value = value.to(self.dtype)
        if self.device is not None and value.device != torch.device(self.device):
            value = value.to(self.device)
        self.value = value
        return self

    def __repr__(self):
        return "Placeholder(name={!r}, dtype={}, shape={})".format(self.name, self.dtype, self.shape)


def _resident(a, placeholder):
    """Dataset as a tensor in the placeholder's dtype/device (one H2D copy)."""
    t = a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a))
    dtype = getattr(placeholder, "dtype", None)
    device = getattr(placeholder, "device", None)
    if dtype is not None:
        t = t.to(dtype)
    if device is not None:
        t = t.to(device)
    return t


def generate_batches(x, y, x_placeholder, y_placeholder, batch_size=20, seed=None):
    """Infinite generator of ``{x_placeholder: X[start:start+B], y_placeholder: Y[start:start+B, None]}``.

    >>> import numpy as np
    >>> N, D = 100, 3
    >>> X, y = np.random.uniform(-10, 10, (N, D)), np.random.choice([0., 1.], N)
    >>> xp, yp = Placeholder(), Placeholder()
    >>> batch = next(generate_batches(X, y, xp, yp, batch_size=10))
    >>> set(batch.keys()) == {xp, yp}
    True
    >>> tuple(batch[xp].shape), tuple(batch[yp].shape)
    ((10, 3), (10, 1))

    A batch size above the dataset size is clamped to it:

    >>> batch = next(generate_batches(X, y, xp, yp, batch_size=1000))
    >>> tuple(batch[xp].shape), tuple(batch[yp].shape)
    ((100, 3), (100, 1))
    """
    assert isinstance(batch_size, int), "generate_batches: batch size must be an integer."
    assert batch_size > 0, "generate_batches: batch size must be greater than zero."
    assert seed is None or isinstance(seed, int), "generate_batches: seed must be an integer or `None`"
    assert seed is None or (0 <= seed <= 2 ** 32 - 1)
    assert y.shape[0] == x.shape[0], "Not exactly one label per datapoint!"

    n_examples = x.shape[0]
    if seed is None:
        seed = np.random.randint(1, 100000)
    rng = np.random.RandomState()
    rng.seed(seed)

    initial_batch_size = batch_size
    batch_size = min(initial_batch_size, n_examples)
    if initial_batch_size != batch_size:
        logging.error("Not enough datapoints to form a minibatch. "
                      "Batchsize was set to %s", batch_size)

    x_dev = _resident(x, x_placeholder)
    y_dev = _resident(y, y_placeholder).reshape(n_examples, -1)[:, :1]
    return WindowBatches(x_dev, y_dev, x_placeholder, y_placeholder, batch_size, rng)


class WindowBatches(object):
    """The infinite minibatch iterator behind :func:`generate_batches`: ``next()`` yields the feed dict of
    one random contiguous window; ``next_starts(n)`` draws the start indices of the next ``n`` windows from
    the SAME RandomState stream (for kernels that gather the windows themselves)."""

    def __init__(self, x_dev, y_dev, x_placeholder, y_placeholder, batch_size, rng):
        self.x_dev, self.y_dev = x_dev, y_dev
        self.x_placeholder, self.y_placeholder = x_placeholder, y_placeholder
        self.batch_size = int(batch_size)
        self.n_examples = int(x_dev.shape[0])
        self._rng = rng

    def __iter__(self):
        return self

    def next_starts(self, n):
        hi = self.n_examples - self.batch_size + 1
        # one vectorised draw = the same values as n successive scalar randint calls (same RandomState stream)
        return self._rng.randint(0, hi, size=int(n)).astype(np.int32)

    def state_dict(self):
        """Position in the window stream (the RandomState), for ``sampler.state_dict()``: a resumed chain sees the same
        minibatches as the uninterrupted one."""
        return {"rng": self._rng.get_state()}

    def load_state_dict(self, state):
        self._rng.set_state(state["rng"])

    def __next__(self):
        start = int(self.next_starts(1)[0])
        return {
            self.x_placeholder: self.x_dev[start:start + self.batch_size],
            self.y_placeholder: self.y_dev[start:start + self.batch_size].reshape(-1, 1),
        }


def generate_shuffled_batches(x, y, x_placeholder, y_placeholder, batch_size=20, seed=None):
    """Like :func:`generate_batches`, rows of each window shuffled (x and y alike;
    ``data_batches.py:132-206``). The permutation stream is
    ``RandomState(seed).permutation`` applied identically to x and y."""
    if seed is None:
        seed = np.random.randint(1, 100000)
    rng = np.random.RandomState()
    rng.seed(seed)

    def batches():
        for batch in generate_batches(x, y, x_placeholder, y_placeholder, batch_size, seed):
            bx, by = batch[x_placeholder], batch[y_placeholder]
            perm = torch.as_tensor(rng.permutation(bx.shape[0]), device=bx.device)
            yield {x_placeholder: bx.index_select(0, perm), y_placeholder: by.index_select(0, perm)}
    return batches()
