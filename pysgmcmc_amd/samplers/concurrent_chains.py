"""Independent chains that SHARE one GPU, stepped concurrently.

The reference runs the chains of an ensemble one after the other (``pysgmcmc/diagnostics/sample_chains.py:369-382``).
Chains are independent, so on the device their steps need no ordering at all: every chain gets its own HIP stream (and, with
``use_hip_graph``, its own hipGraph) and one host thread enqueues the chains round-robin. The kernels of a step each leave part
of the chip idle -- the batch-256 GEMMs of the 10 M-parameter BNN keep the matrix pipe 62 % busy, the launch ramps and tails of
~11 dependent launches per step nothing at all -- and a second chain's launches fill those gaps: two such chains on one MI355X
give 5.9 k samples/s together against 5.0 k for one (``bench.py``'s ``chains_per_gpu`` leg; three or more are host-bound from one
thread). WITHIN a chain the same trick does not work: the update cannot run next to the chain's own backward pass without
cross-stream dependencies, and those cost more than they hide (profiles/HISTORY.md, round 3).

    chains = ConcurrentChains([make_sampler(seed=k) for k in range(2)])
    for step_results in itertools.islice(chains, 1000):     # [(sample, cost) of chain 0, (sample, cost) of chain 1]
        ...
    chains.join()                                            # the current stream now sees every chain's state

Every chain computes exactly what it computes alone (same kernels, same Philox stream); only the interleaving on the device
changes. Use ``sampler.sample_format = "view"`` (or a device format): the default numpy format copies every sample to the host
and synchronises, which serialises the chains.
"""
import torch

__all__ = ["ConcurrentChains"]


class ConcurrentChains(object):
    def __init__(self, samplers):
        samplers = list(samplers)
        if not samplers:
            raise ValueError("ConcurrentChains needs at least one sampler")
        device = samplers[0].device
        if device.type != "cuda" or any(s.device != device for s in samplers):
            raise ValueError("ConcurrentChains: the chains must live on one HIP device")
        self.samplers = samplers
        self.device = device
        current = torch.cuda.current_stream(device)
        self.streams = [torch.cuda.Stream(device=device) for _ in samplers]
        for stream in self.streams:
            stream.wait_stream(current)          # whatever built the chains (parameter init, data upload) comes first

    def __len__(self):
        return len(self.samplers)

    def __iter__(self):
        return self

    def __next__(self):
        """One step of every chain, enqueued round-robin; returns ``[(sample, cost), ...]`` in chain order (device results are
        ordered on the chain's own stream: ``join()`` / ``synchronize()`` before reading them from elsewhere)."""
        out = []
        for sampler, stream in zip(self.samplers, self.streams):
            with torch.cuda.stream(stream):
                out.append(next(sampler))
        return out

    def fork(self):
        """Every chain's stream waits (on the device) for what the current stream has enqueued so far -- e.g. copies of the
        chains' parameters taken after a ``join()``, before the chains move on."""
        current = torch.cuda.current_stream(self.device)
        for stream in self.streams:
            stream.wait_stream(current)

    def steps(self, n_steps):
        """``fork()``, ``n_steps`` steps of every chain, ``join()``: the chains advance concurrently, the caller's stream sees
        the result (the interface of ``FusedBNNChains.steps``)."""
        self.fork()
        last = self.run(n_steps)
        self.join()
        return last

    def run(self, n_steps):
        """``n_steps`` steps of every chain; returns the last step's results."""
        last = None
        for _ in range(int(n_steps)):
            last = next(self)
        return last

    def join(self):
        """The current stream waits (on the device) for everything enqueued on the chains' streams so far."""
        current = torch.cuda.current_stream(self.device)
        for stream in self.streams:
            current.wait_stream(stream)

    def synchronize(self):
        """The host waits for every chain."""
        for stream in self.streams:
            stream.synchronize()
