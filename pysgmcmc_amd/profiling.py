"""Timing of a sampler's fused update kernel, launch by launch (the reference has no profiling hooks, SURVEY.md
section 5; this is what ``bench.py`` builds its ``roofline`` entry from).

    timer = UpdateKernelTimer(reserve=200)
    sampler.kernel_timer = timer
    timer.enabled = True
    for _ in range(200):
        next(sampler)
    torch.cuda.synchronize()
    timer.kernel_us()          # duration of every update launch, from the kernel's own start/stop timestamps

Every timed launch carries a pair of HIP events (``sgmcmc_launch_t.start_event / stop_event`` -> ``hipExtLaunchKernel``)
that receive the KERNEL's start and stop timestamps -- the duration rocprofv3 reports for the kernel. The events are not
free: a launch that carries them costs the 10 M-parameter chain 8 us of device time per step (measured in round 3, profiles/HISTORY.md),
so ``timer.sample_every = k`` times the launches of every k-th step only. ``bracket=True``
additionally records a ``hipEventRecord`` pair around the call on the same stream (it includes ~3-5 us of barrier-packet
and dispatch latency). Timing applies to direct launches (eager stepping and ``use_hip_graph = True``); a launch captured
into a hipGraph (``use_hip_graph = "full"``) cannot carry events and is not timed.
"""
import numpy as np
import torch

from pysgmcmc_amd import kernels

__all__ = ["UpdateKernelTimer"]


class UpdateKernelTimer(object):
    def __init__(self, reserve=0, bracket=False, device=None):
        """``device``: the device of the sampler being timed (events belong to a device; default: the current one)."""
        self.device = device
        self.enabled = False
        self.sample_every = 1      # time every k-th launch only (the events of a timed launch cost a few microseconds of device time)
        self._seen = 0
        self.bracket = bool(bracket)
        self.kevents = []          # one KernelEvents per timed launch, in launch order
        self.tags = []             # per timed launch: None, or (step, lo, hi): the step number and the arena slice of the launch
        self.pairs = []            # (torch event, torch event) brackets, when bracket=True
        self._pool = []
        self._current = None
        if reserve:
            self.reserve(reserve)

    def reserve(self, n):
        """Create the events of ``n`` timed launches up front (no event creation inside a timed loop)."""
        self._pool = [self._new_events() for _ in range(int(n))]

    def _new_events(self):
        with torch.cuda.device(self.device if self.device is not None else torch.cuda.current_device()):
            return (kernels.KernelEvents(self.device), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))

    # -- called by the sampler around its update launch --
    def due(self, tag=None):
        """Called once per launch while enabled: is this launch one of the timed ones? With ``sample_every = k`` the launches
        of every k-th step are (all launches of a step together: the tag's first entry is the step number)."""
        k = max(int(self.sample_every), 1)
        if tag is not None:
            return tag[0] % k == 0
        self._seen += 1
        return (self._seen - 1) % k == 0

    def begin(self, tag=None):
        self._tag = tag
        self._current = self._pool.pop() if self._pool else self._new_events()
        if self.bracket:
            self._current[1].record()

    def launch_config(self, base):
        """The launch geometry ``base`` (or the defaults) plus this launch's timestamp events."""
        base = base if base is not None else kernels._default_launch      # the sweep tools' Python-side default
        geom = base.as_dict() if base is not None else {}
        return kernels.LaunchConfig(events=self._current[0], **geom)

    def end(self, launched=True):
        """``launched=False``: the launch raised -- its never-recorded events are dropped, not kept as a sample."""
        kev, e0, e1 = self._current
        self._current = None
        if not launched:
            return
        if self.bracket:
            e1.record()
            self.pairs.append((e0, e1))
        self.kevents.append(kev)
        self.tags.append(getattr(self, "_tag", None))

    # -- results (after the stream has been synchronised) --
    def kernel_us(self):
        return np.array([k.elapsed_us() for k in self.kevents])

    def step_us(self):
        """Device time of a whole step: from the end of one timed step's LAST update launch to the end of the next timed
        step's, divided by the number of steps in between (with ``sample_every = k`` only every k-th step carries events, so
        consecutive entries are k steps apart; untagged launches count as consecutive steps)."""
        kv, steps = self.last_launch_of_each_step(with_steps=True)
        out = []
        for j in range(len(kv) - 1):
            gap = 1 if (steps[j] is None or steps[j + 1] is None) else max(int(steps[j + 1]) - int(steps[j]), 1)
            out.append(kv[j].us_until(kv[j + 1]) / gap)
        return np.array(out)

    def last_launch_of_each_step(self, with_steps=False):
        out, steps, prev = [], [], object()
        for kev, tag in zip(self.kevents, self.tags):
            step = tag[0] if tag is not None else None
            if tag is not None and step == prev:
                out[-1] = kev
            else:
                out.append(kev)
                steps.append(step)
            prev = step if tag is not None else object()
        return (out, steps) if with_steps else out

    def per_step_kernel_us(self):
        """(sum of the update launches' durations per step, bytes-weighted slices included)."""
        sums, prev = [], object()
        for kev, tag in zip(self.kevents, self.tags):
            us = kev.elapsed_us()
            step = tag[0] if tag is not None else None
            if tag is not None and step == prev:
                sums[-1] += us
            else:
                sums.append(us)
            prev = step if tag is not None else object()
        return np.array(sums)

    def bracket_us(self):
        return np.array([a.elapsed_time(b) * 1e3 for a, b in self.pairs])

    @staticmethod
    def empty_bracket_us(reps=200):
        """Elapsed time of an EMPTY hipEventRecord pair: the fixed cost every bracketed launch carries."""
        torch.cuda.synchronize()
        pairs = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            e1.record()
            pairs.append((e0, e1))
        torch.cuda.synchronize()
        return float(np.median([a.elapsed_time(b) for a, b in pairs])) * 1e3
