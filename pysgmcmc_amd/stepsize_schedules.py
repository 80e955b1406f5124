"""Stepsize schedules: iterators that yield the epsilon fed to each sampler step.

Mirrors ``pysgmcmc/stepsize_schedules.py`` (base :4-34, constant :37-91). The
stepsize is a by-value scalar argument of the fused kernels, so any schedule
works without touching device code.
"""
from abc import ABCMeta, abstractmethod

__all__ = ["StepsizeSchedule", "ConstantStepsizeSchedule", "BurnInRampStepsizeSchedule"]


class StepsizeSchedule(object, metaclass=ABCMeta):
    """Base class: ``next(schedule)`` gives the next stepsize; ``update`` receives
    ``(params, cost)`` after every step (pysgmcmc/samplers/base_classes.py:306)."""

    def __init__(self, initial_value):
        self.initial_value = initial_value

    @abstractmethod
    def __next__(self):
        pass

    def __iter__(self):
        return self

    @abstractmethod
    def update(self, *args, **kwargs):
        pass


class ConstantStepsizeSchedule(StepsizeSchedule):
    """Always the initial value.

    >>> s = ConstantStepsizeSchedule(0.01)
    >>> s.initial_value, next(s)
    (0.01, 0.01)
    >>> from itertools import islice
    >>> list(islice(s, 4))
    [0.01, 0.01, 0.01, 0.01]
    >>> str(ConstantStepsizeSchedule(0.1))
    'ConstantStepsizeSchedule(stepsize=0.1)'
    """

    def __next__(self):
        return self.initial_value

    def __str__(self):
        return "ConstantStepsizeSchedule(stepsize={})".format(self.initial_value)

    def update(self, *args, **kwargs):
        pass


class BurnInRampStepsizeSchedule(StepsizeSchedule):
    """Linear ramp from ``initial_value`` to ``final_value`` over ``burn_in_steps``
    steps, constant afterwards (the "burn-in stepsize schedule" of BASELINE.json
    configs[4]; a new subclass of the reference's schedule API).

    >>> s = BurnInRampStepsizeSchedule(0.1, 0.5, burn_in_steps=4)
    >>> [round(next(s), 3) for _ in range(6)]
    [0.1, 0.2, 0.3, 0.4, 0.5, 0.5]
    """

    def __init__(self, initial_value, final_value, burn_in_steps):
        super().__init__(initial_value)
        assert burn_in_steps >= 0
        self.final_value = final_value
        self.burn_in_steps = int(burn_in_steps)
        self._t = 0

    def __next__(self):
        if self._t >= self.burn_in_steps:
            return self.final_value
        frac = self._t / float(self.burn_in_steps)
        self._t += 1
        return self.initial_value + (self.final_value - self.initial_value) * frac

    def state_dict(self):
        return {"t": self._t}

    def load_state_dict(self, state):
        self._t = int(state["t"])

    def __str__(self):
        return "BurnInRampStepsizeSchedule(initial={}, final={}, burn_in_steps={})".format(
            self.initial_value, self.final_value, self.burn_in_steps)

    def update(self, *args, **kwargs):
        pass
