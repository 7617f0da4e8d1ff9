"""Helpers of the training driver: seeding, two step predicates and a stopwatch. They take the place of what
train_representation.py uses from /root/reference/r3m/utils/utils.py (seeding :34-39, step predicates :85-101, timer :104-118)
under the same names, so `from r3m.utils import utils` keeps working; written for this loop (frame-skip arguments of the
reference's RL ancestry have no meaning here and are gone)."""
import random
import time

import numpy as np
import torch


def set_seed_everywhere(seed):
    """Seed every generator a training process draws from: python, numpy, torch CPU and all visible GPUs."""
    seed = int(seed)
    random.seed(seed)
    np.random.seed(seed % (2 ** 32))
    torch.manual_seed(seed)          # seeds the CUDA/HIP generators too (lazily, also for devices initialised later)


class Until:
    """until(step) is True while step < limit; a limit of None never stops."""

    def __init__(self, limit, _unused=1):
        self.limit = limit

    def __call__(self, step):
        return self.limit is None or step < self.limit


class Every:
    """every(step) is True on multiples of `period`; a period of None (or 0) never fires."""

    def __init__(self, period, _unused=1):
        self.period = period

    def __call__(self, step):
        return bool(self.period) and step % self.period == 0


class Timer:
    """Stopwatch on the monotonic clock: reset() -> (seconds since the previous reset, seconds since construction)."""

    def __init__(self):
        self._t0 = self._lap = time.monotonic()

    def reset(self):
        now = time.monotonic()
        lap, self._lap = now - self._lap, now
        return lap, now - self._t0

    def total_time(self):
        return time.monotonic() - self._t0
