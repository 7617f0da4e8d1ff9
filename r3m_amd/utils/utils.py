"""The three helpers the training driver uses (/root/reference/r3m/utils/utils.py:34-39,85-101,104-118); the rest of that
file is unused DrQ-v2 code and is out of scope."""
import random
import time

import numpy as np
import torch


def set_seed_everywhere(seed):
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


class Until:
    def __init__(self, until, action_repeat=1):
        self._until = until
        self._action_repeat = action_repeat

    def __call__(self, step):
        if self._until is None:
            return True
        return step < self._until // self._action_repeat


class Every:
    def __init__(self, every, action_repeat=1):
        self._every = every
        self._action_repeat = action_repeat

    def __call__(self, step):
        if self._every is None:
            return False
        return step % (self._every // self._action_repeat) == 0


class Timer:
    def __init__(self):
        self._start_time = time.time()
        self._last_time = time.time()

    def reset(self):
        elapsed = time.time() - self._last_time
        self._last_time = time.time()
        return elapsed, time.time() - self._start_time

    def total_time(self):
        return time.time() - self._start_time
