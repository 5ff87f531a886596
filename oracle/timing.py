"""Exclusive wall-clock time per phase of the oracle's step (CPU baseline of bench.py) --
TEST INFRASTRUCTURE.  Off by default: `with phase(name)` costs one attribute test when disabled."""
import time
from contextlib import contextmanager

ACC = {}
ENABLED = False
_stack = []


def reset(enable=True):
    global ENABLED
    ACC.clear()
    del _stack[:]
    ENABLED = enable


@contextmanager
def phase(name):
    if not ENABLED:
        yield
        return
    t0 = time.perf_counter()
    _stack.append(0.0)
    try:
        yield
    finally:
        dt = time.perf_counter() - t0
        inner = _stack.pop()
        ACC[name] = ACC.get(name, 0.0) + dt - inner
        if _stack:
            _stack[-1] += dt


def timed(name):
    """Decorator form of `phase`."""
    import functools

    def deco(fn):
        @functools.wraps(fn)
        def wrapper(*a, **k):
            if not ENABLED:
                return fn(*a, **k)
            with phase(name):
                return fn(*a, **k)
        return wrapper
    return deco
