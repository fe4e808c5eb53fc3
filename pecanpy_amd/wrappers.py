"""Stage timer used by the CLI (same console output as reference src/pecanpy/wrappers.py:5-27)."""
import functools
import time


class Timer:
    """Decorator factory printing ``Took hh:mm:ss.ss to <name>`` after the wrapped call."""

    def __init__(self, name, verbose=True):
        self.name = name
        self.verbose = verbose

    def __call__(self, func):
        if not self.verbose:
            return func

        @functools.wraps(func)
        def timed(*args, **kwargs):
            t0 = time.time()
            result = func(*args, **kwargs)
            dt = time.time() - t0
            h, rem = divmod(dt, 3600)
            m, s = divmod(rem, 60)
            print(f"Took {int(h):02d}:{int(m):02d}:{s:05.2f} to {self.name}")
            return result

        return timed
