"""Console stage timing for the command line (the reference prints one ``Took hh:mm:ss.ss to <stage>``
line per pipeline stage, src/pecanpy/wrappers.py:5-27; same text here, different machinery)."""
import contextlib
import functools
import time


def _clock_text(seconds):
    minutes, sec = divmod(seconds, 60.0)
    hours, minutes = divmod(int(minutes), 60)
    return f"{hours:02d}:{minutes:02d}:{sec:05.2f}"


@contextlib.contextmanager
def stage(label, enabled=True):
    """``with stage("load Graph"): ...`` -- reports the wall time of the block on exit."""
    began = time.perf_counter()
    try:
        yield
    finally:
        if enabled:
            print(f"Took {_clock_text(time.perf_counter() - began)} to {label}")


class Timer:
    """Decorator form (``@Timer("generate walks")``), kept because the reference exposes it."""

    def __init__(self, name, verbose=True):
        self.name, self.verbose = name, verbose

    def __call__(self, fn):
        if not self.verbose:
            return fn   # the reference returns the function untouched when timing is off

        @functools.wraps(fn)
        def wrapped(*a, **kw):
            with stage(self.name, self.verbose):
                return fn(*a, **kw)

        return wrapped
