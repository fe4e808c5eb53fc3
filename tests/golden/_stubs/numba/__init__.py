"""Interpreter stand-in for ``numba`` (golden-vector generation only).

Every ``@njit`` body in the reference is also valid plain Python, so an identity
decorator lets the reference run on NumPy in the interpreter.  This file contains
no reference code; it only exists so ``tests/golden/make_golden.py`` can import the
reference *in the build container* (the reference never travels to the GPU box).
"""
import numpy as np


def njit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def deco(fn):
        return fn

    return deco


jit = njit
prange = range
boolean = np.bool_


def set_num_threads(n):
    return None


def get_num_threads():
    return 1


class _Config:
    NUMBA_DEFAULT_NUM_THREADS = 1
    NUMBA_NUM_THREADS = 1


config = _Config()
