"""ctypes binding of libpecanpy_amd.so (the C ABI declared in include/pecanpy_amd.h).

There is no CPU fallback: if the shared library is missing or no GPU is visible the walk operator
raises, loudly.  ``build()`` compiles the library in-tree with hipcc (cross-compiles without a GPU).
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PECANPY_AMD_LIB") or os.path.join(_HERE, "libpecanpy_amd.so")  # env override: A/B kernel variants
_lib = None


class PwStats(C.Structure):
    _fields_ = [
        ("total_steps", C.c_uint64),
        ("overflow_reads", C.c_uint64),
        ("clamped_reads", C.c_uint64),
        ("dead_end_walks", C.c_uint64),
        ("repair_rounds", C.c_uint64),
        ("walk_kernel_ms", C.c_double),
        ("rng_kernel_ms", C.c_double),
        ("walk_kernel_launches", C.c_uint32),
        ("stream_addressing", C.c_uint32),
        ("lane_kernel", C.c_uint32),
        ("lane_rounds", C.c_uint32),
        ("redo_walks", C.c_uint64),
        ("list_entries_read", C.c_uint64),
        ("ambiguous_steps", C.c_uint64),
        ("lane_kernel_ms", C.c_double),
        ("wave_chain_steps", C.c_uint64),
        ("param_index_ms", C.c_double),
        ("verify_checked", C.c_uint64),
        ("verify_mismatch", C.c_uint64),
        ("verify_dropped", C.c_uint64),
        ("verify_ties", C.c_uint64),
        ("eager_steps", C.c_uint64),
        ("index_max_list", C.c_uint32),
        ("reserved0", C.c_uint32),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


MODE_IDS = {
    "SparseOTF": 0,
    "DenseOTF": 1,
    "PreComp": 2,
    "FirstOrderUnweighted": 3,
    "PreCompFirstOrder": 4,
}

# every symbol include/pecanpy_amd.h declares: (restype, argtypes)
_u32p = C.POINTER(C.c_uint32)
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_SIM_ARGS = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_uint64, C.c_uint32,
             C.c_int, C.c_uint32, C.c_uint64, C.c_void_p, C.POINTER(PwStats)]
SYMBOLS = {
    "pw_version": (C.c_char_p, []),
    "pw_last_error": (C.c_char_p, []),
    "pw_device_count": (C.c_int, []),
    "pw_warmup": (C.c_int, [C.c_int, _f64p]),
    "pw_csr_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int,
                                C.POINTER(C.c_void_p)]),
    "pw_graph_index_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "pw_lane_index_export": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pw_dense_create": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]),
    "pw_dense_create_bits": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "pw_graph_set_thresholds": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pw_graph_destroy": (None, [C.c_void_p]),
    "pw_simulate": (C.c_int, _SIM_ARGS),
    "pw_simulate_device": (C.c_int, _SIM_ARGS),
    "pw_graph_replicate": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "pw_device_mask_to_list": (C.c_int, [C.c_uint64, C.POINTER(C.c_int), C.c_int]),
    "pw_csr_create_multi": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_int), C.c_int,
                                      C.POINTER(C.c_void_p)]),
    "pw_simulate_multi": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_uint64,
                                    C.c_uint32, C.c_int, C.c_uint32, C.c_uint64, C.c_void_p, C.c_int, C.POINTER(PwStats)]),
    "pw_step": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, C.c_uint32, C.c_int, C.c_uint32, C.c_double,
                          _u32p, _u32p]),
    "pw_probs": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, C.c_uint32, C.c_int, C.c_uint32, C.c_void_p,
                           _u32p]),
    "pw_precomp_build": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int]),
    "pw_precomp_export": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]),
    "pw_count_stream_draws": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32,
                                        C.POINTER(C.c_uint64)]),
    "pw_stream_hold": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64]),
    "pw_stream_release": (C.c_int, [C.c_void_p]),
    "pw_sgns_train": (C.c_int, [C.c_int, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_uint32, C.c_void_p]),
    "pw_mt_random_sample": (C.c_int, [C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p]),
    "pw_stream_sample_device": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p]),
    "pw_noise_thresholds_csr": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_double, C.c_void_p]),
    "pw_noise_thresholds_csr_numpy1": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_double, C.c_void_p]),
    "pw_noise_thresholds_dense": (C.c_int, [C.c_void_p, C.c_uint32, C.c_double, C.c_void_p]),
    "pw_edgelist_read": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]),
    "pw_edgelist_shape": (C.c_int, [C.c_void_p] + [C.POINTER(C.c_uint64)] * 4),
    "pw_edgelist_export": (C.c_int, [C.c_void_p] * 7),
    "pw_edgelist_destroy": (None, [C.c_void_p]),
    "pw_selftest_exact_decision": (C.c_int, [C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_void_p, C.c_uint32,
                                             C.c_void_p, C.c_void_p]),
    "pw_selftest_lane": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_void_p, C.c_uint32,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pw_selftest_lane_floats": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_void_p, C.c_uint32,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "pw_selftest_lane_unit_bounded": (C.c_int, [C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "pw_selftest_lane_unit_tight": (C.c_int, [C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pw_selftest_lane_weighted": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "pw_selftest_exact_decision_f64": (C.c_int, [C.c_void_p, C.c_uint32, C.c_double, C.c_double, C.c_void_p,
                                                 C.c_uint32, C.c_void_p, C.c_void_p]),
    "pw_selftest_seqscan_f32": (C.c_int, [C.c_void_p, C.c_uint32, C.c_double, C.c_int, C.c_uint32,
                                          _u32p, _f32p]),
    "pw_selftest_seqscan_f64": (C.c_int, [C.c_void_p, C.c_uint32, C.c_double, C.c_int, C.c_uint32,
                                          _u32p, _f64p]),
}


class PwError(RuntimeError):
    pass


def build(verbose=False):
    """Compile libpecanpy_amd.so for gfx950 with hipcc (in-tree, so it travels with the repo)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc")]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise PwError("building libpecanpy_amd.so failed:\n" + res.stdout)
    return LIB_PATH


def load():
    """Load the shared library and type every exported symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PwError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C pecanpy_amd/csrc` -- the walk engine has no CPU fallback."
        )
    try:  # share torch's HIP runtime (same soname) when torch is used for device buffers / RCCL
        import torch  # noqa: F401
    except Exception:  # pragma: no cover
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


_warm = {"thread": None, "ms": None}


def warmup_async(device=0):
    """Start the library's one-time start-up on `device` (pw_warmup: ~140 ms for the first stream a process creates through the
    library) on a helper thread and return at once: callers with host work in front of their first graph handle -- reading an edge
    list, generating a graph -- call this first and find the runtime warm.  Harmless without a GPU or without the library (nothing
    happens); `warmup_ms()` tells what it took once it is done."""
    import threading

    if _warm["thread"] is not None or not os.path.exists(LIB_PATH):
        return

    def run():
        try:
            ms = C.c_double(0.0)
            if load().pw_warmup(C.c_int(device), C.byref(ms)) == 0:
                _warm["ms"] = ms.value
        except Exception:  # noqa: BLE001 (a warm-up that cannot run changes nothing: the first handle pays the start-up)
            pass

    _warm["thread"] = threading.Thread(target=run, name="pecanpy_amd-warmup", daemon=True)
    _warm["thread"].start()
    import atexit

    atexit.register(lambda: _warm["thread"].join(timeout=5.0))   # (the interpreter does not leave while the runtime is starting)


def warmup_ms():
    """Wall clock of the finished warm-up in ms, or None (not started / still running / no GPU)."""
    t = _warm["thread"]
    if t is not None and not t.is_alive():
        return _warm["ms"]
    return None


def check(rc):
    if rc != 0:
        msg = load().pw_last_error()
        raise PwError(f"libpecanpy_amd error {rc}: {msg.decode() if msg else '?'}")
