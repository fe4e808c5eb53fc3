"""The C ABI library loads without a GPU, exports exactly what include/pecanpy_amd.h declares, and
the walk operator fails loudly (no CPU fallback) when no device is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from pecanpy_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(REPO, "include", "pecanpy_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pw_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_table_agree():
    assert header_functions() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    for name in header_functions():
        assert hasattr(lib, name), name
    assert b"pecanpy_amd" in lib.pw_version()


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/pecanpy_amd.h is a C header (what a cgo / cffi / ctypes binding consumes): a C99
    translation unit that takes the address of every declared entry point compiles warning-free and
    links against the shared library; the host-only services run from C without a GPU."""
    import shutil
    import subprocess

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no C compiler")
    names = header_functions()
    src = tmp_path / "abi.c"
    src.write_text(
        '#include <stdio.h>\n#include "pecanpy_amd.h"\n'
        "int main(void) {\n"
        "    typedef void (*fn)(void);\n"
        "    fn entry[] = {" + ", ".join(f"(fn)&{n}" for n in names) + "};\n"
        "    double d[2];\n"
        "    if (pw_mt_random_sample(0u, 0, 2, d) != PW_OK) return 2;\n"
        '    printf("%zu|%s|%.17g\\n", sizeof(entry) / sizeof(entry[0]), pw_version(), d[0]);\n'
        "    return 0;\n}\n")
    exe = tmp_path / "abi"
    libdir = os.path.dirname(_lib.LIB_PATH)
    cmd = [gcc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", f"-I{os.path.join(REPO, 'include')}",
           str(src), "-o", str(exe), f"-L{libdir}", "-l:" + os.path.basename(_lib.LIB_PATH),
           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-Wl,--unresolved-symbols=ignore-in-shared-libs"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    count, version, first = out.stdout.strip().split("|")
    assert int(count) == len(names) and "pecanpy_amd" in version
    assert float(first) == np.random.RandomState(0).random_sample()   # MT19937 stream from C


def test_reference_interface_citations_present():
    """every entry point documents the reference code it replaces (file:line)"""
    text = open(os.path.join(REPO, "include", "pecanpy_amd.h")).read()
    assert "pecanpy.py:164-210" in text and "graph.py:409-413" in text and "pecanpy.py:442-507" in text


@pytest.mark.skipif(_lib.load().pw_device_count() > 0, reason="needs a box WITHOUT a GPU")
def test_no_cpu_fallback_without_gpu():
    from pecanpy_amd.engine import WalkEngine

    indptr = np.array([0, 1, 2], dtype=np.uint32)
    indices = np.array([1, 0], dtype=np.uint32)
    with pytest.raises(_lib.PwError, match="no HIP device"):
        WalkEngine.from_csr(indptr, indices, None)


@pytest.mark.skipif(_lib.load().pw_device_count() > 0, reason="needs a box WITHOUT a GPU")
def test_warmup_without_a_gpu_is_an_error_in_c_and_harmless_in_the_host_layer():
    """pw_warmup (the library's start-up, meant for a helper thread beside the graph read) reports the missing device like every
    other entry point; the host layer's warmup_async swallows that -- the first graph handle then fails loudly as before."""
    ms = C.c_double(-1.0)
    assert _lib.load().pw_warmup(C.c_int(0), C.byref(ms)) != 0
    assert b"no HIP device" in _lib.load().pw_last_error()
    _lib.warmup_async(0)
    t = _lib._warm["thread"]
    assert t is not None
    t.join(timeout=30.0)
    assert not t.is_alive() and _lib.warmup_ms() is None


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.PwError, match="no CPU fallback"):
        _lib.load()


def test_product_does_not_import_the_oracle():
    """oracle/ is test infrastructure: nothing under pecanpy_amd/ may reference it."""
    for root, _, files in os.walk(os.path.join(REPO, "pecanpy_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hpp", ".hip")):
                src = open(os.path.join(root, f), errors="ignore").read()
                assert "pyoracle" not in src and "liboracle" not in src and "from oracle" not in src, f
