"""CPU: the C-ABI library loads and exports every symbol include/msweep_core.h declares; without a
GPU it fails loudly instead of falling back."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from msweep_amd import core


def _declared():
    src = open(os.path.join(ROOT, "include", "msweep_core.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(msw_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_are_exported():
    lib = core.load_library()
    names = _declared()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/msweep_core.h but not exported"
    assert sorted(core.EXPORTS) == names


def test_version_string():
    lib = core.load_library()
    assert b"gfx950" in lib.msw_core_version()


def test_no_silent_cpu_fallback():
    """On a box without a GPU creating a core must fail with a message; on a GPU box it succeeds."""
    lib = core.load_library()
    h = ctypes.c_void_p()
    rc = lib.msw_core_create(0, ctypes.byref(h))
    if rc != 0:
        msg = lib.msw_last_error(None).decode()
        assert msg, "error text must be set"
        with pytest.raises(core.MswError):
            core.Core(0)
    else:
        lib.msw_core_destroy(h)


def test_product_never_imports_oracle():
    """The shipped package must not import, link or load the oracle (test infrastructure)."""
    pk = os.path.join(ROOT, "msweep_amd")
    bad = re.compile(r"import\s+oracle|from\s+oracle|libmsweep_oracle|msweep_oracle\.h|orc_[a-z_]+\s*\(")
    for dp, _, fs in os.walk(pk):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".inc", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert not bad.search(txt), (f, bad.search(txt).group(0))


def test_reference_call_expressions_compile(tmp_path):
    """The rcgpar call expressions of src/mSWEEP.cpp:194,198,202,420,422, verbatim, compile against the C++
    shim (stand-in seamat types in the test TU); running them needs a GPU (tests/test_gpu_cpp_shim.py)."""
    import subprocess
    for tu in ("reference_calls_test", "shim_test", "device_likelihood_test"):
        subprocess.check_call(["g++", "-std=c++17", "-O0", "-Wall", "-c", "-o", str(tmp_path / (tu + ".o")),
                               os.path.join(ROOT, "tests", "cpp", tu + ".cpp")])


def test_library_carries_the_hash_of_its_sources():
    """msw_core_version() ends with the sha256 prefix of the sources the .so was compiled from; build()
    rebuilds and load_library() refuses a library that does not match the tree."""
    import __graft_entry__ as g
    lib = core.load_library()
    want = g.source_hash()
    assert core.source_hash() == want
    assert lib.msw_core_version().decode().endswith("src " + want)
    assert g.built_hash(g.LIB) == want
