"""GPU: the rcgpar-shaped C++ shim (msweep_amd/cpp/rcgpar_hip.hpp) driven the way
src/mSWEEP.cpp:176-205,419-423 drives rcgpar, compiled with g++ against libmsweep_core.so."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden
from test_gpu_rcg import assert_theta

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def shim_binary(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("shim") / "shim_test")
    lib = os.path.join(ROOT, "msweep_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", out, os.path.join(ROOT, "tests", "cpp", "shim_test.cpp"),
                           "-L" + lib, "-lmsweep_core", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib",
                           "-L/opt/rocm/lib"])
    return out


def _run(binary, mode, L, logc, alpha0):
    G, E = L.shape
    txt = f"{G} {E}\n" + "\n".join(" ".join(repr(float(x)) for x in row) for row in L) + "\n"
    txt += " ".join("-inf" if not np.isfinite(x) else repr(float(x)) for x in logc) + "\n"
    txt += " ".join(repr(float(x)) for x in alpha0) + "\n"
    p = subprocess.run([binary] + ([mode] if mode else []), input=txt, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    out = dict(line.split(" ", 1) for line in p.stdout.strip().splitlines())
    return out


@pytest.mark.parametrize("idx", [1, 2, 4])
def test_rcg_optl_torch_shim(shim_binary, oracle, idx):
    c = load_golden("rcg_golden.json")["cases"][idx]
    L = np.array(c["logl"]); logc = np.array(c["logc"], float); alpha0 = np.array(c["alpha0"])
    out = _run(shim_binary, "", L, logc, alpha0)
    theta = np.array([float(x) for x in out["theta"].split()])
    assert_theta(theta, c["expect"]["theta"])
    assert float(out["colsum_err"]) < 1e-12
    assert int(out["log_lines"]) == (c["expect"]["iters"] + 4) // 5      # one line per 5th iteration
    assert out["error_path"].startswith("ok:")


def test_em_torch_shim(shim_binary, oracle):
    c = load_golden("rcg_golden.json")["cases"][2]
    L = np.array(c["logl"]); logc = np.array(c["logc"], float); alpha0 = np.array(c["alpha0"])
    out = _run(shim_binary, "em", L, logc, alpha0)
    theta = np.array([float(x) for x in out["theta"].split()])
    # the shim, like mSWEEP (src/mSWEEP.cpp:422), forms theta with mixture_components on the returned gamma
    ref = oracle.em_dense(L, logc, alpha0, tol=1e-6, max_iters=5000, want_gamma=True)
    np.testing.assert_allclose(theta, oracle.mixture_components(ref["gamma"], logc), rtol=1e-6, atol=1e-9)
