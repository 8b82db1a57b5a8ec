"""GPU: the rcgpar-shaped C++ shim (msweep_amd/cpp/rcgpar_hip.hpp) driven the way
src/mSWEEP.cpp:176-205,419-423 drives rcgpar, compiled with g++ against libmsweep_core.so."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden
from test_gpu_rcg import assert_theta

pytestmark = pytest.mark.gpu


def _compile(tmp_path_factory, name):
    out = str(tmp_path_factory.mktemp("shim") / name)
    lib = os.path.join(ROOT, "msweep_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-o", out, os.path.join(ROOT, "tests", "cpp", name + ".cpp"),
                           "-L" + lib, "-lmsweep_core", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib",
                           "-L/opt/rocm/lib"])
    return out


@pytest.fixture(scope="module")
def shim_binary(tmp_path_factory):
    return _compile(tmp_path_factory, "shim_test")


@pytest.fixture(scope="module")
def reference_calls_binary(tmp_path_factory):
    return _compile(tmp_path_factory, "reference_calls_test")


@pytest.fixture(scope="module")
def device_likelihood_binary(tmp_path_factory):
    return _compile(tmp_path_factory, "device_likelihood_test")


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


@pytest.mark.parametrize("algorithm", ["rcgcpu", "rcggpu", "emgpu"])
def test_reference_call_expressions_compile_and_run(reference_calls_binary, oracle, algorithm):
    """src/mSWEEP.cpp:194,198,202,420,422 verbatim (tests/cpp/reference_calls_test.cpp) against the shim: the
    reference's default --algorithm rcgcpu (rcgpar::rcg_optl_omp + mixture_components) included."""
    c = load_golden("rcg_golden.json")["cases"][2]
    L = np.array(c["logl"]); logc = np.array(c["logc"], float); alpha0 = np.array(c["alpha0"])
    out = _run(reference_calls_binary, algorithm, L, logc, alpha0)
    theta = np.array([float(x) for x in out["theta"].split()])
    if algorithm == "emgpu":
        ref = oracle.em_dense(L, logc, alpha0, tol=1e-6, max_iters=5000, want_gamma=True)
        np.testing.assert_allclose(theta, oracle.mixture_components(ref["gamma"], logc), rtol=1e-6, atol=1e-9)
    else:
        assert_theta(theta, c["expect"]["theta"])


def test_device_likelihood_build_and_accessors(device_likelihood_binary, gpu_core):
    """msw::DeviceLikelihood::build / log_counts / groups_considered / solve without log counts against the
    Python mirror of the same C entry points."""
    from msweep_amd import synth
    from msweep_amd.likelihood import from_alignment
    p = synth.make_csr_problem(3000, 12, seed=61, max_other=3, theta_support=8)
    aln = synth.csr_to_targets(p)
    E, G = len(p["rowptr"]) - 1, 12
    txt = f"{E} {len(aln['ec_targets'])} {aln['n_targets']} {G} 1\n"
    for a in (aln["ec_tptr"], aln["ec_targets"], aln["target_group"], p["group_sizes"], p["ec_counts"]):
        txt += " ".join(str(int(x)) for x in a) + "\n"
    r = subprocess.run([device_likelihood_binary], input=txt, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    out = dict(line.split(" ", 1) for line in r.stdout.strip().splitlines())
    lik = from_alignment(gpu_core, aln["ec_tptr"], aln["ec_targets"], aln["target_group"], p["group_sizes"],
                         p["ec_counts"], min_hits=1)
    res = gpu_core.solve(None, np.ones(lik.n_groups))
    assert int(out["n_groups"]) == lik.n_groups and int(out["n_ecs"]) == E
    np.testing.assert_array_equal(np.array(out["mask"].split(), int).astype(bool), lik.groups_considered())
    np.testing.assert_array_equal(np.array(out["logc"].split(), float), lik.log_counts())
    assert int(out["iters"]) == res["iters"]
    np.testing.assert_array_equal(np.array(out["theta"].split(), float), res["theta"])
