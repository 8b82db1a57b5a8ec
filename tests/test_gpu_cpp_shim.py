"""GPU: the rcgpar-shaped C++ shim (msweep_amd/cpp/rcgpar_hip.hpp) driven the way
src/mSWEEP.cpp:176-205,419-423 drives rcgpar, compiled with g++ against libmsweep_core.so."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden
from msweep_amd.core import MswError
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


@pytest.mark.parametrize("mode", ["intersection", "union"])
def test_device_likelihood_from_themisto_files(device_likelihood_binary, gpu_core, tmp_path, mode):
    """msw::DeviceLikelihood::build_from_files -- the reader on the device + the build from its resident classes, the C++
    face of Alignment::read + collapse + ConstructAdaptiveLikelihood -- against the Python mirror of the same entry
    points on the same two strands."""
    from msweep_amd import synth
    from msweep_amd.likelihood import from_device_alignment
    p = synth.make_csr_problem(4000, 15, seed=62, max_other=3, theta_support=9)
    aln = synth.csr_to_targets(p)
    E = len(p["ec_counts"])
    rng = np.random.default_rng(5)
    ec_of = rng.permutation(np.repeat(np.arange(E, dtype=np.int64), p["ec_counts"].astype(np.int64)))
    files = [str(tmp_path / "r1.txt"), str(tmp_path / "r2.txt")]
    for k, path in enumerate(files):
        synth.write_themisto(path, ec_of, aln["ec_tptr"], aln["ec_targets"], chunk=1000,
                             extra=(rng, 0.1, aln["n_targets"]) if k else None)
    txt = f"{aln['n_targets']} 15\n" + " ".join(str(int(x)) for x in aln["target_group"]) + "\n" + \
          " ".join(str(int(x)) for x in p["group_sizes"]) + "\n"
    r = subprocess.run([device_likelihood_binary, "files", "1" if mode == "union" else "0", "1"] + files, input=txt,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    out = dict(line.split(" ", 1) for line in r.stdout.strip().splitlines())
    dev = gpu_core.read_alignment(files, int(aln["n_targets"]), mode)
    lik = from_device_alignment(gpu_core, dev, aln["target_group"], p["group_sizes"], min_hits=1, download_log_counts=True)
    res = gpu_core.solve(None, np.ones(lik.n_groups))
    assert (int(out["n_groups"]), int(out["n_ecs"]), int(out["n_reads"]), int(out["n_aligned"])) == \
           (lik.n_groups, dev.n_ecs, dev.n_reads, dev.n_aligned)
    np.testing.assert_array_equal(np.array(out["mask"].split(), int).astype(bool), lik.groups_considered())
    np.testing.assert_array_equal(np.array(out["counts"].split(), np.uint64), dev.ec_counts())
    np.testing.assert_array_equal(np.array(out["logc"].split(), float), lik.log_counts())
    assert int(out["iters"]) == res["iters"]
    np.testing.assert_array_equal(np.array(out["theta"].split(), float), res["theta"])


def test_reference_calls_keep_the_likelihood_resident_over_the_bootstrap_loop(reference_calls_binary, gpu_core, tmp_path):
    """src/mSWEEP.cpp:402,507: the same `ll_mat` goes to rcg_optl() 1 + --iters times.  Through the verbatim call
    expressions: ONE upload for the estimate and 20 replicates (the shim keeps the matrix resident, keyed by address,
    shape and a sampled content hash), every later call served at the cost of its solve + the dense gamma it must
    return; a matrix rewritten in place at the same address is uploaded again."""
    from conftest import dense_from_csr
    from msweep_amd import synth
    from msweep_amd.likelihood import from_dense, precalc_lls
    G, B = 120, 20
    p = synth.make_csr_problem(60000, G, seed=71, max_other=6)
    L = dense_from_csr(p, precalc_lls(p["group_sizes"]))
    E = L.shape[1]
    logc = np.log(p["ec_counts"].astype(float))
    alpha0 = np.ones(G)
    path = tmp_path / "problem.bin"
    with open(path, "wb") as f:
        np.array([G, E], np.uint64).tofile(f)
        L.tofile(f)
        logc.tofile(f)
        alpha0.tofile(f)
    r = subprocess.run([reference_calls_binary, "rcgcpu", str(path), str(B)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = dict(line.split(" ", 1) for line in r.stdout.strip().splitlines())
    print({k: v for k, v in out.items() if not k.startswith("theta")})
    assert int(out["uploads"]) == 1 and int(out["hits"]) == B
    assert int(out["uploads_after_change"]) == 2
    # a single cell edited in place IS caught (the key hashes every cell up to 2^27 cells), and an unchanged matrix
    # is not uploaded again
    assert int(out["uploads_after_one_cell"]) == 3 and int(out["uploads_after_no_change"]) == 3
    first, later, solve = float(out["first_ms"]), float(out["later_ms"]), float(out["later_solve_ms"])
    assert later < 0.6 * first, (first, later)          # the host copy + upload + device compression are paid once
    # the answers: the estimate, replicate 1 (log counts rotated by one) and the rewritten matrix, against the Python
    # mirror on the same inputs
    from_dense(gpu_core, L, logc)
    assert_theta(np.array(out["theta"].split(), float), gpu_core.solve(logc, alpha0)["theta"], rel=1e-9, abs_=1e-12)
    assert_theta(np.array(out["theta_b1"].split(), float), gpu_core.solve(np.roll(logc, -1), alpha0)["theta"], rel=1e-9, abs_=1e-12)
    L2 = np.where(L > -4.0, L - 0.25, L)
    from_dense(gpu_core, L2, logc)
    ref2 = gpu_core.solve(logc, alpha0)["theta"]
    th2 = np.array(out["theta_changed"].split(), float)
    assert_theta(th2, ref2, rel=1e-9, abs_=1e-12)
    assert np.max(np.abs(th2 - np.array(out["theta"].split(), float))) > 1e-6      # and it IS another answer
    print(f"first call {first:.1f} ms, later calls {later:.1f} ms each (solve {solve:.1f} ms + the dense gamma the boundary returns)")


def test_python_mirror_keeps_the_dense_likelihood_resident(gpu_core):
    from conftest import dense_from_csr
    from msweep_amd import rcgpar, synth
    from msweep_amd.likelihood import precalc_lls
    p = synth.make_csr_problem(20000, 40, seed=72, max_other=5)
    L = dense_from_csr(p, precalc_lls(p["group_sizes"]))
    logc = np.log(p["ec_counts"].astype(float))
    alpha0 = np.ones(40)
    rcgpar.forget_likelihood()
    u0, h0 = rcgpar.likelihood_cache_stats()
    a = rcgpar.rcg_optl("rcgcpu", L, logc, alpha0)
    b = rcgpar.rcg_optl("rcggpu", L, np.roll(logc, -1), alpha0)
    c = rcgpar.rcg_optl("emgpu", L, logc, alpha0)
    assert rcgpar.likelihood_cache_stats() == (u0 + 1, h0 + 2)
    assert a.core is b.core is c.core
    first_core = a.core
    L[L > -4.0] -= 0.25                      # rewritten in place: same object, same shape
    d = rcgpar.rcg_optl("rcgcpu", L, logc, alpha0)
    assert rcgpar.likelihood_cache_stats()[0] == u0 + 2 and d.core is not first_core
    assert first_core._h is None             # the handle it replaced was closed, not leaked
    assert np.max(np.abs(d.theta - a.theta)) > 1e-6
    # ONE cell edited in place: caught (the key is an xxh3 over every cell up to 2^27 cells)
    L[17, L.shape[1] // 3 + 5] -= 1e-3
    e = rcgpar.rcg_optl("rcgcpu", L, logc, alpha0)
    assert rcgpar.likelihood_cache_stats()[0] == u0 + 3 and e.core is not d.core
    # gamma() belongs to the call that returned it: after ANOTHER solve on the shared handle it raises instead of
    # handing out the later call's matrix ...
    f = rcgpar.rcg_optl("rcgcpu", L, np.roll(logc, -1), alpha0)
    assert f.core is e.core
    g_f = f.gamma()
    assert g_f.shape == L.shape
    with pytest.raises(MswError, match="solved again"):
        e.gamma()
    # ... unless the caller asked for it to be kept
    k1 = rcgpar.rcg_optl("rcgcpu", L, logc, alpha0, keep_gamma=True)
    k2 = rcgpar.rcg_optl("rcgcpu", L, np.roll(logc, -1), alpha0, keep_gamma=True)
    np.testing.assert_array_equal(k2.gamma(), g_f)
    assert np.max(np.abs(k1.gamma() - k2.gamma())) > 1e-9
    rcgpar.forget_likelihood()
    assert d.core._h is None
    with pytest.raises(MswError, match="closed"):
        f.gamma()
