"""GPU: results of a REAL mSWEEP run, when somebody has dropped them under tests/golden/external/ (README.md there:
the commands, the files, what is compared).  The only road to an oracle pinned by the reference itself -- the
reference tree ships no expected output.  Skips while the directory holds no fixture."""
import glob
import json
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN
from msweep_amd.__main__ import read_likelihood_file
from msweep_amd.core import ALGO_EM, ALGO_RCG, PREC_DOUBLE, PREC_FLOAT
from msweep_amd.likelihood import from_dense

pytestmark = pytest.mark.gpu
EXT = os.path.join(GOLDEN, "external")


def fixtures(root=EXT):
    out = []
    for lik in sorted(glob.glob(os.path.join(root, "*_likelihoods.tsv")) + glob.glob(os.path.join(root, "*_likelihoods.txt"))):
        name = lik[:-len("_likelihoods.tsv")]
        if os.path.exists(name + "_abundances.txt"):
            out.append((os.path.basename(name), lik, name + "_abundances.txt"))
    return out


def read_abundances(path):
    """(names, theta) of a <prefix>_abundances.txt (src/PlainSample.cpp:32-71; bootstrap files: the first column)."""
    names, theta = [], []
    for ln in open(path):
        if ln.startswith("#") or not ln.strip():
            continue
        parts = ln.rstrip("\n").split("\t")
        names.append(parts[0])
        theta.append(float(parts[1]))
    return names, np.array(theta)


def printed_half_unit(x):
    """half a unit of the sixth significant digit of x as C++'s default ostream prints it"""
    x = np.abs(np.asarray(x, float))
    return np.where(x > 0, 0.5 * 10.0 ** (np.floor(np.log10(np.maximum(x, 1e-300))) - 5), 0.5e-11)


def assert_at_file_precision(tag, got, ref):
    tol = np.where(ref >= 1e-4, 1e-6 * ref, 1e-8) + printed_half_unit(ref)
    err = np.abs(got - ref)
    i = int(np.argmax(err / tol))
    print(f"{tag}: worst |diff| / tolerance = {err[i] / tol[i]:.3f} at group {i} (file {ref[i]:g}, got {got[i]:.9g})")
    assert np.all(err <= tol), (tag, i, got[i], ref[i])


def parse_log(path):
    """rcgpar's verbose lines `iter: k, bound: b, |g|: n` (every 5th iteration)"""
    out = []
    for ln in open(path):
        m = re.search(r"iter:\s*(\d+),\s*bound:\s*([-+0-9.eE]+|-?inf|nan),\s*\|g\|:\s*([-+0-9.eE]+|-?inf|nan)", ln)
        if m:
            out.append((int(m.group(1)), float(m.group(2)), float(m.group(3))))
    return out


def run_fixture(core, oracle, name, lik_path, ab_path):
    names, theta_ref = read_abundances(ab_path)
    G = len(open(lik_path).readline().rstrip("\n").split("\t")) - 1
    assert 0 < G <= len(names), "fewer abundance rows than likelihood columns"
    ec_counts, L = read_likelihood_file(lik_path, G)
    logc = np.log(ec_counts.astype(np.float64))
    opts = json.load(open(lik_path.rsplit("_likelihoods", 1)[0] + ".json")) if os.path.exists(
        lik_path.rsplit("_likelihoods", 1)[0] + ".json") else {}
    algo = ALGO_EM if opts.get("algorithm", "rcgcpu") not in ("rcgcpu", "rcggpu") else ALGO_RCG   # src/mSWEEP.cpp:192-204
    alpha0 = np.asarray(opts.get("alphas", np.ones(G)), float)
    tol, max_iters = float(opts.get("tol", 1e-6)), int(opts.get("max_iters", 5000))
    prec = PREC_FLOAT if opts.get("emprecision") == "float" else PREC_DOUBLE
    from_dense(core, L, logc)
    res = core.solve(logc, alpha0, tol, max_iters, algo, prec)
    print(f"{name}: {G} groups x {L.shape[1]} ECs, {res['iters']} iterations on the HIP path")
    assert np.all(theta_ref[G:] == 0.0)          # pruned groups: a literal 0 (src/PlainSample.cpp:56-66)
    assert_at_file_precision(f"{name}: HIP vs the file", res["theta"], theta_ref[:G])
    if algo == ALGO_RCG:
        d = oracle.rcg_optl_dense(L, logc, alpha0, tol, max_iters)
        th = oracle.mixture_components(d["gamma"], logc)
    else:
        d = oracle.em_dense(L, logc, alpha0, tol, max_iters)
        th = d["theta"]
    print(f"{name}: oracle {d['iters']} iterations")
    assert_at_file_precision(f"{name}: oracle vs the file", th, theta_ref[:G])
    log_path = lik_path.rsplit("_likelihoods", 1)[0] + "_log.txt"
    if os.path.exists(log_path) and algo == ALGO_RCG:
        lines = parse_log(log_path)
        assert lines, "no `iter: k, bound: b` line in the log"
        tr = core.trace(res["iters"])
        for k, b, nn in lines:
            assert k < tr["n"], f"the file logs iteration {k}, the HIP path stopped after {res['iters']}"
            assert abs(tr["bound"][k] - b) <= 1e-6 * abs(b) + printed_half_unit(b), (k, tr["bound"][k], b)
            assert abs(tr["newnorm"][k] - nn) <= 1e-4 * abs(nn) + printed_half_unit(nn), (k, tr["newnorm"][k], nn)
        assert lines[-1][0] <= res["iters"] - 1 < lines[-1][0] + 5   # the last logged iteration: k % 5 == 0
        print(f"{name}: {len(lines)} logged iterations match the HIP trace; last logged {lines[-1][0]}, stop {res['iters']}")
    return res


def test_external_fixtures(gpu_core, oracle):
    fx = fixtures()
    if not fx:
        pytest.skip("tests/golden/external/ holds no <name>_likelihoods.tsv + <name>_abundances.txt pair (README.md there)")
    for name, lik, ab in fx:
        run_fixture(gpu_core, oracle, name, lik, ab)


def test_the_hook_itself_on_a_self_made_fixture(gpu_core, oracle, tmp_path):
    """The plumbing of the hook (file formats, tolerances at six printed digits, the log parser) exercised on files this
    repository's own CLI mirror writes -- self-consistency, NOT reference-derived (flagged so in the README)."""
    from test_gpu_cli_toy import _toy
    from msweep_amd.__main__ import main
    _toy(tmp_path, n_reads=600)
    pre = str(tmp_path / "toy")
    assert main(["--themisto-1", str(tmp_path / "toy_1.txt"), "--themisto-2", str(tmp_path / "toy_2.txt"), "-i",
                 str(tmp_path / "clustering.txt"), "-o", pre, "--write-likelihood", "--no-fit-model"]) == 0
    import contextlib
    import io
    err = io.StringIO()
    with contextlib.redirect_stderr(err):
        assert main(["--read-likelihood", pre + "_likelihoods.tsv", "-i", str(tmp_path / "clustering.txt"), "-o", pre,
                     "--verbose"]) == 0
    open(pre + "_log.txt", "w").write(err.getvalue())
    fx = fixtures(str(tmp_path))
    assert [f[0] for f in fx] == ["toy"]
    res = run_fixture(gpu_core, oracle, *fx[0])
    assert res["iters"] > 1 and parse_log(pre + "_log.txt")
