"""GPU: config[0] -- a toy Themisto plaintext alignment through the command line
(`python -m msweep_amd`, the flags of docs/example.md:43-51) against the oracle's restatement of
the same pipeline (Likelihood.hpp counts + LUT + dense matrix, rcgpar loop, bootstrap stream)."""
import io
import os

import numpy as np
import pytest

from msweep_amd.__main__ import main
from msweep_amd.alignment import Alignment
from msweep_amd.reference import read_reference
from test_gpu_rcg import assert_theta

pytestmark = pytest.mark.gpu


def _toy(tmp_path, n_reads=1500, seed=3):
    """4 clusters x ~10 reference sequences, paired-end reads drawn from theta = (.5,.3,.15,.05)."""
    rng = np.random.default_rng(seed)
    sizes = [12, 9, 10, 8]
    names = [f"clust{k + 1}" for k in range(4)]
    indicators = [names[k] for k in range(4) for _ in range(sizes[k])]
    order = rng.permutation(len(indicators))
    indicators = [indicators[i] for i in order]            # group members scattered over target ids
    members = {n: [i for i, x in enumerate(indicators) if x == n] for n in names}
    theta = [0.5, 0.3, 0.15, 0.05]
    l1, l2 = [], []
    for r in range(n_reads):
        g = rng.choice(4, p=theta)
        hit = set(rng.choice(members[names[g]], max(1, rng.binomial(sizes[g], 0.65)), replace=False).tolist())
        for o in range(4):
            if o != g and rng.random() < 0.3:
                hit |= set(rng.choice(members[names[o]], max(1, rng.binomial(sizes[o], 0.15)), replace=False).tolist())
        h1 = sorted(hit | ({int(rng.integers(0, len(indicators)))} if rng.random() < 0.1 else set()))
        h2 = sorted(hit) if rng.random() > 0.05 else []
        l1.append(" ".join(map(str, [r] + h1)))
        l2.append(" ".join(map(str, [r] + h2)))
    (tmp_path / "toy_1.txt").write_text("\n".join(l1) + "\n")
    (tmp_path / "toy_2.txt").write_text("\n".join(l2) + "\n")
    (tmp_path / "clustering.txt").write_text("\n".join(indicators) + "\n")
    return names


def _oracle_pipeline(oracle, tmp_path, min_hits=0):
    grouping = read_reference(open(tmp_path / "clustering.txt"))
    aln = Alignment(len(grouping.group_indicators))
    aln.read("intersection", [open(tmp_path / "toy_1.txt"), open(tmp_path / "toy_2.txt")])
    aln.collapse()
    counts = oracle.group_counts(aln.ec_tptr, aln.ec_targets, grouping.group_indicators, grouping.get_n_groups())
    L, mask = oracle.fill_ll_mat(counts, aln.ec_counts, grouping.get_sizes(), min_hits=min_hits)
    logc = oracle.fill_ec_counts(aln.ec_counts)
    return grouping, aln, L, mask, logc


def _parse(path):
    head, rows = {}, []
    for ln in open(path):
        ln = ln.rstrip("\n")
        if ln.startswith("#"):
            k, _, v = ln.partition("\t")
            head[k] = v
        else:
            p = ln.split("\t")
            rows.append((p[0], [float(x) for x in p[1:]]))
    return head, rows


def test_toy_cli_plain(tmp_path, oracle):
    _toy(tmp_path)
    prefix = str(tmp_path / "toy")
    rc = main(["--themisto-1", str(tmp_path / "toy_1.txt"), "--themisto-2", str(tmp_path / "toy_2.txt"),
               "-i", str(tmp_path / "clustering.txt"), "-t", "1", "-o", prefix])
    assert rc == 0
    grouping, aln, L, mask, logc = _oracle_pipeline(oracle, tmp_path)
    ref = oracle.rcg_optl_dense(L, logc, np.ones(L.shape[0]))
    theta = oracle.mixture_components(ref["gamma"], logc)
    head, rows = _parse(prefix + "_abundances.txt")
    assert head["#num_reads:"] == str(aln.n_reads()) and head["#num_aligned:"] == str(int(aln.ec_counts.sum()))
    assert [r[0] for r in rows] == grouping.get_names()
    got = np.array([r[1][0] for r in rows])
    np.testing.assert_allclose(got, theta, rtol=6e-6)          # the file holds 6 significant digits
    assert got[np.argmax(theta)] > 0.3


def test_toy_cli_bootstrap_and_min_hits(tmp_path, oracle):
    _toy(tmp_path)
    prefix = str(tmp_path / "boot")
    rc = main(["--themisto", f"{tmp_path / 'toy_1.txt'},{tmp_path / 'toy_2.txt'}", "-i", str(tmp_path / "clustering.txt"),
               "-o", prefix, "--iters", "3", "--seed", "42", "--min-hits", "1"])
    assert rc == 0
    grouping, aln, L, mask, logc = _oracle_pipeline(oracle, tmp_path, min_hits=1)
    head, rows = _parse(prefix + "_abundances.txt")
    assert head["#bootstrap_iters:"] == "3"
    est = [n for n, m in zip(grouping.get_names(), mask) if m]
    assert [r[0] for r in rows][:len(est)] == est
    alpha = np.ones(L.shape[0])
    base = oracle.mixture_components(oracle.rcg_optl_dense(L, logc, alpha)["gamma"], logc)
    w = aln.ec_counts.astype(np.uint32)
    counts = oracle.bootstrap_counts(w, 42, int(w.sum()), 3)
    cols = [base]
    for b in range(3):
        with np.errstate(divide="ignore"):
            lc = np.log(counts[b].astype(float))
        cols.append(oracle.mixture_components(oracle.rcg_optl_dense(L, lc, alpha)["gamma"], lc))
    got = np.array([r[1] for r in rows[:len(est)]])
    np.testing.assert_allclose(got, np.array(cols).T, rtol=7e-6, atol=1e-12)


def test_cli_error_paths(tmp_path, capsys):
    _toy(tmp_path)
    bad = tmp_path / "bad.txt"
    bad.write_text("0 1 x\n")
    rc = main(["--themisto-1", str(bad), "-i", str(tmp_path / "clustering.txt")])
    assert rc == 1 and "Reading the pseudoalignments failed" in capsys.readouterr().err
    rc = main(["--themisto-1", str(tmp_path / "toy_1.txt"), "-i", str(tmp_path / "clustering.txt"),
               "--alphas", "1,2"])
    assert rc == 1


def test_cli_likelihood_roundtrip_probs_and_rate(tmp_path, oracle, capsys):
    """--write-likelihood -> --read-likelihood round trip (include/Likelihood.hpp:224-273), --write-probs
    (src/Sample.cpp:63-85) and --run-rate (src/Sample.cpp:99-152, src/mSWEEP.cpp:529-545)."""
    _toy(tmp_path, n_reads=600)
    pre = str(tmp_path / "rt")
    base = ["--themisto-1", str(tmp_path / "toy_1.txt"), "--themisto-2", str(tmp_path / "toy_2.txt"),
            "-i", str(tmp_path / "clustering.txt"), "-o", pre]
    assert main(base + ["--write-likelihood", "--no-fit-model"]) == 0
    assert not os.path.exists(pre + "_abundances.txt")
    grouping, aln, L, mask, logc = _oracle_pipeline(oracle, tmp_path)
    rows = [ln.split("\t") for ln in open(pre + "_likelihoods.tsv").read().splitlines()]
    assert len(rows) == aln.n_ecs() and [int(r[0]) for r in rows] == aln.ec_counts.tolist()
    np.testing.assert_allclose(np.array([[float(x) for x in r[1:]] for r in rows]).T, L, rtol=1e-5)
    # estimate from the written (6-digit) likelihood through the dense path
    pre2 = str(tmp_path / "rt2")
    assert main(["--read-likelihood", pre + "_likelihoods.tsv", "-i", str(tmp_path / "clustering.txt"), "-o", pre2,
                 "--write-probs", "--run-rate"]) == 0
    assert main(base) == 0
    _, rows_direct = _parse(pre + "_abundances.txt")
    head2, rows_rt = _parse(pre2 + "_abundances.txt")
    assert head2["#c_id"] == "mean_theta\tRATE\tKLD"
    np.testing.assert_allclose([r[1][0] for r in rows_rt], [r[1][0] for r in rows_direct], rtol=2e-4)
    rate = np.array([r[1][1] for r in rows_rt])
    assert rate.sum() == pytest.approx(1.0, rel=1e-4) and np.all(rate >= 0)
    probs = [ln.split("\t") for ln in open(pre2 + "_probs.tsv").read().splitlines() if ln]
    assert probs[0] == ["ec_id"] + grouping.get_names() and len(probs) == aln.n_ecs() + 1
    P = np.array([[float(x) for x in r[1:]] for r in probs[1:]])
    np.testing.assert_allclose(P.sum(1), 1.0, rtol=1e-4)


@pytest.fixture(scope="module")
def mini_binary(tmp_path_factory):
    """msweep_amd/cpp/msweep_mini.cpp: the estimation path of mSWEEP's main() as a native program."""
    import subprocess
    from conftest import ROOT
    out = str(tmp_path_factory.mktemp("mini") / "msweep_mini")
    lib = os.path.join(ROOT, "msweep_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", out, os.path.join(lib, "cpp", "msweep_mini.cpp"),
                           "-L" + lib, "-lmsweep_core", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"])
    return out


@pytest.mark.parametrize("extra", [[], ["--min-hits", "400"], ["--iters", "3", "--seed", "42"],
                                   ["--algorithm", "emgpu", "--themisto-mode", "union"],
                                   ["--write-probs"], ["--write-probs", "--min-hits", "400", "--iters", "2", "--seed", "5"],
                                   ["--run-rate"], ["--run-rate", "--min-hits", "400"],
                                   ["--write-likelihood", "--write-probs"], ["--write-likelihood", "--no-fit-model"]])
def test_native_driver_matches_python_cli(tmp_path, mini_binary, extra):
    """Same flags, same files, byte for byte (bootstrap included: the device stream is seeded): abundances, and --
    with their flags -- the probabilities, the likelihood file and the RATE / KLD table."""
    import subprocess
    _toy(tmp_path)
    common = ["--themisto-1", str(tmp_path / "toy_1.txt"), "--themisto-2", str(tmp_path / "toy_2.txt"),
              "-i", str(tmp_path / "clustering.txt")] + extra
    assert main(common + ["-o", str(tmp_path / "py")]) == 0
    p = subprocess.run([mini_binary] + common + ["-o", str(tmp_path / "cc")], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    outputs = [] if "--no-fit-model" in extra else ["abundances.txt"]
    outputs += ["probs.tsv"] if "--write-probs" in extra and "--no-fit-model" not in extra else []
    outputs += ["likelihoods.tsv"] if "--write-likelihood" in extra else []
    for name in outputs:
        got, want = (tmp_path / ("cc_" + name)).read_text(), (tmp_path / ("py_" + name)).read_text()
        assert len(want) > 20 and got == want, name
    if "--no-fit-model" in extra:
        assert not (tmp_path / "cc_abundances.txt").exists() and not (tmp_path / "py_abundances.txt").exists()


def test_native_driver_reads_a_likelihood_file(tmp_path, mini_binary):
    """--read-likelihood in the native program: the file another run wrote, no pseudoalignments given; the same
    abundances and probabilities as the Python mirror on the same file, and -- to the 6 digits the file keeps -- the
    estimate of the run that wrote it.  --print-probs goes to stdout."""
    import subprocess
    _toy(tmp_path)
    common = ["--themisto-1", str(tmp_path / "toy_1.txt"), "--themisto-2", str(tmp_path / "toy_2.txt"),
              "-i", str(tmp_path / "clustering.txt")]
    p = subprocess.run([mini_binary] + common + ["-o", str(tmp_path / "w"), "--write-likelihood"],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    rd = ["-i", str(tmp_path / "clustering.txt"), "--read-likelihood", str(tmp_path / "w_likelihoods.tsv"), "--write-probs"]
    assert main(rd + ["-o", str(tmp_path / "py")]) == 0
    p = subprocess.run([mini_binary] + rd + ["-o", str(tmp_path / "cc")], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    for name in ("abundances.txt", "probs.tsv"):
        assert (tmp_path / ("cc_" + name)).read_text() == (tmp_path / ("py_" + name)).read_text(), name
    h0, r0 = _parse(str(tmp_path / "w_abundances.txt"))
    h1, r1 = _parse(str(tmp_path / "cc_abundances.txt"))
    assert h1["#num_aligned:"] == h0["#num_aligned:"]
    assert [n for n, _ in r1] == [n for n, _ in r0]
    np.testing.assert_allclose([v[0] for _, v in r1], [v[0] for _, v in r0], rtol=2e-4, atol=1e-7)
    p = subprocess.run([mini_binary] + rd[:-1] + ["--print-probs"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert p.stdout.startswith((tmp_path / "cc_probs.tsv").read_text())      # then the abundances, also on stdout
    p = subprocess.run([mini_binary, "-i", str(tmp_path / "clustering.txt"), "--read-likelihood", str(tmp_path / "toy_1.txt")],
                       capture_output=True, text=True, timeout=60)
    assert p.returncode == 1 and "Could not read from the likelihoods file." in p.stderr


def test_native_driver_errors(tmp_path, mini_binary):
    import subprocess
    _toy(tmp_path)
    (tmp_path / "bad.txt").write_text("0 1 2\n1 x\n")
    p = subprocess.run([mini_binary, "--themisto", str(tmp_path / "bad.txt"), "-i", str(tmp_path / "clustering.txt")],
                       capture_output=True, text=True, timeout=60)
    assert p.returncode == 1 and "Reading the pseudoalignments failed" in p.stderr and "line 2" in p.stderr
    p = subprocess.run([mini_binary, "--themisto-1", str(tmp_path / "toy_1.txt"), "-i", str(tmp_path / "clustering.txt"),
                        "--alphas", "1,2"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 1 and "--alphas must have the same number of values" in p.stderr


def test_rate_kld_against_the_restated_reference(gpu_core, oracle):
    """--run-rate (experimental in the reference): the product's host routine (msweep_amd/__main__.py
    dirichlet_kld_rate, fed with the column sums the solve already has) against the oracle's restatement of
    Sample::dirichlet_kld + get_rates on the G x E matrix (src/Sample.cpp:99-152: repeated additions per read,
    the 1e-16 clamp, the log-sum-exp with its shift of max(0, max log KLD))."""
    from msweep_amd.__main__ import dirichlet_kld_rate
    from msweep_amd import synth
    from msweep_amd.likelihood import from_grouped_counts
    p = synth.make_csr_problem(4000, 25, seed=71, max_other=5, dirichlet=0.3)
    lik = from_grouped_counts(gpu_core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
    res = gpu_core.solve(lik.log_counts(), np.ones(25))
    gamma = gpu_core.gamma()
    total = float(p["ec_counts"].sum())
    kld, rate = dirichlet_kld_rate(res["theta"] * total)
    al_o, kld_o, rate_o = oracle.dirichlet_kld_rate(gamma, lik.log_counts())
    np.testing.assert_allclose(res["theta"] * total, al_o, rtol=1e-9)
    np.testing.assert_allclose(kld, kld_o, rtol=1e-6)       # lgamma differences of ~1e4-sized arguments
    np.testing.assert_allclose(rate, rate_o, rtol=1e-6)
    assert rate.sum() == pytest.approx(1.0, rel=1e-12)
    # a group nobody supports: KLD clamps at 1e-16 (src/Sample.cpp:126)
    kld2, rate2 = dirichlet_kld_rate(np.array([5000.0, 1e-300, 3000.0]))
    assert kld2[1] == pytest.approx(1e-16) and rate2.sum() == pytest.approx(1.0)


def test_cli_probs_with_bootstrap_are_the_original_estimate(tmp_path, capsys):
    """--iters N --write-probs: the probabilities written are those of the un-resampled estimate (the
    reference writes them before its replicate loop, src/mSWEEP.cpp:437-493 vs :496)."""
    _toy(tmp_path, n_reads=500)
    base = ["--themisto-1", str(tmp_path / "toy_1.txt"), "--themisto-2", str(tmp_path / "toy_2.txt"),
            "-i", str(tmp_path / "clustering.txt")]
    assert main(base + ["-o", str(tmp_path / "plain"), "--write-probs"]) == 0
    assert main(base + ["-o", str(tmp_path / "boot"), "--write-probs", "--iters", "3", "--seed", "7"]) == 0
    assert open(str(tmp_path / "plain_probs.tsv")).read() == open(str(tmp_path / "boot_probs.tsv")).read()
    head, _ = _parse(str(tmp_path / "boot_abundances.txt"))
    assert head["#bootstrap_iters:"] == "3"
