"""GPU: bootstrap resampling (bit-exact with libstdc++'s stream), the bootstrap driver, and the EM
variant, against the oracle and the committed golden vectors."""
import numpy as np
import pytest

from conftest import dense_from_csr, load_golden, lutidx_of
from msweep_amd import synth
from msweep_amd.core import ALGO_EM, ALGO_RCG
from msweep_amd.likelihood import from_dense, from_grouped_counts, precalc_lls
from test_gpu_rcg import assert_theta

pytestmark = pytest.mark.gpu


def test_resample_counts_match_libstdcxx_golden(gpu_core):
    g = load_golden("bootstrap_golden.json")
    for c in g["cases"]:
        exp = np.array(c["counts"], np.uint32)
        got = gpu_core.resample_counts(c["weights"], c["seed"], c["draws"], 0, len(exp))
        np.testing.assert_array_equal(got, exp)
        # any slice of the ONE sequential stream, asked for on its own (what rank r of N does)
        if len(exp) > 1:
            got1 = gpu_core.resample_counts(c["weights"], c["seed"], c["draws"], 1, len(exp))
            np.testing.assert_array_equal(got1, exp[1:])


def test_resample_counts_match_oracle_large(gpu_core, oracle):
    rng = np.random.default_rng(21)
    w = rng.integers(0, 500, 200_000).astype(np.uint32)
    w[rng.random(len(w)) < 0.3] = 0
    exp = oracle.bootstrap_counts(w, -99, 300_000, 3)
    got = gpu_core.resample_counts(w, -99, 300_000, 0, 3)
    np.testing.assert_array_equal(got, exp)
    # backwards seek re-seeds the stream
    np.testing.assert_array_equal(gpu_core.resample_counts(w, -99, 300_000, 1, 2), exp[1:2])
    assert np.all(got.sum(1) == 300_000)


def test_far_seek_jumps_ahead_like_stepping(gpu_core, monkeypatch):
    """Rank r of a multi-GPU bootstrap starts r * (B / P) * draws words into the stream: beyond 2^27
    words the seek is a GF(2) jump on the host; same counts as stepping there on the device."""
    rng = np.random.default_rng(3)
    w = rng.integers(1, 50, 5000).astype(np.uint32)
    draws, first = 100_000, 1500            # 1.5e8 words in
    jumped = gpu_core.resample_counts(w, 11, draws, first, first + 2)
    jumped_on = gpu_core.resample_counts(w, 11, draws, first + 2, first + 3)   # short forward seek after a jump
    monkeypatch.setenv("MSWEEP_MT_NOJUMP", "1")
    stepped = gpu_core.resample_counts(w, 11, draws, first, first + 3)
    np.testing.assert_array_equal(jumped, stepped[:2])
    np.testing.assert_array_equal(jumped_on, stepped[2:])
    assert np.all(stepped.sum(1) == draws)


def test_bootstrap_driver_matches_replicate_by_replicate_oracle(gpu_core, oracle):
    p = synth.make_csr_problem(30000, 80, seed=14, max_other=6)
    G = 80
    alpha0 = np.ones(G)
    lik = from_grouped_counts(gpu_core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
    w = p["ec_counts"].astype(np.uint32)
    draws = int(w.sum())
    theta, iters = gpu_core.bootstrap(w, 42, draws, 0, 4, alpha0)
    counts = oracle.bootstrap_counts(w, 42, draws, 4)
    lut = precalc_lls(p["group_sizes"])
    for b in range(4):
        with np.errstate(divide="ignore"):
            logc = np.log(counts[b].astype(float))       # -inf for ECs drawn zero times (:70)
        ref = oracle.rcg_optl_csr(p["rowptr"], p["grp"], lutidx_of(p, lut), lut, np.log(0.01), G, logc, alpha0)
        assert iters[b] == ref["iters"]
        assert_theta(theta[b], ref["theta"])
        assert theta[b].sum() == pytest.approx(1.0, abs=1e-12)   # normalised by the resampled total
    # a slice of the replicate stream equals the same rows of the full run (N-GPU invariance)
    # (identical resampled counts; the solves differ only by the order of the LDS atomic adds of
    # the column sums, i.e. by rounding)
    theta2, _ = gpu_core.bootstrap(w, 42, draws, 2, 4, alpha0)
    for b in range(2):
        assert_theta(theta2[b], theta[2 + b])
    # --bootstrap-count smaller than the number of reads
    theta3, _ = gpu_core.bootstrap(w, 7, 5000, 0, 1, alpha0)
    c3 = oracle.bootstrap_counts(w, 7, 5000, 1)[0]
    with np.errstate(divide="ignore"):
        r3 = oracle.rcg_optl_csr(p["rowptr"], p["grp"], lutidx_of(p, lut), lut, np.log(0.01), G,
                                 np.log(c3.astype(float)), alpha0)
    assert_theta(theta3[0], r3["theta"])


@pytest.mark.parametrize("idx", [1, 2, 3, 6])
def test_em_dense_matches_oracle(gpu_core, oracle, idx):
    c = load_golden("rcg_golden.json")["cases"][idx]
    L = np.array(c["logl"])
    logc = np.array([-np.inf if x is None else x for x in c["logc"]])
    alpha0 = np.array(c["alpha0"])
    from_dense(gpu_core, L, logc)
    res = gpu_core.solve(logc, alpha0, tol=1e-9, max_iters=5000, algo=ALGO_EM)
    ref = oracle.em_dense(L, logc, alpha0, tol=1e-9, max_iters=5000, want_gamma=True)
    assert abs(res["iters"] - ref["iters"]) <= 2
    np.testing.assert_allclose(res["theta"], ref["theta"], rtol=1e-6, atol=1e-9)
    assert res["bound"] == pytest.approx(ref["bound"], rel=1e-10)
    np.testing.assert_allclose(np.exp(gpu_core.gamma()), np.exp(ref["gamma"]), atol=1e-6)


def test_em_csr_matches_oracle_and_rcg_region(gpu_core, oracle):
    p = synth.make_csr_problem(20000, 60, seed=15, max_other=6)
    G = 60
    alpha0 = np.ones(G)
    lik = from_grouped_counts(gpu_core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
    em = gpu_core.solve(lik.log_counts(), alpha0, tol=1e-8, max_iters=20000, algo=ALGO_EM)
    lut = precalc_lls(p["group_sizes"])
    ref = oracle.em_dense(dense_from_csr(p, lut), lik.log_counts(), alpha0, tol=1e-8, max_iters=20000)
    assert abs(em["iters"] - ref["iters"]) <= 2
    np.testing.assert_allclose(em["theta"], ref["theta"], rtol=1e-6, atol=1e-9)
    rcg = gpu_core.solve(lik.log_counts(), alpha0, algo=ALGO_RCG)
    np.testing.assert_allclose(em["theta"], rcg["theta"], atol=5e-3)   # different objectives, same region
    # --emprecision float is accepted (served in fp64)
    em32 = gpu_core.solve(lik.log_counts(), alpha0, tol=1e-8, max_iters=20000, algo=ALGO_EM, prec=1)
    np.testing.assert_allclose(em32["theta"], em["theta"], rtol=1e-9, atol=1e-15)


def test_large_slot_area_em_and_bootstrap(gpu_core, oracle):
    """A dense matrix with continuous listed values is re-expressed with one table slot per listed
    cell: far more than 16384 slots, so the per-slot tables are rebuilt by k_tables (many workgroups)
    instead of inside k_step / k_fin / k_em_fin.  EM, RCG and the bootstrap driver on that form."""
    E, G = 12000, 300
    p = synth.make_dense_problem(E, G, seed=23)
    rng = np.random.default_rng(5)
    w = rng.integers(1, 40, E).astype(np.uint32)
    logc = np.log(w.astype(float))
    alpha0 = np.ones(G)
    from_dense(gpu_core, p["logl"], logc)
    assert 16384 < gpu_core.shape()[2] < G * E // 4
    em = gpu_core.solve(logc, alpha0, tol=1e-4, max_iters=5000, algo=ALGO_EM)
    ref = oracle.em_dense(p["logl"], logc, alpha0, tol=1e-4, max_iters=5000)
    assert abs(em["iters"] - ref["iters"]) <= max(2, ref["iters"] // 200)   # a flat tail: the stop is soft
    np.testing.assert_allclose(em["theta"], ref["theta"], rtol=1e-5, atol=1e-8)
    rcg = gpu_core.solve(logc, alpha0)
    s = oracle.rcg_optl_dense_structured(p["logl"], logc, alpha0)
    assert abs(rcg["iters"] - s["iters"]) <= 2
    assert_theta(rcg["theta"], s["theta"])
    # bootstrap replicates: each equals a solve on the oracle's resampled counts
    draws = int(w.sum())
    theta, iters = gpu_core.bootstrap(w, 7, draws, 0, 2, alpha0)
    counts = oracle.bootstrap_counts(w, 7, draws, 2)
    for b in range(2):
        with np.errstate(divide="ignore"):
            lc = np.log(counts[b].astype(float))
        rb = oracle.rcg_optl_dense_structured(p["logl"], lc, alpha0)
        assert abs(iters[b] - rb["iters"]) <= 2
        assert_theta(theta[b], rb["theta"])


@pytest.mark.parametrize("streams", ["1", "3", "16"])
def test_replicates_in_flight_do_not_change_results(gpu_core, monkeypatch, streams):
    """The driver runs several replicates at a time (solver states sharing the resident likelihood,
    one stream each, contiguous blocks of the one random stream): same counts, same solves."""
    p = synth.make_csr_problem(20000, 50, seed=41, max_other=6)
    G = 50
    lik = from_grouped_counts(gpu_core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
    w = p["ec_counts"].astype(np.uint32)
    draws = int(w.sum())
    monkeypatch.setenv("MSWEEP_BOOTSTRAP_STREAMS", "1")
    ref, it_ref = gpu_core.bootstrap(w, 5, draws, 1, 8, np.ones(G))
    monkeypatch.setenv("MSWEEP_BOOTSTRAP_STREAMS", streams)
    got, it = gpu_core.bootstrap(w, 5, draws, 1, 8, np.ones(G))
    assert it.tolist() == it_ref.tolist()
    for b in range(7):
        assert_theta(got[b], ref[b])
    # the handle's own state still serves a plain solve afterwards
    res = gpu_core.solve(lik.log_counts(), np.ones(G))
    assert res["theta"].sum() == pytest.approx(1.0, abs=1e-12)


def test_replicates_in_flight_across_likelihood_changes(gpu_core, monkeypatch):
    """The extra solver states outlive a likelihood: a larger CSR likelihood, then a dense one (kept on
    the dense sweeps), on the same handle -- each time the same replicates as with one in flight."""
    def both(w, draws, G):
        monkeypatch.setenv("MSWEEP_BOOTSTRAP_STREAMS", "1")
        ref, it_ref = gpu_core.bootstrap(w, 9, draws, 0, 5, np.ones(G))
        monkeypatch.setenv("MSWEEP_BOOTSTRAP_STREAMS", "3")
        got, it = gpu_core.bootstrap(w, 9, draws, 0, 5, np.ones(G))
        assert it.tolist() == it_ref.tolist()
        for b in range(5):
            assert_theta(got[b], ref[b])

    for R, G, seed in [(5000, 20, 51), (60000, 120, 52)]:
        p = synth.make_csr_problem(R, G, seed=seed, max_other=5)
        from_grouped_counts(gpu_core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
        w = p["ec_counts"].astype(np.uint32)
        both(w, int(w.sum()), G)
    monkeypatch.setenv("MSWEEP_DENSE_COMPRESS", "0")
    d = synth.make_dense_problem(3000, 70, seed=53)
    from_dense(gpu_core, d["logl"], d["logc"])
    w = np.random.default_rng(2).integers(1, 9, 3000).astype(np.uint32)
    both(w, int(w.sum()), 70)



def test_emprecision_float_is_served_in_double(gpu_core):
    """--emprecision float (src/mSWEEP.cpp:129,202): accepted, computed in fp64 -- the same bits as double
    (msweep_amd/csrc/host_em.inc says why no fp32 variant exists)."""
    from msweep_amd.core import PREC_DOUBLE, PREC_FLOAT
    p = synth.make_csr_problem(20000, 60, seed=33, max_other=5)
    lik = from_grouped_counts(gpu_core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
    d = gpu_core.solve(lik.log_counts(), np.ones(60), algo=ALGO_EM, prec=PREC_DOUBLE, max_iters=300)
    f = gpu_core.solve(lik.log_counts(), np.ones(60), algo=ALGO_EM, prec=PREC_FLOAT, max_iters=300)
    assert d["iters"] == f["iters"] and d["bound"] == f["bound"]
    np.testing.assert_array_equal(d["theta"], f["theta"])
