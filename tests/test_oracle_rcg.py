"""CPU: the optimiser oracle (rcgpar v1.2.1 restated -- parity unpinned by the reference tree,
see oracle/msweep_oracle.h) against the independent numpy twin's golden trajectories, the
structured formulation against the dense-state one, and mathematical properties."""
import numpy as np
import pytest

from conftest import dense_from_csr, load_golden, lutidx_of
from msweep_amd import synth
from msweep_amd.likelihood import precalc_lls


def _case_arrays(c):
    L = np.array(c["logl"])
    logc = np.array([-np.inf if x is None else x for x in c["logc"]])
    return L, logc, np.array(c["alpha0"])


@pytest.mark.parametrize("idx", range(7))
def test_dense_oracle_matches_numpy_twin(oracle, idx):
    c = load_golden("rcg_golden.json")["cases"][idx]
    L, logc, alpha0 = _case_arrays(c)
    r = oracle.rcg_optl_dense(L, logc, alpha0, c["tol"], c["max_iters"], trace=25)
    e = c["expect"]
    assert r["iters"] == e["iters"], c["name"]
    k = min(len(e["trace"]["bound"]), r["iters"])
    np.testing.assert_allclose(r["trace"]["bound"][:k], e["trace"]["bound"][:k], rtol=1e-12)
    nn, en = r["trace"]["newnorm"][:k], np.array(e["trace"]["newnorm"][:k])
    np.testing.assert_allclose(nn[:10], en[:10], rtol=1e-9, atol=1e-300)
    assert r["trace"]["didreset"][:k].tolist() == e["trace"]["didreset"][:k]
    np.testing.assert_allclose(r["trace"]["theta"][:min(k, 10)], np.array(e["trace"]["theta"])[:min(k, 10)], rtol=1e-9)
    theta = oracle.mixture_components(r["gamma"], logc)
    np.testing.assert_allclose(theta, e["theta"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(np.exp(r["gamma"]).sum(0), 1.0, rtol=1e-12)


@pytest.mark.parametrize("idx", range(7))
def test_structured_dense_matches_dense_state(oracle, idx):
    c = load_golden("rcg_golden.json")["cases"][idx]
    L, logc, alpha0 = _case_arrays(c)
    d = oracle.rcg_optl_dense(L, logc, alpha0, trace=25)
    s = oracle.rcg_optl_dense_structured(L, logc, alpha0, trace=25, want_gamma=True)
    assert s["iters"] == d["iters"]
    k = min(20, d["iters"])
    np.testing.assert_allclose(s["trace"]["bound"][:k], d["trace"]["bound"][:k], rtol=1e-11)
    assert s["trace"]["didreset"][:k].tolist() == d["trace"]["didreset"][:k].tolist()
    np.testing.assert_allclose(s["trace"]["theta"][:10], d["trace"]["theta"][:10], rtol=1e-9)
    np.testing.assert_allclose(s["theta"], oracle.mixture_components(d["gamma"], logc), rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(np.exp(s["gamma"]), np.exp(d["gamma"]), atol=1e-7)


def test_csr_structured_matches_dense_state(oracle):
    p = synth.make_csr_problem(4000, 50, seed=5, max_other=6)
    lut = precalc_lls(p["group_sizes"])
    L = dense_from_csr(p, lut)
    logc = np.log(p["ec_counts"].astype(float))
    alpha0 = np.ones(50)
    d = oracle.rcg_optl_dense(L, logc, alpha0, trace=30)
    s = oracle.rcg_optl_csr(p["rowptr"], p["grp"], lutidx_of(p, lut), lut, np.log(0.01), 50, logc, alpha0,
                            trace=30, want_gamma=True)
    assert s["iters"] == d["iters"]
    k = min(20, d["iters"])
    np.testing.assert_allclose(s["trace"]["bound"][:k], d["trace"]["bound"][:k], rtol=1e-11)
    np.testing.assert_allclose(s["trace"]["newnorm"][:10], d["trace"]["newnorm"][:10], rtol=1e-8)
    assert s["trace"]["didreset"][:k].tolist() == d["trace"]["didreset"][:k].tolist()
    th = oracle.mixture_components(d["gamma"], logc)
    np.testing.assert_allclose(s["theta"], th, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(np.exp(s["gamma"]), np.exp(d["gamma"]), atol=1e-7)


def test_known_answers(oracle):
    # G = 1 -> theta = 1
    L = np.log(np.random.default_rng(0).uniform(0.1, 0.9, (1, 7)))
    r = oracle.rcg_optl_dense(L, np.zeros(7), np.ones(1))
    assert oracle.mixture_components(r["gamma"], np.zeros(7))[0] == pytest.approx(1.0, rel=1e-14)
    # (nearly) disjoint support: responsibilities are 0/1, theta_g = c_g / sum c
    G, E = 3, 9
    L = np.full((G, E), -800.0)
    owner = np.arange(E) % G
    L[owner, np.arange(E)] = -1.0
    cnt = np.arange(1, E + 1, dtype=float)
    r = oracle.rcg_optl_dense(L, np.log(cnt), np.ones(G))
    th = oracle.mixture_components(r["gamma"], np.log(cnt))
    exp = np.array([cnt[owner == g].sum() for g in range(G)]) / cnt.sum()
    np.testing.assert_allclose(th, exp, rtol=1e-12)


def test_invariances(oracle):
    rng = np.random.default_rng(4)
    c = load_golden("rcg_golden.json")["cases"][2]
    L, logc, alpha0 = _case_arrays(c)
    base = oracle.rcg_optl_dense_structured(L, logc, alpha0)
    # permutation of groups and of ECs
    pg, pe = rng.permutation(L.shape[0]), rng.permutation(L.shape[1])
    r = oracle.rcg_optl_dense_structured(L[pg][:, pe], logc[pe], alpha0[pg])
    np.testing.assert_allclose(r["theta"], base["theta"][pg], rtol=1e-6, atol=1e-10)
    # bound never decreases over accepted steps; sum N = sum alpha + sum c
    d = oracle.rcg_optl_dense(L, logc, alpha0, trace=100)
    b = d["trace"]["bound"][:d["iters"]]
    assert np.all(np.diff(b) > -1e-9)
    th = oracle.mixture_components(d["gamma"], logc)
    assert th.sum() == pytest.approx(1.0, abs=1e-12)
    # fixed point: gamma_gj proportional to exp(L_gj + digamma(N_g)) at convergence
    N = th * np.exp(logc).sum() + alpha0
    dg = np.array([oracle.digamma(x) for x in N])
    z = L + dg[:, None]
    z -= np.log(np.exp(z).sum(0))[None, :]
    np.testing.assert_allclose(np.exp(z), np.exp(d["gamma"]), atol=5e-4)


def test_em_oracle_converges_to_same_optimum_region(oracle):
    """EM (MAP with alpha0 = 1 -> ML) and RCG-VB optimise different objectives; with many reads
    they agree to within the prior's influence."""
    c = load_golden("rcg_golden.json")["cases"][6]
    L, logc, alpha0 = _case_arrays(c)
    em = oracle.em_dense(L, logc, alpha0, tol=1e-10, max_iters=20000)
    assert em["theta"].sum() == pytest.approx(1.0, abs=1e-12)
    np.testing.assert_allclose(em["theta"], c["expect"]["theta"], atol=2e-3)


def test_check_every_knob_same_trajectory_stop_on_the_grid(oracle):
    """orc_rcg_opts.check_every (every published iteration count is a multiple of 5): the
    trajectory is the default's, only the stop moves to the next iteration on the grid -- in all three loops."""
    p = synth.make_csr_problem(4000, 40, seed=9, max_other=5)
    lut = precalc_lls(p["group_sizes"])
    L = dense_from_csr(p, lut)
    logc = np.log(p["ec_counts"].astype(float))
    alpha0 = np.ones(40)
    runs = [lambda **kw: oracle.rcg_optl_dense(L, logc, alpha0, trace=200, **kw),
            lambda **kw: oracle.rcg_optl_dense_structured(L, logc, alpha0, trace=200, **kw),
            lambda **kw: oracle.rcg_optl_csr(p["rowptr"], p["grp"], lutidx_of(p, lut), lut, np.log(0.01), 40, logc,
                                             alpha0, trace=200, **kw)]
    for run in runs:
        d, g = run(), run(check_every=5)
        # (not necessarily the NEXT grid point: past convergence the gains hover around zero, and a negative one
        # is a reset iteration, which never stops)
        assert g["iters"] % 5 == 0 and d["iters"] <= g["iters"]
        np.testing.assert_array_equal(g["trace"]["bound"][:d["iters"]], d["trace"]["bound"][:d["iters"]])
        # every gain between the default's stop and the grid's was below tol or a reset: nothing else delays it
        assert run(check_every=1)["iters"] == d["iters"]


def test_em_variant_knobs(oracle):
    """orc_em_opts: ML drops the prior's pseudo-counts (equal to MAP at alpha0 = 1); the theta-change stop rule
    ends at a fixed point of the same map."""
    p = synth.make_csr_problem(3000, 30, seed=10, max_other=5)
    lut = precalc_lls(p["group_sizes"])
    L = dense_from_csr(p, lut)
    logc = np.log(p["ec_counts"].astype(float))
    one, prior = np.ones(30), np.full(30, 3.0)
    m1, l1 = oracle.em_dense(L, logc, one, tol=1e-9), oracle.em_dense(L, logc, one, tol=1e-9, prior="ml")
    assert m1["iters"] == l1["iters"]
    np.testing.assert_allclose(m1["theta"], l1["theta"], rtol=1e-12, atol=1e-15)   # (x + 1) - 1 drops what is below 1e-16
    m3, l3 = oracle.em_dense(L, logc, prior, tol=1e-9), oracle.em_dense(L, logc, prior, tol=1e-9, prior="ml")
    np.testing.assert_allclose(l3["theta"], l1["theta"], rtol=1e-12)          # ML ignores alpha0
    assert np.max(np.abs(m3["theta"] - l3["theta"])) > 1e-4                  # MAP does not
    t = oracle.em_dense(L, logc, one, tol=1e-10, stop="theta")
    np.testing.assert_allclose(t["theta"], m1["theta"], atol=1e-6)
    g5 = oracle.em_dense(L, logc, one, tol=1e-9, check_every=5)
    assert g5["iters"] % 5 == 0 and m1["iters"] <= g5["iters"] < m1["iters"] + 5


@pytest.mark.parametrize("name,max_apart", [("seed1_case517", 5e-8), ("seed7_case292", 1e-9), ("seed99_case247", 1e-9)])
def test_extended_oracles_agree_where_the_fp64_ones_drift(oracle, name, max_apart):
    """orc_rcg_opts::extended (round 5): on the inputs of tools/fuzz_parity.py that amplify rounding (a Fletcher-Reeves
    factor ~ 100; tests/golden/fuzz/) the structured and the dense-state restatement carried in x87 extended precision
    agree with each other to 1e-10 .. 3e-8 over twelve iterations, while the fp64 structured oracle is 2e-7 .. 4e-5 away
    from both: the extended pair is the judge of the HIP path there (tests/test_gpu_fuzz.py)."""
    import os
    from conftest import ROOT
    d = np.load(os.path.join(ROOT, "tests", "golden", "fuzz", name + ".npz"))
    G, E, n = len(d["alpha0"]), len(d["rowptr"]) - 1, 12
    lutidx = (d["grp"] * d["lut"].shape[1] + d["cnt"]).astype(np.uint32)
    L = np.full((G, E), np.log(0.01))
    L[d["grp"], np.repeat(np.arange(E), np.diff(d["rowptr"].astype(np.int64)))] = d["lut"][d["grp"], d["cnt"]]
    run_s = lambda ext: oracle.rcg_optl_csr(d["rowptr"], d["grp"], lutidx, d["lut"], np.log(0.01), G, d["logc"], d["alpha0"],  # noqa: E731
                                            tol=-1.0, max_iters=n, trace=n, extended=ext)["trace"]
    s64, s80 = run_s(False), run_s(True)
    d80 = oracle.rcg_optl_dense(L, d["logc"], d["alpha0"], tol=-1.0, max_iters=n, trace=n, extended=True)["trace"]
    rel = lambda a, b: float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300) * (np.abs(b) >= 1e-12)))  # noqa: E731
    assert s80["didreset"].tolist() == d80["didreset"].tolist() == s64["didreset"].tolist()
    apart, drift = rel(s80["theta"], d80["theta"]), rel(s64["theta"], s80["theta"])
    assert apart < max_apart, apart
    assert drift > 4 * apart and drift > 1e-7, (drift, apart)      # (what the fp64 oracle cost the sweep as a judge)


def test_extended_option_changes_nothing_on_a_tame_problem(oracle):
    """Same trajectory over the first ten iterations to 1e-9 where rounding is not amplified (the recursion still
    multiplies differences ~10 x per 10-20 iterations, SURVEY.md 7.3b: 1e-6 over thirty); identical accept / reject
    decisions and stop."""
    p = synth.make_csr_problem(20000, 150, seed=3, max_other=8)
    lut = precalc_lls(p["group_sizes"])
    logc = np.log(p["ec_counts"].astype(float))
    a0 = np.ones(150)
    r = [oracle.rcg_optl_csr(p["rowptr"], p["grp"], lutidx_of(p, lut), lut, np.log(0.01), 150, logc, a0, trace=30,
                             extended=e) for e in (False, True)]
    assert r[0]["iters"] == r[1]["iters"]
    assert r[0]["trace"]["didreset"].tolist() == r[1]["trace"]["didreset"].tolist()
    L = dense_from_csr(p, lut)
    dd = [oracle.rcg_optl_dense(L, logc, a0, trace=30, extended=e) for e in (False, True)]
    for a, b in ((r[0], r[1]), (dd[0], dd[1]), (dd[1], r[1])):
        np.testing.assert_allclose(a["trace"]["theta"][:10], b["trace"]["theta"][:10], rtol=1e-9, atol=1e-16)
        np.testing.assert_allclose(a["trace"]["theta"][:30], b["trace"]["theta"][:30], rtol=1e-6, atol=1e-12)
    # the two extended formulations stay closer to each other than either fp64 one stays to its extended twin
    rel = lambda a, b: float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-12)))  # noqa: E731
    th = lambda x: x["trace"]["theta"][:30]  # noqa: E731
    assert rel(th(dd[1]), th(r[1])) <= max(rel(th(dd[0]), th(dd[1])), rel(th(r[0]), th(r[1])))
    np.testing.assert_allclose(dd[0]["gamma"], dd[1]["gamma"], rtol=0, atol=1e-5)
