"""CPU: the oracle's likelihood pieces against mpmath golden vectors (tests/golden/lut_golden.json)
and the host LUT mirror against the oracle.  Reference: include/Likelihood.hpp:47-60,92-107,198-207;
src/Sample.cpp:87-97."""
import numpy as np
import pytest
from scipy.special import digamma as sp_digamma

from conftest import load_golden
from msweep_amd.likelihood import bb_params, precalc_lls


def test_bb_params_and_ldbb_against_mpmath(oracle):
    g = load_golden("lut_golden.json")
    for c in g["cases"]:
        a, b = oracle.bb_params([c["n"]], c["q"], c["e"])
        assert a[0] == pytest.approx(c["alpha"], rel=1e-13)
        assert b[0] == pytest.approx(c["beta"], rel=1e-13)
        for ent in c["entries"]:
            v = oracle.ldbb_scaled(ent["k"], c["n"], c["alpha"], c["beta"])
            # lgamma differences of O(1e3) magnitude terms: absolute tolerance
            assert abs(v - ent["ldbb"]) < 5e-11 * max(1.0, c["n"] / 10), (c["n"], ent)


def test_lut_table_against_mpmath(oracle):
    g = load_golden("lut_golden.json")
    for c in g["cases"]:
        lut = oracle.precalc_lls([c["n"]], c["q"], c["e"], c["zi"])
        assert lut.shape == (1, c["n"] + 1)
        assert lut[0, 0] == pytest.approx(c["log_zi"], rel=1e-15)
        for ent in c["entries"]:
            assert abs(lut[0, ent["k"]] - ent["lut"]) < 5e-11 * max(1.0, c["n"] / 10)


def test_lut_layout_multiple_groups(oracle):
    sizes = np.array([3, 1, 10, 3, 7], np.uint64)
    lut = oracle.precalc_lls(sizes)
    assert lut.shape == (5, 11)
    assert np.all(lut[:, 0] == np.log(0.01))          # column 0 = log(zi) for every group (:98)
    assert np.array_equal(lut[0], lut[3])              # equal sizes share a row
    assert np.all(lut[1, 2:] == np.log(0.01))          # k > n_g never indexed; kept at log(zi)
    # "scaled": normalised by B(n+alpha, beta), so the k = n entry is exactly log1p(-zi) (:57-60,102)
    for g, n in enumerate(sizes):
        assert lut[g, int(n)] == pytest.approx(np.log1p(-0.01), abs=1e-13)
        assert np.all(np.diff(lut[g, 1:int(n) + 1]) > 0)


def test_host_lut_mirror_matches_oracle(oracle):
    rng = np.random.default_rng(3)
    sizes = (1 + rng.poisson(9, 200)).astype(np.uint64)
    for q, e, zi in [(0.65, 0.01, 0.01), (0.4, 0.1, 0.05)]:
        a, b = bb_params(sizes, q, e)
        ao, bo = oracle.bb_params(sizes, q, e)
        np.testing.assert_allclose(a, ao, rtol=1e-14)
        np.testing.assert_allclose(b, bo, rtol=1e-14)
        np.testing.assert_allclose(precalc_lls(sizes, q, e, zi), oracle.precalc_lls(sizes, q, e, zi),
                                   rtol=0, atol=2e-12)


def test_digamma_series(oracle):
    g = load_golden("lut_golden.json")
    for d in g["digamma"]:
        # the truncated series itself is only good to ~5e-11 (shift to x >= 7, four terms)
        assert oracle.digamma(d["x"]) == pytest.approx(d["psi"], abs=2e-10 + 1e-13 * abs(d["psi"]))
    xs = np.geomspace(1e-3, 1e9, 200)
    got = np.array([oracle.digamma(x) for x in xs])
    assert np.all(np.abs(got - sp_digamma(xs)) <= 2e-10 + 1e-13 * np.abs(sp_digamma(xs)))


def test_group_counts_mask_and_dense_fill(oracle):
    """fill_ll_mat (:109-186): counts, --min-hits mask, compaction order, LUT lookup."""
    # 3 groups (sizes 2,1,3), 6 targets, 4 ECs
    target_group = np.array([0, 2, 1, 0, 2, 2], np.uint32)
    sizes = np.array([2, 1, 3], np.uint64)
    tptr = np.array([0, 2, 3, 6, 6], np.uint64)
    targets = np.array([0, 3, 1, 1, 4, 5, ], np.uint32)   # EC0: g0 x2 ; EC1: g2 x1 ; EC2: g2 x3 ; EC3: none
    counts = oracle.group_counts(tptr, targets, target_group, 3)
    assert counts.tolist() == [[2, 0, 0, 0], [0, 0, 0, 0], [0, 1, 3, 0]]
    ec_counts = np.array([5, 1, 2, 9], np.uint64)
    L, mask = oracle.fill_ll_mat(counts, ec_counts, sizes)
    assert mask.tolist() == [True, True, True] and L.shape == (3, 4)
    lut = oracle.precalc_lls(sizes)
    assert L[0, 0] == lut[0, 2] and L[2, 2] == lut[2, 3] and L[1, 3] == np.log(0.01)
    # min_hits = 3: group 0 has 5 reads, group 1 none, group 2 has 1 + 2 = 3
    L2, mask2 = oracle.fill_ll_mat(counts, ec_counts, sizes, min_hits=3)
    assert mask2.tolist() == [True, False, True] and L2.shape == (2, 4)
    lut2 = oracle.precalc_lls(sizes[[0, 2]])
    assert L2[1, 1] == lut2[1, 1] and L2[0, 0] == lut2[0, 2]
    L3, mask3 = oracle.fill_ll_mat(counts, ec_counts, sizes, min_hits=4)
    assert mask3.tolist() == [True, False, False] and L3.shape == (1, 4)
    np.testing.assert_array_equal(oracle.fill_ec_counts(ec_counts), np.log(ec_counts.astype(float)))
