"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs, against the committed golden fixtures, and through size-independent properties.

Protocol: (i) lock-step gate -- same formulation, theta / bound / |g| after
k <= 20 iterations at rel 1e-9 and identical reset decisions; (ii) convergence gate at --tol 1e-6
-- theta rel <= 1e-6 for weights >= 1e-4 and abs <= 1e-8 below (the north-star tolerance);
(iii) dense-state reference-shaped oracle as cross-check.
"""
import numpy as np
import pytest

from conftest import dense_from_csr, load_golden, lutidx_of
from msweep_amd import synth
from msweep_amd.core import ALGO_RCG, Core, MswError
from msweep_amd.likelihood import from_dense, from_grouped_counts, precalc_lls

pytestmark = pytest.mark.gpu

REL, FLOOR, ABS = 1e-6, 1e-4, 1e-8  # north-star tolerance on the mixture weights


def assert_theta(got, ref, rel=REL, floor=FLOOR, abs_=ABS):
    got, ref = np.asarray(got), np.asarray(ref)
    big = ref >= floor
    worst_rel = np.max(np.abs(got - ref)[big] / ref[big], initial=0.0)
    worst_abs = np.max(np.abs(got - ref)[~big], initial=0.0)
    assert worst_rel <= rel, f"worst rel err {worst_rel:.3e} on weights >= {floor}"
    assert worst_abs <= abs_, f"worst abs err {worst_abs:.3e} on weights < {floor}"


def lockstep(tr, rt, k, rel=1e-9):
    k = min(k, tr["n"], int(np.sum(rt["didreset"] >= 0)))
    assert k > 0
    assert tr["didreset"][:k].tolist() == rt["didreset"][:k].tolist()
    np.testing.assert_allclose(tr["bound"][:k], rt["bound"][:k], rtol=rel)
    np.testing.assert_allclose(tr["newnorm"][:k], rt["newnorm"][:k], rtol=1e-7, atol=1e-300)
    np.testing.assert_allclose(tr["theta"][:k], rt["theta"][:k], rtol=rel, atol=1e-15)


@pytest.fixture(params=["dense", "auto"])
def dense_mode(request, monkeypatch):
    """msw_core_set_dense_logl either keeps the matrix dense ("dense": MSWEEP_DENSE_COMPRESS=0) or may
    re-express it as CSR-of-ECs + value table when it has background structure ("auto", the default)."""
    if request.param == "dense":
        monkeypatch.setenv("MSWEEP_DENSE_COMPRESS", "0")
    else:
        monkeypatch.delenv("MSWEEP_DENSE_COMPRESS", raising=False)
    return request.param


def solve_csr(core, p, alpha0=None, logc=None, trace=20, **kw):
    G = len(p["group_sizes"])
    alpha0 = np.ones(G) if alpha0 is None else alpha0
    lik = from_grouped_counts(core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
    logc = lik.log_counts() if logc is None else logc
    core.set_trace_theta(trace)
    res = core.solve(logc, alpha0, **kw)
    return res, core.trace(trace, with_theta=True), logc, alpha0


@pytest.mark.parametrize("R,G,seed,mo", [(3000, 60, 5, 6), (50000, 300, 6, 6), (200000, 1000, 7, 15)])
def test_csr_lockstep_and_convergence_vs_structured_oracle(gpu_core, oracle, R, G, seed, mo):
    p = synth.make_csr_problem(R, G, seed=seed, max_other=mo)
    res, tr, logc, alpha0 = solve_csr(gpu_core, p)
    lut = precalc_lls(p["group_sizes"])
    ref = oracle.rcg_optl_csr(p["rowptr"], p["grp"], lutidx_of(p, lut), lut, np.log(0.01), G, logc, alpha0, trace=20)
    lockstep(tr, ref["trace"], 20)
    assert res["iters"] == ref["iters"]
    assert res["bound"] == pytest.approx(ref["bound"], rel=1e-11)
    assert_theta(res["theta"], ref["theta"])
    assert res["theta"].sum() == pytest.approx(1.0, abs=1e-12)


def test_csr_vs_dense_state_reference_shaped_oracle(gpu_core, oracle):
    """The HIP CSR path against rcgpar's dense-state algorithm as the reference structures it."""
    p = synth.make_csr_problem(20000, 120, seed=8, max_other=8)
    res, tr, logc, alpha0 = solve_csr(gpu_core, p)
    lut = precalc_lls(p["group_sizes"])
    L = dense_from_csr(p, lut)
    d = oracle.rcg_optl_dense(L, logc, alpha0, trace=20)
    lockstep(tr, d["trace"], 20, rel=1e-8)
    assert abs(res["iters"] - d["iters"]) <= 5  # the reference itself varies by 10 with -t (docs/gpubenchmarks.md:15-17)
    assert_theta(res["theta"], oracle.mixture_components(d["gamma"], logc))
    # gamma materialisation (K6) against the oracle's dense gamma, as probabilities
    g = gpu_core.gamma()
    assert g.shape == L.shape
    np.testing.assert_allclose(np.exp(g).sum(0), 1.0, rtol=1e-12)
    np.testing.assert_allclose(np.exp(g), np.exp(d["gamma"]), atol=1e-6)
    # the resident likelihood expands to exactly the reference's dense matrix
    np.testing.assert_array_equal(gpu_core.get_dense_logl(), L)


@pytest.mark.parametrize("idx", range(7))
def test_golden_fixture_dense_path(gpu_core, idx, dense_mode):
    """HIP dense-L path against the committed golden trajectories (tests/golden/rcg_golden.json)."""
    c = load_golden("rcg_golden.json")["cases"][idx]
    L = np.array(c["logl"])
    logc = np.array([-np.inf if x is None else x for x in c["logc"]])
    alpha0 = np.array(c["alpha0"])
    from_dense(gpu_core, L, logc)
    gpu_core.set_trace_theta(20)
    res = gpu_core.solve(logc, alpha0, c["tol"], c["max_iters"])
    e = c["expect"]
    assert res["iters"] == e["iters"], c["name"]
    tr = gpu_core.trace(20, with_theta=True)
    k = min(10, e["iters"])
    np.testing.assert_allclose(tr["bound"][:k], e["trace"]["bound"][:k], rtol=1e-11)
    assert tr["didreset"][:k].tolist() == e["trace"]["didreset"][:k]
    np.testing.assert_allclose(tr["theta"][:k], np.array(e["trace"]["theta"])[:k], rtol=1e-8)
    assert_theta(res["theta"], e["theta"])


@pytest.mark.parametrize("E,G,seed", [(2000, 40, 3), (20000, 500, 4), (3000, 1000, 5), (513, 65, 6)])
def test_dense_path_vs_dense_state_oracle(gpu_core, oracle, E, G, seed, dense_mode):
    p = synth.make_dense_problem(E, G, seed=seed)
    alpha0 = np.ones(G)
    from_dense(gpu_core, p["logl"], p["logc"])
    # continuous listed values: one table slot per listed cell (the 20000 x 500 case builds its tables with k_tables)
    compressed = gpu_core.shape()[2] < G * E
    if dense_mode == "dense":
        assert not compressed
    elif G > 40:                      # 2000 x 40 sits at the density limit (a quarter of the cells listed)
        assert compressed
    gpu_core.set_trace_theta(20)
    res = gpu_core.solve(p["logc"], alpha0)
    tr = gpu_core.trace(20, with_theta=True)
    ref = oracle.rcg_optl_dense(p["logl"], p["logc"], alpha0, trace=20)
    lockstep(tr, ref["trace"], 15, rel=1e-8)
    assert abs(res["iters"] - ref["iters"]) <= 5
    assert_theta(res["theta"], oracle.mixture_components(ref["gamma"], p["logc"]))
    s = oracle.rcg_optl_dense_structured(p["logl"], p["logc"], alpha0, trace=20)
    lockstep(tr, s["trace"], 20)
    np.testing.assert_array_equal(gpu_core.get_dense_logl(), p["logl"])


@pytest.mark.parametrize("R,G,seed", [(40000, 200, 31), (150000, 9000, 32)])
def test_reference_shaped_dense_matrix_takes_the_csr_sweeps(gpu_core, oracle, R, G, seed):
    """The matrix LL_WOR21::fill_ll_mat hands to rcg_optl (include/Likelihood.hpp:176-185): log(zi)
    background + lookup-table values.  msw_core_set_dense_logl re-expresses it on the device as
    CSR-of-ECs + value table: same solve as the CSR boundary, exact expansion back, any group count."""
    p = synth.make_csr_problem(R, G, seed=seed, max_other=6)
    lut = precalc_lls(p["group_sizes"])
    E = len(p["ec_counts"])
    if G > 8192:                        # keep the dense copy small: the first ECs only
        E = 3000
        p = dict(p, rowptr=p["rowptr"][:E + 1], ec_counts=p["ec_counts"][:E])
        p["grp"], p["cnt"] = p["grp"][:p["rowptr"][E]], p["cnt"][:p["rowptr"][E]]
    L = dense_from_csr(p, lut)
    logc = np.log(p["ec_counts"].astype(float))
    alpha0 = np.ones(G)
    from_dense(gpu_core, L, logc)
    g_, e_, nnz = gpu_core.shape()
    assert (g_, e_) == (G, E) and nnz == p["rowptr"][E]
    gpu_core.set_trace_theta(20)
    res = gpu_core.solve(logc, alpha0)
    tr = gpu_core.trace(20, with_theta=True)
    ref = oracle.rcg_optl_csr(p["rowptr"], p["grp"], lutidx_of(p, lut), lut, np.log(0.01), G, logc, alpha0, trace=20)
    lockstep(tr, ref["trace"], 20)
    assert res["iters"] == ref["iters"]
    assert_theta(res["theta"], ref["theta"])
    np.testing.assert_array_equal(gpu_core.get_dense_logl(), L)
    # the CSR boundary fed with the same cells
    res2, _, _, _ = solve_csr(gpu_core, p)
    assert res2["iters"] == res["iters"]
    if G <= 8192:
        np.testing.assert_allclose(res2["theta"], res["theta"], rtol=1e-9, atol=1e-15)


def test_dense_boundary_edge_matrices(gpu_core, oracle):
    """Shapes the re-expression must get right or leave alone: nothing listed at all, a background that
    holds under half of the cells (stays dense).  (-inf cells turn the RCG update into NaN in the
    reference-shaped oracle as well: not an input of this path.)"""
    G, E = 12, 300
    alpha0 = np.ones(G)
    logc = np.zeros(E)
    # every cell the background: a valid likelihood with nothing listed -> theta = prior mean
    L = np.full((G, E), np.log(0.01))
    from_dense(gpu_core, L, logc)
    assert gpu_core.shape() == (G, E, 0)
    res = gpu_core.solve(logc, alpha0)
    np.testing.assert_allclose(res["theta"], np.full(G, 1.0 / G), rtol=1e-12)
    np.testing.assert_array_equal(gpu_core.get_dense_logl(), L)
    # no value holds half of the sample: kept dense
    rng = np.random.default_rng(8)
    L = rng.normal(-3.0, 1.0, (G, E))
    from_dense(gpu_core, L, logc)
    assert gpu_core.shape()[2] == G * E
    ref = oracle.rcg_optl_dense(L, logc, alpha0)
    assert_theta(gpu_core.solve(logc, alpha0)["theta"], oracle.mixture_components(ref["gamma"], logc))


def test_dense_strided_input_and_gamma(gpu_core, oracle, dense_mode):
    p = synth.make_dense_problem(700, 33, seed=9)
    big = np.zeros((33, 1000))
    big[:, :700] = p["logl"]
    from_dense(gpu_core, big[:, :700], p["logc"])          # ld = 1000 > E
    res = gpu_core.solve(p["logc"], np.ones(33))
    ref = oracle.rcg_optl_dense(p["logl"], p["logc"], np.ones(33))
    assert_theta(res["theta"], oracle.mixture_components(ref["gamma"], p["logc"]))
    np.testing.assert_allclose(np.exp(gpu_core.gamma()), np.exp(ref["gamma"]), atol=1e-6)


def test_edge_cases_csr(gpu_core, oracle):
    lut = precalc_lls(np.array([3, 2, 5, 1], np.uint64))
    lz = np.log(0.01)
    # (a) ECs without any hit (all-background rows), a single-cell row, zero-count ECs (-inf)
    rowptr = np.array([0, 0, 2, 3, 3, 6], np.uint64)
    grp = np.array([0, 2, 1, 3, 0, 2], np.uint32)
    cnt = np.array([3, 1, 2, 1, 1, 5], np.uint32)
    gpu_core.set_csr(rowptr, grp, cnt, lut, lz, 4)
    with np.errstate(divide="ignore"):
        logc = np.log(np.array([4.0, 7.0, 0.0, 2.0, 9.0]))
    alpha0 = np.array([1.0, 0.5, 2.0, 1.0])
    res = gpu_core.solve(logc, alpha0)
    lutidx = (grp * lut.shape[1] + cnt).astype(np.uint32)
    ref = oracle.rcg_optl_csr(rowptr, grp, lutidx, lut, lz, 4, logc, alpha0)
    assert res["iters"] == ref["iters"]
    np.testing.assert_allclose(res["theta"], ref["theta"], rtol=1e-9)
    # (b) one group
    gpu_core.set_csr(np.array([0, 1, 1], np.uint64), np.array([0], np.uint32), np.array([2], np.uint32),
                     lut[:1], lz, 1)
    r1 = gpu_core.solve(np.log([3.0, 5.0]), np.ones(1))
    assert r1["theta"][0] == pytest.approx(1.0, rel=1e-14)
    # (b') one group, ECs with and without the hit: the gradient is exactly zero, as the reference's is (a norm
    # of rounding noise once made beta a ratio of two noises and the state ran away)
    for E, miss in ((3, {0}), (64, {5}), (200, {7, 8})):
        rp = np.concatenate([[0], np.cumsum([0 if j in miss else 1 for j in range(E)])]).astype(np.uint64)
        n = int(rp[-1])
        gpu_core.set_csr(rp, np.zeros(n, np.uint32), np.ones(n, np.uint32), lut[:1], lz, 1)
        lc = np.log(np.arange(1.0, E + 1))
        r1 = gpu_core.solve(lc, np.ones(1))
        o1 = oracle.rcg_optl_csr(rp, np.zeros(n, np.uint32), np.full(n, 1, np.uint32), lut[:1], lz, 1, lc, np.ones(1))
        assert r1["theta"][0] == 1.0 and r1["iters"] == o1["iters"] == 2
        assert r1["bound"] == pytest.approx(o1["bound"], rel=1e-12)
    # (b'') identical groups, symmetric prior: an exactly stationary start (beta = x / 0; the reference-shaped dense
    # oracle returns NaN here).  No momentum for such a step: the weights stay at the symmetric point.
    lut2 = precalc_lls(np.array([2, 2], np.uint64))
    for rp2, c2 in (([0, 2], [57.0]), ([0, 2, 4, 4], [57.0, 3.0, 11.0])):
        rp2 = np.array(rp2, np.uint64)
        n = int(rp2[-1])
        g2, k2 = np.tile([0, 1], n // 2).astype(np.uint32), np.ones(n, np.uint32)
        gpu_core.set_csr(rp2, g2, k2, lut2, lz, 2)
        rs = gpu_core.solve(np.log(c2), np.ones(2))
        os_ = oracle.rcg_optl_csr(rp2, g2, (g2 * lut2.shape[1] + k2).astype(np.uint32), lut2, lz, 2, np.log(c2), np.ones(2))
        np.testing.assert_allclose(rs["theta"], [0.5, 0.5], rtol=1e-12)
        np.testing.assert_allclose(os_["theta"], [0.5, 0.5], rtol=1e-12)
    # (c) a single EC
    gpu_core.set_csr(np.array([0, 2], np.uint64), np.array([1, 3], np.uint32), np.array([1, 1], np.uint32), lut, lz, 4)
    r2 = gpu_core.solve(np.log([10.0]), np.ones(4))
    o2 = oracle.rcg_optl_csr(np.array([0, 2], np.uint64), [1, 3], (np.array([1, 3]) * lut.shape[1] + 1), lut, lz, 4,
                             np.log([10.0]), np.ones(4))
    np.testing.assert_allclose(r2["theta"], o2["theta"], rtol=1e-9)
    # (e) no EC hits anything (nnz = 0): every row is background, the weights follow the prior
    gpu_core.set_csr(np.zeros(4, np.uint64), np.zeros(0, np.uint32), np.zeros(0, np.uint32), lut, lz, 4)
    r4 = gpu_core.solve(np.log([2.0, 3.0, 4.0]), alpha0)
    o4 = oracle.rcg_optl_csr(np.zeros(4, np.uint64), np.zeros(0, np.uint32), np.zeros(0, np.uint32), lut, lz, 4,
                             np.log([2.0, 3.0, 4.0]), alpha0)
    assert r4["iters"] == o4["iters"]
    np.testing.assert_allclose(r4["theta"], o4["theta"], rtol=1e-9)
    # (d) max_iters cap is honoured
    p = synth.make_csr_problem(5000, 40, seed=3, max_other=5)
    res3, _, _, _ = solve_csr(gpu_core, p, max_iters=3)
    assert res3["iters"] == 3


def test_long_rows_and_global_vector_mode(gpu_core, oracle):
    """G > LDS capacity (group vectors in HBM, global atomics) and an EC longer than a tile
    (workgroup-cooperative path)."""
    rng = np.random.default_rng(12)
    G = 9000
    sizes = (1 + rng.poisson(3, G)).astype(np.uint64)
    lut = precalc_lls(sizes)
    rows, cols, cnts = [], [], []
    for j in range(400):
        n = G if j in (7, 200) else int(rng.integers(1, 12))   # two ECs hit every group
        g = np.sort(rng.choice(G, n, replace=False))
        rows.append(n); cols.append(g); cnts.append(rng.integers(1, sizes[g] + 1))
    rowptr = np.concatenate([[0], np.cumsum(rows)]).astype(np.uint64)
    grp = np.concatenate(cols).astype(np.uint32)
    cnt = np.concatenate(cnts).astype(np.uint32)
    logc = np.log(rng.integers(1, 50, 400).astype(float))
    alpha0 = np.ones(G)
    gpu_core.set_csr(rowptr, grp, cnt, lut, np.log(0.01), G)
    gpu_core.set_trace_theta(10)
    res = gpu_core.solve(logc, alpha0)
    tr = gpu_core.trace(10, with_theta=True)
    lutidx = (grp * lut.shape[1] + cnt).astype(np.uint32)
    ref = oracle.rcg_optl_csr(rowptr, grp, lutidx, lut, np.log(0.01), G, logc, alpha0, trace=10)
    lockstep(tr, ref["trace"], 10)
    assert_theta(res["theta"], ref["theta"])


def test_wide_records_large_lut(gpu_core, oracle):
    """More than 65536 LUT slots -> 64-bit records and the table read from HBM/L2."""
    rng = np.random.default_rng(13)
    G = 400
    sizes = rng.permutation(np.arange(100, 100 + G)).astype(np.uint64)   # all sizes distinct: 400 x 500 slots
    lut = precalc_lls(sizes)
    assert lut.size > 65536
    E = 3000
    n = rng.integers(1, 9, E)
    rowptr = np.concatenate([[0], np.cumsum(n)]).astype(np.uint64)
    grp = np.concatenate([np.sort(rng.choice(G, k, replace=False)) for k in n]).astype(np.uint32)
    cnt = rng.integers(1, sizes[grp] + 1).astype(np.uint32)
    logc = np.log(rng.integers(1, 30, E).astype(float))
    gpu_core.set_csr(rowptr, grp, cnt, lut, np.log(0.01), G)
    gpu_core.set_trace_theta(10)
    res = gpu_core.solve(logc, np.ones(G))
    tr = gpu_core.trace(10, with_theta=True)
    lutidx = (grp * lut.shape[1] + cnt).astype(np.uint32)
    ref = oracle.rcg_optl_csr(rowptr, grp, lutidx, lut, np.log(0.01), G, logc, np.ones(G), trace=10)
    lockstep(tr, ref["trace"], 10)
    assert_theta(res["theta"], ref["theta"])


@pytest.mark.parametrize("nbytes", [4, 8])
def test_record_formats_agree_with_oracle(gpu_core, oracle, monkeypatch, nbytes):
    """4-byte (default) and 8-byte SELL records (MSWEEP_RECORD_BYTES developer switch): same
    lock-step trajectory and converged weights; ECs up to 16 cells (every slice length)."""
    monkeypatch.setenv("MSWEEP_RECORD_BYTES", str(nbytes))
    G = 500
    p = synth.make_csr_problem(40000, G, seed=21, max_other=15)
    res, tr, logc, alpha0 = solve_csr(gpu_core, p)
    lut = precalc_lls(p["group_sizes"])
    ref = oracle.rcg_optl_csr(p["rowptr"], p["grp"], lutidx_of(p, lut), lut, np.log(0.01), G, logc, alpha0, trace=20)
    lockstep(tr, ref["trace"], 20)
    assert res["iters"] == ref["iters"]
    assert_theta(res["theta"], ref["theta"])
    # the resident likelihood decodes back to the same dense matrix whatever the record format
    L = gpu_core.get_dense_logl()
    np.testing.assert_array_equal(L[:, :200], dense_from_csr(p, lut)[:, :200])


@pytest.mark.parametrize("shape", ["all_short", "few_long", "forced_on_long", "forced_off"])
def test_passB_short_slice_instantiation(oracle, monkeypatch, shape):
    """Pass B's short-slice instantiation (8 records per lane in registers, 16 wavefronts per workgroup; sweep_kernels.hpp
    RC = 8) is picked when slices of more than 8 rows hold at most a twentieth of the rows.  all_short: ECs of 1..8
    cells; few_long: a handful of ECs of 9..16 cells among them (their slices go through the 8-row kernel's streaming
    branch in two chunks, odd lengths included); forced_on_long: MSWEEP_PASSB_RC=8 on ECs of up to 16 cells (half the
    slices stream); forced_off: MSWEEP_PASSB_RC=16 on the short input.  Every one in lock-step with the oracle; the other
    instantiation on the same layout agrees to rounding (the column sums are integers and the row sums are formed in the
    same order, but W = sum r_j meets over 16 instead of 12 wavefronts); zero counts, the escape byte and an EM run
    included."""
    if shape == "forced_on_long":
        monkeypatch.setenv("MSWEEP_PASSB_RC", "8")
    elif shape == "forced_off":
        monkeypatch.setenv("MSWEEP_PASSB_RC", "16")
    else:
        monkeypatch.delenv("MSWEEP_PASSB_RC", raising=False)
    rng = np.random.default_rng(41)
    G, E = 900, 30000
    sizes = (1 + rng.poisson(6, G)).astype(np.uint64)
    lens = rng.integers(1, 9, E)
    if shape == "few_long":
        lens[rng.choice(E, 200, replace=False)] = rng.integers(9, 17, 200)
    elif shape == "forced_on_long":
        lens = rng.integers(1, 17, E)
    rowptr, grp, cnt, lut = _ragged_problem(rng, E, G, sizes, lens)
    c = rng.integers(1, 20, E).astype(float)
    c[rng.random(E) < 0.2] = 0.0           # a bootstrap replicate's zeros
    c[rng.random(E) < 0.02] = 300.5        # the escape byte
    with np.errstate(divide="ignore"):
        logc = np.log(c)
    alpha0 = rng.uniform(0.5, 2.0, G)
    lutidx = (grp * lut.shape[1] + cnt).astype(np.uint32)
    with Core(0) as core:
        core.set_csr(rowptr, grp, cnt, lut, np.log(0.01), G)
        li = core.layout_info()
        want = {"all_short": 8, "few_long": 8, "forced_on_long": 8, "forced_off": 16}[shape]
        assert li["passB_reg_cells"] == want, li
        assert (li["rows_over_8"] == 0) == (shape in ("all_short", "forced_off")), li
        if shape == "few_long":
            assert 0 < 20 * li["rows_over_8"] <= li["rows"]
        core.set_trace_theta(15)
        res = core.solve(logc, alpha0)
        tr = core.trace(15, with_theta=True)
        em = core.solve(logc, np.maximum(alpha0, 1.0), tol=-1.0, max_iters=25, algo=1)
        h8 = core.layout_hash()
    ref = oracle.rcg_optl_csr(rowptr, grp, lutidx, lut, np.log(0.01), G, logc, alpha0, trace=15)
    lockstep(tr, ref["trace"], 15)
    assert res["iters"] == ref["iters"]
    assert_theta(res["theta"], ref["theta"])
    # the other instantiation on the same layout
    monkeypatch.setenv("MSWEEP_PASSB_RC", "16" if want == 8 else "8")
    with Core(0) as core:
        core.set_csr(rowptr, grp, cnt, lut, np.log(0.01), G)
        assert core.layout_info()["passB_reg_cells"] == (16 if want == 8 else 8) and core.layout_hash() == h8
        core.set_trace_theta(15)
        res2 = core.solve(logc, alpha0)
        tr2 = core.trace(15, with_theta=True)
        em2 = core.solve(logc, np.maximum(alpha0, 1.0), tol=-1.0, max_iters=25, algo=1)
    lockstep(tr2, ref["trace"], 15)
    np.testing.assert_allclose(tr2["theta"][:15], tr["theta"][:15], rtol=1e-11, atol=1e-18)
    assert abs(res2["iters"] - res["iters"]) <= 1
    assert_theta(res2["theta"], res["theta"])
    np.testing.assert_allclose(em2["theta"], em["theta"], rtol=1e-11, atol=1e-18)


def test_fractional_and_large_multiplicities(gpu_core, oracle):
    """Pass B streams the EC multiplicities as bytes and escapes to the fp64 vector for anything
    that is not a small integer: fractional weights, counts >= 255, zeros (bootstrap) in one input."""
    G = 200
    p = synth.make_csr_problem(30000, G, seed=33, max_other=9)
    rng = np.random.default_rng(5)
    E = len(p["ec_counts"])
    c = p["ec_counts"].astype(float)
    kind = rng.integers(0, 4, E)
    c[kind == 1] *= 2.5            # fractional
    c[kind == 2] += 254.0          # >= 255
    c[kind == 3] = 0.0             # -inf in the log
    with np.errstate(divide="ignore"):
        logc = np.log(c)
    res, tr, logc, alpha0 = solve_csr(gpu_core, p, logc=logc)
    lut = precalc_lls(p["group_sizes"])
    ref = oracle.rcg_optl_csr(p["rowptr"], p["grp"], lutidx_of(p, lut), lut, np.log(0.01), G, logc, alpha0, trace=20)
    lockstep(tr, ref["trace"], 20)
    assert res["iters"] == ref["iters"]
    assert_theta(res["theta"], ref["theta"])


def _ragged_problem(rng, E, G, sizes, lens):
    lut = precalc_lls(sizes)
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    grp = np.concatenate([np.sort(rng.choice(G, k, replace=False)) for k in lens] + [np.zeros(0, np.int64)]).astype(np.uint32)
    cnt = rng.integers(1, sizes[grp] + 1).astype(np.uint32)
    return rowptr, grp, cnt, lut


@pytest.mark.parametrize("case", ["short", "ragged", "wide"])
def test_device_packer_matches_host_packer(oracle, monkeypatch, case):
    """The SELL layout is built on the device (pack_kernels.hpp); MSWEEP_HOST_PACK=1 runs the host
    reference implementation.  Same EC order, slice geometry and records, byte for byte -- with
    empty ECs, every slice length, streaming (17..256 cells) and long (> 256) ECs, 8-byte records."""
    rng = np.random.default_rng({"short": 1, "ragged": 2, "wide": 3}[case])
    if case == "short":
        G = 300
        p = synth.make_csr_problem(30000, G, seed=41, max_other=15)
        lut = precalc_lls(p["group_sizes"])
        rowptr, grp, cnt = p["rowptr"], p["grp"], p["cnt"]
    else:
        G = 700 if case == "ragged" else 400
        sizes = (rng.integers(2, 12, G) if case == "ragged" else rng.permutation(np.arange(100, 100 + G))).astype(np.uint64)
        E = 5000
        lens = rng.integers(0, 17, E)
        lens[rng.choice(E, 300, replace=False)] = rng.integers(17, 257, 300)
        lens[rng.choice(E, 12, replace=False)] = rng.integers(257, min(G, 600), 12)
        rowptr, grp, cnt, lut = _ragged_problem(rng, E, G, sizes, lens)
    hashes = []
    for host in (True, False):
        if host:
            monkeypatch.setenv("MSWEEP_HOST_PACK", "1")
        else:
            monkeypatch.delenv("MSWEEP_HOST_PACK")
        with Core(0) as core:
            core.set_csr(rowptr, grp, cnt, lut, np.log(0.01), G)
            hashes.append(core.layout_hash())
            if not host:  # and the packed likelihood still is the likelihood
                E = len(rowptr) - 1
                logc = np.log(rng.integers(1, 20, E).astype(float))
                res = core.solve(logc, np.ones(G))
                lutidx = (grp * lut.shape[1] + cnt).astype(np.uint32)
                ref = oracle.rcg_optl_csr(rowptr, grp, lutidx, lut, np.log(0.01), G, logc, np.ones(G))
                assert_theta(res["theta"], ref["theta"])
    assert hashes[0] == hashes[1]


def test_pack_without_bank_scheduling(oracle, monkeypatch):
    """msw_core_set_pack_schedule(0): the cells keep their CSR order (the faster way to ONE solve) -- device and host
    packer byte for byte, a layout different from the scheduled one, the same answer; the setting holds for the NEXT
    likelihood only from the call on."""
    rng = np.random.default_rng(9)
    G, E = 500, 4000
    sizes = rng.integers(2, 40, G).astype(np.uint64)
    lens = rng.integers(0, 17, E)
    lens[rng.choice(E, 200, replace=False)] = rng.integers(17, 400, 200)
    rowptr, grp, cnt, lut = _ragged_problem(rng, E, G, sizes, lens)
    logc = np.log(rng.integers(1, 20, E).astype(float))
    lutidx = (grp * lut.shape[1] + cnt).astype(np.uint32)
    ref = oracle.rcg_optl_csr(rowptr, grp, lutidx, lut, np.log(0.01), G, logc, np.ones(G))
    hashes = {}
    for sched in (1, 0):
        for host in (True, False):
            if host:
                monkeypatch.setenv("MSWEEP_HOST_PACK", "1")
            else:
                monkeypatch.delenv("MSWEEP_HOST_PACK")
            with Core(0) as core:
                core.set_pack_schedule(sched)
                core.set_csr(rowptr, grp, cnt, lut, np.log(0.01), G)
                assert core.layout_info()["bank_scheduled"] == sched
                hashes[sched, host] = core.layout_hash()
                if not host:
                    res = core.solve(logc, np.ones(G))
                    assert res["iters"] == ref["iters"]
                    assert_theta(res["theta"], ref["theta"])
    assert hashes[1, True] == hashes[1, False] and hashes[0, True] == hashes[0, False]
    assert hashes[0, False] != hashes[1, False]


@pytest.mark.parametrize("G,global_atomics", [(12000, False), (19500, False), (36000, False), (12000, True)])
def test_many_groups_modes(gpu_core, oracle, monkeypatch, G, global_atomics):
    """More groups than the LDS images of the sweeps hold: {e, w} / e_g gathered from memory, the
    column sums still in LDS -- all at once up to ~17 k groups (pass B mode 3), one range of 16 k
    groups per run of the sweep beyond (mode 4: two and three runs here); MSWEEP_GLOBAL_ATOMICS=1
    keeps the column sums in HBM (mode 0)."""
    if global_atomics:
        monkeypatch.setenv("MSWEEP_GLOBAL_ATOMICS", "1")
    p = synth.make_csr_problem(30000, G, seed=17, max_other=8)
    res, tr, logc, alpha0 = solve_csr(gpu_core, p)
    lut = precalc_lls(p["group_sizes"])
    ref = oracle.rcg_optl_csr(p["rowptr"], p["grp"], lutidx_of(p, lut), lut, np.log(0.01), G, logc, alpha0, trace=20)
    lockstep(tr, ref["trace"], 20)
    assert res["iters"] == ref["iters"]
    assert_theta(res["theta"], ref["theta"])


def test_properties_at_scale(gpu_core):
    """Size-independent properties on a problem too large for the dense oracle: sum theta = 1,
    sum N = sum alpha + sum c, EC-splitting invariance, group-permutation equivariance, bound
    monotone over accepted steps, and the fixed-point condition of the RCG-VB optimum."""
    p = synth.make_csr_problem(2_000_000, 5000, seed=2)
    G = 5000
    res, tr, logc, alpha0 = solve_csr(gpu_core, p, trace=0)
    assert res["theta"].sum() == pytest.approx(1.0, abs=1e-11)
    t = gpu_core.trace(4096)
    b = t["bound"]
    ok = t["didreset"] == 0
    assert np.all(np.diff(b)[ok[1:]] > -1e-6)
    # EC splitting: one EC with count c == two identical ECs with counts c1 + c2
    rp = p["rowptr"].astype(np.int64)
    E = len(rp) - 1
    big = np.nonzero(p["ec_counts"] >= 2)[0][:50000]
    lens = np.diff(rp)
    extra_idx = np.concatenate([np.arange(rp[j], rp[j + 1]) for j in big[:2000]])
    sel = big[:2000]
    q = dict(p)
    q["rowptr"] = np.concatenate([rp, rp[-1] + np.cumsum(lens[sel])]).astype(np.uint64)
    q["grp"] = np.concatenate([p["grp"], p["grp"][extra_idx]])
    q["cnt"] = np.concatenate([p["cnt"], p["cnt"][extra_idx]])
    c = p["ec_counts"].copy()
    moved = c[sel] // 2
    c[sel] -= moved
    q["ec_counts"] = np.concatenate([c, moved])
    # (compared after the same number of iterations: the stop test -- a gain of 1e-6 on a bound of 1e7 --
    # is decided by the last bits, and the two problems sum their ECs in different orders)
    res2, _, _, _ = solve_csr(gpu_core, q, trace=0, tol=-1.0, max_iters=res["iters"])
    assert_theta(res2["theta"], res["theta"], rel=5e-6)
    assert abs(solve_csr(gpu_core, q, trace=0)[0]["iters"] - res["iters"]) <= 1
    # group permutation
    perm = np.random.default_rng(1).permutation(G)
    inv = np.argsort(perm)
    r = dict(p)
    r["grp"] = inv[p["grp"]].astype(np.uint32)         # old group g becomes inv[g]
    r["group_sizes"] = p["group_sizes"][perm]
    res3, _, _, _ = solve_csr(gpu_core, r, trace=0, tol=-1.0, max_iters=res["iters"])
    assert_theta(res3["theta"][inv], res["theta"], rel=5e-6)
    assert abs(solve_csr(gpu_core, r, trace=0)[0]["iters"] - res["iters"]) <= 1


def test_error_behaviour(gpu_core):
    """Errors surface as MswError with text (the C++ shim throws std::runtime_error the same way,
    src/mSWEEP.cpp:403-406)."""
    lut = precalc_lls(np.array([2, 2], np.uint64))
    with pytest.raises(MswError, match="group id out of range"):
        gpu_core.set_csr(np.array([0, 1], np.uint64), np.array([5], np.uint32), np.array([1], np.uint32), lut, -4.6, 2)
    with pytest.raises(MswError, match="hit count exceeds"):
        gpu_core.set_csr(np.array([0, 1], np.uint64), np.array([0], np.uint32), np.array([9], np.uint32), lut, -4.6, 2)
    with pytest.raises(MswError, match="rowptr not monotone"):
        gpu_core.set_csr(np.array([0, 2, 1, 2], np.uint64), np.array([0, 1], np.uint32), np.array([1, 1], np.uint32), lut, -4.6, 2)
    with pytest.raises(MswError):
        gpu_core.solve(np.zeros(3), np.ones(2))           # no likelihood resident after the failures
    gpu_core.set_csr(np.array([0, 1], np.uint64), np.array([0], np.uint32), np.array([1], np.uint32), lut, -4.6, 2)
    with pytest.raises(MswError, match="expected logc"):
        gpu_core.solve(np.zeros(3), np.ones(2))
    with pytest.raises(MswError, match="unknown algorithm"):
        gpu_core.solve(np.zeros(1), np.ones(2), algo=7)
    with pytest.raises(MswError):
        Core(99)


@pytest.mark.parametrize("E,G,seed", [(1500, 1500, 7), (400, 5000, 8)])
def test_dense_path_many_groups(gpu_core, oracle, E, G, seed, dense_mode):
    """1024 < G <= 8192: the two-sweep dense kernels (an EC no longer fits a wavefront's registers)."""
    p = synth.make_dense_problem(E, G, seed=seed)
    alpha0 = np.ones(G)
    from_dense(gpu_core, p["logl"], p["logc"])
    gpu_core.set_trace_theta(15)
    res = gpu_core.solve(p["logc"], alpha0)
    tr = gpu_core.trace(15, with_theta=True)
    s = oracle.rcg_optl_dense_structured(p["logl"], p["logc"], alpha0, trace=15, want_gamma=True)
    lockstep(tr, s["trace"], 15, rel=1e-8)
    assert abs(res["iters"] - s["iters"]) <= 3
    assert_theta(res["theta"], s["theta"])
    np.testing.assert_allclose(np.exp(gpu_core.gamma()), np.exp(s["gamma"]), atol=1e-6)


def test_unstructured_dense_matrix_beyond_the_dense_sweeps(gpu_core, oracle):
    """More than 8192 groups and no background structure at all (every cell its own value): the matrix is
    re-expressed as CSR-of-ECs with every cell listed -- any dense matrix the reference's boundary accepts solves."""
    rng = np.random.default_rng(0)
    G, E = 8300, 120
    L = rng.normal(-3.0, 1.0, size=(G, E))
    logc = np.log(rng.integers(1, 30, E).astype(float))
    alpha0 = np.ones(G)
    gpu_core.set_dense_logl(L)
    assert gpu_core.shape()[:2] == (G, E) and gpu_core.shape()[2] >= G * E - E
    gpu_core.set_trace_theta(10)
    res = gpu_core.solve(logc, alpha0)
    tr = gpu_core.trace(10, with_theta=True)
    s = oracle.rcg_optl_dense_structured(L, logc, alpha0, trace=10)
    lockstep(tr, s["trace"], 10, rel=1e-8)
    assert abs(res["iters"] - s["iters"]) <= 3
    assert_theta(res["theta"], s["theta"])
    np.testing.assert_array_equal(gpu_core.get_dense_logl(), L)


@pytest.mark.parametrize("multilane", ["1", "0"])
def test_mid_length_ecs_slice_classes_and_streaming_path(oracle, monkeypatch, multilane):
    """ECs with 17..256 cells.  Default: they take 2, 4, 8 or 16 lanes of their slice (sell.hpp slice classes), every
    slice stays within the 16 rows of the sweeps' register path and the partial row sums meet by DPP.
    MSWEEP_MULTILANE=0: one lane per EC as for short ECs -- slices too long for the register buffers take the
    streaming branch of both sweeps (odd lengths exercise the missing last row).  Both against the oracle, lock-step."""
    monkeypatch.setenv("MSWEEP_MULTILANE", multilane)
    gpu_core = Core(0)
    rng = np.random.default_rng(17)
    G = 700
    sizes = (1 + rng.poisson(4, G)).astype(np.uint64)
    lut = precalc_lls(sizes)
    lens = np.concatenate([rng.integers(17, 257, 300), rng.integers(0, 17, 500), [255, 256, 17, 33, 257, 512, 513, 699]])
    rng.shuffle(lens)
    cols = [np.sort(rng.choice(G, int(n), replace=False)) for n in lens]
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    grp = np.concatenate(cols).astype(np.uint32)
    cnt = rng.integers(1, sizes[grp] + 1).astype(np.uint32)
    E = len(lens)
    logc = np.log(rng.integers(1, 30, E).astype(float))
    alpha0 = rng.uniform(0.5, 2.0, G)
    gpu_core.set_csr(rowptr, grp, cnt, lut, np.log(0.01), G)
    li = gpu_core.layout_info()
    if multilane == "1":
        # ECs of 513..1024 / 257..512 / 129..256 / 65..128 / 33..64 / 17..32 / <= 16 cells: 1, 2, 4, 8, 16, 32, 64 to a slice
        want = [int(np.ceil(np.sum((lens > lo) & (lens <= hi)) / per))
                for lo, hi, per in ((512, 1024, 1), (256, 512, 2), (128, 256, 4), (64, 128, 8), (32, 64, 16), (16, 32, 32), (-1, 16, 64))]
        assert li["slices_by_lanes"] == want and li["max_rows"] <= 16 and li["n_long_ecs"] == 0
    else:
        assert li["slices_by_lanes"][:6] == [0] * 6 and li["max_rows"] == 256 and li["n_long_ecs"] == 4
    gpu_core.set_trace_theta(15)
    res = gpu_core.solve(logc, alpha0)
    tr = gpu_core.trace(15, with_theta=True)
    lutidx = (grp * lut.shape[1] + cnt).astype(np.uint32)
    ref = oracle.rcg_optl_csr(rowptr, grp, lutidx, lut, np.log(0.01), G, logc, alpha0, trace=15, want_gamma=True)
    lockstep(tr, ref["trace"], 15)
    assert res["iters"] == ref["iters"]
    assert_theta(res["theta"], ref["theta"])
    np.testing.assert_allclose(np.exp(gpu_core.gamma()), np.exp(ref["gamma"]), atol=1e-7)
    # fractional and large multiplicities (the escape byte: the EC's lanes all read its double) and a bootstrap-like
    # count vector with zeros
    logc2 = np.log(np.where(rng.random(E) < 0.3, rng.uniform(0.5, 400.0, E), rng.integers(0, 6, E).astype(float)) + 1e-300)
    logc2[np.exp(logc2) < 1e-200] = -np.inf
    res2 = gpu_core.solve(logc2, alpha0)
    ref2 = oracle.rcg_optl_csr(rowptr, grp, lutidx, lut, np.log(0.01), G, logc2, alpha0)
    assert abs(res2["iters"] - ref2["iters"]) <= 1
    assert_theta(res2["theta"], ref2["theta"])
    # bootstrap replicates: the resampled counts reach every lane of their EC (two replicates in flight share the records)
    w = rng.integers(1, 30, E).astype(np.uint32)
    theta, iters = gpu_core.bootstrap(w, 3, int(w.sum()), 0, 3, alpha0)
    counts = oracle.bootstrap_counts(w, 3, int(w.sum()), 3)
    for b in range(3):
        with np.errstate(divide="ignore"):
            refb = oracle.rcg_optl_csr(rowptr, grp, lutidx, lut, np.log(0.01), G, np.log(counts[b].astype(float)), alpha0)
        assert abs(int(iters[b]) - refb["iters"]) <= 1
        assert_theta(theta[b], refb["theta"])
    gpu_core.close()


def test_solves_are_bit_reproducible(gpu_core):
    """Fixed-point integer column sums (sweep_kernels.hpp kFx): the order in which the wavefronts of a
    workgroup reach the LDS atomics no longer matters -- repeated solves, and bootstrap replicates run
    with different numbers of replicates in flight, give the same bits."""
    p = synth.make_csr_problem(400_000, 1500, seed=51, max_other=10)
    lik = from_grouped_counts(gpu_core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
    alpha0 = np.ones(1500)
    first = gpu_core.solve(lik.log_counts(), alpha0)
    for _ in range(4):
        r = gpu_core.solve(lik.log_counts(), alpha0)
        assert r["iters"] == first["iters"] and r["bound"] == first["bound"]
        np.testing.assert_array_equal(r["theta"], first["theta"])
    w = p["ec_counts"].astype(np.uint32)
    runs = []
    for streams in ("1", "4", "2"):
        import os
        os.environ["MSWEEP_BOOTSTRAP_STREAMS"] = streams
        try:
            runs.append(gpu_core.bootstrap(w, 5, int(w.sum()), 0, 5, alpha0))
        finally:
            del os.environ["MSWEEP_BOOTSTRAP_STREAMS"]
    for th, it in runs[1:]:
        np.testing.assert_array_equal(th, runs[0][0])
        np.testing.assert_array_equal(it, runs[0][1])


def test_continue_is_the_same_solve(gpu_core):
    """msw_core_continue (bench.py: W warm-up steps, then K timed steps of the SAME solve): 5 + 20 iterations
    give the bits of 25 in one go."""
    p = synth.make_csr_problem(100_000, 300, seed=52, max_other=8)
    lik = from_grouped_counts(gpu_core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
    alpha0 = np.ones(300)
    gpu_core.set_fixed_iters(True)
    try:
        gpu_core.prepare(lik.log_counts(), alpha0)
        whole = gpu_core.run(max_iters=25)
        gpu_core.run(max_iters=5)
        part = gpu_core.continue_(20)
        tm = gpu_core.last_timing()
    finally:
        gpu_core.set_fixed_iters(False)
    assert whole["iters"] == 25 and part["iters"] == 25 and tm["iters"] == 20
    assert part["bound"] == whole["bound"]
    np.testing.assert_array_equal(part["theta"], whole["theta"])
    with pytest.raises(MswError, match="fixed-iteration"):
        gpu_core.continue_(3)


def test_gamma_block_equals_the_columns_of_gamma(gpu_core, oracle):
    """msw_core_gamma_block: any block of ECs of the log-responsibility matrix (what --write-probs and the
    binning input read EC by EC, src/Sample.cpp:63-85), CSR and dense flavours; blocks tile the matrix."""
    p = synth.make_csr_problem(30000, 90, seed=53, max_other=7)
    lik = from_grouped_counts(gpu_core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
    gpu_core.solve(lik.log_counts(), np.ones(90))
    full = gpu_core.gamma()
    E = full.shape[1]
    for e0, e1 in ((0, 1), (17, 4000), (E - 5, E), (0, E), (123, 123)):
        np.testing.assert_array_equal(gpu_core.gamma_block(e0, e1), full[:, e0:e1])
    lut = precalc_lls(p["group_sizes"])
    ref = oracle.rcg_optl_csr(p["rowptr"], p["grp"], lutidx_of(p, lut), lut, np.log(0.01), 90, lik.log_counts(),
                              np.ones(90), want_gamma=True)
    np.testing.assert_allclose(np.exp(full), np.exp(ref["gamma"]), atol=1e-9)
    with pytest.raises(MswError, match="EC range"):
        gpu_core.gamma_block(5, E + 1)
    d = synth.make_dense_problem(3000, 40, seed=9)
    import os
    os.environ["MSWEEP_DENSE_COMPRESS"] = "0"
    try:
        from_dense(gpu_core, d["logl"], d["logc"])
    finally:
        del os.environ["MSWEEP_DENSE_COMPRESS"]
    gpu_core.solve(d["logc"], np.ones(40))
    full = gpu_core.gamma()
    np.testing.assert_array_equal(gpu_core.gamma_block(100, 2100), full[:, 100:2100])


def test_hbm_stream_rates_are_plausible(gpu_core):
    """msw_core_hbm_stream_rates (bench.py's measured ceiling): between a tenth of the specification and the
    specification itself, and refused below 1 MiB."""
    r, t = gpu_core.hbm_stream_rates(1 << 28, 2)
    assert 800.0 < r < 8000.0 and 800.0 < t < 8000.0
    with pytest.raises(MswError, match="at least 1 MiB"):
        gpu_core.hbm_stream_rates(1000, 1)


def test_rejected_step_at_the_last_allowed_iteration(gpu_core, oracle):
    """--max-iters ends the run on an evaluation whose verdict is still pending (the verdict on a slot's evaluation is
    taken by the next slot's k_finstep, or by a closing verdict-only launch): when THAT evaluation is a rejected step
    the steepest-descent retry has to be re-evaluated before the run may end.  Every max_iters from 1 up to past the
    oracle's resets, bound and theta of the last iteration in lock-step."""
    import os
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "fuzz", "seed1_case25.npz"))   # resets at iteration 4
    G = len(d["alpha0"])
    lutidx = (d["grp"] * d["lut"].shape[1] + d["cnt"]).astype(np.uint32)
    ref = oracle.rcg_optl_csr(d["rowptr"], d["grp"], lutidx, d["lut"], np.log(0.01), G, d["logc"], d["alpha0"], tol=-1.0,
                              max_iters=8, trace=8)["trace"]
    assert ref["didreset"][:8].tolist().count(1) >= 1
    gpu_core.set_csr(d["rowptr"], d["grp"], d["cnt"], d["lut"], np.log(0.01), G)
    for n in range(1, 9):
        gpu_core.set_trace_theta(n)
        res = gpu_core.solve(d["logc"], d["alpha0"], tol=-1.0, max_iters=n)
        tr = gpu_core.trace(n, with_theta=True)
        assert res["iters"] == n and tr["n"] == n
        assert tr["didreset"].tolist() == ref["didreset"][:n].tolist(), n
        np.testing.assert_allclose(tr["bound"], ref["bound"][:n], rtol=1e-9, err_msg=str(n))
        np.testing.assert_allclose(res["theta"], ref["theta"][n - 1], rtol=1e-6, atol=1e-13, err_msg=str(n))
    gpu_core.set_trace_theta(0)
    # the same through fixed-iteration runs continued in pieces (bench.py's W warm-up steps, then K timed steps)
    gpu_core.set_fixed_iters(True)
    gpu_core.prepare(d["logc"], d["alpha0"])
    gpu_core.run(max_iters=3)
    r = gpu_core.continue_(2)           # ends on the rejected step
    assert r["iters"] == 5
    np.testing.assert_allclose(r["theta"], ref["theta"][4], rtol=1e-6, atol=1e-13)
    r = gpu_core.continue_(3)
    assert r["iters"] == 8 and r["bound"] == pytest.approx(ref["bound"][7], rel=1e-9)
    gpu_core.set_fixed_iters(False)
