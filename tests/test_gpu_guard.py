"""GPU: the cancellation guard (SURVEY.md 7.3 ii, msweep_amd/csrc/sell.hpp "guarded ECs").

The sweeps form Z_j = p0 * U + sum_listed e_g (x_gj - p0).  When one group holds (nearly) all of
U = sum e_g and its cells sit far below the background value (x << p0) the two parts cancel: an
isolate / single-lineage sample whose reads hit the dominant lineage weakly, with a likelihood
table that goes far below log(zi) (custom -q / -e, --read-likelihood matrices) and prior counts
below one (so that the other groups' weights underflow).  Such ECs are evaluated without the
background trick.  Checked against the DENSE-STATE oracle (rcgpar's algorithm on the G x E
matrices, log-domain throughout -- no such cancellation) at the north-star tolerance, and in
lock-step against the structured oracle (which carries the same guard).
"""
import numpy as np
import pytest

from msweep_amd.core import MswError
from test_gpu_rcg import assert_theta, lockstep

pytestmark = pytest.mark.gpu
LOGZI = np.log(0.01)


def isolate_problem(seed, n_ecs=6000, G=40, deep=-55.0, n_levels=6, others=(0, 4), other_values=(-6.0, -0.1)):
    """One dominant group (0) listed in most ECs -- strongly (table value near 0) or weakly (`deep`, far
    below log(zi) = -4.6) -- plus a few other groups per EC (`others`: how many, lo..hi-1) with ordinary values."""
    rng = np.random.default_rng(seed)
    lut = np.empty((G, n_levels + 1))
    lut[:, 0] = LOGZI
    lut[:, 1:] = rng.uniform(*other_values, (G, n_levels))
    lut[0, 1] = deep                  # the dominant group's weak hit
    lut[0, 2] = deep / 2
    lut[0, 3:] = rng.uniform(-1.0, -0.05, n_levels - 2)
    lut[1, 1] = deep * 0.8            # a second group with a deep cell
    rows, grp, cnt = [], [], []
    counts = np.empty(n_ecs, np.uint64)
    for j in range(n_ecs):
        kind = rng.random()
        cells = {}
        if kind < 0.55:               # a read of the dominant lineage, strong hit
            cells[0] = int(rng.integers(3, n_levels + 1))
            counts[j] = rng.integers(1, 400)
        elif kind < 0.85:             # weak hit of the dominant lineage only: the cancelling case
            cells[0] = int(rng.integers(1, 3))
            counts[j] = rng.integers(1, 4)
        elif kind < 0.93:             # dominant weak + the second deep group weak
            cells[0] = 1
            cells[1] = 1
            counts[j] = 1
        else:                         # a read that does not list the dominant group at all
            counts[j] = rng.integers(1, 3)
        for g in rng.choice(np.arange(2, G), rng.integers(*others), replace=False):
            cells[int(g)] = int(rng.integers(1, n_levels + 1))
        if not cells:
            cells[int(rng.integers(2, G))] = 1
        for g in sorted(cells):
            grp.append(g)
            cnt.append(cells[g])
        rows.append(len(cells))
    rowptr = np.concatenate([[0], np.cumsum(rows)]).astype(np.uint64)
    return dict(rowptr=rowptr, grp=np.array(grp, np.uint32), cnt=np.array(cnt, np.uint32), lut=lut,
                ec_counts=counts, G=G)


def dense_of(p):
    G, E = p["G"], len(p["rowptr"]) - 1
    L = np.full((G, E), LOGZI)
    rows = np.repeat(np.arange(E), np.diff(p["rowptr"].astype(np.int64)))
    L[p["grp"], rows] = p["lut"][p["grp"], p["cnt"]]
    return L


@pytest.mark.parametrize("alpha,deep,zero_counts", [(1.0, -55.0, False), (0.5, -55.0, True), (0.01, -55.0, False),
                                                    (1e-3, -120.0, True), (0.01, -30.0, False)])
def test_dominant_group_matches_dense_state_oracle(gpu_core, oracle, alpha, deep, zero_counts):
    _check_dominant(gpu_core, oracle, isolate_problem(seed=int(-deep) + int(alpha * 1000), deep=deep), alpha, deep, zero_counts)


@pytest.mark.parametrize("alpha,deep,zero_counts", [(0.5, -55.0, True), (0.01, -55.0, False)])
def test_dominant_group_with_mid_length_ecs(gpu_core, oracle, alpha, deep, zero_counts):
    """The guarded ECs of slices that hold several lanes per EC (17..256 cells: sell.hpp slice classes): the EC's
    first lane hands it to the guard path, which walks the EC's cells over its lanes."""
    p = isolate_problem(seed=77 + int(alpha * 1000), n_ecs=3000, G=150, deep=deep, others=(18, 120), other_values=(-16.0, -8.0))
    visits = _check_dominant(gpu_core, oracle, p, alpha, deep, zero_counts)
    assert sum(gpu_core.layout_info()["slices_by_lanes"][:-1]) > 0
    assert visits > 0          # the guard path of multi-lane slices is what this test is about


def _check_dominant(gpu_core, oracle, p, alpha, deep, zero_counts):
    G = p["G"]
    alpha0 = np.full(G, alpha)
    with np.errstate(divide="ignore"):
        c = p["ec_counts"].astype(np.float64)
        if zero_counts:
            c[::7] = 0.0                                  # -inf log counts (bootstrap replicates, BootstrapSample.cpp:70)
        logc = np.log(c)
    gpu_core.set_csr(p["rowptr"], p["grp"], p["cnt"], p["lut"], LOGZI, G)
    gpu_core.set_trace_theta(20)
    res = gpu_core.solve(logc, alpha0)
    visits = gpu_core.guarded_visits()
    tr = gpu_core.trace(20, with_theta=True)
    gpu_core.set_trace_theta(0)
    lutidx = (p["grp"] * p["lut"].shape[1] + p["cnt"]).astype(np.uint32)
    s = oracle.rcg_optl_csr(p["rowptr"], p["grp"], lutidx, p["lut"], LOGZI, G, logc, alpha0, trace=20)
    d = oracle.rcg_optl_dense(dense_of(p), logc, alpha0, trace=20)
    th_d = oracle.mixture_components(d["gamma"], logc)
    big = th_d >= 1e-4
    worst = np.max(np.abs(res["theta"] - th_d)[big] / th_d[big])
    print(f"alpha {alpha}, deepest cell {deep}: {visits} guarded EC evaluations; theta[0] = 1 - {1 - th_d[0]:.3e}; iterations hip {res['iters']} / "
          f"structured oracle {s['iters']} / dense-state oracle {d['iters']}; worst rel err vs the dense-state oracle "
          f"{worst:.2e} (weights >= 1e-4), worst abs err below {np.max(np.abs(res['theta'] - th_d)[~big], initial=0):.2e}")
    assert th_d[0] > 0.9
    assert np.all(np.isfinite(res["theta"])) and res["theta"].sum() == pytest.approx(1.0, abs=1e-11)
    lockstep(tr, s["trace"], 20)
    assert abs(res["iters"] - s["iters"]) <= 1      # the stop sits in the last bits of the bound
    assert abs(res["iters"] - d["iters"]) <= 5
    # against the dense-state oracle after the SAME number of iterations
    same = gpu_core.solve(logc, alpha0, tol=-1.0, max_iters=d["iters"])
    assert_theta(same["theta"], th_d)
    # gamma (what --write-probs prints): the guarded ECs' columns too
    g = gpu_core.gamma()
    np.testing.assert_allclose(np.exp(g).sum(0), 1.0, rtol=1e-10)
    np.testing.assert_allclose(np.exp(g), np.exp(d["gamma"]), atol=2e-6)
    return visits


def test_isolate_theta_one_minus_1e_9(gpu_core, oracle):
    """A single lineage at theta >= 1 - 1e-9 with the default prior (alpha = 1).  Nearly all reads hit it
    strongly; a few reads hit it weakly (table values << log zi): for those the dominant group's cell
    and the background sum cancel (Z is 1e-13 of p0 * U), and they are what the other groups live on."""
    rng = np.random.default_rng(4)
    G, E = 30, 3000
    lut = np.full((G, 4), LOGZI)
    lut[0, 1:] = [-36.0, -30.0, -0.2]                    # weak, weak, strong hit of the dominant group
    lut[1:, 1] = -40.0                                   # every other group: only far below the background
    lut[1:, 2:] = -30.0
    kind = rng.random(E)
    kind[:4] = [0.7, 0.9, 0.9, 0.7]                     # four weak ECs, one read each; every other EC strong
    kind[4:] *= 0.6
    rows, grp, cnt = [], [], []
    for j in range(E):
        cells = {0: 3 if kind[j] < 0.6 else (1 if kind[j] < 0.8 else 2)}
        for g in rng.choice(np.arange(1, G), rng.integers(0, 3), replace=False):
            cells[int(g)] = int(rng.integers(1, 4))
        for g in sorted(cells):
            grp.append(g)
            cnt.append(cells[g])
        rows.append(len(cells))
    rowptr = np.concatenate([[0], np.cumsum(rows)]).astype(np.uint64)
    grp, cnt = np.array(grp, np.uint32), np.array(cnt, np.uint32)
    counts = np.where(kind < 0.6, rng.integers(1, 50, E).astype(np.uint64) * 100_000, 1).astype(np.uint64)
    logc = np.log(counts.astype(np.float64))
    gpu_core.set_csr(rowptr, grp, cnt, lut, LOGZI, G)
    # 7e9 reads: the bound is ~1e10 and one fp64 ulp of it is 2e-6 -- the stop rule is given a gain it can
    # resolve (the reference's OpenMP path keeps its bound in long double for this reason)
    d = oracle.rcg_optl_dense(dense_of(dict(rowptr=rowptr, grp=grp, cnt=cnt, lut=lut, G=G)), logc, np.ones(G), tol=1e-3)
    own = gpu_core.solve(logc, np.ones(G), tol=1e-3)
    assert abs(own["iters"] - d["iters"]) <= 2
    res = gpu_core.solve(logc, np.ones(G), tol=-1.0, max_iters=d["iters"])   # the same number of iterations
    th_d = oracle.mixture_components(d["gamma"], logc)
    print(f"isolate: theta[0] = 1 - {1 - res['theta'][0]:.3e} (dense-state oracle 1 - {1 - th_d[0]:.3e}), "
          f"iterations {res['iters']} / {d['iters']}; other groups: hip {res['theta'][1:].max():.3e}, oracle {th_d[1:].max():.3e}")
    assert 1 - th_d[0] < 1e-9
    assert_theta(res["theta"], th_d)
    # the other groups' weights themselves (far below the floor of the north-star tolerance), relatively
    np.testing.assert_allclose(res["theta"][1:], th_d[1:], rtol=1e-5, atol=1e-30)


def test_zero_probability_ec_is_an_error_not_a_nan(gpu_core):
    """An EC whose only cell underflows under every group: reported, not propagated as NaN."""
    G = 4
    lut = np.array([[LOGZI, -800.0]] * G)
    rowptr = np.array([0, 1, 2], np.uint64)
    grp = np.array([0, 0], np.uint32)
    cnt = np.array([1, 1], np.uint32)
    gpu_core.set_csr(rowptr, grp, cnt, lut, LOGZI, G)
    # prior counts so small that every other group's weight underflows: nothing is left for these ECs
    try:
        r = gpu_core.solve(np.zeros(2), np.full(G, 1e-5), max_iters=50)
    except MswError as ex:
        assert "underflow" in str(ex) or "not finite" in str(ex)
    else:
        assert np.all(np.isfinite(r["theta"])) and r["theta"].sum() == pytest.approx(1.0, abs=1e-9)
