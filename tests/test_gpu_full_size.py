"""GPU: BASELINE.json's full-size configurations through size-independent properties (the dense
CPU oracle cannot finish these sizes): cfg2 = synthetic 1M ECs x 500 groups dense likelihood,
cfg3 = synthetic 10M reads x 5k groups (CSR-of-ECs).  The fixed point of the RCG-VB optimum,
    N_g = alpha_g + sum_j c_j softmax_g(L_gj + digamma(N_g)),
is evaluated independently with torch tensor ops on the same data."""
import numpy as np
import pytest

from msweep_amd import synth
from msweep_amd.likelihood import from_dense, from_grouped_counts, precalc_lls

pytestmark = pytest.mark.gpu


def _fixed_point_residual_dense(L, logc, alpha0, theta):
    import torch
    dev = torch.device("cuda", 0)
    c = torch.from_numpy(np.exp(logc)).to(dev)
    N = torch.from_numpy(theta * np.exp(logc).sum() + alpha0).to(dev)
    psi = torch.special.digamma(N)
    acc = torch.zeros_like(N)
    E = L.shape[1]
    for j0 in range(0, E, 100_000):                      # slabs of ECs, rows = groups
        Ls = torch.from_numpy(np.ascontiguousarray(L[:, j0:j0 + 100_000])).to(dev)
        q = torch.softmax(Ls + psi[:, None], dim=0)
        acc += (q * c[None, j0:j0 + 100_000]).sum(1)
    N2 = acc + torch.from_numpy(alpha0).to(dev)
    return ((N2 - N).abs() / N).max().item()


def test_cfg2_dense_1M_x_500(gpu_core):
    p = synth.make_dense_problem(1_000_000, 500, seed=1)
    alpha0 = np.ones(500)
    from_dense(gpu_core, p["logl"], p["logc"])
    res = gpu_core.solve(p["logc"], alpha0)
    assert res["iters"] < 5000
    assert res["theta"].sum() == pytest.approx(1.0, abs=1e-11)
    t = gpu_core.trace(min(res["iters"], 4096))
    ok = t["didreset"] == 0
    assert np.all(np.diff(t["bound"])[ok[1:]] > -1e-6)            # ELBO monotone over accepted steps
    # independent fixed-point check (the stop rule bounds the ELBO gain, not the residual: loose)
    assert _fixed_point_residual_dense(p["logl"], p["logc"], alpha0, res["theta"]) < 5e-3
    # a tighter tolerance moves theta only marginally (the stop rule is on the ELBO gain)
    res2 = gpu_core.solve(p["logc"], alpha0, tol=1e-9)
    big = res2["theta"] > 1e-4
    assert np.max(np.abs(res["theta"] - res2["theta"])[big] / res2["theta"][big]) < 2e-2
    tm = gpu_core.last_timing()
    print(f"cfg2 dense: iters {res['iters']} / {res2['iters']}, {tm['solve_ms'] / max(res2['iters'], 1):.3f} ms/iter")


def test_cfg3_csr_10M_x_5k(gpu_core):
    p = synth.make_csr_problem(10_000_000, 5000, seed=2)
    G = 5000
    lik = from_grouped_counts(gpu_core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
    res = gpu_core.solve(lik.log_counts(), np.ones(G))
    assert res["iters"] < 5000
    assert res["theta"].sum() == pytest.approx(1.0, abs=1e-11)
    t = gpu_core.trace(min(res["iters"], 4096))
    ok = t["didreset"] == 0
    assert np.all(np.diff(t["bound"])[ok[1:]] > -1e-5)
    # fixed point on the CSR data, independently in numpy on a 200k-EC sample is not a global
    # check; instead verify the global identity with torch sparse-free arithmetic:
    import torch
    dev = torch.device("cuda", 0)
    lut = torch.from_numpy(precalc_lls(p["group_sizes"])).to(dev)
    grp = torch.from_numpy(p["grp"].astype(np.int64)).to(dev)
    cnt = torch.from_numpy(p["cnt"].astype(np.int64)).to(dev)
    rp = p["rowptr"].astype(np.int64)
    row = torch.repeat_interleave(torch.arange(len(rp) - 1, device=dev), torch.from_numpy(np.diff(rp)).to(dev))
    c = torch.from_numpy(p["ec_counts"].astype(np.float64)).to(dev)
    N = torch.from_numpy(res["theta"] * float(p["ec_counts"].sum()) + 1.0).to(dev)
    psi = torch.special.digamma(N)
    lz = float(np.log(0.01))
    cell = torch.exp(lut[grp, cnt] + psi[grp]) - torch.exp(lz + psi[grp])      # listed cell minus background
    Z = torch.exp(lz + psi).sum() + torch.zeros(len(rp) - 1, dtype=torch.float64, device=dev).index_add_(0, row, cell)
    r = c / Z
    acc = torch.exp(lz + psi) * r.sum() + torch.zeros(G, dtype=torch.float64, device=dev).index_add_(0, grp, r[row] * cell)
    resid = (((acc + 1.0) - N).abs() / N).max().item()
    assert resid < 5e-3
    print(f"cfg3: iters {res['iters']}, fixed-point residual {resid:.2e}")


def test_cfg5_scaled_sparse_min_hits(gpu_core):
    """cfg5 scaled to one test box: sparse 3M reads x 20k groups, theta supported on 2 000 groups,
    <= 8 listed groups per read, --min-hits 1 pruning through the device likelihood build."""
    from msweep_amd.likelihood import from_alignment
    p = synth.make_csr_problem(3_000_000, 20_000, seed=3, max_other=7, theta_support=2000)
    aln = synth.csr_to_targets(p)
    lik = from_alignment(gpu_core, aln["ec_tptr"], aln["ec_targets"], aln["target_group"], p["group_sizes"],
                         p["ec_counts"], min_hits=1)
    hit = np.zeros(20_000, bool)
    hit[p["grp"]] = True
    np.testing.assert_array_equal(lik.groups_considered(), hit)        # groups with >= 1 aligned read
    G2 = lik.n_groups
    assert G2 == int(hit.sum()) and 1900 <= G2 <= 2000            # ~90 % of the groups pruned
    res = gpu_core.solve(lik.log_counts(), np.ones(G2))
    assert res["iters"] < 5000 and res["theta"].sum() == pytest.approx(1.0, abs=1e-11)
    # mass concentrates on the true support
    kept = np.nonzero(hit)[0]
    on_support = p["theta_true"][kept] > 0
    assert res["theta"][on_support].sum() > 0.99
    print(f"cfg5-scaled: {G2} of 20000 groups kept, iters {res['iters']}")
