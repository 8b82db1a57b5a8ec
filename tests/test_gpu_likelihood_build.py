"""GPU: device build of the likelihood from the pseudoalignment (K0-K2) against the oracle's
restatement of LL_WOR21::fill_ll_mat (include/Likelihood.hpp:109-195)."""
import numpy as np
import pytest

from msweep_amd import synth
from msweep_amd.core import MswError
from msweep_amd.likelihood import from_alignment, from_grouped_counts
from test_gpu_rcg import assert_theta

pytestmark = pytest.mark.gpu


def _oracle_dense(oracle, aln, p, min_hits=0, q=0.65, e=0.01, zi=0.01):
    counts = oracle.group_counts(aln["ec_tptr"], aln["ec_targets"], aln["target_group"], len(p["group_sizes"]))
    return oracle.fill_ll_mat(counts, p["ec_counts"], p["group_sizes"], q, e, zi, min_hits), counts


@pytest.mark.parametrize("R,G,seed,mo", [(4000, 50, 31, 6), (20000, 300, 32, 12)])
def test_build_matches_oracle_dense_matrix(gpu_core, oracle, R, G, seed, mo):
    p = synth.make_csr_problem(R, G, seed=seed, max_other=mo)
    aln = synth.csr_to_targets(p)
    lik = from_alignment(gpu_core, aln["ec_tptr"], aln["ec_targets"], aln["target_group"], p["group_sizes"],
                         p["ec_counts"])
    (L, mask), _ = _oracle_dense(oracle, aln, p)
    assert lik.n_groups == G and lik.groups_considered().all() and mask.all()
    np.testing.assert_allclose(lik.log_mat(), L, rtol=0, atol=5e-12)       # device lgamma vs glibc lgamma
    np.testing.assert_array_equal(lik.log_counts(), np.log(p["ec_counts"].astype(float)))
    # same abundances as the pre-counted upload path
    r1 = gpu_core.solve(lik.log_counts(), np.ones(G))
    # logc = NULL: the log counts the build left on the device (no upload per solve)
    r0 = gpu_core.solve(None, np.ones(G))
    assert r0["iters"] == r1["iters"]
    np.testing.assert_allclose(r0["theta"], r1["theta"], rtol=1e-12)
    from_grouped_counts(gpu_core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
    r2 = gpu_core.solve(lik.log_counts(), np.ones(G))
    assert r1["iters"] == r2["iters"]
    assert_theta(r1["theta"], r2["theta"])
    with pytest.raises(MswError, match="null logc"):   # an uploaded likelihood keeps no counts
        gpu_core.solve(None, np.ones(G))


def test_build_long_ecs_and_other_parameters(gpu_core, oracle):
    """ECs with more than 64 aligned targets take the workgroup/LDS-histogram path."""
    rng = np.random.default_rng(5)
    G = 40
    sizes = rng.integers(1, 60, G).astype(np.uint64)
    T = int(sizes.sum())
    target_group = rng.permutation(np.repeat(np.arange(G, dtype=np.uint32), sizes.astype(np.int64)))
    E = 300
    lists = []
    for j in range(E):
        n = int(rng.integers(65, 400)) if j % 7 == 0 else int(rng.integers(0, 30))
        lists.append(rng.choice(T, min(n, T), replace=False))
    tptr = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.uint64)
    targets = np.concatenate(lists).astype(np.uint32)
    ecc = rng.integers(1, 40, E).astype(np.uint64)
    lik = from_alignment(gpu_core, tptr, targets, target_group, sizes, ecc, q=0.5, e=0.05, zero_inflation=0.02)
    counts = oracle.group_counts(tptr, targets, target_group, G)
    L, _ = oracle.fill_ll_mat(counts, ecc, sizes, 0.5, 0.05, 0.02, 0)
    np.testing.assert_allclose(lik.log_mat(), L, rtol=0, atol=2e-11)
    assert gpu_core.shape()[2] == int((counts > 0).sum())


@pytest.mark.parametrize("min_hits", [1, 40, 400])
def test_min_hits_mask_and_compaction(gpu_core, oracle, min_hits):
    p = synth.make_csr_problem(6000, 120, seed=33, max_other=5, theta_support=25)
    aln = synth.csr_to_targets(p)
    lik = from_alignment(gpu_core, aln["ec_tptr"], aln["ec_targets"], aln["target_group"], p["group_sizes"],
                         p["ec_counts"], min_hits=min_hits)
    (L, mask), counts = _oracle_dense(oracle, aln, p, min_hits=min_hits)
    hits = ((counts > 0) * p["ec_counts"][None, :].astype(np.int64)).sum(1)
    assert np.array_equal(mask, hits >= min_hits)
    np.testing.assert_array_equal(lik.groups_considered(), mask)
    assert lik.n_groups == int(mask.sum()) == L.shape[0]
    np.testing.assert_allclose(lik.log_mat(), L, rtol=0, atol=5e-12)
    # estimation on the pruned likelihood matches the oracle on the same pruned matrix
    res = gpu_core.solve(lik.log_counts(), np.ones(lik.n_groups))
    ref = oracle.rcg_optl_dense(L, lik.log_counts(), np.ones(lik.n_groups))
    assert_theta(res["theta"], oracle.mixture_components(ref["gamma"], lik.log_counts()))


def test_build_error_behaviour(gpu_core):
    tptr = np.array([0, 2], np.uint64)
    with pytest.raises(MswError, match="target id out of range"):
        gpu_core.build_likelihood(tptr, np.array([0, 9], np.uint32), np.array([0, 0, 1], np.uint32),
                                  np.array([2, 1], np.uint64), np.array([3], np.uint64))
    with pytest.raises(MswError, match="group indicator out of range"):
        gpu_core.build_likelihood(tptr, np.array([0, 1], np.uint32), np.array([0, 5, 1], np.uint32),
                                  np.array([2, 1], np.uint64), np.array([3], np.uint64))
    # (the pointer and target checks run on the device, in the kernel that reads them: every length class of EC)
    for n_hits in (20, 70):   # an EC of 17..64 hits (one wavefront), of more than 64 (one workgroup)
        tg = np.zeros(n_hits, np.uint32)
        tg[-1] = 9
        with pytest.raises(MswError, match="target id out of range"):
            gpu_core.build_likelihood(np.array([0, n_hits], np.uint64), tg, np.array([0, 0, 1], np.uint32),
                                      np.array([2, 1], np.uint64), np.array([3], np.uint64))
    with pytest.raises(MswError, match="ec_tptr not monotone"):
        # the second EC ends before it starts; the first one would reach past the 3 targets there are
        gpu_core.build_likelihood(np.array([0, 5, 3], np.uint64), np.array([0, 1, 2], np.uint32),
                                  np.array([0, 0, 1], np.uint32), np.array([2, 1], np.uint64), np.array([3, 1], np.uint64))
    with pytest.raises(MswError, match="more sequences of a group"):
        # the same target listed twice: count 2 for a group of size 1
        gpu_core.build_likelihood(tptr, np.array([2, 2], np.uint32), np.array([0, 0, 1], np.uint32),
                                  np.array([2, 1], np.uint64), np.array([3], np.uint64))
    with pytest.raises(MswError, match="removed every group"):
        gpu_core.build_likelihood(tptr, np.array([0, 1], np.uint32), np.array([0, 0, 1], np.uint32),
                                  np.array([2, 1], np.uint64), np.array([3], np.uint64), min_hits=100)
