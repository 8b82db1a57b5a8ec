"""GPU: the EC-sharded single solve.  Two ranks are emulated on ONE GPU as two
host threads with two handles and the in-process communicator (the same code path as RCCL apart
from the transport); the RCCL transport itself is exercised with a 1-rank communicator."""
import threading

import numpy as np
import pytest

from conftest import lutidx_of
from msweep_amd import synth
from msweep_amd.core import ALGO_EM, ALGO_RCG, Comm, Core
from msweep_amd.likelihood import from_grouped_counts, precalc_lls
from msweep_amd.parallel import csr_block, shard_ecs
from test_gpu_rcg import assert_theta

pytestmark = pytest.mark.gpu


def _solve_sharded(p, n_ranks, algo=ALGO_RCG, tol=1e-6):
    G = len(p["group_sizes"])
    bounds = shard_ecs(p["rowptr"], n_ranks)
    comms = Comm.local(n_ranks)
    out = [None] * n_ranks
    err = []

    def work(r):
        try:
            core = Core(0)
            blk = csr_block(p, bounds[r], bounds[r + 1])
            lik = from_grouped_counts(core, blk["rowptr"], blk["grp"], blk["cnt"], blk["ec_counts"], p["group_sizes"])
            core.set_comm(comms[r])
            core.set_trace_theta(10)
            res = core.solve(lik.log_counts(), np.ones(G), tol=tol, algo=algo, max_iters=20000)
            res["trace"] = core.trace(10, with_theta=True)
            out[r] = res
            core.set_comm(None)
            core.close()
        except Exception as ex:      # surface worker failures in the main thread
            err.append(ex)

    th = [threading.Thread(target=work, args=(r,)) for r in range(n_ranks)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not err, err
    assert all(o is not None for o in out)
    return out, bounds


@pytest.mark.parametrize("n_ranks", [2, 3])
def test_sharded_equals_single_and_oracle(gpu_core, oracle, n_ranks):
    p = synth.make_csr_problem(60000, 200, seed=41, max_other=8)
    G = 200
    out, bounds = _solve_sharded(p, n_ranks)
    assert bounds[0] == 0 and bounds[-1] == len(p["rowptr"]) - 1 and np.all(np.diff(bounds) > 0)
    # every rank returns the same answer, bit for bit (identical reductions on identical data)
    for r in range(1, n_ranks):
        np.testing.assert_array_equal(out[r]["theta"], out[0]["theta"])
        assert out[r]["iters"] == out[0]["iters"] and out[r]["bound"] == out[0]["bound"]
    # and it is the single-GPU / oracle answer
    lut = precalc_lls(p["group_sizes"])
    logc = np.log(p["ec_counts"].astype(float))
    ref = oracle.rcg_optl_csr(p["rowptr"], p["grp"], lutidx_of(p, lut), lut, np.log(0.01), G, logc, np.ones(G), trace=10)
    assert out[0]["iters"] == ref["iters"]
    np.testing.assert_allclose(out[0]["trace"]["bound"], ref["trace"]["bound"][:10], rtol=1e-9)
    np.testing.assert_allclose(out[0]["trace"]["theta"], ref["trace"]["theta"][:10], rtol=1e-9)
    assert_theta(out[0]["theta"], ref["theta"])
    assert out[0]["theta"].sum() == pytest.approx(1.0, abs=1e-12)


def test_sharded_em(gpu_core):
    p = synth.make_csr_problem(30000, 80, seed=42, max_other=6)
    out, _ = _solve_sharded(p, 2, algo=ALGO_EM, tol=1e-8)
    lik = from_grouped_counts(gpu_core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
    single = gpu_core.solve(lik.log_counts(), np.ones(80), tol=1e-8, algo=ALGO_EM, max_iters=20000)
    assert abs(out[0]["iters"] - single["iters"]) <= 2
    # compared after the same number of iterations (the stop sits in the last bits of the log-likelihood;
    # one slow EM iteration moves the small weights by 1e-5)
    same = gpu_core.solve(lik.log_counts(), np.ones(80), tol=-1.0, algo=ALGO_EM, max_iters=out[0]["iters"])
    np.testing.assert_allclose(out[0]["theta"], same["theta"], rtol=1e-6, atol=1e-10)
    np.testing.assert_array_equal(out[0]["theta"], out[1]["theta"])


def test_rccl_transport_single_rank(gpu_core):
    """ncclAllReduce on the solve stream (1-rank communicator: transport and stream ordering)."""
    p = synth.make_csr_problem(40000, 100, seed=43, max_other=6)
    lik = from_grouped_counts(gpu_core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
    plain = gpu_core.solve(lik.log_counts(), np.ones(100))
    comm = Comm.rccl(Comm.unique_id(), 0, 1, 0)
    gpu_core.set_comm(comm)
    try:
        sharded = gpu_core.solve(lik.log_counts(), np.ones(100))
    finally:
        gpu_core.set_comm(None)
        comm.close()
    assert sharded["iters"] == plain["iters"]
    assert_theta(sharded["theta"], plain["theta"])


@pytest.mark.parametrize("n_ranks,B", [(2, 7), (3, 7), (3, 2), (3, 1)])
def test_bootstrap_dist_gathers_every_replicate_on_every_rank(gpu_core, n_ranks, B):
    """msw_core_bootstrap_dist (the replicate loop of src/mSWEEP.cpp:496-518 over the GPUs of a node):
    thread-ranks with the in-process communicator; every rank ends with the whole B x G table in
    replicate order, equal to the single-handle run whatever the number of ranks."""
    p = synth.make_csr_problem(30000, 80, seed=44, max_other=6)
    G = 80                   # 7 replicates over 2 / 3 ranks: ragged blocks; 2 and 1 over 3: ranks with nothing to solve
    alpha0 = np.ones(G)
    w = p["ec_counts"].astype(np.uint32)
    draws = int(w.sum())
    from_grouped_counts(gpu_core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
    single, it_single = gpu_core.bootstrap(w, 42, draws, 0, B, alpha0)
    comms = Comm.local(n_ranks)
    out, err = [None] * n_ranks, []

    def work(r):
        try:
            core = Core(0)
            from_grouped_counts(core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
            out[r] = core.bootstrap_dist(comms[r], w, 42, draws, B, alpha0)
            assert comms[r].size() == (n_ranks, r) and comms[r].rccl_count() == 0
            core.close()
        except Exception as ex:
            err.append(ex)

    th = [threading.Thread(target=work, args=(r,)) for r in range(n_ranks)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not err, err
    for r in range(n_ranks):
        theta, iters = out[r]
        assert theta.shape == (B, G)
        np.testing.assert_array_equal(theta, out[0][0])          # the gathered table is the same everywhere
        np.testing.assert_array_equal(iters, it_single)
        for b in range(B):
            assert_theta(theta[b], single[b])


def test_bootstrap_dist_failing_rank_fails_every_rank_and_nobody_waits(gpu_core):
    """A rank whose block of replicates fails (here: a bad draw count on it alone) still joins the all-gather
    and reports through its status word: EVERY rank gets an error, none is left in the collective, and the
    communicator is still good for the next call."""
    p = synth.make_csr_problem(20000, 50, seed=46, max_other=5)
    G, B, n_ranks = 50, 5, 3
    alpha0 = np.ones(G)
    w = p["ec_counts"].astype(np.uint32)
    draws = int(w.sum())
    comms = Comm.local(n_ranks)
    first, second = [None] * n_ranks, [None] * n_ranks

    def work(r):
        core = Core(0)
        try:
            from_grouped_counts(core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
            try:
                core.bootstrap_dist(comms[r], w, 42, 0 if r == 1 else draws, B, alpha0)
                first[r] = "ok"
            except RuntimeError as ex:
                first[r] = str(ex)
            second[r] = core.bootstrap_dist(comms[r], w, 42, draws, B, alpha0)
        except Exception as ex:
            second[r] = ex
        finally:
            core.close()

    th = [threading.Thread(target=work, args=(r,)) for r in range(n_ranks)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
        assert not t.is_alive(), "a rank is still waiting in the exchange"
    assert "bootstrap_count must be" in first[1], first
    assert "rank 1 failed" in first[0] and "rank 1 failed" in first[2], first
    for r in range(n_ranks):
        assert not isinstance(second[r], Exception), second[r]
        np.testing.assert_array_equal(second[r][0], second[0][0])


def test_api_misuse_on_a_sharded_handle_does_not_poison_the_group(gpu_core):
    """An argument / state error on a handle with a communicator attached touches no collective: the
    group must stay usable (the in-process group's failure flag is sticky)."""
    p = synth.make_csr_problem(20000, 40, seed=47, max_other=5)
    G, n_ranks = 40, 2
    bounds = shard_ecs(p["rowptr"], n_ranks)
    comms = Comm.local(n_ranks)
    out, err = [None] * n_ranks, []

    def work(r):
        try:
            core = Core(0)
            blk = csr_block(p, bounds[r], bounds[r + 1])
            lik = from_grouped_counts(core, blk["rowptr"], blk["grp"], blk["cnt"], blk["ec_counts"], p["group_sizes"])
            core.set_comm(comms[r])
            if r == 0:
                with pytest.raises(RuntimeError, match="no solve has run"):
                    core.gamma_block(0, 1)
            out[r] = core.solve(lik.log_counts(), np.ones(G), tol=1e-6)
            core.set_comm(None)
            core.close()
        except Exception as ex:
            err.append(ex)

    th = [threading.Thread(target=work, args=(r,)) for r in range(n_ranks)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not err, err
    np.testing.assert_array_equal(out[0]["theta"], out[1]["theta"])


def test_bootstrap_leaves_the_original_estimate_on_the_handle(gpu_core):
    """The reference writes the probabilities of the UN-resampled estimate (src/mSWEEP.cpp:437-493 come
    before the replicate loop at :496): msw_core_gamma after msw_core_bootstrap still describes it."""
    p = synth.make_csr_problem(5000, 30, seed=45, max_other=4)
    lik = from_grouped_counts(gpu_core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
    res = gpu_core.solve(lik.log_counts(), np.ones(30))
    g0 = gpu_core.gamma()
    w = p["ec_counts"].astype(np.uint32)
    gpu_core.bootstrap(w, 3, int(w.sum()), 0, 3, np.ones(30))
    np.testing.assert_array_equal(gpu_core.gamma(), g0)
    assert gpu_core.trace(res["iters"])["n"] == res["iters"]


def test_rccl_allgather_single_rank():
    comm = Comm.rccl(Comm.unique_id(), 0, 1, 0)
    try:
        assert comm.size() == (1, 0) and comm.rccl_count() == 1
        x = np.arange(5, dtype=np.float64)
        np.testing.assert_array_equal(comm.allgather(x), x[None, :])
    finally:
        comm.close()


@pytest.mark.parametrize("n_ranks", [2, 3])
def test_sharded_build_with_min_hits_equals_single_handle(gpu_core, n_ranks):
    """every rank builds the likelihood of its EC block on the device; the --min-hits
    counts (include/Likelihood.hpp:146-163: hits of a group over ALL ECs) are all-reduced, so every rank
    prunes the same groups; the EC-sharded solve on top gives the single-handle answer."""
    from msweep_amd.likelihood import from_alignment
    p = synth.make_csr_problem(40000, 400, seed=46, max_other=5, theta_support=60)
    aln = synth.csr_to_targets(p)
    G = 400
    min_hits = 2200                                 # every group gets ~1700 spurious hits; a shard sees a part only
    lik = from_alignment(gpu_core, aln["ec_tptr"], aln["ec_targets"], aln["target_group"], p["group_sizes"],
                         p["ec_counts"], min_hits=min_hits)
    single = gpu_core.solve(None, np.ones(lik.n_groups))
    assert 1 < lik.n_groups < 60
    bounds = shard_ecs(p["rowptr"], n_ranks)
    tptr = aln["ec_tptr"].astype(np.int64)
    comms = Comm.local(n_ranks)
    out, err = [None] * n_ranks, []

    def work(r):
        try:
            core = Core(0)
            e0, e1 = bounds[r], bounds[r + 1]
            core.set_comm(comms[r])
            lk = from_alignment(core, (tptr[e0:e1 + 1] - tptr[e0]).astype(np.uint64), aln["ec_targets"][tptr[e0]:tptr[e1]],
                                aln["target_group"], p["group_sizes"], p["ec_counts"][e0:e1], min_hits=min_hits)
            res = core.solve(None, np.ones(lk.n_groups))
            out[r] = (lk.groups_considered().copy(), lk.n_groups, res)
            core.set_comm(None)
            core.close()
        except Exception as ex:
            err.append(ex)

    th = [threading.Thread(target=work, args=(r,)) for r in range(n_ranks)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not err, err
    # a shard on its own would see fewer hits: the test means something only if the global counts matter
    local_hits = np.bincount(p["grp"][int(p["rowptr"][bounds[0]]):int(p["rowptr"][bounds[1]])],
                             weights=np.repeat(p["ec_counts"][bounds[0]:bounds[1]],
                                               np.diff(p["rowptr"][bounds[0]:bounds[1] + 1].astype(np.int64))), minlength=G)
    assert np.any((local_hits >= min_hits) != lik.groups_considered())
    for mask, ng, res in out:
        np.testing.assert_array_equal(mask, lik.groups_considered())
        assert ng == lik.n_groups
        assert res["iters"] == single["iters"]
        assert_theta(res["theta"], single["theta"])
        np.testing.assert_array_equal(res["theta"], out[0][2]["theta"])


@pytest.mark.parametrize("min_hits,bad", [(0, "target"), (2200, "target"), (2200, "tptr"), (0, "count")])
def test_sharded_build_failing_rank_fails_every_rank_and_nobody_waits(gpu_core, min_hits, bad):
    """A rank whose OWN block of the pseudoalignment is bad (a target id out of range / ec_tptr not monotone: found on
    the device after K1; an EC that hits more sequences of a group than the group holds: found after the --min-hits
    all-reduce) fails TOGETHER with its peers through the build's status word -- nobody waits in the all-reduce of the
    group hits or in the first collective of a solve -- and the communicator is good for a second, clean build."""
    from msweep_amd.likelihood import from_alignment
    p = synth.make_csr_problem(40000, 400, seed=46, max_other=5, theta_support=60)
    aln = synth.csr_to_targets(p)
    n_ranks = 3
    bounds = shard_ecs(p["rowptr"], n_ranks)
    tptr = aln["ec_tptr"].astype(np.int64)
    comms = Comm.local(n_ranks)
    first, second = [None] * n_ranks, [None] * n_ranks

    def block(r, spoil):
        e0, e1 = bounds[r], bounds[r + 1]
        tp = (tptr[e0:e1 + 1] - tptr[e0]).astype(np.uint64)
        tg = aln["ec_targets"][tptr[e0]:tptr[e1]].copy()
        sizes = p["group_sizes"].copy()
        if spoil and bad == "target":
            tg[len(tg) // 2] = len(aln["target_group"]) + 7
        if spoil and bad == "tptr":
            tp[len(tp) // 2] = tp[len(tp) // 2 + 1] + 3
        if spoil and bad == "count":          # the rank believes group 0 holds one sequence: some EC hits it more often
            g = int(np.bincount(p["grp"][p["cnt"] > 1]).argmax())
            sizes[g] = 1
        return tp, tg, sizes, p["ec_counts"][e0:e1]

    def work(r):
        core = Core(0)
        try:
            core.set_comm(comms[r])
            tp, tg, sizes, ecc = block(r, spoil=(r == 1))
            try:
                from_alignment(core, tp, tg, aln["target_group"], sizes, ecc, min_hits=min_hits)
                first[r] = "ok"
            except RuntimeError as ex:
                first[r] = str(ex)
            tp, tg, sizes, ecc = block(r, spoil=False)
            lk = from_alignment(core, tp, tg, aln["target_group"], sizes, ecc, min_hits=min_hits)
            second[r] = core.solve(None, np.ones(lk.n_groups))
            core.set_comm(None)
        except Exception as ex:
            second[r] = ex
        finally:
            core.close()

    th = [threading.Thread(target=work, args=(r,)) for r in range(n_ranks)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
        assert not t.is_alive(), "a rank is still waiting in a collective of the sharded build"
    want = {"target": "target id out of range", "tptr": "ec_tptr not monotone", "count": "more sequences of a group"}[bad]
    assert want in first[1], first
    assert "another rank of the sharded build failed" in first[0] and "another rank" in first[2], first
    for r in range(n_ranks):
        assert not isinstance(second[r], Exception), second[r]
        np.testing.assert_array_equal(second[r]["theta"], second[0]["theta"])


def test_bootstrap_bad_arguments_fail_the_call_not_a_table_of_nan(gpu_core):
    """msw_core_bootstrap / _dist: an unknown algorithm id or max_iters = 0 is an ERROR (include/msweep_core.h: "bad
    arguments fail the whole call"); only a numerically failed replicate becomes a row of NaN."""
    p = synth.make_csr_problem(5000, 20, seed=48, max_other=4)
    from_grouped_counts(gpu_core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
    w = p["ec_counts"].astype(np.uint32)
    with pytest.raises(RuntimeError, match="unknown algorithm"):
        gpu_core.bootstrap(w, 1, int(w.sum()), 0, 2, np.ones(20), algo=7)
    with pytest.raises(RuntimeError, match="max_iters"):
        gpu_core.bootstrap(w, 1, int(w.sum()), 0, 2, np.ones(20), max_iters=0)
    comm = Comm.local(1)[0]
    with pytest.raises(RuntimeError):
        gpu_core.bootstrap_dist(comm, w, 1, int(w.sum()), 2, np.ones(20), prec=9)
    theta, _ = gpu_core.bootstrap_dist(comm, w, 1, int(w.sum()), 2, np.ones(20))   # and the communicator still works
    assert np.all(np.isfinite(theta))
