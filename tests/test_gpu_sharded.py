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
    np.testing.assert_allclose(out[0]["theta"], single["theta"], rtol=1e-6, atol=1e-10)
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
