"""Child process of tests/test_gpu_peer_allreduce.py: n thread-ranks on cuda:0 with the communicator the environment
selects (MSWEEP_ALLREDUCE=peer | rccl-less host staging), started fresh so that the variable is read when the
communicators are created and GPU_MAX_HW_QUEUES gives every rank's stream a hardware queue of its own (the peer
kernel waits for its peers: the ranks' kernels must be able to run side by side).  Prints one JSON line."""
import json
import sys
import threading

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from msweep_amd import synth  # noqa: E402
from msweep_amd.core import ALGO_RCG, Comm, Core  # noqa: E402
from msweep_amd.likelihood import from_grouped_counts  # noqa: E402
from msweep_amd.parallel import csr_block, shard_ecs  # noqa: E402


def run_threads(n, work):
    err = []

    def guarded(r):
        try:
            work(r)
        except Exception as ex:  # surface worker failures
            err.append(repr(ex))
    th = [threading.Thread(target=guarded, args=(r,)) for r in range(n)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    return err


def messages(n_ranks, rounds, seed):
    """Messages of changing size (1 word, then growing past the first inbox, then shrinking): both slot parities, the
    collective re-allocation and the integer / floating-point halves."""
    rng = np.random.default_rng(seed)
    sizes = [(0, 1), (7, 4), (3 * 200, 4), (0, 1), (3 * 5000, 4), (300, 0), (3 * 5000, 4), (0, 1)] * rounds
    msgs = []
    for ni, nr in sizes:
        ints = rng.integers(0, 2**62, (n_ranks, ni), dtype=np.uint64)
        reals = rng.standard_normal((n_ranks, nr)) * 10.0 ** rng.integers(-8, 8, (n_ranks, nr))
        msgs.append((ints, reals))
    return msgs


def main_process_rank():
    """proc-messages | proc-solve  <n_ranks> <rank> <segment>: ONE rank per process (Comm.shm), all on cuda:0 -- the
    process-per-rank set-up of the sharded solve and, with MSWEEP_ALLREDUCE=peer, the hipIpc-mapped inboxes."""
    mode, n_ranks, rank, name = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    out = {"mode": mode, "n_ranks": n_ranks, "rank": rank}
    core = Core(0)
    comm = Comm.shm(name, rank, n_ranks)
    if mode == "proc-messages":
        msgs = messages(n_ranks, 2, 5)
        sums = []
        for ints, reals in msgs:
            a, b, _ = comm.allreduce(ints[rank], reals[rank])
            sums.append((a, b))
        _, _, ms = comm.allreduce(msgs[4][0][rank], msgs[4][1][rank], repeats=200)
        ok = True
        for (ints, reals), (a, b) in zip(msgs, sums):
            want_r = np.zeros(reals.shape[1])
            for r in range(n_ranks):
                want_r = want_r + reals[r]
            ok = ok and np.array_equal(a, ints.sum(axis=0, dtype=np.uint64)) and np.array_equal(b, want_r)
        out.update(ok=bool(ok), n_messages=len(msgs), ms_per_call_15004_words=ms)
    else:
        G = 200
        p = synth.make_csr_problem(60000, G, seed=41, max_other=8)
        bounds = shard_ecs(p["rowptr"], n_ranks)
        blk = csr_block(p, bounds[rank], bounds[rank + 1])
        lik = from_grouped_counts(core, blk["rowptr"], blk["grp"], blk["cnt"], blk["ec_counts"], p["group_sizes"])
        core.set_comm(comm)
        core.set_profiling(True)
        rr = core.solve(lik.log_counts(), np.ones(G), tol=1e-6, algo=ALGO_RCG, max_iters=20000)
        t = core.last_timing()
        out.update(theta=rr["theta"].tolist(), iters=int(rr["iters"]), bound=float(rr["bound"]),
                   collective_ms=float(t["collective_ms"]), collectives=int(t["collectives"]))
        core.set_comm(None)
    comm.close()
    core.close()
    print(json.dumps(out))


def main():
    if sys.argv[1].startswith("proc-"):
        return main_process_rank()
    mode, n_ranks = sys.argv[1], int(sys.argv[2])
    out = {"mode": mode, "n_ranks": n_ranks}
    comms = Comm.local(n_ranks)
    if mode == "messages":
        msgs = messages(n_ranks, 3, 5)
        got = [[None] * len(msgs) for _ in range(n_ranks)]
        timing = [0.0] * n_ranks

        def work(r):
            Core(0).close()          # device context on this thread
            for k, (ints, reals) in enumerate(msgs):
                a, b, _ = comms[r].allreduce(ints[r], reals[r])
                got[r][k] = (a, b)
            _, _, timing[r] = comms[r].allreduce(msgs[4][0][r], msgs[4][1][r], repeats=200)
        err = run_threads(n_ranks, work)
        ok = not err
        if ok:
            for k, (ints, reals) in enumerate(msgs):
                want_i = ints.sum(axis=0, dtype=np.uint64)
                want_r = np.zeros(reals.shape[1])
                for r in range(n_ranks):                 # rank order, as every transport sums
                    want_r = want_r + reals[r]
                for r in range(n_ranks):
                    ok = ok and np.array_equal(got[r][k][0], want_i) and np.array_equal(got[r][k][1], want_r)
        out.update(ok=bool(ok), err=err, n_messages=len(msgs), ms_per_call_15004_words=max(timing))
    elif mode == "solve":
        G = 200
        p = synth.make_csr_problem(60000, G, seed=41, max_other=8)
        bounds = shard_ecs(p["rowptr"], n_ranks)
        res = [None] * n_ranks

        def work(r):
            core = Core(0)
            blk = csr_block(p, bounds[r], bounds[r + 1])
            lik = from_grouped_counts(core, blk["rowptr"], blk["grp"], blk["cnt"], blk["ec_counts"], p["group_sizes"])
            core.set_comm(comms[r])
            rr = core.solve(lik.log_counts(), np.ones(G), tol=1e-6, algo=ALGO_RCG, max_iters=20000)
            core.set_profiling(True)
            rr2 = core.solve(lik.log_counts(), np.ones(G), tol=1e-6, algo=ALGO_RCG, max_iters=20000)
            t = core.last_timing()
            res[r] = {"theta": rr["theta"].tolist(), "iters": int(rr["iters"]), "bound": float(rr["bound"]),
                      "again_same": bool(np.array_equal(rr["theta"], rr2["theta"])),
                      "collective_ms": float(t["collective_ms"]), "collectives": int(t["collectives"])}
            core.set_comm(None)
            core.close()
        err = run_threads(n_ranks, work)
        out.update(err=err, ranks=res)
    elif mode == "timeout":
        # MSWEEP_PEER_TEST_SKIP_RANK=1 (set by the test): rank 1 joins the set-up but never launches its kernel --
        # rank 0's wait must end in an error on the host after MSWEEP_PEER_TIMEOUT_MS, not in a hung grid
        said = [None] * n_ranks

        def work(r):
            Core(0).close()
            try:
                comms[r].allreduce([1, 2, 3], [1.0])
                said[r] = "returned"
            except Exception as ex:
                said[r] = str(ex)
        err = run_threads(n_ranks, work)
        out.update(err=err, said=said)
    for c in comms:
        c.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
