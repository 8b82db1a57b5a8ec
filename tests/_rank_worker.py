"""One rank of the multi-GPU tests (tests/test_gpu_multi.py): a fresh process that owns GPU `rank`.
usage: _rank_worker.py RANK WORLD DIR [shm] -- the RCCL unique id travels through DIR/uid (rank 0 writes it), the
results through DIR/out_RANK.npz.  `shm`: every rank on GPU 0, the ranks meet in a shared-memory segment
(msw_comm_create_shm) instead of an RCCL communicator, which refuses two ranks on one device."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from msweep_amd import synth  # noqa: E402
from msweep_amd.core import Comm, Core  # noqa: E402
from msweep_amd.likelihood import from_grouped_counts  # noqa: E402
from msweep_amd.parallel import csr_block, shard_ecs  # noqa: E402


def problem():
    return synth.make_csr_problem(60000, 200, seed=41, max_other=8), 200


def main():
    rank, world, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    shm = len(sys.argv) > 4 and sys.argv[4] == "shm"
    uid_path = os.path.join(d, "uid")
    if shm:
        uid = None
    elif rank == 0:
        uid = Comm.unique_id()
        with open(uid_path + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(uid_path + ".tmp", uid_path)
    else:
        t0 = time.time()
        while not os.path.exists(uid_path):
            if time.time() - t0 > 120:
                sys.exit("rank %d: no unique id after 120 s" % rank)
            time.sleep(0.05)
        uid = open(uid_path, "rb").read()
    if shm:
        comm = Comm.shm("/msweep_rankworker_" + os.path.basename(d), rank, world, 0)
        assert comm.size() == (world, rank) and comm.rccl_count() == 0
    else:
        comm = Comm.rccl(uid, rank, world, rank)
        assert comm.size() == (world, rank) and comm.rccl_count() == world
    p, G = problem()
    alpha0 = np.ones(G)
    core = Core(0 if shm else rank)
    # (a) the replicate loop over the GPUs: every rank holds the whole likelihood
    from_grouped_counts(core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
    w = p["ec_counts"].astype(np.uint32)
    B = 2 * world + 1                                   # ragged blocks
    theta_b, iters_b = core.bootstrap_dist(comm, w, 42, int(w.sum()), B, alpha0)
    one, _ = core.bootstrap_dist(comm, w, 42, int(w.sum()), 1, alpha0)   # fewer replicates than ranks
    # (b) ONE solve with the ECs sharded over the GPUs
    b = shard_ecs(p["rowptr"], world)
    blk = csr_block(p, b[rank], b[rank + 1])
    lik = from_grouped_counts(core, blk["rowptr"], blk["grp"], blk["cnt"], blk["ec_counts"], p["group_sizes"])
    core.set_comm(comm)
    res = core.solve(lik.log_counts(), alpha0, tol=1e-6, max_iters=20000)
    core.set_comm(None)
    np.savez(os.path.join(d, f"out_{rank}.npz"), theta_b=theta_b, iters_b=iters_b, one=one, theta=res["theta"],
             iters=res["iters"], bound=res["bound"])
    core.close()
    comm.close()


if __name__ == "__main__":
    main()
