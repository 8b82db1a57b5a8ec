"""CPU: the N > 1 path (replicate sharding + all-gather) with world_size 2 on the gloo backend."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from msweep_amd.parallel import all_gather_rows, bootstrap_sharded, replicate_slice


def test_replicate_slices_partition():
    for n in (0, 1, 7, 8, 1000):
        for w in (1, 2, 3, 8):
            sl = [replicate_slice(n, r, w) for r in range(w)]
            assert sl[0][0] == 0 and sl[-1][1] == n
            assert all(sl[i][1] == sl[i + 1][0] for i in range(w - 1))
            assert max(e - b for b, e in sl) - min(e - b for b, e in sl) <= 1


def _fake_solver(begin, end, G=5):
    # deterministic function of the replicate index, stands in for the per-replicate solve
    return np.array([[b * 10.0 + g for g in range(G)] for b in range(begin, end)]).reshape(end - begin, G)


def _worker(rank, world, port, n_rep, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = bootstrap_sharded(_fake_solver, n_rep, 5, dist)
    q.put((rank, full))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_rep", [7, 2, 1])
def test_world2_gloo_matches_single_process(n_rep):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_rep, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single = bootstrap_sharded(_fake_solver, n_rep, 5, None)
    np.testing.assert_array_equal(res[0], single)
    np.testing.assert_array_equal(res[1], single)
    np.testing.assert_array_equal(all_gather_rows(single, n_rep), single)
