"""GPU, needs >= 2 devices (skips cleanly on a one-GPU box): the two multi-GPU modes over REAL RCCL ranks --
one fresh process per GPU (tests/_rank_worker.py), the library's own communicator (msw_comm_create_rccl),
checked against the single-GPU answers computed here.
  * msw_core_bootstrap_dist (src/mSWEEP.cpp:496-518 over the GPUs): the gathered B x G table is the same on
    every rank and BIT-IDENTICAL to the single-GPU run (every replicate is a whole solve on one GPU; only the
    placement changes), ragged blocks and fewer replicates than ranks included;
  * the EC-sharded solve: every rank returns the same bits; against the single-GPU solve the iteration count
    and theta at the north-star tolerance."""
import os
import subprocess
import sys

import numpy as np
import pytest

from msweep_amd import synth
from msweep_amd.likelihood import from_grouped_counts
from test_gpu_rcg import assert_theta

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def n_devices():
    # through the HIP runtime this process ALREADY uses (the library is loaded RTLD_GLOBAL, its dependencies with it; the
    # `gpu_core` fixture has initialised it).  Not through torch.cuda -- the torch wheel brings its own copy of the ROCm
    # runtime: a process that first used the system's and then torch's aborts at exit ("double free or corruption" after
    # pytest's summary, exit code 134) -- and not by loading /opt/rocm's libamdhip64.so by path: where torch came first
    # (pytest imports every test module when it collects), that one does not resolve against torch's older HSA runtime
    import ctypes
    from msweep_amd.core import load_library
    load_library()
    n = ctypes.c_int(0)
    rc = ctypes.CDLL(None).hipGetDeviceCount(ctypes.byref(n))
    return n.value if rc == 0 else 0


def _run_ranks(tmp_path, world, env, shm=False):
    d = str(tmp_path)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_rank_worker.py"), str(r), str(world), d]
                              + (["shm"] if shm else []), env=env) for r in range(world)]
    try:
        rcs = [p.wait(timeout=600) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert rcs == [0] * world, rcs
    return [np.load(os.path.join(d, f"out_{r}.npz")) for r in range(world)]


def _check_against_single_gpu(gpu_core, out, world):
    p = synth.make_csr_problem(60000, 200, seed=41, max_other=8)
    G, B = 200, 2 * world + 1
    lik = from_grouped_counts(gpu_core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
    w = p["ec_counts"].astype(np.uint32)
    single_b, iters_b = gpu_core.bootstrap(w, 42, int(w.sum()), 0, B, np.ones(G))
    single = gpu_core.solve(lik.log_counts(), np.ones(G), tol=1e-6, max_iters=20000)
    for r in range(world):
        np.testing.assert_array_equal(out[r]["theta_b"], single_b)
        np.testing.assert_array_equal(out[r]["iters_b"], iters_b)
        np.testing.assert_array_equal(out[r]["one"], single_b[:1])
        np.testing.assert_array_equal(out[r]["theta"], out[0]["theta"])
        assert int(out[r]["iters"]) == int(out[0]["iters"]) and float(out[r]["bound"]) == float(out[0]["bound"])
    assert int(out[0]["iters"]) == single["iters"]
    assert_theta(out[0]["theta"], single["theta"])


# transport of the sharded solve's two all-reduces per iteration: ncclAllReduce (the default), or the one-shot kernel
# that writes into the peers' inboxes over xGMI (hipIpc-mapped; msweep_amd/csrc/peer_comm.hpp) -- same bits either way
@pytest.mark.parametrize("transport", ["rccl", "peer"])
@pytest.mark.parametrize("world", [2, 4, 8])   # 8: the size of the machine the driver runs the scaling curve on
def test_rccl_ranks_bootstrap_and_sharded_solve(gpu_core, tmp_path, world, transport):
    if n_devices() < world:
        pytest.skip(f"needs {world} GPUs, {n_devices()} visible")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MSWEEP_ALLREDUCE=transport)
    _check_against_single_gpu(gpu_core, _run_ranks(tmp_path, world, env), world)


@pytest.mark.parametrize("transport", ["rccl", "peer"])   # ("rccl" = the set-up communicator's own, host-staged here)
def test_process_ranks_on_one_gpu_bootstrap_and_sharded_solve(gpu_core, tmp_path, transport):
    """The same two modes with one PROCESS per rank on ONE GPU (runs on the one-GPU box): the ranks meet in a
    shared-memory segment (msw_comm_create_shm) because RCCL refuses two ranks on one device.  What this covers that the
    thread-ranks do not: a process per rank with its own HIP context, the block partition of the replicate stream by
    rank, the gather of the table across processes, the hipIpc-mapped inboxes of the peer transport."""
    world = 3
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MSWEEP_ALLREDUCE=transport, GPU_MAX_HW_QUEUES="8")
    _check_against_single_gpu(gpu_core, _run_ranks(tmp_path, world, env, shm=True), world)


@pytest.mark.parametrize("mode,transport", [("replicates", "rccl"), ("shard", "peer")])
def test_bench_with_two_ranks_on_one_gpu(mode, transport):
    """bench.py's N > 1 code (its own launcher, the per-rank replicate, the barriers and the max over ranks around the
    timed region, the sharded workload, the one JSON line from rank 0) rehearsed with two rank processes on ONE GPU:
    MSWEEP_BENCH_ONE_GPU=1 -- torch.distributed over gloo, the library's ranks in a shared-memory segment.  A functional
    check, not a measurement (the line says so in `rccl_ranks`)."""
    import json
    env = dict(os.environ, MSWEEP_BENCH_ONE_GPU="1", MSWEEP_ALLREDUCE=transport, GPU_MAX_HW_QUEUES="8",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline",
           "--reads", "300000", "--groups", "300", "--mode", mode]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["value"] > 0 and "rehearsal" in d["rccl_ranks"]
    if mode == "shard":
        assert d["shard_split_ms_per_step"]["allreduce"] == "peer" and d["shard_split_ms_per_step"]["collectives_per_step"] >= 2
    else:
        assert d["bootstrap_cfg4"]["replicates"] == 2 * d["bootstrap_cfg4"]["per_rank"]


def test_bench_under_torchrun_with_two_ranks_on_one_gpu():
    """The way the driver launches N > 1 -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ...` -- rehearsed with two ranks on ONE GPU (MSWEEP_BENCH_ONE_GPU=1: a
    functional check, not a measurement).  The rank processes read RANK / WORLD_SIZE / MASTER_PORT from the environment,
    meet through their own unix socket (named after their common parent, the launcher's agent) and never import torch
    (bench.py asserts so at the end of every rank): round 4's two-ROCm-runtimes hazard is out of the path."""
    import json
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MSWEEP_BENCH_ONE_GPU="1", GPU_MAX_HW_QUEUES="8", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--no-cpu-baseline", "--reads", "300000", "--groups", "300"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=500, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["value"] > 0 and "rehearsal" in d["rccl_ranks"]
    assert len(d["ms_per_step_runs"]) == 5 and d["bootstrap_cfg4"]["replicates"] == 2 * d["bootstrap_cfg4"]["per_rank"]
