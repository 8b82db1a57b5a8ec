#!/usr/bin/env python3
"""bench.py -- headline benchmark of the abundance-estimation hot path (BASELINE.json).

Workload (config.workload "cfg3"): synthetic 10M reads x 5k groups collapsed to a CSR-of-ECs
likelihood (seeded generator msweep_amd/synth.py, seed 2), resident in HBM before the timed
region.  A "step" = ONE RCG iteration of the hot path over the whole likelihood: pass A
(natural-gradient norm sweep) + pass B (per-EC softmax, column sums N_g, ELBO sweep) + the O(G)
digamma / Fletcher-Reeves / bound kernels.  `value` = cells of the EC x group likelihood matrix
the reference would hold (E * G) processed per second, summed over all ranks.

N > 1 (one process per GPU): the path shards over bootstrap replicates (src/mSWEEP.cpp:496-518,
independent solves on the same likelihood): rank r runs K iterations on replicate r's resampled EC
counts -- no data-path collective; the per-replicate abundances are exchanged with one RCCL
all-gather at the end (msw_comm_allgather, inside the timed region).  Scaling is "weak" (per-GPU
work fixed).  Launched either by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in
the environment) or plainly as `python bench.py --gpus N`: the parent then starts the N rank
processes itself BEFORE anything touches the GPU (it never initialises HIP), relays rank 0's line
and exits non-zero if any rank fails.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

OUT = sys.stdout
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
TRAFFIC_JSON = "r02_traffic_pmc.json"  # HBM bytes per launch of the sweeps (rocprofv3 PMC, committed)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--groups", type=int, default=5000)
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the EM and time-to-convergence runs after the timed region (profiling runs)")
    ap.add_argument("--cpu-sample-ecs", type=int, default=500_000)
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--launch-selftest", action="store_true",
                    help="(tests) exercise the rank launcher and the rendezvous on CPU/gloo only: no GPU, no workload")
    ap.add_argument("--mode", choices=["replicates", "shard"], default="replicates",
                    help="N > 1: 'replicates' = one bootstrap replicate per GPU (weak scaling, default); "
                         "'shard' = ONE solve with the ECs sharded over the GPUs, RCCL all-reduce of the "
                         "column sums every iteration (strong scaling)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed / RCCL even with one rank (exercises the N > 1 code path)")
    return ap.parse_args()


def cpu_share():
    """CPUs this process may really use: the cgroup quota when there is one (a GPU box exposes all its
    logical CPUs but grants a share), else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def cpu_info():
    info = {"logical_cpus": os.cpu_count(), "usable_cpus": cpu_share()}
    try:
        txt = open("/proc/cpuinfo").read()
        models = [ln.split(":", 1)[1].strip() for ln in txt.splitlines() if ln.startswith("model name")]
        phys = {ln.split(":", 1)[1].strip() for ln in txt.splitlines() if ln.startswith("physical id")}
        cores = [ln.split(":", 1)[1].strip() for ln in txt.splitlines() if ln.startswith("cpu cores")]
        info["model"] = models[0] if models else None
        info["sockets_x_cores"] = f"{max(len(phys), 1)} x {cores[0]}" if cores else None
    except OSError:
        pass
    return info


def cpu_baseline(prob, lut, n_ecs, iters):
    """Oracle leg (test infrastructure, the ONLY place bench.py touches oracle/): the dense-state
    RCG exactly as the reference structures it (rcgpar::rcg_optl_omp restated), timed on the host
    cores over the first `n_ecs` ECs of the same workload expanded to a dense G x E matrix."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    from oracle import Oracle
    O = Oracle()
    G = len(prob["group_sizes"])
    rp = prob["rowptr"].astype(np.int64)
    E = min(n_ecs, len(rp) - 1)
    nz = rp[E]
    L = np.full((G, E), np.log(0.01))
    rows = np.repeat(np.arange(E), np.diff(rp[:E + 1]))
    L[prob["grp"][:nz], rows] = lut[prob["grp"][:nz], prob["cnt"][:nz]]
    logc = np.log(prob["ec_counts"][:E].astype(float))
    cores = cpu_share()
    O.set_num_threads(cores)
    # per-iteration time = (run of `iters` + 1 iterations) - (run of 1 iteration): the one-off cost of
    # allocating and first-touching the three G x E state matrices is not an iteration
    t0 = time.perf_counter()
    O.rcg_optl_dense(L, logc, np.ones(G), tol=-1.0, max_iters=1)
    t1 = time.perf_counter()
    r = O.rcg_optl_dense(L, logc, np.ones(G), tol=-1.0, max_iters=iters + 1)
    t2 = time.perf_counter()
    dt = max((t2 - t1) - (t1 - t0), 1e-9)
    out = {"value": E * G * iters / dt, "unit": "cells/s", "cores": cores, "kind": "port",
           "sample": f"first {E} ECs of the cfg3 workload x {G} groups as a dense fp64 matrix "
                     f"({E * G * 8 / 1e9:.1f} GB; the reference's four G x E matrices), {iters} RCG iterations of "
                     f"the dense-state restatement of rcgpar::rcg_optl_omp on {cores} OpenMP threads, {dt:.1f} s "
                     f"(+ {t1 - t0:.1f} s for set-up and one iteration); iters/s on the sample = {iters / dt:.3f}",
           "cpu": cpu_info()}
    return out


def cpu_baseline_structured(prob, lut, iters):
    """Second CPU line (apples to apples): the oracle's structured CSR restatement -- the same
    O(nnz) algorithm the GPU runs -- single-threaded on the FULL workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    from oracle import Oracle
    O = Oracle()
    G = len(prob["group_sizes"])
    E = len(prob["rowptr"]) - 1
    lutidx = (prob["grp"].astype(np.uint32) * lut.shape[1] + prob["cnt"]).astype(np.uint32)
    logc = np.log(prob["ec_counts"].astype(float))
    cores = cpu_share()
    O.set_num_threads(cores)
    t0 = time.perf_counter()
    O.rcg_optl_csr(prob["rowptr"], prob["grp"], lutidx, lut, np.log(0.01), G, logc, np.ones(G), tol=-1.0, max_iters=1)
    t1 = time.perf_counter()
    r = O.rcg_optl_csr(prob["rowptr"], prob["grp"], lutidx, lut, np.log(0.01), G, logc, np.ones(G), tol=-1.0,
                       max_iters=iters + 1)
    t2 = time.perf_counter()
    dt = max((t2 - t1) - (t1 - t0), 1e-9)
    return {"value": float(E) * G * iters / dt, "unit": "cells/s", "cores": cores, "kind": "port",
            "listed_cells_per_sec": float(len(prob["grp"])) * iters / dt,
            "sample": f"full cfg3 workload, {iters} iterations of the structured CSR restatement "
                      f"(oracle/rcg_oracle.cpp orc_rcg_optl_csr: the O(nnz) algorithm the GPU runs) on {cores} "
                      f"OpenMP threads, {dt:.1f} s; iters/s = {iters / dt:.3f}"}


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher: start the N rank processes (fresh interpreters,
    one GPU each) from a parent that never touches the GPU, relay rank 0's JSON line, and fail if any
    rank fails."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # wait for all; a rank that fails takes the others down with it (they would wait in the rendezvous or in a
    # collective until a timeout): only the processes started here are signalled, by their own handles
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = (r, p.returncode)
        time.sleep(0.2)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
    reader.join(timeout=10)
    rcs = [p.wait() for p in procs]
    sys.stdout.write(b"".join(c for c in chunks if c).decode())
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        print(f"bench.py: ranks failed (rank, exit code): {bad}" + (f"; first to fail: rank {failed[0]}" if failed else ""),
              file=sys.stderr)
        sys.exit(1)
    sys.exit(0)


def launch_selftest(a, rank, world):
    """CPU-only rehearsal of the launcher + rendezvous (tests/test_parallel_cpu.py): gloo, no GPU."""
    import torch
    import torch.distributed as dist
    if os.environ.get("MSWEEP_SELFTEST_FAIL_RANK") == str(rank):   # (tests) a rank that dies before the rendezvous
        sys.exit(3)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)])
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"selftest": True, "n_gpus": world, "gathered": [float(x.item()) for x in out]}), flush=True)
    dist.destroy_process_group()


def main():
    a = parse()
    if a.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        launch_ranks(a)          # never returns
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:          # before anything touches the GPU
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch {a.gpus} ranks "
                 f"(plain `python bench.py --gpus {a.gpus}` starts them itself)")
    if a.launch_selftest:
        return launch_selftest(a, rank, world)
    # ONE JSON line on stdout: libraries that print banners there (RCCL's version block at its first
    # initialisation) are sent to stderr; the line itself goes to the saved descriptor
    global OUT
    sys.stdout.flush()
    OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    dist = None
    if world > 1 or a.force_dist:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    n_gpus = world

    import numpy as np
    from msweep_amd import synth
    from msweep_amd.core import Core
    from msweep_amd.likelihood import from_grouped_counts, precalc_lls

    t0 = time.time()
    prob = synth.make_csr_problem(a.reads, a.groups, seed=a.seed)
    G = a.groups
    E = len(prob["rowptr"]) - 1
    nnz = len(prob["grp"])
    t_gen = time.time() - t0
    log(f"generated cfg3: E={E} nnz={nnz} in {t_gen:.1f}s")

    core = Core(local_rank)
    shard = dist is not None and a.mode == "shard"
    comm = None
    rccl_ranks = None
    if dist is not None:
        # the library's own RCCL communicator (C ABI: msw_comm_create_rccl); torch.distributed only
        # carries the unique id and the barriers around the timed region
        from msweep_amd.core import Comm
        uid = [Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        comm = Comm.rccl(uid[0], rank, world, local_rank)
        rccl_ranks = comm.rccl_count()     # ncclCommCount
        if rccl_ranks != world:
            sys.exit(f"bench.py: RCCL communicator spans {rccl_ranks} ranks, expected {world}")
    if shard:
        from msweep_amd.parallel import csr_block, shard_ecs
        b = shard_ecs(prob["rowptr"], world)
        blk = csr_block(prob, b[rank], b[rank + 1])
        lik = from_grouped_counts(core, blk["rowptr"], blk["grp"], blk["cnt"], blk["ec_counts"], prob["group_sizes"])
        core.set_comm(comm)
    else:
        lik = from_grouped_counts(core, prob["rowptr"], prob["grp"], prob["cnt"], prob["ec_counts"],
                                  prob["group_sizes"])
    alpha0 = np.ones(G)
    if shard:
        logc = lik.log_counts()
    elif dist is not None:
        # replicate `rank` of the bootstrap, drawn on the device from the reference's ONE sequential
        # mt19937_64(--seed 42) stream (src/BootstrapSample.cpp:60-73): rank r owns draws
        # [r * n_reads, (r + 1) * n_reads), exactly what a single-GPU run would give replicate r
        w = prob["ec_counts"].astype(np.uint32)
        counts = core.resample_counts(w, 42, int(w.sum()), rank, rank + 1)[0].astype(np.float64)
        with np.errstate(divide="ignore"):
            logc = np.log(counts)
    else:
        logc = lik.log_counts()
    log("likelihood resident")
    # SURVEY 8(d), second figure, measured FIRST: time to convergence at the reference's defaults
    # (--tol 1e-6, --max-iters 5000), host inputs handed over per call as at the reference's boundary
    # (PCIe-inclusive); twice, both reported.  Besides being a figure of its own this is ~80 ms of the same
    # sweeps: a cold MI355X reaches its working clocks only after 60-80 ms of activity (tools/ramp_check.py:
    # the SAME twenty iterations take 0.214 ms each on a cold chip, 0.181 ms once it has been busy for 40 ms),
    # and the timed region below is to measure the path, not the governor.
    conv = None
    if not shard and not a.no_extras:
        runs = []
        for _ in range(2 + int(os.environ.get("MSWEEP_BENCH_PRESOLVES", "0"))):  # developer switch: more of them
            t1 = time.perf_counter()
            rc = core.solve(logc, alpha0, tol=1e-6, max_iters=5000)
            runs.append(((time.perf_counter() - t1) * 1e3, core.last_timing()["solve_ms"]))
        conv = {"iters": int(rc["iters"]), "ms": min(r[0] for r in runs), "device_ms": min(r[1] for r in runs),
                "runs_ms": [round(r[0], 3) for r in runs], "tol": 1e-6,
                "includes": "upload of log counts and prior, download of theta"}
    core.set_fixed_iters(True)
    # the EM optimiser (--algorithm emgpu: one pass-B sweep + one O(G) kernel per iteration), W + K steps on the
    # same resident inputs; reported beside the headline, which is the reference's default RCG
    em = None
    if not shard and not a.no_extras:
        from msweep_amd.core import ALGO_EM
        core.run(max_iters=max(a.warmup, 1), algo=ALGO_EM)
        t1 = time.perf_counter()
        core.run(max_iters=a.steps, algo=ALGO_EM)
        t_em = time.perf_counter() - t1
        em = {"ms_per_step": t_em * 1e3 / a.steps, "iters_per_sec": a.steps / t_em}
    if conv is None:
        core.prepare(logc, alpha0)           # inputs resident in HBM before the timed region
    # (after the convergence run they already are: msw_core_solve = msw_core_prepare + msw_core_run)
    # W untimed warm-up steps: the first W iterations of the solve (with its set-up: the evaluation of the
    # initial state and the first iteration's rejected step); the K timed steps are the NEXT K iterations of
    # the same solve (msw_core_continue) -- every kernel slot they need, rejected steps included
    core.run(max_iters=max(a.warmup, 1))
    if dist is not None and not shard:
        comm.allgather(np.zeros(G))           # RCCL sets its connections up on the first collective: not timed

    def sync():
        # barrier + device synchronisation on both sides of the timed region.  core.run() itself
        # returns only after its HIP stream has drained (it downloads theta), so with one rank there
        # is nothing left to wait for and torch is not even imported.
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    sync()
    t0 = time.perf_counter()
    res = core.continue_(a.steps)             # exactly K steps
    if dist is not None and not shard:
        gathered = comm.allgather(res["theta"])   # (world, G): the per-replicate abundances, RCCL all-gather
        assert gathered.shape == (world, G)
    sync()
    dt = time.perf_counter() - t0
    tm0 = core.last_timing()
    log(f"timed {a.steps} steps in {dt:.3f}s")
    assert tm0["iters"] == a.steps, (tm0["iters"], a.steps)
    # K more steps with HIP events around every sweep launch (on the solve stream) for the
    # per-kernel durations of the roofline object.  Kept out of the timed run: every event record
    # is a barrier packet that costs ~6 us of idle GPU between two kernels.
    core.set_profiling(True)
    core.run(max_iters=max(a.warmup, 1))      # the same iterations as the timed ones: W, then K
    core.continue_(a.steps)
    tm = core.last_timing()
    core.set_profiling(False)
    if dist is not None:
        import torch
        tt = torch.tensor([dt], dtype=torch.float64).cuda()
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # the practical ceiling beside the specification: the rates this very device reaches with a
    # read-only 16-byte-load sweep and with the triad -- on a buffer the size of the sweeps' record stream (what
    # a kernel that does nothing but read that stream once would get, launch ramp and tail included) and on 4 GiB
    stream_read = stream_triad = stream_read_4g = stream_triad_4g = None
    if rank == 0:
        stream_read, stream_triad = core.hbm_stream_rates(max(int(tm["bytes_passB"]), 1 << 20), 5)
        stream_read_4g, stream_triad_4g = core.hbm_stream_rates(1 << 32, 3)
    if rank == 0:
        cells = float(E) * G * a.steps * (1 if shard else n_gpus)
        msA = tm["passA_ms"] / max(tm["passA_launches"], 1)
        msB = tm["passB_ms"] / max(tm["passB_launches"], 1)
        dom, ms_dom, b_dom = ("k_passB", msB, tm["bytes_passB"]) if msB >= msA else ("k_passA", msA, tm["bytes_passA"])
        achieved = b_dom / (ms_dom * 1e-3) / 1e9 if ms_dom > 0 else 0.0
        # HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE),
        # collected offline on this exact workload and committed under profiles/
        traffic, traffic_src = None, None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", TRAFFIC_JSON)))
            if (a.reads, G, a.seed) == (10_000_000, 5000, 2):  # the profiled workload
                traffic = next(v["hbm_bytes_per_launch"] for k, v in tj.items() if dom in k)
                traffic_src = f"profiles/{TRAFFIC_JSON} (offline rocprofv3 --pmc passes on this workload, not this run)"
        except Exception:
            traffic = None
        line = {
            "metric": "EM iters/sec + reads×groups cells/sec, 10M reads × 5k groups",
            "value": cells / dt, "unit": "cells/s",
            "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt * 1e3 / a.steps, "higher_is_better": True, "scaling": "strong" if shard else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "cfg3: synthetic 10M reads x 5k groups, CSR-of-ECs likelihood, RCG-VB "
                                   "(--algorithm rcggpu), fixed iteration count",
                       "algorithm": "rcg", "reads": a.reads, "groups": G, "ecs": E, "nnz": nnz, "seed": a.seed,
                       "sharding": "single solve" if n_gpus == 1 else (
                           f"one solve, ECs sharded over {n_gpus} GPUs, RCCL all-reduce of (G+4) fp64 per iteration"
                           if shard else f"bootstrap replicates, 1 per GPU x {n_gpus}: every rank solves on its "
                                         "replicate's RESAMPLED counts (a third of the ECs at zero, one in fifty at four or "
                                         "more), a different trajectory from the original counts the single-GPU line "
                                         "solves on")},
            "iters_per_sec": a.steps * (1 if shard else n_gpus) / dt,
            "listed_cells_per_sec": float(nnz) * a.steps * (1 if shard else n_gpus) / dt,
            "value_counts": "logical EC x group cells of the matrix the reference holds (E * G per iteration); "
                            "listed_cells_per_sec counts the cells the CSR-of-ECs form stores (nnz per iteration)",
            "rccl_ranks": rccl_ranks,
            "reads_x_groups_cells_per_sec": float(a.reads) * G * a.steps * (1 if shard else n_gpus) / dt,
            "device_ms_per_step": tm0["solve_ms"] / a.steps,
            "kernels": {"k_passA_ms": msA, "k_passB_ms": msB, "passA_launches": tm["passA_launches"],
                        "passB_launches": tm["passB_launches"]},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": b_dom, "avg_launch_ms": ms_dom,
                         "measured_stream_GBs": {"read_only_same_size": stream_read, "triad_same_size": stream_triad,
                                                 "read_only_4GiB": stream_read_4g, "triad_4GiB": stream_triad_4g,
                                                 "what": "msw_core_hbm_stream_rates on this device, best of 7 shapes x "
                                                         "5 launches; same_size = a buffer of algorithmic_bytes_per_launch"},
                         "frac_of_measured_read": achieved / stream_read if stream_read else None,
                         "kernels": {"k_passA": {"achieved": tm["bytes_passA"] / (msA * 1e-3) / 1e9 if msA > 0 else 0.0,
                                                 "frac": tm["bytes_passA"] / (msA * 1e-3) / 1e9 / HBM_PEAK_GBS if msA > 0 else 0.0},
                                     "k_passB": {"achieved": tm["bytes_passB"] / (msB * 1e-3) / 1e9 if msB > 0 else 0.0,
                                                 "frac": tm["bytes_passB"] / (msB * 1e-3) / 1e9 / HBM_PEAK_GBS if msB > 0 else 0.0}}},
            "setup_s": {"generate": t_gen},
        }
        if conv is not None:
            line["time_to_convergence"] = conv
        if em is not None:
            line["em_algorithm"] = em
        if not a.no_cpu_baseline and n_gpus == 1:   # rank 0 at N = 1 only: the host cores are shared by the ranks
            log("cpu baseline ...")
            try:
                line["cpu_baseline"] = cpu_baseline(prob, precalc_lls(prob["group_sizes"]), a.cpu_sample_ecs,
                                                    a.cpu_iters)
            except Exception as ex:  # the baseline is reporting only
                line["cpu_baseline"] = {"value": None, "unit": "cells/s", "cores": 0, "kind": "port",
                                        "sample": f"failed: {ex}"}
            try:
                line["cpu_baseline_structured"] = cpu_baseline_structured(prob, precalc_lls(prob["group_sizes"]), 5)
            except Exception as ex:
                line["cpu_baseline_structured"] = {"value": None, "unit": "cells/s", "cores": 0, "kind": "port",
                                                   "sample": f"failed: {ex}"}
        print(json.dumps(line), file=OUT, flush=True)
    core.set_comm(None) if shard else None
    core.close()
    if dist is not None:
        dist.barrier()
        comm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
