#!/usr/bin/env python3
"""bench.py -- headline benchmark of the abundance-estimation hot path (BASELINE.json).

Default workload (config.workload "cfg3", the configuration BASELINE.json's metric is quoted on): synthetic
10M reads x 5k groups collapsed to a CSR-of-ECs likelihood (seeded generator msweep_amd/synth.py, seed 2),
resident in HBM before the timed region.  A "step" = ONE RCG iteration of the hot path over the whole
likelihood: pass A (natural-gradient norm sweep) + pass B (per-EC softmax, column sums N_g, ELBO sweep) + the
O(G) digamma / Fletcher-Reeves / bound kernels.  `value` = cells of the EC x group likelihood matrix the
reference would hold (E * G) processed per second, summed over all ranks.

--config selects the other single-GPU configurations of BASELINE.json, same JSON object:
  cfg2  synthetic 1M ECs x 500 groups DENSE likelihood through rcg_optl's own dense boundary
        (msw_core_set_dense_logl; the library re-expresses it as CSR-of-ECs on the device);
  cfg5  sparse 50M reads x 20k groups through msw_core_build_likelihood with --min-hits 1 (build timed
        separately), one GPU -- or, with --gpus N --mode shard, BASELINE's "8 x MI355X": every rank expands and
        builds its own block of ECs (the --min-hits counts all-reduced inside the build) and the ONE solve
        all-reduces its column sums every iteration;
  cfg4  cfg3 + bootstrap: a "step" is ONE bootstrap replicate (src/mSWEEP.cpp:496-518) through
        msw_core_bootstrap_dist -- K replicates per rank: stream seek / GF(2) jump-ahead, resampling, solve to
        --tol 1e-6, and the all-gather of the abundances inside the timed region.

N > 1 (one process per GPU): the path shards over bootstrap replicates (independent solves on the same
likelihood): with cfg3 rank r runs K iterations on replicate r's resampled EC counts -- no data-path collective;
the per-replicate abundances are exchanged with one RCCL all-gather at the end (msw_comm_allgather, inside the
timed region).  Every cfg3 line (any N) also carries `bootstrap_cfg4`: a short run of the REAL replicate loop
(msw_core_bootstrap_dist, fixed replicates per rank).  Scaling is "weak" (per-GPU work fixed).  Launched either by
torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) or plainly as
`python bench.py --gpus N`: the parent then starts the N rank processes itself BEFORE anything touches the GPU
(it never initialises HIP), relays rank 0's line and exits non-zero if any rank fails.  Either way a rank process
never imports torch (round 5): the barriers around the timed region, the max over ranks and the all-gather of the
abundances are collectives of the library's own RCCL communicator, whose unique id travels from rank 0 to the others
through a unix socket of their own (class Star).

At N = 1 the cfg3 line also carries `text_to_abundances` (--no-text skips it): the same reads written as two Themisto
plaintext strands and taken from text to abundances.txt -- the reader on the device (msw_alignment_read_device), the
likelihood build from its device-resident classes, the solve to --tol 1e-6 -- with the host reader on the same files
beside it; `--config e2e` is the long form of that leg (Python mirror and gzip pair included).

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
METRIC = "EM iters/sec + reads×groups cells/sec, 10M reads × 5k groups"
# HBM bytes per launch of the sweeps (rocprofv3 PMC passes, committed): newest file that covers the workload
TRAFFIC_JSON = {"cfg3": ["r05_traffic_pmc.json", "r04_traffic_pmc.json", "r03_traffic_pmc.json"],
                "cfg2": ["r05_cfg2_traffic_pmc.json", "r04_cfg2_traffic_pmc.json", "r03_cfg2_traffic_pmc.json"],
                "cfg5": ["r05_cfg5_traffic_pmc.json", "r04_cfg5_traffic_pmc.json", "r03_cfg5_traffic_pmc.json"],
                "cfg4": ["r05_traffic_pmc.json", "r04_traffic_pmc.json", "r03_traffic_pmc.json"],
                "cfg2-dense": ["r05_cfg2_dense_traffic_pmc.json"],       # MSWEEP_DENSE_COMPRESS=0: k_dense_passA / B
                "cfg3-diverse": ["r05_diverse_traffic_pmc.json", "r04_diverse_traffic_pmc.json", "r03_diverse_traffic_pmc.json"],
                "cfg4-diverse": ["r05_diverse_traffic_pmc.json", "r04_diverse_traffic_pmc.json", "r03_diverse_traffic_pmc.json"]}
N_CU, N_SE, SIMD_PER_CU = 256, 32, 4   # MI355X: 8 XCDs x 32 CUs, 4 shader engines per XCD (the SQ_BUSY_CYCLES instances)


def pmc_bound(entry):
    """Which resource the kernel keeps busy, from its committed rocprofv3 --pmc averages per launch (tools/pmc_traffic.py):
    kernel cycles = SQ_BUSY_CYCLES / 32 (one instance per shader engine); LDS busy = SQ_LDS_IDX_ACTIVE (summed over the
    CUs) / (256 x kernel cycles); VALU busy = 4 x SQ_ACTIVE_INST_VALU (quad-cycles, summed over the SIMDs) /
    (1024 x kernel cycles); conflicts = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE."""
    try:
        cyc = entry["SQ_BUSY_CYCLES"] / N_SE
        return {"lds_busy_frac": entry["SQ_LDS_IDX_ACTIVE"] / (N_CU * cyc),
                "lds_conflict_frac": entry["SQ_LDS_BANK_CONFLICT"] / max(entry["SQ_LDS_IDX_ACTIVE"], 1.0),
                "valu_busy_frac": 4.0 * entry["SQ_ACTIVE_INST_VALU"] / (N_CU * SIMD_PER_CU * cyc),
                "kernel_cycles": cyc}
    except (KeyError, ZeroDivisionError, TypeError):
        return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 100; cfg4: 4 replicates per rank)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default 20; cfg4: 1)")
    ap.add_argument("--config", choices=["cfg2", "cfg3", "cfg4", "cfg5", "e2e"], default="cfg3",
                    help="BASELINE.json configuration (default cfg3, the one the metric is quoted on); e2e: cfg3's reads "
                         "from Themisto plaintext files to abundances.txt (SURVEY 8f-1: reader + collapse, device build, "
                         "solve, writer -- its own JSON object)")
    ap.add_argument("--reads", type=int, default=None, help="cfg3/4/5: reads (default 10M; cfg5 50M); cfg2: ECs (1M)")
    ap.add_argument("--groups", type=int, default=None, help="default 5000 (cfg2: 500, cfg5: 20000)")
    ap.add_argument("--seed", type=int, default=None, help="generator seed (default: 2; cfg2: 1; cfg5: 3)")
    ap.add_argument("--group-sizes", choices=["poisson", "diverse"], default="poisson",
                    help="cfg3/cfg4 generator: group sizes 1 + Poisson(9) (SURVEY 8d, default) or log-normal up to 400 as "
                         "real groupings have them (thousands of used table slots: the hybrid slot area)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-text", action="store_true",
                    help="cfg3: skip the text_to_abundances leg (two Themisto strands written and read back: ~40 s)")
    ap.add_argument("--no-extras", "--no-prewarm", dest="no_extras", action="store_true",
                    help="skip the convergence solves and the EM leg that run BEFORE the timed region (and warm "
                         "the clocks: `prewarm` in the line) and the bootstrap leg after it")
    ap.add_argument("--cpu-sample-ecs", type=int, default=500_000)
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--bootstrap-per-rank", type=int, default=4,
                    help="cfg3: replicates per rank of the bootstrap_cfg4 leg (0 = skip)")
    ap.add_argument("--cpu-legs-only", action="store_true",
                    help="(internal) the CPU baseline legs of --config cfg3 / cfg2 in a process of their own: bench.py "
                         "starts it with OMP_PROC_BIND=close and OMP_PLACES set (BASELINE.md 2); no GPU is touched")
    ap.add_argument("--cpu-probe", action="store_true", help="(internal, with --cpu-legs-only) a two-second OpenMP probe")
    ap.add_argument("--launch-selftest", action="store_true",
                    help="(tests) exercise the rank launcher and the rendezvous on CPU/gloo only: no GPU, no workload")
    ap.add_argument("--mode", choices=["replicates", "shard"], default="replicates",
                    help="N > 1: 'replicates' = one bootstrap replicate per GPU (weak scaling, default); "
                         "'shard' = ONE solve with the ECs sharded over the GPUs, RCCL all-reduce of the "
                         "column sums every iteration (strong scaling)")
    ap.add_argument("--force-dist", action="store_true",
                    help="create the RCCL communicator even with one rank (exercises the N > 1 code path)")
    a = ap.parse_args()
    dflt = {"cfg2": (1_000_000, 500, 1), "cfg3": (10_000_000, 5000, 2), "cfg4": (10_000_000, 5000, 2),
            "cfg5": (50_000_000, 20_000, 3), "e2e": (10_000_000, 5000, 2)}[a.config]
    a.default_shape = (a.reads, a.groups, a.seed) == (None, None, None)   # (of its --config and --group-sizes)
    a.reads = dflt[0] if a.reads is None else a.reads
    a.groups = dflt[1] if a.groups is None else a.groups
    a.seed = dflt[2] if a.seed is None else a.seed
    if a.steps is None:
        a.steps = 4 if a.config == "cfg4" else 100
    if a.warmup is None:
        a.warmup = 1 if a.config == "cfg4" else 20
    return a


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
    info = {"logical_cpus": os.cpu_count(), "usable_cpus": cpu_share(), "OMP_PROC_BIND": os.environ.get("OMP_PROC_BIND", "unset"),
            "OMP_PLACES": os.environ.get("OMP_PLACES", "unset"), "OMP_WAIT_POLICY": os.environ.get("OMP_WAIT_POLICY", "unset")}
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


# ---- OMP_PROC_BIND=close OMP_PLACES=cores for the CPU legs ---------------------------------------
# A GPU box shows ALL hardware threads of a host its other tenants use too and grants a CPU share of them (cgroup
# quota).  `OMP_PLACES=cores` there means the first cores of the host, whoever is busy on them: round 4 tried it in
# process and the baseline stalled for minutes (spinning barriers burn the quota while a bound thread waits for a
# busy core).  So: the legs run in a CHILD process -- libgomp reads the binding when it loads, and a stall can be
# ended -- bound `close` to explicit places = the least busy physical cores of the last 0.4 s, one hardware thread
# each, with passive waits -- AFTER a two-second probe (a small dense-state run in a child of its own) has shown that
# the binding works here at all: on the GPU boxes of round 5 it does not (the bound probe never finishes: some of the
# host's cores are not ours to run on, whatever the affinity mask says), and the legs then run unbound at once, with the
# probe's outcome in the line.  If a bound child falls silent all the same it is ended and the legs run again unbound.
def pick_idle_cores(n):
    """n physical cores (one hardware thread each, ascending ids) with the least non-idle time over 0.4 s."""
    def snap():
        out = {}
        for ln in open("/proc/stat"):
            if ln.startswith("cpu") and ln[3].isdigit():
                f = ln.split()
                v = [int(x) for x in f[1:9]]
                out[int(f[0][3:])] = (sum(v), v[3] + v[4])
        return out
    allowed = sorted(os.sched_getaffinity(0))
    a = snap()
    time.sleep(0.4)
    b = snap()
    busy = {c: 1.0 - (b[c][1] - a[c][1]) / max(b[c][0] - a[c][0], 1) for c in allowed if c in a and c in b}
    cores = {}
    for c in allowed:
        try:
            sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
        except OSError:
            sib = str(c)
        cores.setdefault(sib, []).append(c)
    ranked = sorted(cores.values(), key=lambda hw: (max(busy.get(c, 0.0) for c in hw), hw[0]))
    return sorted(hw[0] for hw in ranked[:n])


def cpu_legs_in_child(a):
    """The CPU legs of cfg3 / cfg2 in a child process bound as BASELINE.md 2 prescribes; None if that cannot be had."""
    import subprocess
    import threading
    n = cpu_share()
    try:
        places = pick_idle_cores(n)
    except Exception:  # (no /proc/stat, no topology files)
        places = []
    base = [sys.executable, os.path.abspath(__file__), "--cpu-legs-only", "--config", a.config, "--reads", str(a.reads),
            "--groups", str(a.groups), "--seed", str(a.seed), "--group-sizes", a.group_sizes,
            "--cpu-sample-ecs", str(a.cpu_sample_ecs), "--cpu-iters", str(a.cpu_iters)]
    tries = []
    probe_note = "no probe (no /proc/stat or topology files)"
    if len(places) == n:
        bound_env = dict(OMP_PROC_BIND="close", OMP_PLACES=",".join("{%d}" % c for c in places), OMP_WAIT_POLICY="passive",
                         OMP_NUM_THREADS=str(n))

        def probe(extra, limit):
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "OMP_PROC_BIND", "OMP_PLACES")}
            env.update(extra or {})
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-legs-only", "--cpu-probe"], env=env,
                                   capture_output=True, text=True, timeout=limit)
                return float(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else None
            except (subprocess.TimeoutExpired, ValueError, IndexError):
                return None
        t_free = probe(dict(OMP_WAIT_POLICY="passive", OMP_NUM_THREADS=str(n)), 120)   # (same wait policy: the binding alone differs)
        t_bound = probe(bound_env, max(20.0, 4.0 * (t_free or 5.0)))
        if t_free is not None and t_bound is not None and t_bound <= 1.5 * t_free:
            tries.append(bound_env)
            probe_note = f"probe: bound {t_bound:.2f} s, unbound {t_free:.2f} s"
        else:
            probe_note = (f"probe: bound {'did not finish' if t_bound is None else '%.2f s' % t_bound}, unbound "
                          f"{'did not finish' if t_free is None else '%.2f s' % t_free}: OMP_PROC_BIND=close is not usable on this host")
        log("cpu baseline: " + probe_note)
    tries.append(None)      # unbound, as rounds 1-4
    for env_extra in tries:
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "OMP_PROC_BIND", "OMP_PLACES")}
        env.update(env_extra or {})
        p = subprocess.Popen(base, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        last = [time.time()]

        def relay():
            for ln in p.stderr:
                last[0] = time.time()
                sys.stderr.write(ln)
                sys.stderr.flush()
        th = threading.Thread(target=relay, daemon=True)
        th.start()
        out = []
        rd = threading.Thread(target=lambda: out.append(p.stdout.read()), daemon=True)
        rd.start()
        t0 = time.time()
        stalled = False
        while p.poll() is None:
            time.sleep(0.5)
            # every leg logs when it starts: silence of 150 s, or 420 s in all, is a stall (a leg is ~25 s)
            if time.time() - last[0] > 150 or time.time() - t0 > 420:
                stalled = True
                p.kill()
                break
        p.wait()
        rd.join(timeout=10)
        if not stalled and p.returncode == 0 and out and out[0].strip():
            try:
                lines = json.loads(out[0].strip().splitlines()[-1])
            except ValueError:
                continue
            for v in lines.values():
                if isinstance(v, dict):
                    v["binding"] = ("OMP_PROC_BIND=close on the %d least busy physical cores (OMP_PLACES one hardware thread each), "
                                    "OMP_WAIT_POLICY=passive, child process (%s)" % (n, probe_note)) if env_extra else \
                                   "unbound, child process (" + probe_note + ")"
            return lines
        log("cpu baseline: the %s child %s; %s" % ("bound" if env_extra else "unbound", "stalled" if stalled else "failed",
                                                   "running the legs again unbound" if env_extra else "giving up"))
    return None


def cpu_legs_only(a):
    """Child mode (--cpu-legs-only): regenerate the workload on the host and run its CPU legs; ONE JSON object on stdout."""
    import numpy as np
    if a.cpu_probe:   # seconds of a small all-thread dense-state run under this process's OpenMP environment
        O, _ = _oracle()
        rng = np.random.default_rng(1)
        L = rng.normal(-3.0, 1.0, (200, 40000))
        t0 = time.perf_counter()
        O.rcg_optl_dense(L, np.zeros(40000), np.ones(200), tol=-1.0, max_iters=3)
        print(f"{time.perf_counter() - t0:.4f}", flush=True)
        return
    from msweep_amd import synth
    from msweep_amd.likelihood import precalc_lls
    t0 = time.time()
    if a.config == "cfg2":
        E, G = a.reads, a.groups
        p = synth.make_dense_problem(E, G, seed=a.seed)
        log(f"cpu legs (child): cfg2 regenerated in {time.time() - t0:.0f}s")
        out = cpu_cfg2_lines(a, p, E, G)
    else:
        G = a.groups
        prob = synth.make_csr_problem(a.reads, G, seed=a.seed,
                                      group_sizes=synth.diverse_group_sizes if a.group_sizes == "diverse" else None)
        log(f"cpu legs (child): {a.config} regenerated in {time.time() - t0:.0f}s")
        out = cpu_cfg3_lines(a, prob, G)
    print(json.dumps(out), flush=True)


def cpu_cfg3_lines(a, prob, G):
    import numpy as np
    from msweep_amd.likelihood import precalc_lls
    E = len(prob["rowptr"]) - 1
    lut = precalc_lls(prob["group_sizes"])
    n = min(a.cpu_sample_ecs, E)
    lutidx = (prob["grp"].astype(np.uint32) * lut.shape[1] + prob["cnt"]).astype(np.uint32)
    logc = np.log(prob["ec_counts"].astype(float))
    out = cpu_dense_lines(lambda m: dense_sample(prob["rowptr"], prob["grp"], prob["cnt"], lut, G, m), logc, n, a.cpu_iters,
                          lambda m: f"first {m} ECs of the cfg3 workload x {G} groups")
    out["cpu_baseline_structured"] = cpu_structured_csr(prob["rowptr"], prob["grp"], lutidx, lut, G, logc, 5,
                                                        "full cfg3 workload")
    return out


def cpu_cfg2_lines(a, p, E, G):
    import numpy as np
    out = cpu_dense_lines(lambda n: p["logl"] if n == E else np.ascontiguousarray(p["logl"][:, :n]), p["logc"], E,
                          a.cpu_iters, lambda n: "the full cfg2 workload" if n == E else f"first {n} ECs of the cfg2 workload")
    O, cores = _oracle()
    dt, _ = _timed_iters(lambda n: O.rcg_optl_dense_structured(p["logl"], p["logc"], np.ones(G), tol=-1.0,
                                                               max_iters=n), 5)
    out["cpu_baseline_structured"] = {
        "value": float(E) * G * 5 / dt, "unit": "cells/s", "cores": cores, "kind": "port",
        "sample": f"full cfg2 workload, 5 iterations of the structured restatement on the dense matrix "
                  f"(oracle/rcg_oracle.cpp: two read-only passes over L per iteration, no G x E state) on "
                  f"{cores} OpenMP threads, {dt:.1f} s; iters/s = {5 / dt:.3f}"}
    return out


# ---- CPU baselines (oracle legs: test infrastructure, the ONLY place bench.py touches oracle/) ------------------
def _oracle(threads=None, fastmath=False):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from oracle import Oracle
    O = Oracle(fastmath=fastmath)
    cores = cpu_share() if threads is None else threads
    O.set_num_threads(cores)
    return O, cores


def _timed_iters(fn, iters):
    """per-iteration time = (run of `iters` + 1 iterations) - (run of 1 iteration): the one-off cost of
    allocating and first-touching the state is not an iteration"""
    t0 = time.perf_counter()
    fn(1)
    t1 = time.perf_counter()
    fn(iters + 1)
    t2 = time.perf_counter()
    return max((t2 - t1) - (t1 - t0), 1e-9), t1 - t0


def cpu_dense_state(L, logc, iters, what, threads=None, fastmath=False):
    """The dense-state RCG exactly as the reference structures it (rcgpar::rcg_optl_omp restated: L, gamma, step,
    oldstep as G x E fp64), timed on the host cores on the dense matrix L.  threads: OpenMP threads (default: the
    box's CPU share); fastmath: the oracle built with the reference's Release flags (-ffast-math, CMakeLists.txt:20)."""
    import numpy as np
    O, cores = _oracle(threads, fastmath)
    G, E = L.shape
    dt, t_first = _timed_iters(lambda n: O.rcg_optl_dense(L, logc, np.ones(G), tol=-1.0, max_iters=n), iters)
    return {"value": E * G * iters / dt, "unit": "cells/s", "cores": cores, "kind": "port",
            "build": "-O3 -march=x86-64-v3 -fopenmp" + (" -ffast-math -funroll-loops" if fastmath else ""),
            "sample": f"{what} as a dense fp64 matrix ({E} ECs x {G} groups, {E * G * 8 / 1e9:.1f} GB; the reference's "
                      f"four G x E matrices), {iters} RCG iterations of the dense-state restatement of "
                      f"rcgpar::rcg_optl_omp on {cores} OpenMP thread{'s' if cores != 1 else ''}, {dt:.1f} s (+ {t_first:.1f} s "
                      f"for set-up and one iteration); iters/s on the sample = {iters / dt:.3f}",
            "cpu": cpu_info()}


def cpu_dense_lines(make_L, logc, n_full, iters, what):
    """BASELINE.md 2's three CPU lines of the dense-state restatement: all usable cores, the same with the
    reference's -ffast-math build on the SAME sample, and ONE thread on a tenth of it (the same rows of the same
    matrix: one thread on the whole sample would take minutes)."""
    L = make_L(n_full)
    log(f"cpu baseline: dense-state restatement, {L.shape[1]} ECs x {L.shape[0]} groups, all cores ...")
    out = {"cpu_baseline": cpu_dense_state(L, logc[:n_full], iters, what(n_full))}
    log(f"cpu baseline: {out['cpu_baseline']['value']:.3g} cells/s; the -ffast-math build ...")
    out["cpu_baseline_fastmath"] = cpu_dense_state(L, logc[:n_full], iters, what(n_full), fastmath=True)
    n1 = max(n_full // 10, 1)
    log(f"cpu baseline: {out['cpu_baseline_fastmath']['value']:.3g} cells/s; one thread on {n1} ECs ...")
    out["cpu_baseline_1thread"] = cpu_dense_state(np_ascontig(L[:, :n1]), logc[:n1], iters, what(n1), threads=1)
    log(f"cpu baseline: {out['cpu_baseline_1thread']['value']:.3g} cells/s")
    return out


def np_ascontig(a):
    import numpy as np
    return np.ascontiguousarray(a)


def cpu_structured_csr(rowptr, grp, lutidx, lut, G, logc, iters, what):
    """Second CPU line (apples to apples): the oracle's structured CSR restatement -- the same O(nnz) algorithm
    the GPU runs -- on the FULL workload."""
    import numpy as np
    O, cores = _oracle()
    E = len(rowptr) - 1
    dt, _ = _timed_iters(lambda n: O.rcg_optl_csr(rowptr, grp, lutidx, lut, np.log(0.01), G, logc, np.ones(G),
                                                  tol=-1.0, max_iters=n), iters)
    return {"value": float(E) * G * iters / dt, "unit": "cells/s", "cores": cores, "kind": "port",
            "listed_cells_per_sec": float(len(grp)) * iters / dt,
            "sample": f"{what}, {iters} iterations of the structured CSR restatement (oracle/rcg_oracle.cpp "
                      f"orc_rcg_optl_csr: the O(nnz) algorithm the GPU runs) on {cores} OpenMP threads, {dt:.1f} s; "
                      f"iters/s = {iters / dt:.3f}"}


def dense_sample(rowptr, grp, cnt, lut, G, n_ecs):
    import numpy as np
    rp = rowptr.astype(np.int64)
    E = min(n_ecs, len(rp) - 1)
    nz = rp[E]
    L = np.full((G, E), np.log(0.01))
    rows = np.repeat(np.arange(E), np.diff(rp[:E + 1]))
    L[grp[:nz], rows] = lut[grp[:nz], cnt[:nz]]
    return L


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


class Star:
    """Host-side meeting point of the rank processes of ONE node -- what carries the library's RCCL unique id from
    rank 0 to the others (torch.distributed did that until round 4: the torch wheel bundles a second ROCm runtime,
    and the two in one process aborted at exit or failed to load depending on the import order).  Rank 0 listens on
    an ABSTRACT unix socket named after the ranks' common parent (this file's launcher, or torch.distributed.run's
    agent) and MASTER_PORT; the name dies with rank 0, so a crashed run leaves nothing a later one could join."""

    def __init__(self, rank, world, timeout=300.0):
        import socket
        self.rank, self.world, self.peers, self.sock = rank, world, {}, None
        name = "\0msweep_bench_%d_%s" % (os.getppid(), os.environ.get("MASTER_PORT", "0"))
        if world == 1:
            return
        if rank == 0:
            srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            srv.bind(name)
            srv.listen(world)
            srv.settimeout(timeout)
            while len(self.peers) < world - 1:
                c, _ = srv.accept()
                c.settimeout(timeout)
                self.peers[int.from_bytes(self._recv(c, 4), "little")] = c
            srv.close()
        else:
            t_end = time.time() + timeout
            while True:
                c = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
                try:
                    c.connect(name)
                    break
                except (ConnectionRefusedError, FileNotFoundError):
                    c.close()
                    if time.time() > t_end:
                        raise RuntimeError("bench.py: rank 0 never opened the rendezvous socket")
                    time.sleep(0.05)
            c.settimeout(timeout)
            c.sendall(rank.to_bytes(4, "little"))
            self.sock = c

    @staticmethod
    def _recv(c, n):
        buf = b""
        while len(buf) < n:
            part = c.recv(n - len(buf))
            if not part:
                raise RuntimeError("bench.py: a rank left the rendezvous")
            buf += part
        return buf

    def _send_msg(self, c, data):
        c.sendall(len(data).to_bytes(8, "little") + data)

    def _recv_msg(self, c):
        return self._recv(c, int.from_bytes(self._recv(c, 8), "little"))

    def allgather(self, obj):
        """[rank 0's obj, rank 1's, ...] on every rank (JSON-serialisable objects; a barrier as a side effect)."""
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            out = [obj] + [json.loads(self._recv_msg(self.peers[r])) for r in range(1, self.world)]
            data = json.dumps(out).encode()
            for r in range(1, self.world):
                self._send_msg(self.peers[r], data)
            return out
        self._send_msg(self.sock, json.dumps(obj).encode())
        return json.loads(self._recv_msg(self.sock))

    def bcast_bytes(self, data):
        """rank 0's bytes on every rank"""
        return bytes.fromhex(self.allgather(data.hex() if self.rank == 0 else None)[0])

    def close(self):
        for c in list(self.peers.values()) + ([self.sock] if self.sock else []):
            c.close()


def launch_selftest(a, rank, world):
    """CPU-only rehearsal of the launcher + rendezvous (tests/test_parallel_cpu.py): no GPU, no library."""
    if os.environ.get("MSWEEP_SELFTEST_FAIL_RANK") == str(rank):   # (tests) a rank that dies before the rendezvous
        sys.exit(3)
    star = Star(rank, world)
    token = star.bcast_bytes(os.urandom(128) if rank == 0 else b"")     # what the unique id travels as
    out = star.allgather(float(rank + 1))
    same = star.allgather(token.hex())
    if rank == 0:
        print(json.dumps({"selftest": True, "n_gpus": world, "gathered": out, "token_bytes": len(token),
                          "token_same_everywhere": len(set(same)) == 1, "torch_imported": "torch" in sys.modules}),
              flush=True)
    star.close()


# ---- end to end from Themisto text ---------------------------------------------------------------
def run_e2e(a):
    """cfg3's reads as two Themisto plaintext strands -> msw_alignment_read_device (text to HBM; parse, paired-end
    intersection, collapse into equivalence classes as kernels) -> msw_core_build_likelihood_aln -> solve to --tol 1e-6
    -> abundances.txt.  One GPU.  Beside it: the same files through the host reader (msw_alignment_read +
    msw_core_build_likelihood: the same abundances.txt, byte for byte), through the Python mirror of
    include/mSWEEP_alignment.hpp (msweep_amd/alignment.py) on a bounded prefix, and a gzip pair (single-threaded zlib
    inflate in front of the host parser)."""
    import gzip
    import io
    import shutil
    import tempfile
    import numpy as np
    from msweep_amd import synth
    from msweep_amd.alignment import Alignment
    from msweep_amd.core import Core, read_alignment
    from msweep_amd.likelihood import from_alignment, from_device_alignment
    from msweep_amd.sample import PlainSample
    G, R = a.groups, a.reads
    tmp = tempfile.mkdtemp(prefix="msweep_e2e_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        t0 = time.time()
        prob = synth.make_csr_problem(R, G, seed=a.seed)
        aln = synth.csr_to_targets(prob, shuffle=False)
        E = len(prob["ec_counts"])
        log(f"e2e: {R} reads x {G} groups generated ({E} ECs, {aln['n_targets']} targets) in {time.time() - t0:.0f}s; writing text ...")
        rng = np.random.default_rng(11)
        ec_of = rng.permutation(np.repeat(np.arange(E, dtype=np.int64), prob["ec_counts"].astype(np.int64)))
        f1, f2 = os.path.join(tmp, "reads_1.txt"), os.path.join(tmp, "reads_2.txt")
        t0 = time.time()
        nbytes = 0
        for k, path in enumerate((f1, f2)):
            # the second strand disagrees with its mate on a tenth of the reads (one more target: gone after the
            # intersection), so the merge has something to do
            nbytes += synth.write_themisto(path, ec_of, aln["ec_tptr"], aln["ec_targets"], chunk=1_000_000,
                                           extra=(rng, 0.1, aln["n_targets"]) if k else None)
            log(f"e2e: strand {k + 1} written ({os.path.getsize(path) / 1e6:.0f} MB, {time.time() - t0:.0f}s)")
        n_targets = int(aln["n_targets"])
        target_group, sizes = aln["target_group"], prob["group_sizes"]
        names = [f"g{g}" for g in range(G)]
        del aln, ec_of
        core = Core(0)
        core.set_pack_schedule(False)   # one solve: the drivers skip the bank ordering below five bootstrap replicates
        stages = {}
        best = None
        def one_pass(device_reader):
            t1 = time.perf_counter()
            if device_reader:
                al = core.read_alignment([f1, f2], n_targets, "intersection")
                t2 = time.perf_counter()
                lik = from_device_alignment(core, al, target_group, sizes)
                n_reads, n_aligned, n_ecs, n_hits = al.n_reads, al.n_aligned, al.n_ecs, al.n_hits
            else:
                al = read_alignment([f1, f2], n_targets, "intersection")
                t2 = time.perf_counter()
                lik = from_alignment(core, al["ec_tptr"], al["ec_targets"], target_group, sizes, al["ec_counts"],
                                     download_log_counts=False)
                n_reads, n_aligned, n_ecs, n_hits = al["n_reads"], len(al["ec_reads"]), len(al["ec_counts"]), len(al["ec_targets"])
            t3 = time.perf_counter()
            res = core.solve(None, np.ones(lik.n_groups))
            t4 = time.perf_counter()
            out = io.StringIO()
            smp = PlainSample(n_reads, n_aligned)
            smp.store_abundances(res["theta"])
            smp.write_abundances(names, out)
            with open(os.path.join(tmp, "e2e_abundances.txt"), "w") as f:
                f.write(out.getvalue())
            t5 = time.perf_counter()
            cur = {"read_collapse_s": t2 - t1, "build_likelihood_s": t3 - t2, "solve_s": t4 - t3, "write_abundances_s": t5 - t4,
                   "total_s": t5 - t1, "iters": int(res["iters"]), "ecs": int(n_ecs), "target_hits": int(n_hits)}
            # (a few ECs fewer than the generator made: the reference keys reads by a 64-bit XOR-shift hash of their
            # target set alone and merges what collides -- include/mSWEEP_alignment.hpp:152-156,186 -- as this reader does)
            assert E - 64 <= cur["ecs"] <= E and abs(res["theta"].sum() - 1.0) < 1e-9
            return cur, out.getvalue()

        both, first = {}, {}
        for device_reader in (False, True):
            best = None
            for rep in range(2 if not device_reader else 3):  # (the first pass also warms the page cache and the allocator)
                cur, text = one_pass(device_reader)
                first.setdefault(device_reader, cur)
                log(f"e2e pass {rep} ({'device' if device_reader else 'host'} reader): " +
                    ", ".join(f"{k} {v:.3f}" if isinstance(v, float) else f"{k} {v}" for k, v in cur.items()))
                if best is None or cur["total_s"] < best["total_s"]:
                    best = cur
            both[device_reader] = (best, text)
        assert both[False][1] == both[True][1], "abundances.txt differs between the host and the device reader"
        stages, stages_host = both[True][0], both[False][0]
        threads = int(os.environ.get("MSWEEP_READER_THREADS", "0")) or min(16, cpu_share())
        # the Python mirror of the reference's reader on a prefix of the same files
        n_py = min(R, 100_000)
        heads = []
        for path in (f1, f2):
            with open(path) as f:
                heads.append("".join(f.readline() for _ in range(n_py)))
        t1 = time.perf_counter()
        pa = Alignment(n_targets)
        pa.read("intersection", [io.StringIO(h) for h in heads])
        pa.collapse()
        t_py = time.perf_counter() - t1
        py_bytes = sum(len(h) for h in heads)
        # a gzip pair of a prefix (1 M reads): zlib inflates on one thread, then the same parser
        n_gz = min(R, 1_000_000)
        gz_bytes = 0
        gz_paths = []
        for path in (f1, f2):
            with open(path, "rb") as f:
                buf = b"".join(f.readline() for _ in range(n_gz))
            gz_bytes += len(buf)
            gp = path + ".gz"
            with gzip.open(gp, "wb", compresslevel=1) as g:
                g.write(buf)
            gz_paths.append(gp)
        t1 = time.perf_counter()
        ag = read_alignment(gz_paths, n_targets, "intersection")
        t_gz = time.perf_counter() - t1
        gz_file_bytes = sum(os.path.getsize(g) for g in gz_paths)
        core.close()
        line = {
            "metric": "e2e: Themisto plaintext -> abundances.txt, 10M reads x 5k groups (reads/s of the whole pipeline)",
            "value": R / stages["total_s"], "unit": "reads/s", "n_gpus": 1, "higher_is_better": True, "data": "synthetic",
            "config": {"workload": f"cfg3's {R} reads x {G} groups as two Themisto plaintext strands ({nbytes / 1e9:.2f} GB of "
                                   "text, --themisto-mode intersection), msw_alignment_read_device -> msw_core_build_likelihood_aln -> "
                                   "msw_core_solve(--tol 1e-6) -> abundances.txt", "reads": R, "groups": G, "seed": a.seed},
            "stages_s": stages,
            "stages_first_pass_s": first[True],
            "stages_what": "stages_s: the best of three passes on one handle (from the second pass on the reader's device "
                           "memory comes from the handle's pool: no allocation); stages_first_pass_s: the first pass of "
                           "the device reader in this process -- what a single run of the drivers pays (page cache warm "
                           "from the host reader's passes)",
            "reader": {"text_MB_per_s": nbytes / 1e6 / stages["read_collapse_s"],
                       "reads_per_s": R / stages["read_collapse_s"], "text_bytes": nbytes, "host_threads": threads,
                       "what": "msw_alignment_read_device: pread into pinned staging on the host threads, the text to HBM "
                               "two chunks in flight, then kernels -- tokens, rows by read id, sorted sets, paired-end "
                               "intersection, the reference's 64-bit hash, radix sort, equivalence classes "
                               "(host_reader.inc); the classes stay in device memory for msw_core_build_likelihood_aln"},
            "host_reader": {"stages_s": stages_host, "threads": threads,
                            "text_MB_per_s": nbytes / 1e6 / stages_host["read_collapse_s"],
                            "reads_per_s": R / stages_host["read_collapse_s"],
                            "what": "the same files through msw_alignment_read (host_alignment.inc: mmap, chunk-parallel "
                                    "parse, parallel merge sort) + msw_core_build_likelihood from host arrays: the same "
                                    "abundances.txt, byte for byte"},
            "python_mirror": {"reads": n_py, "seconds": t_py, "text_MB_per_s": py_bytes / 1e6 / t_py, "reads_per_s": n_py / t_py,
                              "what": "msweep_amd/alignment.py Alignment.read + collapse (the mirror of include/"
                                      "mSWEEP_alignment.hpp:54-215) on the first lines of the same files: the stated baseline"},
            "gzip": {"reads": n_gz, "seconds": t_gz, "uncompressed_MB_per_s": gz_bytes / 1e6 / t_gz,
                     "compressed_MB": gz_file_bytes / 1e6, "ecs": int(len(ag["ec_counts"])),
                     "what": "the first reads of both strands gzip-compressed (level 1): zlib inflate on ONE thread per "
                             "file in front of the same parser"},
            "cpu": cpu_info(),
        }
        print(json.dumps(line), file=OUT, flush=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# ---- workloads --------------------------------------------------------------------------------------------------
def load_workload(a, core, shard, rank, world):
    """Generate the configuration's synthetic input and make its likelihood resident on `core`.
    Returns dict(E, G, nnz, logc (host vector or None = resident), w (uint32 EC counts or None), desc,
    setup_s, cpu (callable -> dict of CPU baselines), ...)."""
    import numpy as np
    from msweep_amd import synth
    from msweep_amd.likelihood import from_alignment, from_dense, from_grouped_counts, precalc_lls
    t0 = time.time()
    if a.config == "cfg2":
        E, G = a.reads, a.groups
        p = synth.make_dense_problem(E, G, seed=a.seed)
        t_gen = time.time() - t0
        t0 = time.time()
        from_dense(core, p["logl"], p["logc"])
        t_up = time.time() - t0
        nnz = core.shape()[2]
        log(f"cfg2: dense {G} x {E} generated in {t_gen:.1f}s, resident in {t_up:.2f}s (listed cells {nnz})")

        def cpu():
            return cpu_legs_in_child(a) or cpu_cfg2_lines(a, p, E, G)
        return dict(E=E, G=G, nnz=nnz, logc=p["logc"], w=None, cpu=cpu, reads=E,
                    setup_s={"generate": t_gen, "set_dense_logl": t_up},
                    desc=f"cfg2: synthetic {E} ECs x {G} groups dense fp64 likelihood (4.0 GB) through the dense boundary "
                         + ("(msw_core_set_dense_logl with MSWEEP_DENSE_COMPRESS=0: kept DENSE on the device, EC-major, the "
                            "sweeps k_dense_passA / k_dense_passB -- what a matrix without background structure gets), "
                            if os.environ.get("MSWEEP_DENSE_COMPRESS") == "0" else
                            "(msw_core_set_dense_logl: re-expressed on the device as CSR-of-ECs, one table slot per listed "
                            "cell), ") + "RCG-VB, fixed iteration count")
    if a.config == "cfg5":
        G = a.groups
        p = synth.make_csr_problem(a.reads, G, seed=a.seed, max_other=7, theta_support=max(G // 10, 1), chunk=2_000_000)
        t_gen = time.time() - t0
        t0 = time.time()
        E, nnz = len(p["rowptr"]) - 1, len(p["grp"])
        if shard:
            # EC-sharded build: every rank expands and builds its own block of ECs; the --min-hits
            # counts of a group are summed over ALL ECs (one all-reduce of G integers inside the build call)
            from msweep_amd.parallel import csr_block, shard_ecs
            b = shard_ecs(p["rowptr"], world)
            blk = csr_block(p, b[rank], b[rank + 1])
            aln = synth.csr_to_targets(blk, shuffle=False)
            ecc = blk["ec_counts"]
        else:
            aln = synth.csr_to_targets(p, shuffle=False)
            ecc = p["ec_counts"]
        t_aln = time.time() - t0
        t0 = time.time()
        lik = from_alignment(core, aln["ec_tptr"], aln["ec_targets"], aln["target_group"], p["group_sizes"],
                             ecc, min_hits=1)
        t_build = time.time() - t0
        hits = int(len(aln["ec_targets"]))
        G2 = lik.n_groups
        keep = {"aln": aln}
        del aln

        def first_theta():
            """msw_core_build_likelihood (--min-hits 1) + solve to --tol 1e-6 on a fresh handle of this (warm) process."""
            from msweep_amd.core import Core
            al = keep.pop("aln")
            out = {}
            for sched in (False, True):
                with Core(core.device) as c2:
                    c2.set_pack_schedule(sched)
                    t1 = time.perf_counter()
                    lk = from_alignment(c2, al["ec_tptr"], al["ec_targets"], al["target_group"], p["group_sizes"], ecc,
                                        min_hits=1, download_log_counts=False)
                    t2 = time.perf_counter()
                    r = c2.solve(None, np.ones(lk.n_groups))
                    t3 = time.perf_counter()
                out["bank_scheduled" if sched else "unscheduled"] = {
                    "build_likelihood_ms": (t2 - t1) * 1e3, "solve_ms": (t3 - t2) * 1e3, "total_ms": (t3 - t1) * 1e3,
                    "iters": int(r["iters"])}
            out["ms"] = out["unscheduled"]["total_ms"]
            out["what"] = ("msw_core_build_likelihood(--min-hits 1) + msw_core_solve(--tol 1e-6, log counts resident) on a "
                           "fresh handle, host wall clock, the pseudoalignment in pageable host memory; `unscheduled` = "
                           "msw_core_set_pack_schedule(0), the drivers' choice for one solve")
            return out
        log(f"cfg5: E={E} nnz={nnz} generated in {t_gen:.0f}+{t_aln:.0f}s; build (upload of {hits} target hits, "
            f"K0-K2, --min-hits 1 mask, SELL packing) {t_build:.2f}s; {G2} of {G} groups kept")

        def cpu():
            kept = np.nonzero(lik.groups_considered())[0]
            newid = np.cumsum(lik.groups_considered()) - 1
            grp2 = newid[p["grp"]].astype(np.uint32)
            lut = precalc_lls(p["group_sizes"][kept])
            lutidx = (grp2 * lut.shape[1] + p["cnt"]).astype(np.uint32)
            n = min(200_000, E)
            out = cpu_dense_lines(lambda m: dense_sample(p["rowptr"], grp2, p["cnt"], lut, G2, m), lik.log_counts(), n,
                                  a.cpu_iters, lambda m: f"first {m} ECs of the cfg5 workload x the {G2} groups --min-hits 1 keeps")
            out["cpu_baseline_structured"] = cpu_structured_csr(p["rowptr"], grp2, lutidx, lut, G2, lik.log_counts(), 3,
                                                                "full cfg5 workload (compacted to the kept groups)")
            return out
        return dict(E=E, G=G2, nnz=nnz, logc=lik.log_counts() if shard else None, w=None, cpu=cpu, reads=a.reads,
                    first_theta=None if shard else first_theta,
                    setup_s={"generate": t_gen, "expand_to_targets": t_aln, "build_likelihood": t_build},
                    build={"seconds": t_build, "target_hits": hits, "groups_in": G, "groups_kept": G2,
                           "what": "msw_core_build_likelihood: upload of the pseudoalignment (ec_tptr / ec_targets / "
                                   "group indicators), K1 EC x group counts, K2 --min-hits 1 mask + compaction, K0 lookup "
                                   "table, log counts, SELL-64 packing -- host wall clock of the one call"},
                    desc=f"cfg5: synthetic {a.reads} reads x {G} groups, theta supported on {max(G // 10, 1)} groups, <= 8 listed "
                         f"groups per read, msw_core_build_likelihood with --min-hits 1 ({G2} groups kept), CSR-of-ECs, one "
                         "GPU, RCG-VB, fixed iteration count")
    # cfg3 / cfg4
    G = a.groups
    prob = synth.make_csr_problem(a.reads, G, seed=a.seed,
                                  group_sizes=synth.diverse_group_sizes if a.group_sizes == "diverse" else None)
    E, nnz = len(prob["rowptr"]) - 1, len(prob["grp"])
    t_gen = time.time() - t0
    log(f"generated {a.config}: E={E} nnz={nnz} in {t_gen:.1f}s")
    t0 = time.time()
    if shard:
        from msweep_amd.parallel import csr_block, shard_ecs
        b = shard_ecs(prob["rowptr"], world)
        blk = csr_block(prob, b[rank], b[rank + 1])
        lik = from_grouped_counts(core, blk["rowptr"], blk["grp"], blk["cnt"], blk["ec_counts"], prob["group_sizes"])
    else:
        lik = from_grouped_counts(core, prob["rowptr"], prob["grp"], prob["cnt"], prob["ec_counts"], prob["group_sizes"])
    t_up = time.time() - t0

    def cpu():
        return cpu_legs_in_child(a) or cpu_cfg3_lines(a, prob, G)

    def first_theta():
        """Time to the FIRST estimate of a likelihood the host holds as CSR-of-ECs: msw_core_set_csr (upload, slot
        plan, SELL packing -- without the LDS-bank ordering of the cells, which pays from the ~1000th iteration on:
        what the drivers do for runs of one solve) + the solve to --tol 1e-6, on a fresh handle of this (warm) process."""
        from msweep_amd.core import Core
        lut = precalc_lls(prob["group_sizes"])
        logc = lik.log_counts()
        out = {}
        for sched in (False, True):
            with Core(core.device) as c2:
                c2.set_pack_schedule(sched)
                t1 = time.perf_counter()
                c2.set_csr(prob["rowptr"], prob["grp"], prob["cnt"], lut, np.log(0.01), G)
                t2 = time.perf_counter()
                r = c2.solve(logc, np.ones(G))
                t3 = time.perf_counter()
            out["bank_scheduled" if sched else "unscheduled"] = {
                "set_csr_ms": (t2 - t1) * 1e3, "solve_ms": (t3 - t2) * 1e3, "total_ms": (t3 - t1) * 1e3, "iters": int(r["iters"])}
        out["ms"] = out["unscheduled"]["total_ms"]
        out["what"] = ("msw_core_set_csr + msw_core_solve(--tol 1e-6) on a fresh handle, host wall clock, inputs in "
                       "pageable host memory; `unscheduled` = msw_core_set_pack_schedule(0), the drivers' choice for "
                       "one solve; setup_s.set_csr is the process's FIRST upload (cold allocator) incl. the Python "
                       "mirror's table and log-count work")
        return out

    def from_text():
        """The same reads as two Themisto plaintext strands -> abundances.txt: the reader on the device + the build from
        its device-resident classes + the solve to --tol 1e-6, beside the host reader on the same files (one pass).
        The compact form of `bench.py --config e2e`."""
        import io
        import shutil
        import tempfile
        from msweep_amd.core import Core, read_alignment
        from msweep_amd.likelihood import from_alignment, from_device_alignment
        from msweep_amd.sample import PlainSample
        tmp = tempfile.mkdtemp(prefix="msweep_text_", dir=os.environ.get("TMPDIR", "/tmp"))
        try:
            aln = synth.csr_to_targets(prob, shuffle=False)
            rng = np.random.default_rng(11)
            ec_of = rng.permutation(np.repeat(np.arange(E, dtype=np.int64), prob["ec_counts"].astype(np.int64)))
            files, nbytes = [os.path.join(tmp, "reads_1.txt"), os.path.join(tmp, "reads_2.txt")], 0
            for k, path in enumerate(files):
                nbytes += synth.write_themisto(path, ec_of, aln["ec_tptr"], aln["ec_targets"], chunk=1_000_000,
                                               extra=(rng, 0.1, aln["n_targets"]) if k else None)
            n_targets, target_group, names = int(aln["n_targets"]), aln["target_group"], [f"g{g}" for g in range(G)]
            del aln, ec_of

            def run(c2, device_reader):
                t1 = time.perf_counter()
                if device_reader:
                    al = c2.read_alignment(files, n_targets, "intersection")
                    t2 = time.perf_counter()
                    lk = from_device_alignment(c2, al, target_group, prob["group_sizes"])
                    n_reads, n_aligned, n_ecs = al.n_reads, al.n_aligned, al.n_ecs
                else:
                    al = read_alignment(files, n_targets, "intersection")
                    t2 = time.perf_counter()
                    lk = from_alignment(c2, al["ec_tptr"], al["ec_targets"], target_group, prob["group_sizes"], al["ec_counts"],
                                        download_log_counts=False)
                    n_reads, n_aligned, n_ecs = al["n_reads"], len(al["ec_reads"]), len(al["ec_counts"])
                t3 = time.perf_counter()
                r = c2.solve(None, np.ones(lk.n_groups))
                t4 = time.perf_counter()
                out = io.StringIO()
                smp = PlainSample(n_reads, n_aligned)
                smp.store_abundances(r["theta"])
                smp.write_abundances(names, out)
                with open(os.path.join(tmp, "abundances.txt"), "w") as f:
                    f.write(out.getvalue())
                t5 = time.perf_counter()
                return {"read_collapse_s": t2 - t1, "build_likelihood_s": t3 - t2, "solve_s": t4 - t3,
                        "write_abundances_s": t5 - t4, "total_s": t5 - t1, "iters": int(r["iters"]), "ecs": int(n_ecs)}, out.getvalue()

            with Core(core.device) as c2:
                c2.set_pack_schedule(False)   # one solve: as the drivers
                host, text_host = run(c2, False)
                passes = [run(c2, True) for _ in range(3)]
            best = min((x[0] for x in passes), key=lambda d: d["total_s"])
            return {"seconds": best["total_s"], "reads_per_sec": a.reads / best["total_s"], "stages_s": best,
                    "first_pass_s": passes[0][0], "host_reader": host, "text_bytes": nbytes,
                    "same_abundances_txt": all(x[1] == text_host for x in passes),
                    "what": "the workload's reads as two Themisto plaintext strands (--themisto-mode intersection) -> "
                            "msw_alignment_read_device (text to HBM, parse / merge / hash / sort / classes as kernels) -> "
                            "msw_core_build_likelihood_aln (classes read in device memory) -> msw_core_solve(--tol 1e-6) -> "
                            "abundances.txt; best of three passes on one handle, first_pass_s = the first of them; "
                            "host_reader = one pass of msw_alignment_read + msw_core_build_likelihood on the same files "
                            "(page cache warm for both); `bench.py --config e2e` is the long form"}
        finally:
            shutil.rmtree(tmp, ignore_errors=True)

    if a.group_sizes == "diverse":
        return dict(E=E, G=G, nnz=nnz, logc=lik.log_counts(), w=prob["ec_counts"].astype(np.uint32), cpu=cpu, reads=a.reads,
                    first_theta=first_theta,
                    setup_s={"generate": t_gen, "set_csr": t_up},
                    desc=f"{a.config} with DIVERSE group sizes (log-normal, up to 400 sequences per group: "
                         f"{int(prob['group_sizes'].max())} here; thousands of used lookup-table slots -> index records + hybrid "
                         "slot area): synthetic 10M reads x 5k groups, CSR-of-ECs likelihood, RCG-VB, fixed iteration count")
    return dict(E=E, G=G, nnz=nnz, logc=lik.log_counts(), w=prob["ec_counts"].astype(np.uint32), cpu=cpu, reads=a.reads,
                first_theta=first_theta, from_text=from_text if a.config == "cfg3" and not shard else None,
                setup_s={"generate": t_gen, "set_csr": t_up},
                desc="cfg3: synthetic 10M reads x 5k groups, CSR-of-ECs likelihood, RCG-VB (--algorithm rcggpu), "
                     "fixed iteration count" if a.config == "cfg3" else
                     "cfg4: cfg3's likelihood (synthetic 10M reads x 5k groups, CSR-of-ECs) + bootstrap: a step = one "
                     "replicate of the one mt19937_64(--seed 42) stream -- seek / jump-ahead, resampling, RCG solve to "
                     "--tol 1e-6 -- through msw_core_bootstrap_dist, abundances all-gathered")


def bootstrap_leg(core, comm, w, per_rank, world, G):
    """The real replicate loop (msw_core_bootstrap_dist): per_rank * world replicates over the ranks of comm."""
    import numpy as np
    draws = int(w.sum())
    B = per_rank * world
    t0 = time.perf_counter()
    theta, iters = core.bootstrap_dist(comm, w, 42, draws, B, np.ones(G))
    dt = time.perf_counter() - t0
    bt = core.last_bootstrap_timing()
    assert theta.shape == (B, G) and np.all(np.abs(theta.sum(1) - 1.0) < 1e-9)
    return dt, iters, bt


def iteration_histogram(iters):
    """RCG iterations to --tol 1e-6 over the replicates: extremes, quartiles and counts per bin of 5."""
    import numpy as np
    it = np.asarray(iters, np.int64)
    if it.size == 0:
        return None
    lo = int(it.min()) // 5 * 5
    bins = np.bincount((it - lo) // 5)
    return {"min": int(it.min()), "q25": float(np.percentile(it, 25)), "median": float(np.median(it)),
            "q75": float(np.percentile(it, 75)), "max": int(it.max()), "mean": float(it.mean()),
            "bins_of_5": {str(lo + 5 * i): int(n) for i, n in enumerate(bins) if n}}


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
    if a.cpu_legs_only:
        return cpu_legs_only(a)
    if a.config == "e2e":
        if world > 1:
            sys.exit("bench.py: --config e2e is a single-GPU line")
        return run_e2e(a)
    if a.config == "cfg2" and (world > 1 or a.mode == "shard"):
        sys.exit("bench.py: --config cfg2 is a single-GPU line; the N > 1 modes run on cfg3 / cfg4 / cfg5")
    if a.config == "cfg5" and world > 1 and a.mode != "shard":
        sys.exit("bench.py: --config cfg5 over several GPUs is ONE solve with the ECs sharded (BASELINE.json: 8 x MI355X): "
                 "add --mode shard")
    if a.config == "cfg4" and a.mode == "shard":
        sys.exit("bench.py: --config cfg4 shards whole replicates, not ECs")
    # ONE JSON line on stdout: libraries that print banners there (RCCL's version block at its first
    # initialisation) are sent to stderr; the line itself goes to the saved descriptor
    global OUT
    sys.stdout.flush()
    OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    # MSWEEP_BENCH_ONE_GPU=1 (developer switch, a FUNCTIONAL rehearsal of the N > 1 code of this file on a box with one
    # GPU, never a measurement): every rank on device 0, the library's ranks meeting in a shared-memory segment
    # (msw_comm_create_shm) -- RCCL refuses two ranks on one device
    one_gpu = os.environ.get("MSWEEP_BENCH_ONE_GPU", "0") == "1" and world > 1
    if one_gpu:
        local_rank = 0
    multi = world > 1 or a.force_dist
    n_gpus = world

    import numpy as np
    from msweep_amd.core import Comm, Core

    core = Core(local_rank)
    shard = multi and a.mode == "shard"
    comm = None
    rccl_ranks = None
    star = None
    if multi:
        # the library's own RCCL communicator (C ABI: msw_comm_create_rccl) carries every collective of this file:
        # the barriers around the timed region (msw_comm_allreduce), the max over ranks and the abundances
        # (msw_comm_allgather).  No torch in a rank process: only the 128-byte unique id travels outside RCCL, through
        # the ranks' own unix socket (Star).
        os.environ.setdefault("MASTER_PORT", "29533")
        star = Star(rank, world)
        if one_gpu:
            comm = Comm.shm("/msweep_bench_%d_%s" % (os.getppid(), os.environ["MASTER_PORT"]), rank, world, 0)
            rccl_ranks = "none: MSWEEP_BENCH_ONE_GPU rehearsal, all ranks on one device (NOT a measurement)"
        else:
            uid = star.bcast_bytes(Comm.unique_id() if rank == 0 else b"")
            comm = Comm.rccl(uid, rank, world, local_rank)
            rccl_ranks = comm.rccl_count()     # ncclCommCount
            if rccl_ranks != world:
                sys.exit(f"bench.py: RCCL communicator spans {rccl_ranks} ranks, expected {world}")
    boot_comm = comm if comm is not None else Comm.local(1)[0]   # one rank: the in-process communicator
    if shard:
        core.set_comm(comm)      # before the likelihood: cfg5's --min-hits counts are all-reduced over the EC shards
    wl = load_workload(a, core, shard, rank, world)
    E, G, nnz = wl["E"], wl["G"], wl["nnz"]
    alpha0 = np.ones(G)
    logc = wl["logc"]
    if multi and not shard and a.config == "cfg3":
        # replicate `rank` of the bootstrap, drawn on the device from the reference's ONE sequential
        # mt19937_64(--seed 42) stream (src/BootstrapSample.cpp:60-73): rank r owns draws
        # [r * n_reads, (r + 1) * n_reads), exactly what a single-GPU run would give replicate r
        counts = core.resample_counts(wl["w"], 42, int(wl["w"].sum()), rank, rank + 1)[0].astype(np.float64)
        with np.errstate(divide="ignore"):
            logc = np.log(counts)
    log("likelihood resident")

    def sync():
        # barrier + device synchronisation on both sides of the timed region.  The library calls themselves return
        # only after their HIP streams have drained (they download theta), and the barrier is a collective of the
        # library's communicator on the device (a one-word all-reduce, synchronised before it returns): every
        # rank's GPU is idle when the last rank arrives.  With one rank there is nothing to wait for.
        if multi:
            comm.allreduce([0], [0.0])      # (one integer + one double: the shape the sharded solve's all-reduce has)

    def max_over_ranks(x):
        return float(comm.allgather(np.array([x])).max()) if multi else x

    # Pre-warm, BEFORE the timed region and reported in the line (`prewarm`): SURVEY 8(d)'s second figure --
    # time to convergence at the reference's defaults (--tol 1e-6, --max-iters 5000), host inputs handed over
    # per call as at the reference's boundary (PCIe-inclusive), twice -- and the EM leg.  Besides being figures
    # of their own these are ~100 ms of the same sweeps: a cold MI355X reaches its working clocks only after
    # 60-80 ms of activity (tools/timing.py ramp: the SAME twenty iterations take 0.214 ms each on a cold chip,
    # 0.181 ms once it has been busy for 40 ms), and the timed region is to measure the path, not the governor.
    conv = em = em_float = None
    prewarm = {"ms": 0.0, "what": "none (--no-prewarm)"}
    t_pre = time.perf_counter()
    if not shard and not a.no_extras:
        runs = []
        for _ in range(2 + int(os.environ.get("MSWEEP_BENCH_PRESOLVES", "0"))):  # developer switch: more of them
            t1 = time.perf_counter()
            rc = core.solve(logc, alpha0, tol=1e-6, max_iters=5000)
            runs.append(((time.perf_counter() - t1) * 1e3, core.last_timing()["solve_ms"]))
        conv = {"iters": int(rc["iters"]), "ms": min(r[0] for r in runs), "device_ms": min(r[1] for r in runs),
                "runs_ms": [round(r[0], 3) for r in runs], "tol": 1e-6,
                "includes": "upload of log counts and prior, download of theta"}
        if a.config != "cfg4":
            # the EM optimiser (--algorithm emgpu: one pass-B sweep + one O(G) kernel per iteration), W + K steps
            # on the same resident inputs; reported beside the headline, which is the reference's default RCG
            from msweep_amd.core import ALGO_EM
            core.set_fixed_iters(True)
            core.run(max_iters=max(a.warmup, 1), algo=ALGO_EM)
            t1 = time.perf_counter()
            core.run(max_iters=a.steps, algo=ALGO_EM)
            t_em = time.perf_counter() - t1
            em = {"ms_per_step": t_em * 1e3 / a.steps, "iters_per_sec": a.steps / t_em}
            # ... and --emprecision float (the reference's fastest GPU mode, docs/gpubenchmarks.md:22,25): the fp32 sweep
            # where the layout allows (msw_timing::em_float_kernels), same W + K steps; then to --tol 1e-6, where the float
            # log-likelihood stops growing at float resolution long before the double run stops
            from msweep_amd.core import PREC_FLOAT
            core.run(max_iters=max(a.warmup, 1), algo=ALGO_EM, prec=PREC_FLOAT)
            t1 = time.perf_counter()
            core.run(max_iters=a.steps, algo=ALGO_EM, prec=PREC_FLOAT)
            t_emf = time.perf_counter() - t1
            fp32 = bool(core.last_timing()["em_float_kernels"])
            core.set_fixed_iters(False)
            t1 = time.perf_counter()
            rf = core.run(tol=1e-6, max_iters=5000, algo=ALGO_EM, prec=PREC_FLOAT)
            t_conv = time.perf_counter() - t1
            core.set_fixed_iters(True)
            em_float = {"ms_per_step": t_emf * 1e3 / a.steps, "iters_per_sec": a.steps / t_emf, "fp32_kernels": fp32,
                        "to_tol_1e-6": {"iters": int(rf["iters"]), "ms": t_conv * 1e3},
                        "what": "msw_core_run(MSW_ALGO_EM, MSW_PREC_FLOAT): fp32 likelihood values, weights, row sums and "
                                "quotients, exact integer column sums, the log-likelihood rounded to float for the stop "
                                "rule (msweep_amd/csrc/em_f32_kernels.hpp)" if fp32 else
                                "this layout is not served by the fp32 kernels: the fp64 kernels under MSW_PREC_FLOAT"}
        prewarm = {"ms": (time.perf_counter() - t_pre) * 1e3,
                   "what": f"{len(runs)} convergence solves at --tol 1e-6 ({conv['iters']} iterations each)"
                           + (f" + EM leg of {max(a.warmup, 1)} + {a.steps} iterations" if em else "")
                           + ", host wall clock, before the W warm-up steps; --no-prewarm skips them"}

    line_extra = {}
    if a.config == "cfg4":
        # ---- a step = one bootstrap replicate ---------------------------------------------------------------
        w = wl["w"]
        bootstrap_leg(core, boot_comm, w, max(a.warmup, 1), world, G)   # W untimed replicates per rank (builds the
        sync()                                                          # cumulative table, opens RCCL's connections)
        t0 = time.perf_counter()
        dt_call, iters, bt = bootstrap_leg(core, boot_comm, w, a.steps, world, G)
        sync()
        dt = time.perf_counter() - t0
        steps_total = a.steps * world
        tot_iters = int(iters.sum())
        cells = float(E) * G * tot_iters
        line_extra = {"replicates": steps_total, "replicates_per_sec": steps_total / dt,
                      "iterations_per_replicate": [int(x) for x in iters],
                      "iterations_histogram": iteration_histogram(iters),
                      "rank0_split_ms": {k: bt[k] for k in ("table_ms", "solve_ms", "gather_ms", "table_reused")},
                      "iters_per_sec": tot_iters / dt, "listed_cells_per_sec": float(nnz) * tot_iters / dt,
                      "reads_x_groups_cells_per_sec": float(wl["reads"]) * G * tot_iters / dt}
        # per-kernel durations for the roofline object: a fixed-iteration solve with events, as for cfg3
        core.prepare(logc, alpha0)
        core.set_fixed_iters(True)
        core.set_profiling(True)
        core.run(max_iters=20)
        core.continue_(100)
        tm = core.last_timing()
        tm0 = None
        core.set_profiling(False)
    else:
        core.set_fixed_iters(True)
        if conv is None:
            core.prepare(logc, alpha0)           # inputs resident in HBM before the timed region
        # (after the convergence run they already are: msw_core_solve = msw_core_prepare + msw_core_run)
        # W untimed warm-up steps: the first W iterations of the solve (with its set-up: the evaluation of the
        # initial state and the first iteration's rejected step); the K timed steps are the NEXT K iterations of
        # the same solve (msw_core_continue) -- every kernel slot they need, rejected steps included
        # The timed region is a few milliseconds at the driver's K = 20 (one scheduler hiccup on the host moves it by
        # several per cent): it is taken REPEATS times -- every time the W warm-up iterations again (untimed), then the
        # same K iterations, barrier + synchronisation on both sides, max over ranks -- and the MEDIAN is the line's
        # ms_per_step / value; all of them are in ms_per_step_runs.
        REPEATS = 5
        dts, dev = [], []
        for rep in range(REPEATS):
            core.run(max_iters=max(a.warmup, 1))
            if multi and not shard and rep == 0:
                comm.allgather(np.zeros(G))       # RCCL sets its connections up on the first collective: not timed
            sync()
            t0 = time.perf_counter()
            res = core.continue_(a.steps)         # exactly K steps
            if multi and not shard:
                gathered = comm.allgather(res["theta"])   # (world, G): the per-replicate abundances, RCCL all-gather
                assert gathered.shape == (world, G)
            sync()
            dts.append(max_over_ranks(time.perf_counter() - t0))
            tm0 = core.last_timing()
            dev.append(tm0["solve_ms"] / a.steps)
            assert tm0["iters"] == a.steps, (tm0["iters"], a.steps)
        dt = float(np.median(dts))
        log(f"timed {a.steps} steps x {REPEATS}: {', '.join(f'{x * 1e3:.3f}' for x in dts)} ms; median {dt * 1e3:.3f} ms")
        # K more steps with HIP events around every sweep launch (on the solve stream) for the
        # per-kernel durations of the roofline object.  Kept out of the timed run: every event record
        # is a barrier packet that costs ~6 us of idle GPU between two kernels.
        core.set_profiling(True)
        core.run(max_iters=max(a.warmup, 1))      # the same iterations as the timed ones: W, then K
        core.continue_(a.steps)
        tm = core.last_timing()
        core.set_profiling(False)
        mult = 1 if shard else n_gpus
        cells = float(E) * G * a.steps * mult
        line_extra = {"iters_per_sec": a.steps * mult / dt, "listed_cells_per_sec": float(nnz) * a.steps * mult / dt,
                      "reads_x_groups_cells_per_sec": float(wl["reads"]) * G * a.steps * mult / dt,
                      "device_ms_per_step": float(np.median(dev)),
                      "ms_per_step_runs": [x * 1e3 / a.steps for x in dts],
                      "ms_per_step_what": f"median of {REPEATS} repeats of the K-step region (each: W untimed warm-up "
                                          "iterations, barrier + synchronisation, K timed iterations, synchronisation + barrier; "
                                          "max over ranks per repeat)"}
        if shard:
            # where an iteration of the EC-sharded solve goes on rank 0 (the profiled run: events around the sweeps and
            # around the two all-reduces with their packing kernels): the first hardware run says which part to fix
            it = max(tm["iters"], 1)
            line_extra["shard_split_ms_per_step"] = {
                "passA": tm["passA_ms"] / it, "passB": tm["passB_ms"] / it, "collectives": tm["collective_ms"] / it,
                "chain_and_gaps": (tm["solve_ms"] - tm["passA_ms"] - tm["passB_ms"] - tm["collective_ms"]) / it,
                "collectives_per_step": tm["collectives"] / it,
                "allreduce": os.environ.get("MSWEEP_ALLREDUCE", "rccl"),     # `peer`: msweep_amd/csrc/peer_comm.hpp
                "what": "rank 0, profiled run (an event pair around every sweep and every all-reduce adds ~6 us of idle "
                        "GPU each: the sum exceeds ms_per_step); collectives = k_sum_scalar + all-reduce of 1 double "
                        "after pass A, k_colsum + all-reduce of 3 G integers + 4 doubles after pass B; chain_and_gaps = "
                        "k_finstep, k_redfin (redundant on every rank) and the launch boundaries"}
    if a.config == "cfg4":
        dt = max_over_ranks(dt)

    # the real replicate loop beside every cfg3 line: msw_core_bootstrap_dist, fixed replicates per rank
    boot = None
    if a.config == "cfg3" and not shard and not a.no_extras and a.bootstrap_per_rank > 0:
        core.set_fixed_iters(False)
        w = wl["w"]
        bootstrap_leg(core, boot_comm, w, 1, world, G)      # not timed: cumulative table, connections, clones
        sync()
        t1 = time.perf_counter()
        _, iters, bt = bootstrap_leg(core, boot_comm, w, a.bootstrap_per_rank, world, G)
        sync()
        tb = time.perf_counter() - t1
        tb = max_over_ranks(tb)
        B = a.bootstrap_per_rank * world
        boot = {"replicates": B, "per_rank": a.bootstrap_per_rank, "seconds": tb, "replicates_per_sec": B / tb,
                "iterations": [int(x) for x in iters], "cells_per_sec": float(E) * G * float(iters.sum()) / tb,
                "rank0_split_ms": {k: bt[k] for k in ("table_ms", "solve_ms", "gather_ms", "table_reused")},
                "what": "msw_core_bootstrap_dist(--seed 42): every rank seeks (GF(2) jump-ahead) to its block of the one "
                        "mt19937_64 stream, resamples and solves its replicates to --tol 1e-6, ONE all-gather of the "
                        "B x (G + 1) table; max over ranks of the host wall clock, barrier on both sides"}

    # the practical ceiling beside the specification: the rates this very device reaches with a
    # read-only 16-byte-load sweep and with the triad -- on a buffer the size of the sweeps' record stream (what
    # a kernel that does nothing but read that stream once would get, launch ramp and tail included) and on 4 GiB
    if rank == 0:
        stream_read, stream_triad = core.hbm_stream_rates(max(int(tm["bytes_passB"]), 1 << 20), 5)
        stream_read_4g, stream_triad_4g = core.hbm_stream_rates(1 << 32, 3)
        msA = tm["passA_ms"] / max(tm["passA_launches"], 1)
        msB = tm["passB_ms"] / max(tm["passB_launches"], 1)
        dom, ms_dom, b_dom = ("k_passB", msB, tm["bytes_passB"]) if msB >= msA else ("k_passA", msA, tm["bytes_passA"])
        achieved = b_dom / (ms_dom * 1e-3) / 1e9 if ms_dom > 0 else 0.0
        # HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE),
        # collected offline on this exact workload and committed under profiles/
        traffic, traffic_src, busy = None, None, None
        if a.default_shape:
            tkey = a.config + ("-diverse" if a.group_sizes == "diverse" else "")
            if a.config == "cfg2" and os.environ.get("MSWEEP_DENSE_COMPRESS") == "0":
                tkey = "cfg2-dense"
            for name in TRAFFIC_JSON.get(tkey, []):
                try:
                    tj = json.load(open(os.path.join(ROOT, "profiles", name)))
                    ent = next(v for k, v in tj.items() if dom[2:] in k)     # ("passB": k_passB<..>, k_dense_passB<..>)
                    traffic = ent["hbm_bytes_per_launch"]
                    busy = pmc_bound(ent)
                    traffic_src = f"profiles/{name} (offline rocprofv3 --pmc passes on this workload, not this run)"
                    break
                except Exception:
                    continue
        # what bounds the dominant kernel: the record stream is NOT it where the counters say that LDS (gathers, integer
        # atomics, half of the cycles bank conflicts) and VALU issue are both busy and overlap only in part
        bound = "hbm"
        if busy is not None:
            lb, vb = busy["lds_busy_frac"], busy["valu_busy_frac"]
            if vb >= 0.75 and lb < 0.3:
                bound = "valu (hbm frac reported)"          # the dense sweeps: one fp64 exp per cell
            elif lb >= 0.75 and vb < 0.3:
                bound = "lds (hbm frac reported)"
            elif max(lb, vb) >= 0.4:
                bound = "lds+valu (hbm frac reported)"
        gb = lambda b, ms: b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        sharding = "single solve" if n_gpus == 1 else (
            f"one solve, ECs sharded over {n_gpus} GPUs, {os.environ.get('MSWEEP_ALLREDUCE', 'rccl')} all-reduce of 1 double and "
            "of 3 G integers + 4 doubles per iteration" if shard else
            f"bootstrap replicates, 1 per GPU x {n_gpus}: every rank solves on its replicate's RESAMPLED counts (a third of "
            "the ECs at zero, one in fifty at four or more), a different trajectory from the original counts the "
            "single-GPU line solves on")
        if a.config == "cfg4":
            sharding = f"{a.steps} bootstrap replicates per GPU x {n_gpus}, contiguous blocks of the one random stream"
        line = {
            "metric": METRIC, "value": cells / dt, "unit": "cells/s",
            "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt * 1e3 / a.steps, "higher_is_better": True, "scaling": "strong" if shard else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl["desc"], "algorithm": "rcg", "reads": wl["reads"], "groups": G, "ecs": E, "nnz": nnz,
                       "seed": a.seed, "sharding": sharding},
            "value_counts": "logical EC x group cells of the matrix the reference holds (E * G per iteration); "
                            "listed_cells_per_sec counts the cells the CSR-of-ECs form stores (nnz per iteration)",
            "rccl_ranks": rccl_ranks,
            "prewarm": prewarm,
            "kernels": {"k_passA_ms": msA, "k_passB_ms": msB, "passA_launches": tm["passA_launches"],
                        "passB_launches": tm["passB_launches"]},
            "roofline": {"bound": bound, "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "lds_busy_frac": busy and busy["lds_busy_frac"], "lds_conflict_frac": busy and busy["lds_conflict_frac"],
                         "valu_busy_frac": busy and busy["valu_busy_frac"],
                         "bound_what": "achieved / peak / frac are ALGORITHMIC bytes of the kernel's record stream against the "
                                       "HBM peak, as the contract asks; lds_busy / valu_busy / lds_conflict are the kernel's "
                                       "SQ counters from the same committed PMC file as `traffic`: LDS-array cycles and VALU "
                                       "issue cycles as fractions of the kernel's duration, conflict cycles as a fraction "
                                       "of the LDS cycles (bench.py pmc_bound)",
                         "algorithmic_bytes_per_launch": b_dom, "avg_launch_ms": ms_dom,
                         "measured_stream_GBs": {"read_only_same_size": stream_read, "triad_same_size": stream_triad,
                                                 "read_only_4GiB": stream_read_4g, "triad_4GiB": stream_triad_4g,
                                                 "what": "msw_core_hbm_stream_rates on this device, best of 7 shapes x "
                                                         "5 launches; same_size = a buffer of algorithmic_bytes_per_launch"},
                         "frac_of_measured_read": achieved / stream_read if stream_read else None,
                         "kernels": {"k_passA": {"achieved": gb(tm["bytes_passA"], msA), "frac": gb(tm["bytes_passA"], msA) / HBM_PEAK_GBS},
                                     "k_passB": {"achieved": gb(tm["bytes_passB"], msB), "frac": gb(tm["bytes_passB"], msB) / HBM_PEAK_GBS}}},
            "setup_s": wl["setup_s"],
        }
        try:
            line["layout"] = core.layout_info()
        except Exception:
            pass
        line.update(line_extra)
        if "build" in wl:
            line["likelihood_build"] = wl["build"]
        if conv is not None:
            line["time_to_convergence"] = conv
        if wl.get("first_theta") is not None and n_gpus == 1 and not a.no_extras:
            try:
                line["time_to_first_theta"] = wl["first_theta"]()
            except Exception as ex:  # reporting only
                line["time_to_first_theta"] = {"ms": None, "what": f"failed: {ex}"}
        if wl.get("from_text") is not None and n_gpus == 1 and not a.no_extras and not a.no_text:
            log("text -> abundances ...")
            try:
                line["text_to_abundances"] = wl["from_text"]()
            except Exception as ex:  # reporting only
                line["text_to_abundances"] = {"seconds": None, "what": f"failed: {ex}"}
        if em is not None:
            line["em_algorithm"] = em
        if em_float is not None:
            line["em_algorithm_float"] = em_float
        if boot is not None:
            line["bootstrap_cfg4"] = boot
        if not a.no_cpu_baseline and n_gpus == 1:   # rank 0 at N = 1 only: the host cores are shared by the ranks
            log("cpu baseline ...")
            try:
                line.update(wl["cpu"]())
            except Exception as ex:  # the baseline is reporting only
                line.setdefault("cpu_baseline", {"value": None, "unit": "cells/s", "cores": 0, "kind": "port",
                                                 "sample": f"failed: {ex}"})
        print(json.dumps(line), file=OUT, flush=True)
    core.set_comm(None) if shard else None
    core.close()
    if comm is None:
        boot_comm.close()
    if multi:
        sync()
        comm.close()
        star.allgather(None)     # nobody tears its process down while a peer is still inside the last collective
        star.close()
    # (tests/test_gpu_multi.py, tests/test_parallel_cpu.py) the rank processes never import torch: its wheel bundles a
    # second ROCm runtime (libamdhip64 / librccl / HSA) next to the system one the library is linked against
    assert "torch" not in sys.modules, "bench.py: a rank process imported torch"


if __name__ == "__main__":
    main()
