"""Developer diagnostics on a GPU box: parity vs the oracle + quick timings (not a test)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from msweep_amd import synth
from msweep_amd.core import Core
from msweep_amd.likelihood import from_grouped_counts, precalc_lls, from_dense
from oracle import Oracle

O = Oracle()
O.set_num_threads(min(8, O.num_threads()))

def csr_case(R, G, seed, max_other=6, trace=30):
    p = synth.make_csr_problem(R, G, seed=seed, max_other=max_other)
    alpha = np.ones(G)
    lut = precalc_lls(p["group_sizes"])
    lutidx = (p["grp"].astype(np.uint32) * lut.shape[1] + p["cnt"]).astype(np.uint32)
    logc = np.log(p["ec_counts"].astype(float))
    with Core(0) as core:
        from_grouped_counts(core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
        core.set_trace_theta(trace)
        t = time.time(); res = core.solve(logc, alpha); dt = time.time() - t
        tr = core.trace(trace, with_theta=True)
        tm = core.last_timing()
    ref = O.rcg_optl_csr(p["rowptr"], p["grp"], lutidx, lut, np.log(0.01), G, logc, alpha, trace=trace)
    k = min(tr["n"], ref["iters"], trace)
    rt = ref["trace"]
    print(f"[csr R={R} G={G}] E={len(logc)} nnz={len(p['grp'])} iters gpu={res['iters']} oracle={ref['iters']} "
          f"bound gpu={res['bound']:.9f} oracle={ref['bound']:.9f} wall={dt:.3f}s solve_ms={tm['solve_ms']:.2f}")
    for kk in sorted(set([0, 1, 4, 9, 19, k - 1])):
        if 0 <= kk < k:
            print(f"   k={kk} dbound={tr['bound'][kk]-rt['bound'][kk]:.3e} nn_rel={tr['newnorm'][kk]/rt['newnorm'][kk]-1:.3e} "
                  f"reset {tr['didreset'][kk]}/{rt['didreset'][kk]} theta_rel={np.max(np.abs(tr['theta'][kk]-rt['theta'][kk])/rt['theta'][kk]):.3e}")
    err = np.abs(res["theta"] - ref["theta"])
    print(f"   final theta: max rel (theta>=1e-4) {np.max((err/ref['theta'])[ref['theta']>=1e-4], initial=0):.3e}  max abs {err.max():.3e}  sum {res['theta'].sum():.15f}")

def dense_case(E, G, seed, trace=30):
    p = synth.make_dense_problem(E, G, seed=seed)
    alpha = np.ones(G)
    with Core(0) as core:
        from_dense(core, p["logl"], p["logc"])
        core.set_trace_theta(trace)
        t = time.time(); res = core.solve(p["logc"], alpha); dt = time.time() - t
        tr = core.trace(trace, with_theta=True)
        tm = core.last_timing()
        g = core.gamma()
    ref = O.rcg_optl_dense(p["logl"], p["logc"], alpha, trace=trace)
    th = O.mixture_components(ref["gamma"], p["logc"])
    rt = ref["trace"]; k = min(tr["n"], ref["iters"], trace)
    print(f"[dense E={E} G={G}] iters gpu={res['iters']} oracle={ref['iters']} bound gpu={res['bound']:.9f} oracle={ref['bound']:.9f} wall={dt:.3f}s solve_ms={tm['solve_ms']:.2f}")
    for kk in sorted(set([0, 1, 4, 9, 19, k - 1])):
        if 0 <= kk < k:
            print(f"   k={kk} dbound={tr['bound'][kk]-rt['bound'][kk]:.3e} nn_rel={tr['newnorm'][kk]/rt['newnorm'][kk]-1:.3e} "
                  f"reset {tr['didreset'][kk]}/{rt['didreset'][kk]} theta_rel={np.max(np.abs(tr['theta'][kk]-rt['theta'][kk])/rt['theta'][kk]):.3e}")
    err = np.abs(res["theta"] - th)
    print(f"   final theta: max rel (theta>=1e-4) {np.max((err/th)[th>=1e-4], initial=0):.3e} max abs {err.max():.3e}; gamma max abs diff (prob) {np.abs(np.exp(g)-np.exp(ref['gamma'])).max():.3e}")

def timing_case(R, G, iters=50):
    t = time.time(); p = synth.make_csr_problem(R, G, seed=2); tg = time.time() - t
    alpha = np.ones(G); logc = np.log(p["ec_counts"].astype(float))
    E = len(logc); nnz = len(p["grp"])
    with Core(0) as core:
        t = time.time(); from_grouped_counts(core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"]); tu = time.time() - t
        core.set_fixed_iters(True); core.set_profiling(True)
        core.solve(logc, alpha, max_iters=5)
        res = core.solve(logc, alpha, max_iters=iters)
        tm = core.last_timing()
    print(f"[timing R={R} G={G}] gen {tg:.1f}s upload {tu:.1f}s E={E} nnz={nnz} iters={tm['iters']} solve_ms={tm['solve_ms']:.2f} "
          f"ms/iter={tm['solve_ms']/tm['iters']:.4f} passA avg ms={tm['passA_ms']/max(tm['passA_launches'],1):.4f} passB avg ms={tm['passB_ms']/max(tm['passB_launches'],1):.4f} "
          f"A GB/s={tm['bytes_passA']/(tm['passA_ms']/max(tm['passA_launches'],1)*1e-3)/1e9:.1f} B GB/s={tm['bytes_passB']/(tm['passB_ms']/max(tm['passB_launches'],1)*1e-3)/1e9:.1f}")

if __name__ == "__main__":
    print("start", flush=True)
    csr_case(3000, 60, 5)
    csr_case(50000, 300, 6)
    csr_case(200000, 1000, 7, max_other=15)
    dense_case(2000, 40, 3)
    dense_case(20000, 500, 4)
    timing_case(1_000_000, 5000)
    if "--big" in sys.argv:
        timing_case(10_000_000, 5000)
