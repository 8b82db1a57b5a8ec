import time, numpy as np, sys, os
sys.path.insert(0, os.getcwd())
from msweep_amd import synth
from msweep_amd.core import Core
from msweep_amd.likelihood import from_grouped_counts, precalc_lls
core = Core(0)
p = synth.make_csr_problem(10_000_000, 5000, seed=2)
lut = precalc_lls(p["group_sizes"])
for rep in range(3):
    t0 = time.perf_counter()
    core.set_csr(p["rowptr"], p["grp"], p["cnt"], lut, np.log(0.01), 5000)
    t1 = time.perf_counter()
    lc = np.log(p["ec_counts"].astype(float))
    res = core.solve(lc, np.ones(5000))
    t2 = time.perf_counter()
    print(f"cfg3 msw_core_set_csr {1e3*(t1-t0):.1f} ms; solve to --tol 1e-6 ({res['iters']} iterations) {1e3*(t2-t1):.1f} ms", flush=True)
