import time, numpy as np, sys, os
sys.path.insert(0, os.getcwd())
from msweep_amd import synth
from msweep_amd.core import Core
from msweep_amd.likelihood import from_grouped_counts, precalc_lls
core = Core(0)
p = synth.make_csr_problem(10_000_000, 5000, seed=2)
lut = precalc_lls(p["group_sizes"])
lc = np.log(p["ec_counts"].astype(float))
for rep in range(4):
    core.set_pack_schedule(rep < 2)       # two calls with the bank scheduling (default), two without
    t0 = time.perf_counter()
    core.set_csr(p["rowptr"], p["grp"], p["cnt"], lut, np.log(0.01), 5000)
    t1 = time.perf_counter()
    res = core.solve(lc, np.ones(5000))
    t2 = time.perf_counter()
    print(f"cfg3 bank scheduling {'on' if rep < 2 else 'off'}: msw_core_set_csr {1e3*(t1-t0):.1f} ms; solve to --tol 1e-6 ({res['iters']} iterations) {1e3*(t2-t1):.1f} ms", flush=True)
