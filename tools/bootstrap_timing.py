"""Developer timing of cfg4 (bootstrap at 10M reads x 5k groups): resampling vs solve per replicate."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from msweep_amd import synth
from msweep_amd.core import Core
from msweep_amd.likelihood import from_grouped_counts

R = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
p = synth.make_csr_problem(R, 5000, seed=2)
core = Core(0)
lik = from_grouped_counts(core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
w = p["ec_counts"].astype(np.uint32)
draws = int(w.sum())
t = time.perf_counter(); c = core.resample_counts(w, 42, draws, 0, 1); t1 = time.perf_counter() - t
t = time.perf_counter(); c = core.resample_counts(w, 42, draws, 1, 3); t2 = time.perf_counter() - t
print(f"resample: first replicate {t1*1e3:.1f} ms (includes cp table build), next two {t2*1e3/2:.1f} ms each (incl. 37 MB D2H)")
t = time.perf_counter(); th, it = core.bootstrap(w, 42, draws, 0, 4, np.ones(5000)); t3 = time.perf_counter() - t
print(f"bootstrap: 4 replicates in {t3:.3f} s = {t3/4*1e3:.1f} ms per replicate, iterations {it.tolist()}")
t = time.perf_counter(); r = core.solve(lik.log_counts(), np.ones(5000)); t4 = time.perf_counter() - t
print(f"plain solve: {t4*1e3:.1f} ms, {r['iters']} iterations")
