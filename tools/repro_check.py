"""Run-to-run reproducibility of a solve at cfg3 size (10 M reads x 5 k groups): N solves of the same
inputs on one handle + N bootstrap runs of the same replicates; iteration counts and theta must be
IDENTICAL (fixed-point integer column sums, sweep_kernels.hpp kFx).  With MSWEEP_CORE_LIB pointing at
an MSW_FX=0 build (fp64 LDS atomics) the spread of the old path is printed instead.
usage: repro_check.py [n_runs] [reads]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from msweep_amd import synth  # noqa: E402
from msweep_amd.core import Core  # noqa: E402
from msweep_amd.likelihood import from_grouped_counts  # noqa: E402

n_runs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
R, G, B = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000, 5000, 3
p = synth.make_csr_problem(R, G, seed=2)
core = Core(0)
lik = from_grouped_counts(core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
w = p["ec_counts"].astype(np.uint32)
draws = int(w.sum())
print("library:", os.environ.get("MSWEEP_CORE_LIB", "default (fixed-point column sums)"))
solves = [core.solve(lik.log_counts(), np.ones(G)) for _ in range(n_runs)]
its = [s["iters"] for s in solves]
same = all(np.array_equal(s["theta"], solves[0]["theta"]) and s["bound"] == solves[0]["bound"] for s in solves)
big = solves[0]["theta"] > 1e-6
spread = max(np.max(np.abs(s["theta"] - solves[0]["theta"])[big] / solves[0]["theta"][big]) for s in solves)
print(f"{n_runs} solves: iterations {its}; theta and bound bit-identical: {same}; max rel spread on theta > 1e-6: {spread:.2e}")
boots = []
for n in ["1", "2"] * (n_runs // 2):
    os.environ["MSWEEP_BOOTSTRAP_STREAMS"] = n
    boots.append(core.bootstrap(w, 42, draws, 0, B, np.ones(G)))
same_b = all(np.array_equal(t, boots[0][0]) and np.array_equal(i, boots[0][1]) for t, i in boots)
print(f"{len(boots)} bootstrap runs of {B} replicates (1 and 2 replicates in flight alternating): iterations "
      f"{[i.tolist() for _, i in boots]}; bit-identical: {same_b}")
core.close()
sys.exit(0 if (same and same_b) or "MSWEEP_CORE_LIB" in os.environ else 1)
