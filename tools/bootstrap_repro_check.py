"""Developer check: run-to-run spread of bootstrap replicates at cfg4 size, 1 stream vs 1 stream and
1 stream vs 2 streams (same replicates): iteration counts and theta."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from msweep_amd import synth
from msweep_amd.core import Core
from msweep_amd.likelihood import from_grouped_counts

R, G, B = 10_000_000, 5000, 6
p = synth.make_csr_problem(R, G, seed=2)
core = Core(0)
lik = from_grouped_counts(core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
w = p["ec_counts"].astype(np.uint32)
draws = int(w.sum())
runs = []
for n in ("1", "1", "2", "2"):
    os.environ["MSWEEP_BOOTSTRAP_STREAMS"] = n
    th, it = core.bootstrap(w, 42, draws, 0, B, np.ones(G))
    runs.append((n, th, it))
    print("streams", n, "iterations", it.tolist(), flush=True)
ref = runs[0][1]
for n, th, it in runs[1:]:
    for floor in (1e-8, 1e-6, 1e-4):
        big = ref > floor
        print(f"streams {n} vs first run: max rel diff on theta > {floor:g}: {np.max(np.abs(th - ref)[big] / ref[big]):.2e}")
