"""Developer: one small solve (typical single sample) for rocprofv3 --kernel-trace --stats."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from msweep_amd import synth
from msweep_amd.core import Core
from msweep_amd.likelihood import from_grouped_counts

R, G = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000, int(sys.argv[2]) if len(sys.argv) > 2 else 200
core = Core(0)
p = synth.make_csr_problem(R, G, seed=2, max_other=8)
lik = from_grouped_counts(core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
core.set_fixed_iters(True)
core.prepare(lik.log_counts(), np.ones(G))
core.run(max_iters=20)
import time
t = time.perf_counter(); core.run(max_iters=500); dt = time.perf_counter() - t
print(f"R={R} G={G}: {dt / 500 * 1e6:.1f} us/iter")
