"""Developer timing of the group-count cliff: above ~9.3k groups the per-group vectors no longer fit
the LDS image of the sweeps and are gathered from L2 / accumulated with global atomics."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from msweep_amd import synth
from msweep_amd.core import Core
from msweep_amd.likelihood import from_grouped_counts
for R, G in [(2_000_000, 8000), (2_000_000, 12000), (2_000_000, 20000)]:
    p = synth.make_csr_problem(R, G, seed=2, max_other=8)
    with Core(0) as core:
        lik = from_grouped_counts(core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
        core.set_fixed_iters(True); core.set_profiling(True)
        core.prepare(lik.log_counts(), np.ones(G)); core.run(max_iters=5); core.run(max_iters=30)
        tm = core.last_timing()
        print(f"R={R} G={G} nnz={len(p['grp'])}: {tm['solve_ms']/tm['iters']*1e3:.0f} us/iter; pass A {tm['passA_ms']/tm['passA_launches']*1e3:.0f} us, "
              f"pass B {tm['passB_ms']/tm['passB_launches']*1e3:.0f} us", flush=True)
