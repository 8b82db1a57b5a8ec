"""Developer timing of small / medium problems (typical single-sample mSWEEP runs)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from msweep_amd import synth
from msweep_amd.core import Core
from msweep_amd.likelihood import from_grouped_counts

core = Core(0)
for R, G in [(20_000, 50), (100_000, 200), (500_000, 1000), (2_000_000, 3000)]:
    p = synth.make_csr_problem(R, G, seed=2, max_other=8)
    lik = from_grouped_counts(core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
    core.solve(lik.log_counts(), np.ones(G))
    t = time.perf_counter(); r = core.solve(lik.log_counts(), np.ones(G)); dt = time.perf_counter() - t
    tm = core.last_timing()
    print(f"R={R} G={G} E={len(p['rowptr'])-1}: {r['iters']} iters, wall {dt*1e3:.2f} ms, device {tm['solve_ms']:.2f} ms, "
          f"{tm['solve_ms']/r['iters']*1e3:.1f} us/iter")
