import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from msweep_amd import synth
from msweep_amd.core import Core
from msweep_amd.likelihood import from_grouped_counts
core = Core(0)
for R in (7500, 15000, 30000, 60000, 120000):
    p = synth.make_csr_problem(R, 3000, seed=2, max_other=2500)
    E, nnz = len(p["rowptr"]) - 1, len(p["grp"])
    lik = from_grouped_counts(core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
    core.set_fixed_iters(True); core.prepare(lik.log_counts(), np.ones(3000)); core.run(max_iters=5)
    core.set_profiling(True); core.run(max_iters=30); tm = core.last_timing(); core.set_profiling(False); core.set_fixed_iters(False)
    a = tm["passA_ms"] / tm["passA_launches"]; b = tm["passB_ms"] / tm["passB_launches"]
    print(f"E={E} nnz={nnz}: pass A {a*1e3:.1f} us, pass B {b*1e3:.1f} us", flush=True)
