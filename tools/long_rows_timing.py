"""Developer timing: ECs with many listed groups (conserved sequence hits most groups): the streaming
branch (17..256 cells per EC) and the long-EC path (> 256) of the sweeps."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from msweep_amd import synth
from msweep_amd.core import Core
from msweep_amd.likelihood import from_grouped_counts

core = Core(0)
CASES = [(2_000_000, 1000, 6), (2_000_000, 1000, 40), (1_000_000, 1000, 200), (200_000, 1000, 800), (60_000, 3000, 2500)]
if len(sys.argv) == 2:
    CASES = CASES[int(sys.argv[1]):]
elif len(sys.argv) > 2:   # reads groups max_other [reads groups max_other ...]
    a = [int(x) for x in sys.argv[1:]]
    CASES = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)]
print("MSWEEP_LONG_ROW =", os.environ.get("MSWEEP_LONG_ROW", "(default)"))
for R, G, mo in CASES:
    p = synth.make_csr_problem(R, G, seed=2, max_other=mo)
    E, nnz = len(p["rowptr"]) - 1, len(p["grp"])
    lik = from_grouped_counts(core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
    core.set_fixed_iters(True)
    core.prepare(lik.log_counts(), np.ones(G))
    core.run(max_iters=5)
    core.set_profiling(True)
    core.run(max_iters=50)
    tm = core.last_timing()
    core.set_profiling(False)
    core.set_fixed_iters(False)
    a = tm["passA_ms"] / tm["passA_launches"]; b = tm["passB_ms"] / tm["passB_launches"]
    print(f"R={R} G={G} max_other={mo}: E={E} nnz={nnz} ({nnz / E:.1f}/EC, {nnz / (E * G):.1%} of G x E): "
          f"pass A {a * 1e3:.1f} us ({nnz / a / 1e6:.0f} Gcell/s), pass B {b * 1e3:.1f} us ({nnz / b / 1e6:.0f} Gcell/s), "
          f"{tm['solve_ms'] / tm['iters'] * 1e3:.1f} us/iter", flush=True)
