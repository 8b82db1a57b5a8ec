"""Developer tool: gains (bound - oldbound) of the HIP path and of the oracle around the stop, on a
bootstrap replicate of cfg3 (why do the two stop one iteration apart?)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import cpu_share, lutidx_of  # noqa: E402
from msweep_amd import synth  # noqa: E402
from msweep_amd.core import Core  # noqa: E402
from msweep_amd.likelihood import from_grouped_counts, precalc_lls  # noqa: E402
from oracle import Oracle  # noqa: E402

reads, G = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, 5000
p = synth.make_csr_problem(reads, G, seed=2)
lut = precalc_lls(p["group_sizes"])
O = Oracle()
O.set_num_threads(cpu_share())
core = Core(0)
from_grouped_counts(core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
w = p["ec_counts"].astype(np.uint32)
counts = core.resample_counts(w, 42, int(w.sum()), 0, 1)[0]
with np.errstate(divide="ignore"):
    logc = np.log(counts.astype(float))
for name, lc in (("original counts", np.log(p["ec_counts"].astype(float))), ("replicate 0", logc)):
    core.set_trace_theta(0)
    res = core.solve(lc, np.ones(G))
    k = res["iters"]
    tr = core.trace(k)
    rt = O.rcg_optl_csr(p["rowptr"], p["grp"], lutidx_of(p, lut), lut, np.log(0.01), G, lc, np.ones(G), tol=-1.0,
                        max_iters=k + 2, trace=k + 2)["trace"]
    print(name, "hip iterations", k)
    for i in range(max(k - 5, 1), k + 2):
        gh = tr["bound"][i] - tr["bound"][i - 1] if i < k else float("nan")
        go = rt["bound"][i] - rt["bound"][i - 1]
        print(f"  iteration {i + 1}: gain hip {gh:.4e} oracle {go:.4e}   bound hip-oracle "
              f"{(tr['bound'][i] - rt['bound'][i]) if i < k else float('nan'):.3e}")
core.close()
