"""Developer tool: HIP vs structured oracle vs dense-state oracle per iteration on a saved fuzz case."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from msweep_amd.core import Core  # noqa: E402
from oracle import Oracle  # noqa: E402

d = np.load(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
G = len(d["alpha0"])
E = len(d["rowptr"]) - 1
core = Core(0)
core.set_csr(d["rowptr"], d["grp"], d["cnt"], d["lut"], np.log(0.01), G)
core.set_trace_theta(n)
core.solve(d["logc"], d["alpha0"], tol=-1.0, max_iters=n)
tr = core.trace(n, with_theta=True)
lutidx = (d["grp"] * d["lut"].shape[1] + d["cnt"]).astype(np.uint32)
O = Oracle()
s = O.rcg_optl_csr(d["rowptr"], d["grp"], lutidx, d["lut"], np.log(0.01), G, d["logc"], d["alpha0"], tol=-1.0, max_iters=n, trace=n)["trace"]
L = np.full((G, E), np.log(0.01))
L[d["grp"], np.repeat(np.arange(E), np.diff(d["rowptr"].astype(np.int64)))] = d["lut"][d["grp"], d["cnt"]]
dd = O.rcg_optl_dense(L, d["logc"], d["alpha0"], tol=-1.0, max_iters=n, trace=n)["trace"]


def rel(a, b):
    m = b >= 1e-6
    return np.max(np.abs(a - b)[m] / b[m], initial=0.0)


for i in range(n):
    print(f"it {i}: resets {tr['didreset'][i]}/{s['didreset'][i]}/{dd['didreset'][i]}  hip-struct {rel(tr['theta'][i], s['theta'][i]):.2e}  "
          f"hip-dense {rel(tr['theta'][i], dd['theta'][i]):.2e}  struct-dense {rel(s['theta'][i], dd['theta'][i]):.2e}  "
          f"bound {tr['bound'][i]:.10e} {s['bound'][i]:.10e} {dd['bound'][i]:.10e}  beta {tr['beta'][i]:.3e}")
core.close()
