"""Developer tool: per-iteration differences HIP vs structured oracle vs dense-state oracle on the
dominant-group problems of tests/test_gpu_guard.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_guard import LOGZI, dense_of, isolate_problem  # noqa: E402
from msweep_amd.core import Core  # noqa: E402
from oracle import Oracle  # noqa: E402

alpha, deep = float(sys.argv[1]), float(sys.argv[2])
p = isolate_problem(seed=int(-deep) + int(alpha * 1000), deep=deep)
G = p["G"]
alpha0 = np.full(G, alpha)
logc = np.log(p["ec_counts"].astype(np.float64))
core = Core(0)
core.set_csr(p["rowptr"], p["grp"], p["cnt"], p["lut"], LOGZI, G)
core.set_trace_theta(30)
res = core.solve(logc, alpha0)
tr = core.trace(30, with_theta=True)
O = Oracle()
lutidx = (p["grp"] * p["lut"].shape[1] + p["cnt"]).astype(np.uint32)
s = O.rcg_optl_csr(p["rowptr"], p["grp"], lutidx, p["lut"], LOGZI, G, logc, alpha0, trace=30)["trace"]
d = O.rcg_optl_dense(dense_of(p), logc, alpha0, trace=30)["trace"]
csum = p["ec_counts"].sum()
for k in range(min(30, tr["n"])):
    a, b, c = tr["theta"][k], s["theta"][k], d["theta"][k]
    i = np.argmax(np.abs(a - b) / np.maximum(b, 1e-300))
    j = np.argmax(np.abs(a - c) / np.maximum(c, 1e-300))
    print(f"it {k} reset {tr['didreset'][k]}: hip-struct max rel {np.abs(a - b)[i] / b[i]:.2e} (g {i}, Nc {b[i] * csum:.3e}); "
          f"hip-dense {np.abs(a - c)[j] / c[j]:.2e} (g {j}, Nc {c[j] * csum:.3e}); struct-dense "
          f"{np.max(np.abs(b - c) / np.maximum(c, 1e-300)):.2e}; bound diff hs {tr['bound'][k] - s['bound'][k]:.2e} "
          f"hd {tr['bound'][k] - d['bound'][k]:.2e}")
for k in range(min(30, tr["n"])):
    a, b = tr["theta"][k], s["theta"][k]
    bad = np.nonzero(np.abs(a - b) > 1e-9 * np.abs(b) + 1e-15)[0]
    if len(bad):
        i = bad[np.argmax(np.abs(a - b)[bad])]
        print(f"it {k}: {len(bad)} violations; worst abs: g {i} hip {a[i]:.6e} struct {b[i]:.6e} dense {d['theta'][k][i]:.6e} "
              f"(reads: {a[i] * csum:.4e} vs {b[i] * csum:.4e})")
core.close()
