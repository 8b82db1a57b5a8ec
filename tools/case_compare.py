"""Developer tool: where do the HIP path and the structured oracle part ways on a saved fuzz case?"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from msweep_amd.core import Core  # noqa: E402
from oracle import Oracle  # noqa: E402

d = np.load(sys.argv[1])
G = len(d["alpha0"])
core = Core(0)
core.set_csr(d["rowptr"], d["grp"], d["cnt"], d["lut"], np.log(0.01), G)
core.set_trace_theta(1024)
res = core.solve(d["logc"], d["alpha0"], max_iters=1000)
k = res["iters"]
tr = core.trace(k, with_theta=True)
lutidx = (d["grp"] * d["lut"].shape[1] + d["cnt"]).astype(np.uint32)
O = Oracle()
rt = O.rcg_optl_csr(d["rowptr"], d["grp"], lutidx, d["lut"], np.log(0.01), G, d["logc"], d["alpha0"], tol=-1.0,
                    max_iters=k, trace=k)["trace"]
print("iterations", k)
last = -1
for i in range(k):
    a, b = tr["theta"][i], rt["theta"][i]
    big = b >= 1e-4
    rel = np.max(np.abs(a - b)[big] / b[big], initial=0.0)
    flag = tr["didreset"][i] != rt["didreset"][i]
    if flag or rel > 10 ** (last + 1) * 1e-14 or i < 3 or i == k - 1:
        print(f"it {i}: rel {rel:.2e} resets {tr['didreset'][i]}/{rt['didreset'][i]} bound {tr['bound'][i]:.12e} / {rt['bound'][i]:.12e} "
              f"gain {tr['bound'][i] - tr['bound'][i - 1] if i else 0:.3e} / {rt['bound'][i] - rt['bound'][i - 1] if i else 0:.3e} beta {tr['beta'][i]:.6e} / {rt['beta'][i]:.6e}")
        while rel > 10 ** (last + 1) * 1e-14:
            last += 1
core.close()
