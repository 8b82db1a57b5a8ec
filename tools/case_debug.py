"""Developer tool: bounds / theta of the first iterations of a saved fuzz case (tools/fuzz_parity.py FUZZ_ONLY)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from msweep_amd.core import Core  # noqa: E402

d = np.load(sys.argv[1])
core = Core(0)
G = len(d["alpha0"])
core.set_csr(d["rowptr"], d["grp"], d["cnt"], d["lut"], np.log(0.01), G)
core.set_trace_theta(8)
core.set_fixed_iters(True)
core.prepare(d["logc"], d["alpha0"])
r = core.run(max_iters=8)
t = core.trace(8, with_theta=True)
print("lib", os.environ.get("MSWEEP_CORE_LIB", "default"))
print("bound", t["bound"].tolist())
print("reset", t["didreset"].tolist())
print("theta sums", t["theta"].sum(1).tolist(), "min", t["theta"].min(1).tolist())
core.close()
