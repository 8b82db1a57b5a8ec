"""Developer tool for rocprofv3 --kernel-trace: three bursts of 20 fixed iterations on cfg3."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from msweep_amd import synth  # noqa: E402
from msweep_amd.core import Core  # noqa: E402
from msweep_amd.likelihood import from_grouped_counts  # noqa: E402

p = synth.make_csr_problem(10_000_000, 5000, seed=2)
core = Core(0)
lik = from_grouped_counts(core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
core.set_fixed_iters(True)
core.prepare(lik.log_counts(), np.ones(5000))
for k in (200, 20, 20, 20):
    core.run(max_iters=k)
    print(k, core.last_timing()["solve_ms"] / k)
core.close()
