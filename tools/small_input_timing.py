"""Developer tool: per-iteration time of small (single-sample sized) inputs -- where the iteration is bounded by
the five kernel boundaries, not by the sweeps.  usage: small_input_timing.py [reads ...]   (with rocprofv3
--kernel-trace: per-kernel durations through tools/trace_gaps.py)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from msweep_amd import synth  # noqa: E402
from msweep_amd.core import Core  # noqa: E402
from msweep_amd.likelihood import from_grouped_counts  # noqa: E402

sizes = [int(x) for x in sys.argv[1:]] or [100_000, 1_000_000]
for R in sizes:
    for G in (500, 5000):
        p = synth.make_csr_problem(R, G, seed=2)
        core = Core(0)
        lik = from_grouped_counts(core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
        core.set_fixed_iters(True)
        core.prepare(lik.log_counts(), np.ones(G))
        out = []
        for k in (200, 200, 200):
            core.run(max_iters=k)
            out.append(core.last_timing()["solve_ms"] / k * 1e3)
        print(f"reads {R} groups {G} ECs {core.shape()[1]} listed cells {core.shape()[2]}: "
              + " ".join(f"{x:.1f}" for x in out) + " us/iteration", flush=True)
        core.close()
