"""Developer timing: what the slot area costs when it does not fit LDS -- the same reads and groups with group
sizes 1 + Poisson(9) (a few hundred used (size, count) pairs: LDS tables) and log-normal up to 400 (thousands of pairs:
the hybrid slot area of sell.hpp, or -- MSWEEP_HYBRID=0 -- per-slot tables gathered from memory)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from msweep_amd import synth
from msweep_amd.core import Core
from msweep_amd.likelihood import from_grouped_counts
shapes = [(2_000_000, 1000), (10_000_000, 5000)] if len(sys.argv) < 3 else [(int(sys.argv[1]), int(sys.argv[2]))]
for name, fn in (("poisson", None), ("diverse", synth.diverse_group_sizes)):
    for R, G in shapes:
        p = synth.make_csr_problem(R, G, seed=2, group_sizes=fn)
        core = Core(0)
        lik = from_grouped_counts(core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
        pairs = len(np.unique(p["group_sizes"][p["grp"]].astype(np.int64) * 1000 + p["cnt"]))
        core.set_fixed_iters(True); core.prepare(lik.log_counts(), np.ones(G))
        core.run(max_iters=100); core.run(max_iters=100)
        us = core.last_timing()["solve_ms"] * 10
        core.set_profiling(True); core.run(max_iters=50); t = core.last_timing(); core.set_profiling(False)
        print(name, R, G, "max size", int(p["group_sizes"].max()), "used (size,count) pairs", pairs, "cells", core.shape()[2],
              "us/iter", round(us, 1), "passA us", round(t["passA_ms"] / t["passA_launches"] * 1e3, 1),
              "passB us", round(t["passB_ms"] / t["passB_launches"] * 1e3, 1), core.layout_info(), flush=True)
        core.close()
