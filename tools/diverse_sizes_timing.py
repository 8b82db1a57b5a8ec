"""Developer timing: what the slot area costs when it does not fit LDS -- the same reads and groups with group
sizes 1 + Poisson(9) (a few hundred used (size, count) pairs: LDS tables) and log-normal up to 400 (thousands of pairs:
per-slot tables gathered from memory)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from msweep_amd import synth
from msweep_amd.core import Core
from msweep_amd.likelihood import from_grouped_counts
orig = synth._group_sizes
for name, fn in (("poisson", orig), ("diverse", lambda rng, G: np.minimum(1 + rng.lognormal(3.0, 1.2, G).astype(np.int64), 400).astype(np.uint64))):
    synth._group_sizes = fn
    for R, G in ((2_000_000, 1000), (10_000_000, 5000)):
        p = synth.make_csr_problem(R, G, seed=2)
        core = Core(0)
        lik = from_grouped_counts(core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
        pairs = len(np.unique(p["group_sizes"][p["grp"]].astype(np.int64) * 1000 + p["cnt"]))
        core.set_fixed_iters(True); core.prepare(lik.log_counts(), np.ones(G))
        core.run(max_iters=100); core.run(max_iters=100)
        print(name, R, G, "max size", int(p["group_sizes"].max()), "used (size,count) pairs", pairs, "cells", core.shape()[2],
              "us/iter", round(core.last_timing()["solve_ms"] * 10, 1), flush=True)
        core.close()
