"""Developer timing: bootstrap replicates per second against the number of replicates in flight
(MSWEEP_BOOTSTRAP_STREAMS; default = heuristic of host_bootstrap.inc), small and cfg4-sized inputs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from msweep_amd import synth
from msweep_amd.core import Core
from msweep_amd.likelihood import from_grouped_counts

CASES = [(100_000, 200, 48), (1_000_000, 1000, 32), (3_000_000, 5000, 24), (10_000_000, 5000, 16)]
if len(sys.argv) > 1:
    CASES = [CASES[int(a)] for a in sys.argv[1:]]
core = Core(0)
for R, G, B in CASES:
    p = synth.make_csr_problem(R, G, seed=2)
    lik = from_grouped_counts(core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
    w = p["ec_counts"].astype(np.uint32)
    draws = int(w.sum())
    ref = None
    for n in ("1", "2", "4", "8", None):
        if n is None:
            os.environ.pop("MSWEEP_BOOTSTRAP_STREAMS", None)
        else:
            os.environ["MSWEEP_BOOTSTRAP_STREAMS"] = n
        core.bootstrap(w, 42, draws, 0, 2, np.ones(G))            # warm: streams, clones, tables
        t = time.perf_counter(); th, it = core.bootstrap(w, 42, draws, 0, B, np.ones(G)); dt = time.perf_counter() - t
        if ref is None:
            ref = th
        big = ref > 1e-8
        same = f"{np.max(np.abs(th - ref)[big] / ref[big]):.1e}"
        print(f"R={R} G={G}: streams={n or 'default'}: {B} replicates in {dt:.3f} s = {dt / B * 1e3:.2f} ms each "
              f"(mean {it.mean():.0f} iterations); max rel diff of theta (> 1e-8) to 1 stream: {same}", flush=True)
