"""Developer timing of the dense (--read-likelihood) path at cfg2 scale: per-sweep times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from msweep_amd import synth
from msweep_amd.core import Core
from msweep_amd.likelihood import from_dense

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 500
p = synth.make_dense_problem(E, G, seed=1)
for mode in ("0", "1"):
  os.environ["MSWEEP_DENSE_COMPRESS"] = mode
  with Core(0) as core:
    t0 = time.time()
    from_dense(core, p["logl"], p["logc"])
    print(f"MSWEEP_DENSE_COMPRESS={mode}: set_dense_logl {time.time() - t0:.3f} s, cells resident {core.shape()[2]}")
    core.set_fixed_iters(True)
    core.prepare(p["logc"], np.ones(G))
    core.run(max_iters=5)
    t0 = time.time()
    core.run(max_iters=50)
    print(f"  {(time.time() - t0) / 50 * 1e3:.3f} ms/iter (wall)")
    core.set_profiling(True)
    core.run(max_iters=50)
    tm = core.last_timing()
    a = tm["passA_ms"] / tm["passA_launches"]; b = tm["passB_ms"] / tm["passB_launches"]
    print(f"dense E={E} G={G}: {tm['solve_ms'] / tm['iters']:.3f} ms/iter (profiled); pass A {a:.3f} ms = "
          f"{tm['bytes_passA'] / a / 1e6:.0f} GB/s, pass B {b:.3f} ms = {tm['bytes_passB'] / b / 1e6:.0f} GB/s")
