"""Developer tool: is the per-iteration time of iterations 5..24 of a solve a property of those
iterations (data) or of how long the GPU has been busy (clocks)?  Repeats [fresh run(5), continue(20)]."""
import os
import sys
import time

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
for rep in range(8):
    core.run(max_iters=5)
    a = core.last_timing()["solve_ms"]
    core.continue_(20)
    t = core.last_timing()
    print(f"rep {rep}: run(5) {a:.3f} ms; continue(20): {t['solve_ms'] / 20:.4f} ms/slot, slots {t['passB_launches']}", flush=True)
print("late iterations of one solve:")
core.run(max_iters=100)
for rep in range(3):
    core.continue_(20)
    t = core.last_timing()
    print(f"  iterations {100 + 20 * rep}..: {t['solve_ms'] / 20:.4f} ms/slot, slots {t['passB_launches']}", flush=True)
print("and early ones again:")
for rep in range(2):
    core.run(max_iters=5)
    core.continue_(20)
    t = core.last_timing()
    print(f"  continue(20) after run(5): {t['solve_ms'] / 20:.4f} ms/slot, slots {t['passB_launches']}", flush=True)
print("bench.py's order: a full solve to convergence (polling every 16 iterations), then run(5) + continue(20):")
time.sleep(1.0)
core.set_fixed_iters(False)
t0 = time.perf_counter()
r = core.solve(lik.log_counts(), np.ones(5000))
print(f"  solve: {r['iters']} iterations, device {core.last_timing()['solve_ms'] / r['iters']:.4f} ms/iter, wall {(time.perf_counter() - t0) * 1e3:.1f} ms")
core.set_fixed_iters(True)
for rep in range(3):
    core.run(max_iters=5)
    core.continue_(20)
    t = core.last_timing()
    print(f"  continue(20) after run(5): {t['solve_ms'] / 20:.4f} ms/slot, slots {t['passB_launches']}", flush=True)
core.close()
