"""Developer tool: why is a 20-step timed region slower per step than a 200-step one?  Times bursts of
K fixed iterations back to back on the resident cfg3 likelihood (wall and device time per step, and the
sweeps' per-launch durations), with and without idle gaps in between."""
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


def burst(k, prof=False):
    core.set_profiling(prof)
    t0 = time.perf_counter()
    core.run(max_iters=k)
    dt = time.perf_counter() - t0
    tm = core.last_timing()
    s = f"K={k:4d}: wall {dt * 1e3 / k:.4f} ms/step, device {tm['solve_ms'] / k:.4f} ms/step"
    if prof:
        s += f", passA {tm['passA_ms'] / max(tm['passA_launches'], 1) * 1e3:.1f} us, passB {tm['passB_ms'] / max(tm['passB_launches'], 1) * 1e3:.1f} us"
    return s


def cont(k):
    core.set_profiling(False)
    t0 = time.perf_counter()
    core.continue_(k)
    dt = time.perf_counter() - t0
    tm = core.last_timing()
    return f"continue K={k:4d}: wall {dt * 1e3 / k:.4f} ms/step, device {tm['solve_ms'] / k:.4f} ms/step, slots {tm['passB_launches']}"


print("one solve, continued:")
print("  ", burst(5), flush=True)
for k in (20, 20, 20, 20, 100, 20, 20):
    print("  ", cont(k), flush=True)
print("cold start:")
for k in (5, 20, 20, 20, 200, 20, 20, 1000, 20, 20):
    print("  ", burst(k), flush=True)
print("after 2 s idle:")
time.sleep(2.0)
for k in (5, 20, 20, 200, 20):
    print("  ", burst(k), flush=True)
print("profiled bursts:")
for k in (20, 200, 20):
    print("  ", burst(k, True), flush=True)
print("after 0.2 s idle each:")
for k in (20, 20, 20):
    time.sleep(0.2)
    print("  ", burst(k), flush=True)
core.close()
