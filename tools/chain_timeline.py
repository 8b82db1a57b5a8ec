#!/usr/bin/env python3
"""Developer tool: where the time of one RCG iteration goes BETWEEN and INSIDE the four kernels of its chain
(k_passA -> k_finstep -> k_passB -> k_redfin), from s_memrealtime stamps (100 MHz) of a diagnostic build:

    python tools/ab_build.py stamps "-DMSW_STAMPS=1"
    MSWEEP_CORE_LIB=build_ab/lib_stamps.so python tools/chain_timeline.py [reads groups]

CAVEAT: with the stamps compiled in, k_finstep no longer fits its registers (it spills ~450 bytes per lane; the product
build spills nothing) and runs several times slower: read the sweeps', k_redfin's and the boundaries' numbers here, and
take k_finstep's duration from a rocprofv3 kernel trace of the product build (tools/collect_profiles.sh).
Prints the mean timeline of the iterations of a fixed-iteration solve (us from the start of pass A).  The stamps exist
only under MSW_STAMPS; the product build carries none."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msweep_amd import synth  # noqa: E402
from msweep_amd.core import Core  # noqa: E402
from msweep_amd.likelihood import from_grouped_counts  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
N = 60
p = synth.make_csr_problem(R, G, seed=2)
with Core(0) as core:
    L = core._L
    if not hasattr(L, "msw_debug_stamps"):
        sys.exit("this library carries no stamps: build with -DMSW_STAMPS=1 and pass it through MSWEEP_CORE_LIB")
    L.msw_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lik = from_grouped_counts(core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
    core.set_fixed_iters(True)
    core.prepare(lik.log_counts(), np.ones(G))
    core.run(max_iters=40)                      # warm clocks
    L.msw_debug_stamps(core._h, None, 1)
    core.continue_(N)
    st = np.zeros(64 * 40, np.uint64)
    L.msw_debug_stamps(core._h, st.ctypes.data_as(C.c_void_p), 0)
    t = core.last_timing()
st = st.reshape(64, 5, 8).astype(np.int64)
# One row of stamps = one EVALUATION r (Scalars::iter as the kernels read it): pass B and k_redfin of slot r, then pass A
# and k_finstep of slot r + 1 (the verdict on evaluation r is taken there and advances the count).
names = {(2, 0): "passB entry (wg 0)", (2, 1): "passB LDS filled (wg 0)", (2, 6): "passB last wg swept", (2, 7): "passB last wg rows out",
         (3, 0): "k_redfin entry (wg 0)", (3, 1): "k_redfin rows loaded (wg 0)", (3, 2): "k_redfin sums (wg 0)", (3, 7): "k_redfin last wg done",
         (0, 0): "passA entry (wg 0)", (0, 1): "passA LDS filled (wg 0)", (0, 7): "passA last wg done",
         (1, 0): "k_finstep entry", (1, 1): "k_finstep state + loads in", (1, 2): "k_finstep six sums (verdict follows)",
         (1, 3): "k_finstep u updated", (1, 4): "k_finstep tables built", (1, 5): "k_finstep max u", (1, 6): "k_finstep e_g, U"}
order = [k for kk in (2, 3, 0, 1) for k in sorted(names) if k[0] == kk]
rows = [i for i in range(64) if st[i, 2, 0] and st[i, 1, 6] and st[(i + 1) % 64, 2, 0] > st[i, 2, 0]]
print(f"{R} reads x {G} groups: {len(rows)} complete iterations; device time per iteration {t['solve_ms'] / max(t['iters'], 1) * 1e3:.1f} us")
acc = {k: [] for k in order}
nxt = []
for i in rows:
    t0 = st[i, 2, 0]
    for k in order:
        if st[i][k]:
            acc[k].append((st[i][k] - t0) / 100.0)
    nxt.append((st[(i + 1) % 64, 2, 0] - t0) / 100.0)
prev = 0.0
for k in order:
    if acc[k]:
        m = float(np.mean(acc[k]))
        print(f"  {m:8.2f} us  (+{m - prev:6.2f})  {names[k]}")
        prev = m
print(f"  {np.mean(nxt):8.2f} us  (+{np.mean(nxt) - prev:6.2f})  next iteration's passB entry")
