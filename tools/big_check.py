"""One-off scale check (cfg5-sized after --min-hits pruning): 50M reads x 2000 groups through set_csr,
the device packer and a solve; prints sizes, times and the invariants."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from msweep_amd import synth
from msweep_amd.core import Core
from msweep_amd.likelihood import from_grouped_counts

R = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
t = time.time(); p = synth.make_csr_problem(R, G, seed=3, max_other=7); tg = time.time() - t
E = len(p["rowptr"]) - 1; nnz = len(p["grp"])
print(f"generated R={R} G={G}: E={E} nnz={nnz} in {tg:.0f}s", flush=True)
with Core(0) as core:
    t = time.time(); lik = from_grouped_counts(core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"]); tu = time.time() - t
    print(f"set_csr + device pack: {tu:.2f}s", flush=True)
    t = time.time(); r = core.solve(lik.log_counts(), np.ones(G)); ts = time.time() - t
    tm = core.last_timing()
    N = r["theta"] * p["ec_counts"].sum()
    print(f"solve: {r['iters']} iterations in {ts*1e3:.0f} ms ({tm['solve_ms']/r['iters']*1e3:.0f} us/iter), sum theta = {r['theta'].sum():.15f}, "
          f"max |theta - theta_true| = {np.abs(r['theta'] - p['theta_true']).max():.2e}")
