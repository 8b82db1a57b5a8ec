"""Developer timing: a reference-shaped dense matrix (log(zi) background + lookup-table values, what
LL_WOR21::fill_ll_mat hands to rcg_optl) through msw_core_set_dense_logl, kept dense
(MSWEEP_DENSE_COMPRESS=0) vs re-expressed as CSR-of-ECs on the device (default)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from msweep_amd import synth
from msweep_amd.core import Core
from msweep_amd.likelihood import from_dense, precalc_lls
from conftest import dense_from_csr

R = int(sys.argv[1]) if len(sys.argv) > 1 else 1_200_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 500
p = synth.make_csr_problem(R, G, seed=2, max_other=15)
lut = precalc_lls(p["group_sizes"])
L = dense_from_csr(p, lut)
E = L.shape[1]
logc = np.log(p["ec_counts"].astype(float))
print(f"G={G} E={E} dense {L.nbytes / 1e9:.2f} GB, listed cells {len(p['grp'])} ({len(p['grp']) / L.size:.2%})", flush=True)
theta = {}
for mode in ("0", None):
    if mode is None:
        os.environ.pop("MSWEEP_DENSE_COMPRESS", None)
    else:
        os.environ["MSWEEP_DENSE_COMPRESS"] = mode
    with Core(0) as core:
        t0 = time.time()
        from_dense(core, L, logc)
        t_set = time.time() - t0
        nnz = core.shape()[2]
        core.set_fixed_iters(True)
        core.prepare(logc, np.ones(G))
        core.run(max_iters=5)
        t0 = time.time()
        core.run(max_iters=100)
        t_it = (time.time() - t0) / 100
        core.set_fixed_iters(False)
        t0 = time.time()
        res = core.solve(logc, np.ones(G))
        t_solve = time.time() - t0
        theta[mode] = res["theta"]
        print(f"{'dense sweeps' if mode == '0' else 'auto        '}: set_dense_logl {t_set:.3f} s, cells resident {nnz}, "
              f"{t_it * 1e3:.3f} ms/iter, solve to tol {t_solve * 1e3:.1f} ms / {res['iters']} iters", flush=True)
big = theta["0"] > 1e-8
print("max rel diff of theta between the two:", np.max(np.abs(theta[None] - theta["0"])[big] / theta["0"][big]))
