"""Developer tool: randomised sweep of the device likelihood build (msw_core_build_likelihood: K0 lookup table,
K1 counts per (EC, group), K2 --min-hits mask + compaction) against the oracle's restatement of
LL_WOR21::fill_ll_mat (include/Likelihood.hpp:92-195): mask, expanded matrix, log counts, shape.
usage: fuzz_build.py [n_cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from msweep_amd.core import Core, MswError  # noqa: E402
from msweep_amd.likelihood import from_alignment  # noqa: E402
from oracle import Oracle  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
O = Oracle()
core = Core(0)
for case in range(n_cases):
    G = int(rng.choice([1, 2, 7, 64, 65, 300, 2000]))
    smax = int(rng.choice([1, 3, 40, 300]))
    sizes = rng.integers(1, smax + 1, G).astype(np.uint64)
    T = int(sizes.sum())
    target_group = rng.permutation(np.repeat(np.arange(G, dtype=np.uint32), sizes.astype(np.int64))).astype(np.uint32)
    E = int(rng.choice([1, 2, 63, 64, 65, 500, 5000]))
    E = max(1, min(E, int(4e6 // max(G, 1))))
    tmax = int(rng.choice([1, 5, 64, 65, 200, 1000]))
    lists = []
    for j in range(E):
        n = min(int(rng.integers(0, tmax + 1)), T)
        if rng.random() < 0.05:
            n = min(tmax, T)
        lists.append(np.sort(rng.choice(T, n, replace=False)) if n else np.zeros(0, np.int64))
    tptr = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.uint64)
    targets = (np.concatenate(lists) if tptr[-1] else np.zeros(0)).astype(np.uint32)
    ecc = rng.integers(1, int(rng.choice([2, 50, 100000])), E).astype(np.uint64)
    q, e, zi = float(rng.choice([0.65, 0.5, 0.9])), float(rng.choice([0.01, 0.05, 0.001])), float(rng.choice([0.01, 0.02, 0.3]))
    min_hits = int(rng.choice([0, 0, 1, 3, 100, 10**7]))
    tag = f"case {case}: G {G} sizes<= {smax} E {E} targets/EC<= {tmax} hits {len(targets)} min_hits {min_hits} q {q} e {e} zi {zi}"
    counts = O.group_counts(tptr, targets, target_group, G)
    hits = (counts > 0).astype(np.uint64) @ ecc                       # reads of the ECs that hit the group (:146-163)
    want_mask = hits >= min_hits if min_hits > 0 else np.ones(G, bool)
    try:
        try:
            lik = from_alignment(core, tptr, targets, target_group, sizes, ecc, q, e, zi, min_hits)
        except MswError as ex:
            assert not want_mask.any() and "removed every group" in str(ex), str(ex)
            continue
        L, mask = O.fill_ll_mat(counts, ecc, sizes, q, e, zi, min_hits)
        np.testing.assert_array_equal(mask, want_mask)
        np.testing.assert_array_equal(lik.groups_considered(), mask)
        assert lik.n_groups == int(mask.sum())
        np.testing.assert_allclose(lik.log_mat(), L, rtol=0, atol=1e-10)      # device lgamma vs glibc lgamma
        np.testing.assert_allclose(lik.log_counts(), np.log(ecc.astype(np.float64)), rtol=4e-16)   # device log: 1 ulp
        assert core.shape() == (int(mask.sum()), E, int((counts[mask] > 0).sum()))
        r = core.solve(None, np.ones(lik.n_groups), max_iters=30)
        assert np.all(np.isfinite(r["theta"])) and abs(r["theta"].sum() - 1.0) < 1e-9
        if lik.n_groups * E <= 2_000_000:                 # gamma, whole and in blocks (Sample.cpp:63-85's input)
            g = core.gamma()
            np.testing.assert_allclose(np.exp(g).sum(0), 1.0, rtol=1e-10)
            e0 = int(rng.integers(0, E))
            e1 = int(rng.integers(e0, E + 1))
            np.testing.assert_array_equal(core.gamma_block(e0, e1), g[:, e0:e1])
            # theta is the count-weighted mean of the responsibilities (rcgpar::mixture_components)
            np.testing.assert_allclose(np.exp(g) @ ecc.astype(np.float64) / ecc.sum(), r["theta"], rtol=1e-9, atol=1e-15)
    except Exception as ex:  # noqa: BLE001
        print("FAILED", tag, "::", str(ex)[:600], flush=True)
        sys.exit(1)
    if case % 10 == 0:
        print("ok", tag, flush=True)
print(f"{n_cases} cases passed")
core.close()
