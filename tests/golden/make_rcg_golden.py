"""Generates rcg_golden.json: trajectories of the RCG loop (rcgpar v1.2.1's
published rcg_optl_mat) from an independent pure-numpy restatement written for this purpose.

NOT reference-derived: the reference tree holds no tests or vectors for this path and rcgpar is
not vendored ("parity unpinned", oracle/msweep_oracle.h).  The fixture pins the C oracle, the
structured formulation and the HIP path to one another and to this twin: early iterations tightly,
the converged theta loosely (the recursion amplifies rounding).
Run:  python tests/golden/make_rcg_golden.py
"""
import json
import os

import numpy as np
from scipy.special import gammaln, logsumexp


def digamma_ref(x):  # src/Sample.cpp:87-97
    r = 0.0
    while x < 7:
        r -= 1 / x
        x += 1
    x -= 0.5
    xx = 1.0 / x
    xx2 = xx * xx
    xx4 = xx2 * xx2
    return r + np.log(x) + (1. / 24.) * xx2 - (7.0 / 960.0) * xx4 + (31.0 / 8064.0) * xx4 * xx2 - (127.0 / 30720.0) * xx4 * xx4


def rcg_numpy(L, logc, alpha0, tol=1e-6, max_iters=5000, init_bound=-100000.0, record=25):
    G, E = L.shape
    c = np.exp(logc)
    gamma = np.full((G, E), np.log(1.0 / G))
    oldstep = np.zeros((G, E))
    oldnorm, bound, didreset = 1.0, init_bound, False
    bound_const = gammaln(alpha0.sum()) - gammaln(alpha0.sum() + c.sum()) - gammaln(alpha0).sum()

    def Nk(g):
        return (np.exp(g) * c[None, :]).sum(1) + alpha0

    def elbo(g, N):
        w = np.exp(g) * c[None, :]
        return bound_const + np.where(w > 0, w * (L - g), 0.0).sum() + gammaln(N).sum()

    N = Nk(gamma)
    tr = dict(bound=[], newnorm=[], beta=[], didreset=[], theta=[])
    it = 0
    for k in range(max_iters):
        d = np.array([digamma_ref(x) for x in N]) - 1.0
        step = L + d[:, None] - gamma
        q = np.exp(gamma)
        colsum = (step * q).sum(0)
        newnorm = (q * (step - colsum[None, :]) * step).sum()
        beta = newnorm / oldnorm
        oldnorm = newnorm
        if didreset:
            oldstep = oldstep * 0.0
        elif beta > 0:
            oldstep = oldstep * beta
            step = step + oldstep
        didreset = False
        gamma = gamma + step
        m = logsumexp(gamma, axis=0)
        gamma = gamma - m[None, :]
        N = Nk(gamma)
        oldbound = bound
        bound = elbo(gamma, N)
        if bound < oldbound:
            didreset = True
            gamma = gamma + m[None, :]
            if beta > 0:
                gamma = gamma - oldstep
            m = logsumexp(gamma, axis=0)
            gamma = gamma - m[None, :]
            N = Nk(gamma)
            bound = elbo(gamma, N)
        else:
            oldstep = step
        if k < record:
            tr["bound"].append(float(bound)); tr["newnorm"].append(float(newnorm)); tr["beta"].append(float(beta))
            tr["didreset"].append(int(didreset)); tr["theta"].append(((N - alpha0) / c.sum()).tolist())
        it = k + 1
        if bound - oldbound < tol and not didreset:
            break
    theta = (np.exp(gamma) * c[None, :]).sum(1) / c.sum()
    return dict(iters=it, bound=float(bound), theta=theta.tolist(), trace=tr)


def problem(rng, G, E, zi=0.01, maxc=40, zero_counts=False):
    L = np.full((G, E), np.log(zi))
    for j in range(E):
        ns = rng.integers(1, min(G, 5) + 1)
        gs = rng.choice(G, ns, replace=False)
        L[gs, j] = np.clip(rng.normal(-2.0, 1.5, ns), -15, -0.05)
    cnt = rng.integers(1, maxc, E).astype(float)
    if zero_counts:
        cnt[rng.random(E) < 0.3] = 0.0
    with np.errstate(divide="ignore"):
        logc = np.log(cnt)
    return L, logc


if __name__ == "__main__":
    rng = np.random.Generator(np.random.PCG64(20231))
    cases = []
    specs = [("tiny", 4, 12, False, 1.0), ("small", 7, 40, False, 1.0), ("medium", 16, 200, False, 1.0),
             ("zero_counts", 6, 60, True, 1.0), ("alphas", 5, 50, False, None), ("one_group", 1, 9, False, 1.0),
             ("big_counts_reset", 8, 120, False, 1.0)]
    for name, G, E, zc, al in specs:
        L, logc = problem(rng, G, E, zero_counts=zc, maxc=4000 if name.startswith("big") else 40)
        alpha0 = np.full(G, al) if al is not None else rng.uniform(0.2, 3.0, G)
        r = rcg_numpy(L, logc, alpha0)
        cases.append(dict(name=name, G=G, E=E, logl=L.tolist(),
                          logc=[None if not np.isfinite(x) else float(x) for x in logc],
                          alpha0=alpha0.tolist(), tol=1e-6, max_iters=5000, expect=r))
        print(name, "iters", r["iters"], "bound", r["bound"], "resets", sum(r["trace"]["didreset"]))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rcg_golden.json")
    json.dump(dict(generator="tests/golden/make_rcg_golden.py (pure-numpy twin; NOT reference-derived)",
                   cases=cases), open(out, "w"))
    print("wrote", out, os.path.getsize(out), "bytes")
