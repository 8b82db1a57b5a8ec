"""Generates lut_golden.json: high-precision (mpmath, 50 digits) evaluation of the reference's
likelihood formulas -- ldbb_scaled (include/Likelihood.hpp:47-60), update_bb_parameters
(:198-207) and the precalc_lls table (:92-107) -- plus digamma values (src/Sample.cpp:87-97
approximates the true digamma).  Run:  python tests/golden/make_lut_golden.py
"""
import json
import os

import mpmath as mp

mp.mp.dps = 50


def lbeta(x, y):
    return mp.loggamma(x) + mp.loggamma(y) - mp.loggamma(x + y)


def ldbb_scaled(k, n, a, b):
    lbc = mp.loggamma(n + 1) - mp.loggamma(k + 1) - mp.loggamma(n - k + 1)
    return lbc + lbeta(k + a, n - k + b) - lbeta(n + a, b)


def bb_params(n, q, e):
    ex = mp.mpf(n) * q
    phi = 1 / (n - ex + e)
    beta = phi * (n - ex)
    alpha = (ex * beta) / (n - ex)
    return alpha, beta


cases = []
for (q, e, zi) in [(0.65, 0.01, 0.01), (0.5, 0.05, 0.001), (0.9, 0.001, 0.1)]:
    q_, e_, zi_ = mp.mpf(str(q)), mp.mpf(str(e)), mp.mpf(str(zi))
    for n in [1, 2, 3, 7, 10, 31, 100, 255, 1000]:
        a, b = bb_params(n, q_, e_)
        ks = sorted(set([1, 2, n // 2, n - 1, n]) & set(range(1, n + 1)))
        rows = []
        for k in ks:
            v = ldbb_scaled(k, n, a, b)
            rows.append({"k": k, "ldbb": float(v), "lut": float(v + mp.log1p(-zi_))})
        cases.append({"q": q, "e": e, "zi": zi, "n": n, "alpha": float(a), "beta": float(b),
                      "log_zi": float(mp.log(zi_)), "entries": rows})

digamma = [{"x": x, "psi": float(mp.digamma(mp.mpf(str(x))))}
           for x in [0.001, 0.05, 0.5, 1.0, 1.5, 2.0, 6.9, 7.0, 7.5, 10.0, 123.456, 1e4, 1e7, 3.3e9]]

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lut_golden.json")
json.dump({"generator": "tests/golden/make_lut_golden.py (mpmath %s, dps=50)" % mp.__version__,
           "cases": cases, "digamma": digamma}, open(out, "w"), indent=1)
print("wrote", out, len(cases), "cases")
