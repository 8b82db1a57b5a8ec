"""Mirror of the rcgpar entry points mSWEEP's rcg_optl() dispatches to
(src/mSWEEP.cpp:176-205) and of rcgpar::mixture_components (src/mSWEEP.cpp:419-423).

Same names, argument order and error behaviour as the call sites; `logl` is a
`msweep_amd.likelihood.Likelihood` (device resident) or a dense G x E ndarray.  The returned
object plays the role of the `seamat::DenseMatrix<double>` gamma: theta is available without
materialising it, `.gamma()` materialises rows = groups on demand.
"""
import numpy as np

from .core import ALGO_EM, ALGO_RCG, PREC_DOUBLE, PREC_FLOAT, Core, MswError
from .likelihood import Likelihood, from_dense


class EcProbs:
    """Result of one estimation call (what Sample::store_probs receives, src/mSWEEP.cpp:402)."""

    def __init__(self, core, theta, iters, bound, log_times_observed):
        self.core, self.theta, self.iters, self.bound = core, theta, iters, bound
        self.log_times_observed = np.asarray(log_times_observed, np.float64)

    def gamma(self):
        return self.core.gamma()


# The dense `logl` of the unmodified call sites stays resident between calls: mSWEEP passes the same matrix once for
# the estimate and once per bootstrap replicate (src/mSWEEP.cpp:402,507).  One entry, keyed by (object identity,
# shape, a hash of 65 536 sampled cells, device); replaced -- and its handle closed -- when another matrix arrives.
# A matrix rewritten in place is caught with the probability of the sample: forget_likelihood() / MSWEEP_SHIM_CACHE=0.
_cache = {"key": None, "core": None, "uploads": 0, "hits": 0}


def _sample_hash(a):
    n = a.size
    if n == 0:
        return 0
    rng = np.random.Generator(np.random.PCG64(0x9e3779b97f4a7c15))
    pos = rng.integers(0, n, 65536)
    flat = a.reshape(-1) if a.flags.c_contiguous else None
    vals = flat[pos] if flat is not None else a[pos // a.shape[1], pos % a.shape[1]]
    corners = np.array([a[0, 0], a[0, -1], a[-1, 0], a[-1, -1]])
    return hash(np.concatenate([vals, corners]).tobytes())


def forget_likelihood():
    """Close the handle the dense call sites keep resident (the next call uploads again)."""
    if _cache["core"] is not None:
        _cache["core"].close()
    _cache["key"], _cache["core"] = None, None


def likelihood_cache_stats():
    return _cache["uploads"], _cache["hits"]


def _resolve(logl, log_times_observed, device):
    if isinstance(logl, Likelihood):
        return logl.core
    import os
    a = np.asarray(logl, np.float64)
    if a.ndim != 2:
        raise MswError("rcg_optl: expected a G x E matrix")
    if os.environ.get("MSWEEP_SHIM_CACHE", "1") == "0":
        forget_likelihood()
        key = None
    else:
        key = (id(logl), a.shape, _sample_hash(a), device)
        if _cache["core"] is not None and _cache["key"] == key:
            _cache["hits"] += 1
            return _cache["core"]
        forget_likelihood()
    core = Core(device)
    try:
        core.set_pack_schedule(False)     # the shortest way to the first estimate (msw_core_set_pack_schedule)
        from_dense(core, a, log_times_observed)
    except Exception:
        core.close()
        raise
    _cache["key"], _cache["core"] = key, core
    _cache["uploads"] += 1
    return core


def _solve(logl, log_times_observed, alpha0, tol, max_iters, log, algo, prec, device):
    core = _resolve(logl, log_times_observed, device)
    r = core.solve(log_times_observed, alpha0, tol, max_iters, algo, prec)
    if log is not None:
        t = core.trace(min(r["iters"], 4096))
        for k in range(0, t["n"], 5):  # rcgpar logs every 5th iteration
            log.write(f"  iter: {k}, bound: {t['bound'][k]}, |g|: {t['newnorm'][k]}\n")
    return EcProbs(core, r["theta"], r["iters"], r["bound"], log_times_observed)


def rcg_optl_torch(logl, log_times_observed, alpha0, tol, max_iters, log=None, device=0):
    """--algorithm rcggpu (src/mSWEEP.cpp:192-195)."""
    return _solve(logl, log_times_observed, alpha0, tol, max_iters, log, ALGO_RCG, PREC_DOUBLE, device)


def rcg_optl_omp(logl, log_times_observed, alpha0, tol, max_iters, log=None, device=0):
    """--algorithm rcgcpu, the reference's default (src/mSWEEP.cpp:196-199): the same RCG algorithm, served
    by the GPU kernels (this library has no CPU path)."""
    return rcg_optl_torch(logl, log_times_observed, alpha0, tol, max_iters, log, device)


def em_torch(logl, log_times_observed, alpha0, tol, max_iters, log=None, precision="double", device=0):
    """--algorithm emgpu / anything else (src/mSWEEP.cpp:200-203); --emprecision float|double."""
    if precision not in ("double", "float"):
        raise MswError(f"em_torch: unknown precision `{precision}`")
    prec = PREC_FLOAT if precision == "float" else PREC_DOUBLE
    return _solve(logl, log_times_observed, alpha0, tol, max_iters, log, ALGO_EM, prec, device)


def mixture_components_torch(probs, log_times_observed=None):
    """rcgpar::mixture_components_torch (src/mSWEEP.cpp:422,515): theta_g = sum_j exp(gamma_gj +
    logc_j) / sum_j c_j.  The column sums were already reduced on the device by the solve."""
    if isinstance(probs, EcProbs):
        return probs.theta
    raise MswError("mixture_components_torch: expected the EcProbs returned by rcg_optl_torch / em_torch")


def mixture_components(probs, log_times_observed=None):
    """rcgpar::mixture_components (src/mSWEEP.cpp:420,513)."""
    return mixture_components_torch(probs, log_times_observed)


def rcg_optl(algorithm, ll_mat, log_ec_counts, prior_counts, tol=1e-6, max_iters=5000, emprecision="double",
             log=None, device=0):
    """The dispatch wrapper itself (src/mSWEEP.cpp:176-205): rcggpu, rcgcpu (the default; the same
    algorithm, run by the GPU kernels), anything else -> EM."""
    if algorithm == "rcggpu":
        return rcg_optl_torch(ll_mat, log_ec_counts, prior_counts, tol, max_iters, log, device)
    if algorithm == "rcgcpu":
        return rcg_optl_omp(ll_mat, log_ec_counts, prior_counts, tol, max_iters, log, device)
    return em_torch(ll_mat, log_ec_counts, prior_counts, tol, max_iters, log, emprecision, device)
