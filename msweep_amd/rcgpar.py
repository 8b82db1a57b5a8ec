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
    """Result of one estimation call (what Sample::store_probs receives, src/mSWEEP.cpp:402).

    The reference returns the dense gamma by value; here it is materialised on demand from the handle's solver state,
    which belongs to the LAST solve on that handle -- and the drop-ins below share one handle per likelihood.  So
    gamma() is valid until the next estimation call on the same likelihood and raises afterwards (it never hands out
    another call's matrix); callers that need it later pass keep_gamma=True to the call, which materialises it then."""

    def __init__(self, core, theta, iters, bound, log_times_observed, keep_gamma=False):
        self.core, self.theta, self.iters, self.bound = core, theta, iters, bound
        self.log_times_observed = np.asarray(log_times_observed, np.float64)
        self._generation = core.generation
        self._gamma = core.gamma() if keep_gamma else None

    def gamma(self):
        if self._gamma is not None:
            return self._gamma
        if self.core._h is None:
            raise MswError("EcProbs.gamma(): the likelihood handle of this result was closed (another matrix was passed to "
                           "the drop-ins, or forget_likelihood()); call gamma() before that, or pass keep_gamma=True")
        if self.core.generation != self._generation:
            raise MswError("EcProbs.gamma(): the handle has solved again since this result was returned -- its state "
                           "now describes the later call; call gamma() before the next estimation call on the same "
                           "likelihood, or pass keep_gamma=True")
        return self.core.gamma()


# The dense `logl` of the unmodified call sites stays resident between calls: mSWEEP passes the same matrix once for
# the estimate and once per bootstrap replicate (src/mSWEEP.cpp:402,507).  One entry, keyed by (object identity,
# shape, a hash of the matrix, device); replaced -- and its handle closed -- when another matrix arrives.  The hash is
# an xxh3 over EVERY cell up to 2^27 cells (1 GB: ~50 ms; a single cell edited in place is caught); above that, 65 536
# sampled cells + corners, which catch a rewrite with the probability of the sample: forget_likelihood() /
# MSWEEP_SHIM_CACHE=0 for callers that rewrite such a matrix in place.  MSWEEP_SHIM_FULL_HASH_CELLS moves the limit.
_cache = {"key": None, "core": None, "uploads": 0, "hits": 0}


def _matrix_hash(a):
    import os
    limit = int(os.environ.get("MSWEEP_SHIM_FULL_HASH_CELLS", 1 << 27))
    if 0 < a.size <= limit:
        import xxhash
        return ("full", xxhash.xxh3_64(memoryview(np.ascontiguousarray(a)).cast("B")).intdigest())
    return ("sampled", _sample_hash(a))


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
        key = (id(logl), a.shape, _matrix_hash(a), device)
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


def _solve(logl, log_times_observed, alpha0, tol, max_iters, log, algo, prec, device, keep_gamma=False):
    core = _resolve(logl, log_times_observed, device)
    r = core.solve(log_times_observed, alpha0, tol, max_iters, algo, prec)
    if log is not None:
        t = core.trace(min(r["iters"], 4096))
        for k in range(0, t["n"], 5):  # rcgpar logs every 5th iteration
            log.write(f"  iter: {k}, bound: {t['bound'][k]}, |g|: {t['newnorm'][k]}\n")
    return EcProbs(core, r["theta"], r["iters"], r["bound"], log_times_observed, keep_gamma)


def rcg_optl_torch(logl, log_times_observed, alpha0, tol, max_iters, log=None, device=0, keep_gamma=False):
    """--algorithm rcggpu (src/mSWEEP.cpp:192-195)."""
    return _solve(logl, log_times_observed, alpha0, tol, max_iters, log, ALGO_RCG, PREC_DOUBLE, device, keep_gamma)


def rcg_optl_omp(logl, log_times_observed, alpha0, tol, max_iters, log=None, device=0, keep_gamma=False):
    """--algorithm rcgcpu, the reference's default (src/mSWEEP.cpp:196-199): the same RCG algorithm, served
    by the GPU kernels (this library has no CPU path)."""
    return rcg_optl_torch(logl, log_times_observed, alpha0, tol, max_iters, log, device, keep_gamma)


def em_torch(logl, log_times_observed, alpha0, tol, max_iters, log=None, precision="double", device=0, keep_gamma=False):
    """--algorithm emgpu / anything else (src/mSWEEP.cpp:200-203); --emprecision float|double."""
    if precision not in ("double", "float"):
        raise MswError(f"em_torch: unknown precision `{precision}`")
    prec = PREC_FLOAT if precision == "float" else PREC_DOUBLE
    return _solve(logl, log_times_observed, alpha0, tol, max_iters, log, ALGO_EM, prec, device, keep_gamma)


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
             log=None, device=0, keep_gamma=False):
    """The dispatch wrapper itself (src/mSWEEP.cpp:176-205): rcggpu, rcgcpu (the default; the same
    algorithm, run by the GPU kernels), anything else -> EM."""
    if algorithm == "rcggpu":
        return rcg_optl_torch(ll_mat, log_ec_counts, prior_counts, tol, max_iters, log, device, keep_gamma)
    if algorithm == "rcgcpu":
        return rcg_optl_omp(ll_mat, log_ec_counts, prior_counts, tol, max_iters, log, device, keep_gamma)
    return em_torch(ll_mat, log_ec_counts, prior_counts, tol, max_iters, log, emprecision, device, keep_gamma)
