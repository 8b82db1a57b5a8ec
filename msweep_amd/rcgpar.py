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


def _resolve(logl, log_times_observed, device):
    if isinstance(logl, Likelihood):
        return logl.core
    core = Core(device)
    from_dense(core, logl, log_times_observed)
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
