"""Host-side mirror of the reference's Likelihood interface for the hot path.

`Likelihood` keeps the accessor names of mSWEEP::Likelihood<T> (include/Likelihood.hpp:62-80:
log_mat(), log_counts(), groups_considered()) but owns a device-resident CSR-of-ECs (or
dense) matrix inside a `Core` handle instead of a seamat::DenseMatrix.
"""
import numpy as np

from .core import Core, MswError


def bb_params(group_sizes, q=0.65, e=0.01):
    """update_bb_parameters (include/Likelihood.hpp:198-207); bb_constants = {q, e}."""
    n = np.asarray(group_sizes, np.float64)
    ex = n * q
    phi = 1.0 / (n - ex + e)
    beta = phi * (n - ex)
    alpha = (ex * beta) / (n - ex)
    return alpha, beta


def _lbeta(x, y):
    from scipy.special import gammaln   # (imported where it is used: the drivers build on the device and never need it)
    return gammaln(x) + gammaln(y) - gammaln(x + y)


def precalc_lls(group_sizes, q=0.65, e=0.01, zero_inflation=0.01):
    """precalc_lls (include/Likelihood.hpp:92-107): G x (max_size+1) table, column 0 = log(zi),
    T[g][k] = ldbb_scaled(k, n_g, alpha_g, beta_g) + log1p(-zi) for 1 <= k <= n_g.  Entries
    k > n_g are never indexed (the reference leaves lgamma-of-negative garbage there); they are
    filled with log(zi)."""
    from scipy.special import gammaln
    sizes = np.asarray(group_sizes, np.int64)
    G, mx = len(sizes), int(sizes.max())
    alpha, beta = bb_params(sizes, q, e)
    k = np.arange(mx + 1, dtype=np.float64)[None, :]
    n = sizes[:, None].astype(np.float64)
    a, b = alpha[:, None], beta[:, None]
    with np.errstate(invalid="ignore", divide="ignore"):
        lbc = gammaln(n + 1) - gammaln(k + 1) - gammaln(np.maximum(n - k, 0) + 1)
        val = lbc + _lbeta(k + a, np.maximum(n - k, 0) + b) - _lbeta(n + a, b) + np.log1p(-zero_inflation)
    lut = np.where((k >= 1) & (k <= n), val, np.log(zero_inflation))
    return np.ascontiguousarray(lut)


class Likelihood:
    """Device-resident likelihood of one grouping (the object main() keeps at
    src/mSWEEP.cpp:294,346 and passes to rcg_optl at :402 and :507)."""

    def __init__(self, core: Core, log_counts, groups_considered, n_groups, n_ecs, ec_counts=None):
        self.core = core
        # (log_counts None: formed on first use from the EC counts -- a device build leaves them resident for the
        # solves, so the 8 * E byte download is only paid by callers that ask for the host vector)
        self._log_counts = None if log_counts is None else np.asarray(log_counts, np.float64)
        self._ec_counts = ec_counts
        self._mask = np.asarray(groups_considered, bool)
        self.n_groups = int(n_groups)
        self.n_ecs = int(n_ecs)

    # accessor names of include/Likelihood.hpp:72-79
    def log_counts(self):
        if self._log_counts is None:
            self._log_counts = np.log(np.asarray(self._ec_counts, np.float64))   # fill_ec_counts, include/Likelihood.hpp:188-195
        return self._log_counts

    def groups_considered(self):
        return self._mask

    def log_mat(self):
        """Dense G' x E matrix (rows = groups); materialised from the device on demand."""
        return self.core.get_dense_logl()


def from_grouped_counts(core: Core, rowptr, grp, cnt, ec_counts, group_sizes, q=0.65, e=0.01,
                        zero_inflation=0.01):
    """Upload an already-counted CSR-of-ECs (per EC: groups hit and how many sequences of each)
    -- the state after include/Likelihood.hpp:122-139 -- without --min-hits masking."""
    lut = precalc_lls(group_sizes, q, e, zero_inflation)
    core.set_csr(rowptr, grp, cnt, lut, np.log(zero_inflation), len(group_sizes))
    logc = np.log(np.asarray(ec_counts, np.float64))
    return Likelihood(core, logc, np.ones(len(group_sizes), bool), len(group_sizes), len(rowptr) - 1)


def from_alignment(core: Core, ec_tptr, ec_targets, target_group, group_sizes, ec_counts, q=0.65, e=0.01,
                   zero_inflation=0.01, min_hits=0, download_log_counts=True):
    """ConstructAdaptiveLikelihood (include/Likelihood.hpp:333-380) on the device.  download_log_counts=False: the
    log counts stay on the device only (solve(None, ...) uses them there); log_counts() then forms the host vector
    on demand."""
    n_kept, mask, logc = core.build_likelihood(ec_tptr, ec_targets, target_group, group_sizes, ec_counts,
                                               q, e, zero_inflation, min_hits, want_logc=download_log_counts)
    return Likelihood(core, logc, mask, n_kept, len(ec_tptr) - 1, ec_counts=None if download_log_counts else ec_counts)


class _LazyCounts:
    """ec_counts of a DeviceAlignment, copied out of device memory only if log_counts() is asked for on the host"""

    def __init__(self, aln):
        self._aln = aln

    def __array__(self, dtype=None, copy=None):
        a = self._aln.ec_counts()
        return a.astype(dtype) if dtype is not None else a


def from_device_alignment(core: Core, aln, target_group, group_sizes, q=0.65, e=0.01, zero_inflation=0.01, min_hits=0,
                          download_log_counts=False):
    """from_alignment on a DeviceAlignment (Core.read_alignment): classes, targets and read counts are consumed in
    device memory (msw_core_build_likelihood_aln); nothing of the pseudoalignment crosses PCIe twice."""
    n_kept, mask, logc = core.build_likelihood_aln(aln, target_group, group_sizes, q, e, zero_inflation, min_hits,
                                                   want_logc=download_log_counts)
    return Likelihood(core, logc, mask, n_kept, aln.n_ecs, ec_counts=None if download_log_counts else _LazyCounts(aln))


def from_dense(core: Core, logl, log_counts):
    """--read-likelihood path (include/Likelihood.hpp:224-253): arbitrary dense G x E matrix."""
    logl = np.asarray(logl, np.float64)
    if logl.ndim != 2:
        raise MswError("from_dense: expected a G x E matrix")
    core.set_dense_logl(logl)
    return Likelihood(core, log_counts, np.ones(logl.shape[0], bool), logl.shape[0], logl.shape[1])
