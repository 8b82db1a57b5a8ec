"""ctypes loader for the CPU ORACLE (oracle/libmsweep_oracle.so).

TEST INFRASTRUCTURE ONLY.  Importers allowed: tests/, __graft_entry__.smoke(), and the
`cpu_baseline` leg of bench.py.  Nothing under msweep_amd/ may import this module.

Parity status: see oracle/msweep_oracle.h ("parity unpinned" for the rcgpar loop; the
likelihood LUT, digamma and bootstrap stream are pinned by tests/golden/).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(force=False):
    """Compile the oracle with g++ (make).  Building the checker is not using it."""
    so = os.path.join(_HERE, "libmsweep_oracle.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


class _Opts(C.Structure):
    _fields_ = [("init_bound", C.c_double), ("weight_newnorm", C.c_int), ("max_trace", C.c_int),
                ("check_every", C.c_int), ("extended", C.c_int)]


class _EmOpts(C.Structure):
    _fields_ = [("prior_mode", C.c_int), ("stop_rule", C.c_int), ("check_every", C.c_int)]


class _Trace(C.Structure):
    _fields_ = [("bound", C.c_void_p), ("newnorm", C.c_void_p), ("beta", C.c_void_p),
                ("didreset", C.c_void_p), ("theta", C.c_void_p)]


_dp = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


class Oracle:
    def __init__(self, fastmath=False):
        build()
        name = "libmsweep_oracle_fastmath.so" if fastmath else "libmsweep_oracle.so"
        L = self.lib = C.CDLL(os.path.join(_HERE, name))
        L.orc_digamma.restype = C.c_double
        L.orc_digamma.argtypes = [C.c_double]
        L.orc_lbeta.restype = C.c_double
        L.orc_lbeta.argtypes = [C.c_double, C.c_double]
        L.orc_ldbb_scaled.restype = C.c_double
        L.orc_ldbb_scaled.argtypes = [C.c_uint64, C.c_uint64, C.c_double, C.c_double]
        L.orc_bb_params.argtypes = [_u64p, C.c_size_t, C.c_double, C.c_double, _dp, _dp]
        L.orc_precalc_lls.argtypes = [_u64p, C.c_size_t, C.c_double, C.c_double, C.c_double, _dp, C.c_size_t]
        L.orc_group_counts.argtypes = [_u64p, _u32p, C.c_size_t, _u32p, C.c_size_t, _u32p]
        L.orc_fill_ll_mat.restype = C.c_size_t
        L.orc_fill_ll_mat.argtypes = [_u32p, _u64p, C.c_size_t, _u64p, C.c_size_t, C.c_double, C.c_double,
                                      C.c_double, C.c_size_t, _dp, _u8p]
        L.orc_fill_ec_counts.argtypes = [_u64p, C.c_size_t, _dp]
        L.orc_rcg_optl_dense.restype = C.c_size_t
        L.orc_rcg_optl_dense.argtypes = [_dp, C.c_size_t, C.c_size_t, _dp, _dp, C.c_double, C.c_size_t,
                                         C.POINTER(_Opts), _dp, C.POINTER(C.c_double), C.POINTER(_Trace)]
        L.orc_rcg_optl_csr.restype = C.c_size_t
        L.orc_rcg_optl_csr.argtypes = [_u64p, _u32p, _u32p, _dp, C.c_size_t, C.c_double, C.c_size_t,
                                       C.c_size_t, _dp, _dp, C.c_double, C.c_size_t, C.POINTER(_Opts),
                                       _dp, C.c_void_p, C.POINTER(C.c_double), C.POINTER(_Trace)]
        L.orc_rcg_optl_dense_structured.restype = C.c_size_t
        L.orc_rcg_optl_dense_structured.argtypes = [_dp, C.c_size_t, C.c_size_t, _dp, _dp, C.c_double,
                                                    C.c_size_t, C.POINTER(_Opts), _dp, C.c_void_p,
                                                    C.POINTER(C.c_double), C.POINTER(_Trace)]
        L.orc_mixture_components.argtypes = [_dp, C.c_size_t, C.c_size_t, _dp, _dp]
        L.orc_em_dense.restype = C.c_size_t
        L.orc_em_dense.argtypes = [_dp, C.c_size_t, C.c_size_t, _dp, _dp, C.c_double, C.c_size_t,
                                   C.c_void_p, _dp, C.POINTER(C.c_double)]
        L.orc_em_dense_opts.restype = C.c_size_t
        L.orc_em_dense_opts.argtypes = [_dp, C.c_size_t, C.c_size_t, _dp, _dp, C.c_double, C.c_size_t,
                                        C.POINTER(_EmOpts), C.c_void_p, _dp, C.POINTER(C.c_double)]
        L.orc_bootstrap_counts_stdlib.argtypes = [_u32p, C.c_size_t, C.c_int32, C.c_size_t, C.c_size_t, _u32p]
        L.orc_bootstrap_counts_restated.argtypes = [_u32p, C.c_size_t, C.c_int32, C.c_size_t, C.c_size_t, _u32p]
        L.orc_bootstrap_counts_stdlib_from_state.argtypes = [_u32p, C.c_size_t, _u64p, C.c_uint64, C.c_size_t, C.c_size_t, _u32p]
        L.orc_discrete_cp.argtypes = [_u32p, C.c_size_t, _dp]
        L.orc_mt19937_64_words.argtypes = [C.c_uint64, C.c_size_t, C.c_size_t, _u64p]
        L.orc_dirichlet_kld_rate.argtypes = [_dp, C.c_size_t, C.c_size_t, _dp, _dp, _dp, _dp]
        L.orc_num_threads.restype = C.c_int
        L.orc_set_num_threads.argtypes = [C.c_int]

    # ---- scalars / likelihood ------------------------------------------------------
    def digamma(self, x):
        return self.lib.orc_digamma(float(x))

    def ldbb_scaled(self, k, n, alpha, beta):
        return self.lib.orc_ldbb_scaled(int(k), int(n), float(alpha), float(beta))

    def bb_params(self, sizes, q=0.65, e=0.01):
        sizes = np.ascontiguousarray(sizes, np.uint64)
        a = np.empty(len(sizes)); b = np.empty(len(sizes))
        self.lib.orc_bb_params(sizes, len(sizes), q, e, a, b)
        return a, b

    def precalc_lls(self, sizes, q=0.65, e=0.01, zi=0.01):
        sizes = np.ascontiguousarray(sizes, np.uint64)
        ld = int(sizes.max()) + 1
        lut = np.empty((len(sizes), ld))
        self.lib.orc_precalc_lls(sizes, len(sizes), q, e, zi, lut, ld)
        return lut

    def group_counts(self, tptr, targets, target_group, n_groups):
        tptr = np.ascontiguousarray(tptr, np.uint64)
        E = len(tptr) - 1
        counts = np.empty((n_groups, E), np.uint32)
        self.lib.orc_group_counts(tptr, np.ascontiguousarray(targets, np.uint32), E,
                                  np.ascontiguousarray(target_group, np.uint32), n_groups, counts)
        return counts

    def fill_ll_mat(self, counts, ec_counts, sizes, q=0.65, e=0.01, zi=0.01, min_hits=0):
        counts = np.ascontiguousarray(counts, np.uint32)
        G, E = counts.shape
        L = np.empty((G, E))
        mask = np.zeros(G, np.uint8)
        n = self.lib.orc_fill_ll_mat(counts, np.ascontiguousarray(ec_counts, np.uint64), E,
                                     np.ascontiguousarray(sizes, np.uint64), G, q, e, zi, min_hits, L, mask)
        return np.ascontiguousarray(L[:n]), mask.astype(bool)

    def fill_ec_counts(self, ec_counts):
        ec = np.ascontiguousarray(ec_counts, np.uint64)
        out = np.empty(len(ec))
        self.lib.orc_fill_ec_counts(ec, len(ec), out)
        return out

    # ---- optimiser -----------------------------------------------------------------
    def _opts_trace(self, G, trace, init_bound, weight_newnorm, check_every=1, extended=False):
        o = _Opts(init_bound, int(weight_newnorm), int(trace), int(check_every), int(bool(extended)))
        t = None
        arrs = None
        if trace:
            arrs = dict(bound=np.full(trace, np.nan), newnorm=np.full(trace, np.nan),
                        beta=np.full(trace, np.nan), didreset=np.full(trace, -1, np.int32),
                        theta=np.full((trace, G), np.nan))
            t = _Trace(*[arrs[k].ctypes.data for k in ("bound", "newnorm", "beta", "didreset", "theta")])
        return o, t, arrs

    def rcg_optl_dense(self, logl, logc, alpha0, tol=1e-6, max_iters=5000, trace=0,
                       init_bound=-100000.0, weight_newnorm=0, check_every=1, extended=False):
        """rcgpar::rcg_optl_omp restated; logl is G x E (rows = groups).  extended: the four G x E matrices and every
        sum over them in x87 extended precision (orc_rcg_opts::extended)."""
        logl = np.ascontiguousarray(logl, np.float64)
        G, E = logl.shape
        gamma = np.empty((G, E))
        o, t, arrs = self._opts_trace(G, trace, init_bound, weight_newnorm, check_every, extended)
        b = C.c_double()
        it = self.lib.orc_rcg_optl_dense(logl, G, E, np.ascontiguousarray(logc, np.float64),
                                         np.ascontiguousarray(alpha0, np.float64), tol, max_iters,
                                         C.byref(o), gamma, C.byref(b), C.byref(t) if t else None)
        return dict(gamma=gamma, iters=it, bound=b.value, trace=arrs)

    def rcg_optl_csr(self, rowptr, grp, lutidx, lut, logzi, G, logc, alpha0, tol=1e-6,
                     max_iters=5000, trace=0, want_gamma=False, init_bound=-100000.0, weight_newnorm=0,
                     check_every=1, extended=False):
        """the structured restatement on CSR-of-ECs; extended: its sweeps in x87 extended precision (orc_rcg_opts)"""
        rowptr = np.ascontiguousarray(rowptr, np.uint64)
        E = len(rowptr) - 1
        theta = np.empty(G)
        gamma = np.empty((G, E)) if want_gamma else None
        o, t, arrs = self._opts_trace(G, trace, init_bound, weight_newnorm, check_every, extended)
        b = C.c_double()
        lut = np.ascontiguousarray(lut, np.float64).ravel()
        it = self.lib.orc_rcg_optl_csr(rowptr, np.ascontiguousarray(grp, np.uint32),
                                       np.ascontiguousarray(lutidx, np.uint32), lut, len(lut), logzi, G, E,
                                       np.ascontiguousarray(logc, np.float64),
                                       np.ascontiguousarray(alpha0, np.float64), tol, max_iters, C.byref(o),
                                       theta, gamma.ctypes.data if want_gamma else None, C.byref(b),
                                       C.byref(t) if t else None)
        return dict(theta=theta, gamma=gamma, iters=it, bound=b.value, trace=arrs)

    def rcg_optl_dense_structured(self, logl, logc, alpha0, tol=1e-6, max_iters=5000, trace=0,
                                  want_gamma=False, init_bound=-100000.0, weight_newnorm=0, check_every=1):
        logl = np.ascontiguousarray(logl, np.float64)
        G, E = logl.shape
        theta = np.empty(G)
        gamma = np.empty((G, E)) if want_gamma else None
        o, t, arrs = self._opts_trace(G, trace, init_bound, weight_newnorm, check_every)
        b = C.c_double()
        it = self.lib.orc_rcg_optl_dense_structured(logl, G, E, np.ascontiguousarray(logc, np.float64),
                                                    np.ascontiguousarray(alpha0, np.float64), tol, max_iters,
                                                    C.byref(o), theta,
                                                    gamma.ctypes.data if want_gamma else None, C.byref(b),
                                                    C.byref(t) if t else None)
        return dict(theta=theta, gamma=gamma, iters=it, bound=b.value, trace=arrs)

    def mixture_components(self, gamma, logc):
        gamma = np.ascontiguousarray(gamma, np.float64)
        G, E = gamma.shape
        theta = np.empty(G)
        self.lib.orc_mixture_components(gamma, G, E, np.ascontiguousarray(logc, np.float64), theta)
        return theta

    def em_dense(self, logl, logc, alpha0, tol=1e-6, max_iters=5000, want_gamma=False, prior="map",
                 stop="gain", check_every=1):
        """rcgpar::em_torch restated [UPSTREAM-UNVERIFIED]; prior: "map" (alpha0 - 1 pseudo-counts) | "ml";
        stop: "gain" (log-likelihood gain < tol) | "theta" (largest move of a weight < tol)."""
        logl = np.ascontiguousarray(logl, np.float64)
        G, E = logl.shape
        theta = np.empty(G)
        gamma = np.empty((G, E)) if want_gamma else None
        b = C.c_double()
        eo = _EmOpts({"map": 0, "ml": 1}[prior], {"gain": 0, "theta": 1}[stop], int(check_every))
        it = self.lib.orc_em_dense_opts(logl, G, E, np.ascontiguousarray(logc, np.float64),
                                        np.ascontiguousarray(alpha0, np.float64), tol, max_iters, C.byref(eo),
                                        gamma.ctypes.data if want_gamma else None, theta, C.byref(b))
        return dict(theta=theta, gamma=gamma, iters=it, bound=b.value)

    def em_dense_f32(self, logl, logc, alpha0, tol=1e-6, max_iters=5000, prior="map", stop="gain", check_every=1, trace=0):
        """em_torch with precision "float" restated (orc_em_dense_f32): theta, iterations, the float log-likelihood."""
        logl = np.ascontiguousarray(logl, np.float64)
        G, E = logl.shape
        theta = np.empty(G)
        tr = np.full((trace, G), np.nan) if trace else None
        b = C.c_double()
        eo = _EmOpts({"map": 0, "ml": 1}[prior], {"gain": 0, "theta": 1}[stop], int(check_every))
        self.lib.orc_em_dense_f32.restype = C.c_size_t
        self.lib.orc_em_dense_f32.argtypes = [_dp, C.c_size_t, C.c_size_t, _dp, _dp, C.c_double, C.c_size_t, C.c_void_p,
                                              _dp, C.c_void_p, C.c_void_p, C.c_size_t]
        it = self.lib.orc_em_dense_f32(logl, G, E, np.ascontiguousarray(logc, np.float64),
                                       np.ascontiguousarray(alpha0, np.float64), tol, max_iters, C.addressof(eo), theta,
                                       C.addressof(b), tr.ctypes.data if trace else None, int(trace))
        return dict(theta=theta, iters=it, bound=b.value, theta_trace=tr)

    # ---- bootstrap -----------------------------------------------------------------
    def bootstrap_counts(self, weights, seed, bootstrap_count, n_reps, restated=False):
        w = np.ascontiguousarray(weights, np.uint32)
        out = np.empty((n_reps, len(w)), np.uint32)
        f = self.lib.orc_bootstrap_counts_restated if restated else self.lib.orc_bootstrap_counts_stdlib
        f(w, len(w), int(seed), int(bootstrap_count), int(n_reps), out)
        return out

    def bootstrap_counts_from_state(self, weights, state312, pos, bootstrap_count, n_reps=1):
        """libstdc++'s replicate loop from a state libstdc++ printed (tests/golden/mt_deep_state.json)."""
        w = np.ascontiguousarray(weights, np.uint32)
        st = np.ascontiguousarray(state312, np.uint64)
        assert st.shape == (312,)
        out = np.empty((n_reps, len(w)), np.uint32)
        self.lib.orc_bootstrap_counts_stdlib_from_state(w, len(w), st, int(pos), int(bootstrap_count), int(n_reps), out)
        return out

    def discrete_cp(self, weights):
        w = np.ascontiguousarray(weights, np.uint32)
        cp = np.empty(len(w))
        self.lib.orc_discrete_cp(w, len(w), cp)
        return cp

    def mt_words(self, seed, skip, n):
        out = np.empty(n, np.uint64)
        self.lib.orc_mt19937_64_words(int(seed) & 0xFFFFFFFFFFFFFFFF, int(skip), int(n), out)
        return out

    def dirichlet_kld_rate(self, gamma, logc):
        """Sample::dirichlet_kld + get_rates (src/Sample.cpp:99-152): (alphas, KLD, RATE)."""
        gamma = np.ascontiguousarray(gamma, np.float64)
        G, E = gamma.shape
        al, kld, rate = np.empty(G), np.empty(G), np.empty(G)
        self.lib.orc_dirichlet_kld_rate(gamma, G, E, np.ascontiguousarray(logc, np.float64), al, kld, rate)
        return al, kld, rate

    def num_threads(self):
        return self.lib.orc_num_threads()

    def set_num_threads(self, n):
        self.lib.orc_set_num_threads(int(n))
