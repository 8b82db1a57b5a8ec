"""ctypes binding of libmsweep_core.so (the C ABI in include/msweep_core.h).

Fails loudly when the HIP extension is missing or no GPU is visible: there is no CPU
fallback in the product path (the CPU oracle under oracle/ is test infrastructure and is
never imported from here).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MSWEEP_CORE_LIB: developer override for A/B timing of two builds in one job
LIB_PATH = os.environ.get("MSWEEP_CORE_LIB") or os.path.join(_HERE, "libmsweep_core.so")

ALGO_RCG, ALGO_EM = 0, 1
PREC_DOUBLE, PREC_FLOAT = 0, 1
# msw_core_set_option ids (include/msweep_core.h): the knobs of rcgpar's loops that are restated from memory
OPT_CHECK_EVERY, OPT_INIT_BOUND, OPT_EM_PRIOR, OPT_EM_STOP = 0, 1, 2, 3
_OPTS = {"check_every": OPT_CHECK_EVERY, "init_bound": OPT_INIT_BOUND, "em_prior": OPT_EM_PRIOR, "em_stop": OPT_EM_STOP}

# every symbol include/msweep_core.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "msw_core_create", "msw_core_destroy", "msw_last_error", "msw_core_version",
    "msw_core_set_dense_logl", "msw_core_set_csr", "msw_core_build_likelihood",
    "msw_core_get_dense_logl", "msw_core_layout_hash", "msw_core_shape", "msw_alignment_read",
    "msw_alignment_shape", "msw_alignment_export", "msw_alignment_view", "msw_alignment_destroy", "msw_alignment_last_error",
    "msw_alignment_read_device", "msw_core_build_likelihood_aln", "msw_core_trim", "msw_alignment_on_device", "msw_core_solve", "msw_core_prepare", "msw_core_run",
    "msw_core_gamma",
    "msw_core_trace", "msw_core_set_trace_theta", "msw_core_bootstrap",
    "msw_core_resample_counts", "msw_core_set_profiling", "msw_core_last_timing",
    "msw_core_set_fixed_iters", "msw_core_hbm_stream_rates", "msw_comm_unique_id", "msw_comm_create_rccl", "msw_comm_create_local",
    "msw_comm_destroy", "msw_core_set_comm", "msw_comm_last_error", "msw_core_bootstrap_dist",
    "msw_comm_size", "msw_comm_rccl_count", "msw_comm_allgather", "msw_comm_allreduce", "msw_comm_create_shm", "msw_core_continue", "msw_core_gamma_block",
    "msw_core_last_bootstrap_timing", "msw_core_layout_info", "msw_core_guarded_visits", "msw_core_set_pack_schedule",
    "msw_core_set_option", "msw_core_get_option",
]


class MswError(RuntimeError):
    """Non-zero return from the C ABI (mirrors the std::runtime_error the C++ shim throws)."""


class Timing(C.Structure):
    _fields_ = [("solve_ms", C.c_double), ("passA_ms", C.c_double), ("passB_ms", C.c_double),
                ("passA_launches", C.c_uint64), ("passB_launches", C.c_uint64), ("iters", C.c_uint64),
                ("bytes_passA", C.c_uint64), ("bytes_passB", C.c_uint64), ("collective_ms", C.c_double),
                ("collectives", C.c_uint64), ("em_float_kernels", C.c_uint64)]


class LayoutInfo(C.Structure):
    _fields_ = [("record_bytes", C.c_int32), ("index_records", C.c_int32), ("groups_in_lds", C.c_int32),
                ("table_in_lds", C.c_int32), ("passB_mode", C.c_int32), ("slot_entries", C.c_uint32),
                ("slot_entries_in_lds", C.c_uint32), ("n_slices", C.c_uint32), ("n_long_ecs", C.c_uint32),
                ("rows", C.c_uint64), ("rows_from_memory", C.c_uint64), ("slices_by_lanes", C.c_uint32 * 7),
                ("max_rows", C.c_uint32), ("bank_scheduled", C.c_int32), ("passB_reg_cells", C.c_int32),
                ("rows_over_8", C.c_uint64)]


class BootstrapTiming(C.Structure):
    _fields_ = [("table_ms", C.c_double), ("solve_ms", C.c_double), ("gather_ms", C.c_double),
                ("replicates", C.c_uint64), ("iterations", C.c_uint64), ("table_reused", C.c_uint64)]


_lib = None


def source_hash():
    """sha256 prefix over the library's sources (the same function as __graft_entry__.source_hash)."""
    import hashlib
    csrc = os.path.join(_HERE, "csrc")
    h = hashlib.sha256()
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".hpp", ".inc", ".h")))
    for f in files + [os.path.join(os.path.dirname(_HERE), "include", "msweep_core.h")]:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _check_fresh(L):
    """A library that was not compiled from the sources in this tree is refused (MSWEEP_CORE_LIB, the
    developer override for A/B builds, is exempt): a stale prebuilt .so cannot pass silently."""
    if os.environ.get("MSWEEP_CORE_LIB"):
        return
    ver = L.msw_core_version().decode()
    try:
        want = source_hash()
    except OSError:
        return          # installed without its sources: nothing to compare with
    if not ver.endswith("src " + want):
        raise MswError(f"{LIB_PATH} is stale: built from other sources ({ver}; the tree hashes to {want}). "
                       "Rebuild: python -c 'import __graft_entry__ as g; g.build()'")


def load_library():
    """dlopen the in-tree extension; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MswError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, sz, dp = C.c_void_p, C.c_size_t, C.c_double
    L.msw_core_version.restype = C.c_char_p
    _check_fresh(L)
    L.msw_core_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.msw_core_destroy.argtypes = [vp]
    L.msw_core_destroy.restype = None
    L.msw_last_error.argtypes = [vp]
    L.msw_last_error.restype = C.c_char_p
    L.msw_core_version.restype = C.c_char_p
    L.msw_core_set_dense_logl.argtypes = [vp, vp, sz, sz, sz]
    L.msw_core_set_csr.argtypes = [vp, vp, vp, vp, vp, sz, dp, sz, sz]
    L.msw_core_build_likelihood.argtypes = [vp, vp, vp, sz, vp, sz, vp, sz, vp, dp, dp, dp, sz,
                                            C.POINTER(sz), vp, vp]
    L.msw_core_get_dense_logl.argtypes = [vp, vp, sz]
    L.msw_core_layout_hash.argtypes = [vp, vp]
    L.msw_core_shape.argtypes = [vp, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]
    L.msw_core_solve.argtypes = [vp, vp, vp, dp, sz, C.c_int, C.c_int, vp, C.POINTER(sz), C.POINTER(dp)]
    L.msw_core_prepare.argtypes = [vp, vp, vp]
    L.msw_core_run.argtypes = [vp, dp, sz, C.c_int, C.c_int, vp, C.POINTER(sz), C.POINTER(dp)]
    L.msw_core_continue.argtypes = [vp, sz, vp, C.POINTER(sz), C.POINTER(dp)]
    L.msw_core_gamma.argtypes = [vp, vp, sz]
    L.msw_core_gamma_block.argtypes = [vp, sz, sz, vp, sz]
    L.msw_core_trace.argtypes = [vp, sz, vp, vp, vp, vp, vp, C.POINTER(sz)]
    L.msw_core_set_trace_theta.argtypes = [vp, sz]
    L.msw_core_bootstrap.argtypes = [vp, vp, C.c_int32, sz, sz, sz, vp, dp, sz, C.c_int, C.c_int, vp, vp]
    L.msw_core_resample_counts.argtypes = [vp, vp, sz, C.c_int32, sz, sz, sz, vp]
    L.msw_core_bootstrap_dist.argtypes = [vp, vp, vp, C.c_int32, sz, sz, vp, dp, sz, C.c_int, C.c_int, vp, vp]
    L.msw_comm_size.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.msw_comm_rccl_count.argtypes = [vp, C.POINTER(C.c_int)]
    L.msw_comm_allgather.argtypes = [vp, vp, sz, vp]
    L.msw_comm_allreduce.argtypes = [vp, vp, sz, vp, sz, C.c_int, C.POINTER(dp)]
    L.msw_comm_create_shm.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.msw_core_set_profiling.argtypes = [vp, C.c_int]
    L.msw_core_set_fixed_iters.argtypes = [vp, C.c_int]
    L.msw_core_set_pack_schedule.argtypes = [vp, C.c_int]
    L.msw_core_set_option.argtypes = [vp, C.c_int, C.c_double]
    L.msw_core_get_option.argtypes = [vp, C.c_int, C.POINTER(C.c_double)]
    L.msw_core_hbm_stream_rates.argtypes = [vp, C.c_size_t, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.msw_core_last_timing.argtypes = [vp, C.POINTER(Timing)]
    L.msw_core_last_bootstrap_timing.argtypes = [vp, C.POINTER(BootstrapTiming)]
    L.msw_core_guarded_visits.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.msw_core_layout_info.argtypes = [vp, C.POINTER(LayoutInfo)]
    L.msw_comm_unique_id.argtypes = [vp]
    L.msw_comm_create_rccl.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.msw_comm_create_local.argtypes = [C.c_int, C.POINTER(vp)]
    L.msw_comm_destroy.argtypes = [vp]
    L.msw_comm_destroy.restype = None
    L.msw_core_set_comm.argtypes = [vp, vp]
    L.msw_comm_last_error.restype = C.c_char_p
    L.msw_alignment_read.argtypes = [C.POINTER(C.c_char_p), sz, sz, C.c_int, C.POINTER(vp)]
    L.msw_alignment_shape.argtypes = [vp, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]
    L.msw_alignment_view.argtypes = [vp] + [C.POINTER(vp)] * 5
    L.msw_alignment_export.argtypes = [vp, vp, vp, vp, vp, vp]
    L.msw_alignment_destroy.argtypes = [vp]
    L.msw_core_trim.argtypes = [vp]
    L.msw_alignment_on_device.argtypes = [vp]
    L.msw_alignment_read_device.argtypes = [vp, C.POINTER(C.c_char_p), sz, sz, C.c_int, C.POINTER(vp)]
    L.msw_core_build_likelihood_aln.argtypes = [vp, vp, vp, sz, vp, sz, C.c_double, C.c_double, C.c_double, sz,
                                                C.POINTER(sz), vp, vp]
    L.msw_alignment_destroy.restype = None
    L.msw_alignment_last_error.restype = C.c_char_p
    _lib = L
    return L


class _AlignmentOwner:
    """Keeps a native alignment handle alive for as long as an array that views its storage is."""

    def __init__(self, L, h):
        self._L, self._h = L, h

    def __del__(self):
        try:
            if self._h:
                self._L.msw_alignment_destroy(self._h)
                self._h = None
        except Exception:
            pass


def read_alignment(paths, n_targets, merge_mode="intersection", copy=False):
    """Native Themisto plaintext reader + EC collapse (msw_alignment_*, host code of the library):
    dict(ec_tptr, ec_targets, ec_counts, ec_rptr, ec_reads, n_reads).  Errors carry the reference's
    messages (include/mSWEEP_alignment.hpp:84-91, :131).  The arrays are read-only VIEWS of the native handle's own
    storage (msw_alignment_view: no 0.9 GB copy at cfg3), which lives as long as any of them does; copy=True gives
    ordinary writable arrays (msw_alignment_export)."""
    L = load_library()
    if merge_mode not in ("intersection", "union"):
        raise MswError(f"Unrecognized option `{merge_mode}` for --themisto-mode")
    arr = (C.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
    h = C.c_void_p()
    if L.msw_alignment_read(arr, len(paths), int(n_targets), 0 if merge_mode == "intersection" else 1, C.byref(h)):
        raise MswError(L.msw_alignment_last_error().decode())
    owner = _AlignmentOwner(L, h)
    ne, nr, nh, na = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
    L.msw_alignment_shape(h, C.byref(ne), C.byref(nr), C.byref(nh), C.byref(na))
    if copy:
        out = dict(ec_tptr=np.empty(ne.value + 1, np.uint64), ec_targets=np.empty(nh.value, np.uint32),
                   ec_counts=np.empty(ne.value, np.uint64), ec_rptr=np.empty(ne.value + 1, np.uint64),
                   ec_reads=np.empty(na.value, np.uint32), n_reads=int(nr.value))
        if L.msw_alignment_export(h, _ptr(out["ec_tptr"]), _ptr(out["ec_targets"]), _ptr(out["ec_counts"]),
                                  _ptr(out["ec_rptr"]), _ptr(out["ec_reads"])):
            raise MswError("msw_alignment_export failed")
        return out
    return _alignment_views(L, h, owner, ne.value, nh.value, na.value, nr.value)


class DeviceAlignment:
    """An alignment read by msw_alignment_read_device: its arrays live in the memory of the GPU that parsed the text
    (Core.read_alignment).  Core.build_likelihood_aln consumes them there; arrays() copies them out on first use
    (the dict read_alignment returns, as read-only views)."""

    def __init__(self, L, h):
        self._L, self._h = L, h
        self._owner = _AlignmentOwner(L, h)
        ne, nr, nh, na = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
        L.msw_alignment_shape(h, C.byref(ne), C.byref(nr), C.byref(nh), C.byref(na))
        self.n_ecs, self.n_reads, self.n_hits, self.n_aligned = ne.value, nr.value, nh.value, na.value
        self._arrays = None
        self._counts = None
        self.on_device = bool(L.msw_alignment_on_device(h))   # False: the host parser took the text (its word is final)

    def ec_counts(self):
        """reads per class (uint64): the one array the drivers need on the host -- 8 bytes per class leave the device"""
        if self._counts is None:
            out = np.empty(self.n_ecs, np.uint64)
            if self._L.msw_alignment_export(self._h, None, None, _ptr(out), None, None):
                raise MswError("msw_alignment_export failed")
            self._counts = out
        return self._counts

    def arrays(self):
        if self._arrays is None:
            self._arrays = _alignment_views(self._L, self._h, self._owner, self.n_ecs, self.n_hits, self.n_aligned, self.n_reads)
        return self._arrays


def _alignment_views(L, h, owner, n_ecs, n_hits, n_aligned, n_reads):
    p = [C.c_void_p() for _ in range(5)]
    if L.msw_alignment_view(h, *[C.byref(x) for x in p]):
        raise MswError("msw_alignment_view failed")

    def view(ptr, n, ctype, dtype):
        if n == 0 or not ptr.value:
            return np.empty(0, dtype)
        buf = (ctype * n).from_address(ptr.value)
        buf._owner = owner               # array -> ctypes buffer -> owner -> handle
        a = np.frombuffer(buf, dtype=dtype)
        a.flags.writeable = False
        return a
    return dict(ec_tptr=view(p[0], n_ecs + 1, C.c_uint64, np.uint64), ec_targets=view(p[1], n_hits, C.c_uint32, np.uint32),
                ec_counts=view(p[2], n_ecs, C.c_uint64, np.uint64), ec_rptr=view(p[3], n_ecs + 1, C.c_uint64, np.uint64),
                ec_reads=view(p[4], n_aligned, C.c_uint32, np.uint32), n_reads=int(n_reads))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _arr(x, dt):
    return np.ascontiguousarray(x, dtype=dt)


class Core:
    """One handle = one GPU + one resident likelihood (the object rcg_optl() would receive)."""

    def __init__(self, device=0):
        self._L = load_library()
        self._h = C.c_void_p()
        rc = self._L.msw_core_create(int(device), C.byref(self._h))
        if rc != 0:
            raise MswError(self._L.msw_last_error(None).decode())
        self.device = device
        self.generation = 0     # solves started on this handle (rcgpar.EcProbs: whose state gamma() would read)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.msw_core_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise MswError(self._L.msw_last_error(self._h).decode())

    # ---- likelihood -------------------------------------------------------------------
    def set_dense_logl(self, logl):
        """logl: G x E array, rows = groups (seamat layout of Likelihood::log_mat())."""
        logl = np.asarray(logl, dtype=np.float64)
        if logl.ndim != 2:
            raise MswError("set_dense_logl: expected a 2-D G x E array")
        if logl.strides[1] != 8 or logl.strides[0] % 8 or logl.strides[0] < 8 * logl.shape[1]:
            logl = np.ascontiguousarray(logl)
        G, E = logl.shape
        self._check(self._L.msw_core_set_dense_logl(self._h, _ptr(logl), G, E, logl.strides[0] // 8 if G else E))

    def set_csr(self, rowptr, grp, cnt, lut, logzi, n_groups):
        rowptr = _arr(rowptr, np.uint64)
        grp = _arr(grp, np.uint32)
        cnt = _arr(cnt, np.uint32)
        lut = _arr(lut, np.float64)
        if lut.ndim != 2 or lut.shape[0] != n_groups:
            raise MswError("set_csr: lut must be n_groups x lut_ld")
        if len(grp) != len(cnt) or (len(rowptr) and int(rowptr[-1]) != len(grp)):
            raise MswError("set_csr: rowptr / grp / cnt lengths disagree")
        E = len(rowptr) - 1
        self._check(self._L.msw_core_set_csr(self._h, _ptr(rowptr), _ptr(grp), _ptr(cnt), _ptr(lut),
                                             lut.shape[1], float(logzi), int(n_groups), E))

    def build_likelihood(self, ec_tptr, ec_targets, target_group, group_sizes, ec_counts, q=0.65, e=0.01,
                         zero_inflation=0.01, min_hits=0, want_logc=True):
        """Device build from the pseudoalignment (replaces LL_WOR21::fill_ll_mat).
        Returns (n_groups_kept, mask[G] bool, logc[E] or None when want_logc is False)."""
        ec_tptr = _arr(ec_tptr, np.uint64)
        ec_targets = _arr(ec_targets, np.uint32)
        target_group = _arr(target_group, np.uint32)
        group_sizes = _arr(group_sizes, np.uint64)
        ec_counts = _arr(ec_counts, np.uint64)
        E, G = len(ec_tptr) - 1, len(group_sizes)
        if len(ec_counts) != E:
            raise MswError("build_likelihood: ec_counts length != n_ecs")
        n_out = C.c_size_t()
        mask = np.zeros(G, np.uint8)
        logc = np.empty(E, np.float64) if want_logc else None
        self._check(self._L.msw_core_build_likelihood(
            self._h, _ptr(ec_tptr), _ptr(ec_targets), E, _ptr(target_group), len(target_group),
            _ptr(group_sizes), G, _ptr(ec_counts), q, e, zero_inflation, int(min_hits), C.byref(n_out),
            _ptr(mask), _ptr(logc)))
        return n_out.value, mask.astype(bool), logc

    def read_alignment(self, paths, n_targets, merge_mode="intersection"):
        """The Themisto plaintext reader ON THIS GPU (msw_alignment_read_device): the text goes to device memory as it
        is read; parse, rows by read id, paired-end merge, the reference's hash, sort and classes are kernels.  Returns
        a DeviceAlignment (build_likelihood_aln reads it where it lies; .arrays() = what read_alignment() returns).
        Errors carry the reference's messages: text the kernels do not judge goes to the host reader."""
        if merge_mode not in ("intersection", "union"):
            raise MswError(f"Unrecognized option `{merge_mode}` for --themisto-mode")
        arr = (C.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
        h = C.c_void_p()
        if self._L.msw_alignment_read_device(self._h, arr, len(paths), int(n_targets),
                                             0 if merge_mode == "intersection" else 1, C.byref(h)):
            raise MswError(self._L.msw_alignment_last_error().decode())
        return DeviceAlignment(self._L, h)

    def trim(self):
        """Gives the device temporaries of read_alignment back (kept on the handle between calls: msw_core_trim)."""
        self._check(self._L.msw_core_trim(self._h))

    def build_likelihood_aln(self, aln, target_group, group_sizes, q=0.65, e=0.01, zero_inflation=0.01, min_hits=0,
                             want_logc=True):
        """build_likelihood on a DeviceAlignment: its classes, targets and read counts are read in device memory.
        Returns (n_groups_kept, mask[G] bool, logc[E] or None)."""
        target_group = _arr(target_group, np.uint32)
        group_sizes = _arr(group_sizes, np.uint64)
        G = len(group_sizes)
        n_out = C.c_size_t()
        mask = np.zeros(G, np.uint8)
        logc = np.empty(aln.n_ecs, np.float64) if want_logc else None
        self._check(self._L.msw_core_build_likelihood_aln(
            self._h, aln._h, _ptr(target_group), len(target_group), _ptr(group_sizes), G, q, e, zero_inflation,
            int(min_hits), C.byref(n_out), _ptr(mask), _ptr(logc)))
        return n_out.value, mask.astype(bool), logc

    def shape(self):
        g, e, n = C.c_size_t(), C.c_size_t(), C.c_size_t()
        self._check(self._L.msw_core_shape(self._h, C.byref(g), C.byref(e), C.byref(n)))
        return g.value, e.value, n.value

    def layout_info(self):
        """How the resident CSR-of-ECs likelihood is laid out for the sweeps (msw_core_layout_info)."""
        t = LayoutInfo()
        self._check(self._L.msw_core_layout_info(self._h, C.byref(t)))
        return {k: (list(getattr(t, k)) if k == "slices_by_lanes" else getattr(t, k)) for k, _ in LayoutInfo._fields_}

    def layout_hash(self):
        out = C.c_uint64(0)
        self._check(self._L.msw_core_layout_hash(self._h, C.byref(out)))
        return int(out.value)

    def get_dense_logl(self):
        G, E, _ = self.shape()
        out = np.empty((G, E))
        self._check(self._L.msw_core_get_dense_logl(self._h, _ptr(out), E))
        return out

    # ---- solve --------------------------------------------------------------------------
    def solve(self, logc, alpha0, tol=1e-6, max_iters=5000, algo=ALGO_RCG, prec=PREC_DOUBLE):
        """logc=None: the log counts build_likelihood() left on the device (no upload per solve)."""
        G, E, _ = self.shape()
        logc = None if logc is None else _arr(logc, np.float64)
        alpha0 = _arr(alpha0, np.float64)
        if (logc is not None and len(logc) != E) or len(alpha0) != G:
            raise MswError(f"solve: expected logc[{E}] and alpha0[{G}], got {len(logc)} and {len(alpha0)}")
        theta = np.empty(G)
        it, b = C.c_size_t(), C.c_double()
        self.generation += 1
        self._check(self._L.msw_core_solve(self._h, _ptr(logc), _ptr(alpha0), float(tol), int(max_iters),
                                           int(algo), int(prec), _ptr(theta), C.byref(it), C.byref(b)))
        return dict(theta=theta, iters=it.value, bound=b.value)

    def prepare(self, logc, alpha0):
        """logc=None: the log counts build_likelihood() left on the device."""
        G, E, _ = self.shape()
        logc = None if logc is None else _arr(logc, np.float64)
        alpha0 = _arr(alpha0, np.float64)
        if (logc is not None and len(logc) != E) or len(alpha0) != G:
            raise MswError(f"prepare: expected logc[{E}] and alpha0[{G}], got "
                           f"{None if logc is None else len(logc)} and {len(alpha0)}")
        self._check(self._L.msw_core_prepare(self._h, _ptr(logc), _ptr(alpha0)))

    def run(self, tol=1e-6, max_iters=5000, algo=ALGO_RCG, prec=PREC_DOUBLE):
        G, _, _ = self.shape()
        theta = np.empty(G)
        it, b = C.c_size_t(), C.c_double()
        self.generation += 1
        self._check(self._L.msw_core_run(self._h, float(tol), int(max_iters), int(algo), int(prec), _ptr(theta),
                                         C.byref(it), C.byref(b)))
        return dict(theta=theta, iters=it.value, bound=b.value)

    def continue_(self, n_iters):
        """n_iters more iterations of the fixed-iteration RCG solve that last ran (msw_core_continue)."""
        G, _, _ = self.shape()
        theta = np.empty(G)
        it, b = C.c_size_t(), C.c_double()
        self.generation += 1
        self._check(self._L.msw_core_continue(self._h, int(n_iters), _ptr(theta), C.byref(it), C.byref(b)))
        return dict(theta=theta, iters=it.value, bound=b.value)

    def gamma(self):
        G, E, _ = self.shape()
        out = np.empty((G, E))
        self._check(self._L.msw_core_gamma(self._h, _ptr(out), E))
        return out

    def gamma_block(self, ec_begin, ec_end):
        """G x (ec_end - ec_begin) block of the log-responsibilities (msw_core_gamma_block)."""
        G, E, _ = self.shape()
        w = int(ec_end) - int(ec_begin)
        out = np.empty((G, max(w, 0)))
        self._check(self._L.msw_core_gamma_block(self._h, int(ec_begin), int(ec_end), _ptr(out), max(w, 1)))
        return out

    def set_trace_theta(self, n):
        self._check(self._L.msw_core_set_trace_theta(self._h, int(n)))

    def trace(self, n, with_theta=False):
        G, _, _ = self.shape()
        bound, nn, beta = np.full(n, np.nan), np.full(n, np.nan), np.full(n, np.nan)
        rs = np.full(n, -1, np.int32)
        th = np.full((n, G), np.nan) if with_theta else None
        got = C.c_size_t()
        self._check(self._L.msw_core_trace(self._h, n, _ptr(bound), _ptr(nn), _ptr(beta), _ptr(rs), _ptr(th),
                                           C.byref(got)))
        k = got.value
        return dict(n=k, bound=bound[:k], newnorm=nn[:k], beta=beta[:k], didreset=rs[:k],
                    theta=None if th is None else th[:k])

    # ---- bootstrap ------------------------------------------------------------------------
    def bootstrap(self, ec_counts, seed, bootstrap_count, rep_begin, rep_end, alpha0, tol=1e-6,
                  max_iters=5000, algo=ALGO_RCG, prec=PREC_DOUBLE):
        G, E, _ = self.shape()
        ec_counts = _arr(ec_counts, np.uint32)
        alpha0 = _arr(alpha0, np.float64)
        if len(ec_counts) != E or len(alpha0) != G:
            raise MswError("bootstrap: ec_counts / alpha0 length mismatch")
        n = int(rep_end) - int(rep_begin)
        theta = np.empty((max(n, 0), G))
        iters = np.zeros(max(n, 0), np.uint64)
        self._check(self._L.msw_core_bootstrap(self._h, _ptr(ec_counts), int(seed), int(bootstrap_count),
                                               int(rep_begin), int(rep_end), _ptr(alpha0), float(tol),
                                               int(max_iters), int(algo), int(prec), _ptr(theta), _ptr(iters)))
        return theta, iters

    def bootstrap_dist(self, comm, ec_counts, seed, bootstrap_count, n_replicates, alpha0, tol=1e-6,
                       max_iters=5000, algo=ALGO_RCG, prec=PREC_DOUBLE):
        """msw_core_bootstrap_dist: all replicates over the ranks of `comm`; every rank gets the whole
        n_replicates x G table (replicate order) and the iteration counts."""
        G, E, _ = self.shape()
        ec_counts = _arr(ec_counts, np.uint32)
        alpha0 = _arr(alpha0, np.float64)
        if len(ec_counts) != E or len(alpha0) != G:
            raise MswError("bootstrap_dist: ec_counts / alpha0 length mismatch")
        n = int(n_replicates)
        theta = np.empty((n, G))
        iters = np.zeros(n, np.uint64)
        self._check(self._L.msw_core_bootstrap_dist(self._h, comm._c, _ptr(ec_counts), int(seed),
                                                    int(bootstrap_count), n, _ptr(alpha0), float(tol),
                                                    int(max_iters), int(algo), int(prec), _ptr(theta), _ptr(iters)))
        return theta, iters

    def resample_counts(self, ec_counts, seed, bootstrap_count, rep_begin, rep_end):
        ec_counts = _arr(ec_counts, np.uint32)
        n = int(rep_end) - int(rep_begin)
        out = np.empty((max(n, 0), len(ec_counts)), np.uint32)
        self._check(self._L.msw_core_resample_counts(self._h, _ptr(ec_counts), len(ec_counts), int(seed),
                                                     int(bootstrap_count), int(rep_begin), int(rep_end),
                                                     _ptr(out)))
        return out

    # ---- EC-sharded solve -------------------------------------------------------------------
    def set_comm(self, comm):
        """Attach a communicator (Comm) -- this handle then holds one rank's block of ECs."""
        self._check(self._L.msw_core_set_comm(self._h, comm._c if comm is not None else None))
        self._comm = comm      # keep it alive

    # ---- measurement ------------------------------------------------------------------------
    def set_profiling(self, on):
        self._check(self._L.msw_core_set_profiling(self._h, int(bool(on))))

    def set_fixed_iters(self, on):
        self._check(self._L.msw_core_set_fixed_iters(self._h, int(bool(on))))

    def hbm_stream_rates(self, n_bytes=1 << 30, reps=5):
        """(read-only GB/s, triad GB/s) of this device on n_bytes of HBM: the practical ceiling (measurement only)."""
        r, t = C.c_double(), C.c_double()
        self._check(self._L.msw_core_hbm_stream_rates(self._h, int(n_bytes), int(reps), C.byref(r), C.byref(t)))
        return r.value, t.value

    def last_timing(self):
        t = Timing()
        self._check(self._L.msw_core_last_timing(self._h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in Timing._fields_}

    def set_pack_schedule(self, enabled):
        """Order the cells of the NEXT likelihood for the LDS banks (default) or keep their CSR order (one solve only:
        the upload is faster than the iterations are slower).  msw_core_set_pack_schedule."""
        self._check(self._L.msw_core_set_pack_schedule(self._h, 1 if enabled else 0))

    def set_option(self, name, value):
        """msw_core_set_option: "check_every" n | "init_bound" b | "em_prior" 0 (MAP) / 1 (ML) | "em_stop" 0 (gain) /
        1 (largest move of a weight).  Persists on the handle."""
        if name not in _OPTS:
            raise MswError(f"set_option: unknown option `{name}`")
        self._check(self._L.msw_core_set_option(self._h, _OPTS[name], float(value)))

    def get_option(self, name):
        if name not in _OPTS:
            raise MswError(f"get_option: unknown option `{name}`")
        v = C.c_double()
        self._check(self._L.msw_core_get_option(self._h, _OPTS[name], C.byref(v)))
        return v.value

    def guarded_visits(self):
        """ECs pass B has taken through the cancellation guard since the likelihood became resident (all iterations)."""
        out = C.c_uint64(0)
        self._check(self._L.msw_core_guarded_visits(self._h, C.byref(out)))
        return int(out.value)

    def last_bootstrap_timing(self):
        t = BootstrapTiming()
        self._check(self._L.msw_core_last_bootstrap_timing(self._h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in BootstrapTiming._fields_}


class Comm:
    """Communicator of the EC-sharded solve (include/msweep_core.h, "EC-sharded single solve")."""

    def __init__(self, c, owner=True):
        self._L = load_library()
        self._c = c
        self._owner = owner

    @staticmethod
    def unique_id():
        L = load_library()
        buf = (C.c_ubyte * 128)()
        if L.msw_comm_unique_id(buf) != 0:
            raise MswError(L.msw_comm_last_error().decode())
        return bytes(buf)

    @classmethod
    def rccl(cls, uid, rank, nranks, device):
        L = load_library()
        out = C.c_void_p()
        buf = (C.c_ubyte * 128).from_buffer_copy(uid)
        if L.msw_comm_create_rccl(buf, int(rank), int(nranks), int(device), C.byref(out)) != 0:
            raise MswError(L.msw_comm_last_error().decode())
        return cls(out)

    @classmethod
    def local(cls, nranks):
        """nranks communicators for nranks host threads of this process."""
        L = load_library()
        arr = (C.c_void_p * nranks)()
        if L.msw_comm_create_local(int(nranks), arr) != 0:
            raise MswError(L.msw_comm_last_error().decode())
        return [cls(C.c_void_p(arr[i])) for i in range(nranks)]

    @classmethod
    def shm(cls, name, rank, nranks, device=0):
        """Rank `rank` of `nranks` PROCESSES of this host meeting in the shared-memory segment `name` ("/...")."""
        L = load_library()
        out = C.c_void_p()
        if L.msw_comm_create_shm(name.encode(), int(rank), int(nranks), int(device), C.byref(out)) != 0:
            raise MswError(L.msw_comm_last_error().decode())
        return cls(out)

    def size(self):
        n, r = C.c_int(), C.c_int()
        if self._L.msw_comm_size(self._c, C.byref(n), C.byref(r)) != 0:
            raise MswError(self._L.msw_comm_last_error().decode())
        return n.value, r.value

    def rccl_count(self):
        """ncclCommCount of the communicator (0 for an in-process one)."""
        n = C.c_int()
        if self._L.msw_comm_rccl_count(self._c, C.byref(n)) != 0:
            raise MswError(self._L.msw_comm_last_error().decode())
        return n.value

    def allgather(self, send):
        """(nranks, len(send)) array: row r = rank r's `send` (host buffers; ncclAllGather underneath)."""
        send = _arr(send, np.float64).ravel()
        out = np.empty((self.size()[0], len(send)))
        if self._L.msw_comm_allgather(self._c, _ptr(send), len(send), _ptr(out)) != 0:
            raise MswError(self._L.msw_comm_last_error().decode())
        return out

    def allreduce(self, ints, reals, repeats=1):
        """Sums over the ranks of a uint64 and a float64 vector (msw_comm_allreduce); returns (ints, reals, ms per call)."""
        a = np.array(ints, np.uint64).ravel()
        b = np.array(reals, np.float64).ravel()
        ms = C.c_double()
        if self._L.msw_comm_allreduce(self._c, _ptr(a) if len(a) else None, len(a), _ptr(b) if len(b) else None, len(b),
                                      int(repeats), C.byref(ms)) != 0:
            raise MswError(self._L.msw_comm_last_error().decode())
        return a, b, ms.value

    def close(self):
        if self._c and self._owner:
            self._L.msw_comm_destroy(self._c)
        self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
