"""GPU: value records (sell.hpp) -- a dense `logl` whose listed values are (nearly) all different, as
msw_core_set_dense_logl receives it from --read-likelihood or cfg2's generator, is re-expressed as CSR-of-ECs whose
12-byte records carry the cell's log-likelihood itself; the sweeps form exp(a (T - tref)) per cell (no slot table).
Against the structured and the dense-state oracle in lock-step, against the slot-per-cell layout it replaces, the
matrix read back bit for bit, gamma, EM, bootstrap, -inf log counts, ECs of every length class."""
import numpy as np
import pytest

from msweep_amd import synth
from msweep_amd.core import ALGO_EM, Core
from msweep_amd.likelihood import from_dense
from test_gpu_rcg import assert_theta, lockstep

pytestmark = pytest.mark.gpu


def continuous_problem(E, G, seed, max_support=20):
    return synth.make_dense_problem(E, G, seed=seed, max_support=max_support)


def solve_dense(core, p, logc=None, alpha0=None, trace=20, **kw):
    G = p["logl"].shape[0]
    logc = p["logc"] if logc is None else logc
    alpha0 = np.ones(G) if alpha0 is None else alpha0
    from_dense(core, p["logl"], logc)
    core.set_trace_theta(trace)
    res = core.solve(logc, alpha0, **kw)
    return res, core.trace(trace, with_theta=True), logc, alpha0


@pytest.mark.parametrize("E,G,support", [(20000, 40, 12), (120000, 300, 20), (9000, 700, 300)])
def test_value_records_lockstep(oracle, E, G, support):
    """(the third shape: ECs of up to 300 listed cells -- the streaming and the wavefront-per-EC paths)"""
    p = continuous_problem(E, G, 31, max_support=support)
    with Core(0) as core:
        res, tr, logc, alpha0 = solve_dense(core, p)
        li = core.layout_info()
        assert li["record_bytes"] == 12 and li["slot_entries"] == 0, li
        ref = oracle.rcg_optl_dense_structured(p["logl"], logc, alpha0, trace=20)
        lockstep(tr, ref["trace"], 20)
        assert res["iters"] == ref["iters"]
        assert_theta(res["theta"], ref["theta"])
        assert res["theta"].sum() == pytest.approx(1.0, abs=1e-12)
        # the resident matrix IS the matrix that was handed over
        np.testing.assert_array_equal(core.get_dense_logl(), p["logl"])
        if E <= 20000:
            g_ref = oracle.rcg_optl_dense_structured(p["logl"], logc, alpha0, want_gamma=True)["gamma"]
            np.testing.assert_allclose(core.gamma_block(50, 350), g_ref[:, 50:350], rtol=1e-9, atol=1e-9)
            # EM rides on the same records
            em = core.solve(logc, alpha0, algo=ALGO_EM, tol=1e-8, max_iters=400)
            em_ref = oracle.em_dense(p["logl"], logc, alpha0, tol=1e-8, max_iters=400)
            assert em["iters"] == em_ref["iters"]
            assert_theta(em["theta"], em_ref["theta"])


def test_value_records_without_bank_scheduling(oracle):
    """msw_core_set_pack_schedule(0) on value records: the same solve from cells in CSR order."""
    p = continuous_problem(30000, 90, 36, max_support=40)
    with Core(0) as core:
        core.set_pack_schedule(0)
        res, tr, logc, alpha0 = solve_dense(core, p)
        li = core.layout_info()
        assert li["record_bytes"] == 12 and li["bank_scheduled"] == 0, li
        ref = oracle.rcg_optl_dense_structured(p["logl"], logc, alpha0, trace=20)
        lockstep(tr, ref["trace"], 20)
        assert res["iters"] == ref["iters"]
        assert_theta(res["theta"], ref["theta"])
        np.testing.assert_array_equal(core.get_dense_logl(), p["logl"])


def test_value_records_equal_the_slot_per_cell_layout(monkeypatch):
    """MSWEEP_VALUE_RECORDS=0: the previous layout (one table slot per listed cell, tables in memory).  Same
    exponentials of the same values: same iteration count, abundances equal to rounding."""
    p = continuous_problem(60000, 120, 32)
    out = []
    for vr in ("1", "0"):
        monkeypatch.setenv("MSWEEP_VALUE_RECORDS", vr)
        with Core(0) as core:
            res, tr, logc, alpha0 = solve_dense(core, p)
            out.append((res, core.layout_info()["record_bytes"]))
    assert out[0][1] == 12 and out[1][1] in (4, 8)
    assert out[0][0]["iters"] == out[1][0]["iters"]
    assert_theta(out[0][0]["theta"], out[1][0]["theta"], rel=1e-9, abs_=1e-12)


def test_value_records_bootstrap_and_zero_counts(oracle):
    """Resampled counts (a third of the ECs at zero: -inf log counts) and counts beyond the byte image."""
    p = continuous_problem(15000, 60, 33)
    rng = np.random.default_rng(3)
    counts = rng.integers(0, 400, 15000).astype(np.uint32)
    with np.errstate(divide="ignore"):
        logc = np.log(counts.astype(float))
    with Core(0) as core:
        res, tr, logc, alpha0 = solve_dense(core, p, logc=logc)
        ref = oracle.rcg_optl_dense_structured(p["logl"], logc, alpha0, trace=20)
        lockstep(tr, ref["trace"], 20)
        # (counts up to 400: a bound of ~1e7 whose last gains sit at the 1e-6 of the stop rule -- the stop may fall
        # one iteration apart, tests/test_gpu_full_size.py (iii))
        assert abs(res["iters"] - ref["iters"]) <= 1
        if res["iters"] == ref["iters"]:
            assert_theta(res["theta"], ref["theta"])
        theta_b, iters_b = core.bootstrap(counts, 11, int(counts.sum()), 0, 2, alpha0)
        cb = oracle.bootstrap_counts(counts, 11, int(counts.sum()), 2)
        for b in range(2):
            with np.errstate(divide="ignore"):
                lc = np.log(cb[b].astype(float))
            rb = oracle.rcg_optl_dense_structured(p["logl"], lc, alpha0)
            assert abs(int(iters_b[b]) - rb["iters"]) <= 2
            if int(iters_b[b]) == rb["iters"]:
                assert_theta(theta_b[b], rb["theta"])


def test_value_records_with_many_groups(oracle):
    """Continuous values AND more groups than the LDS images hold (12 000: pass B mode 3, {e, w} / e_g from memory)."""
    rng = np.random.default_rng(35)
    G, E = 12000, 6000
    L = np.full((G, E), np.log(0.01))
    for j in range(E):
        g = rng.choice(G, int(rng.integers(1, 30)), replace=False)
        L[g, j] = np.clip(rng.normal(-2.0, 1.0, len(g)), -12.0, -0.1)
    logc = np.log(rng.integers(1, 6, E).astype(float))
    alpha0 = np.ones(G)
    with Core(0) as core:
        from_dense(core, L, logc)
        li = core.layout_info()
        assert li["record_bytes"] == 12 and li["groups_in_lds"] == 0 and li["passB_mode"] in (3, 4), li
        core.set_trace_theta(20)
        res = core.solve(logc, alpha0)
        tr = core.trace(20, with_theta=True)
        ref = oracle.rcg_optl_dense_structured(L, logc, alpha0, trace=20)
        lockstep(tr, ref["trace"], 20)
        assert abs(res["iters"] - ref["iters"]) <= 1
        if res["iters"] == ref["iters"]:
            assert_theta(res["theta"], ref["theta"])
        np.testing.assert_array_equal(core.get_dense_logl(), L)
