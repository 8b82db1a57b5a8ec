"""GPU: the hybrid slot area (sell.hpp, index records) -- likelihoods whose per-slot tables do not fit LDS beside
the group vectors (real groupings: group sizes up to hundreds, include/Likelihood.hpp:92-107 makes the table
G x (max_size + 1)): 4-byte (group, entry) records, the most-used entries in LDS, every slice cut into a hot
segment (LDS gathers) and a short cold one (table entries from memory).  Parity against the structured oracle in
lock-step, the device packer against its host reference, gamma blocks, EM and the bootstrap on the same layout."""
import numpy as np
import pytest

from conftest import lutidx_of
from msweep_amd import synth
from msweep_amd.core import ALGO_EM, Core
from msweep_amd.likelihood import from_grouped_counts, precalc_lls
from test_gpu_rcg import assert_theta, lockstep, solve_csr

pytestmark = pytest.mark.gpu


def diverse(R, G, seed, **kw):
    return synth.make_csr_problem(R, G, seed=seed, group_sizes=synth.diverse_group_sizes, **kw)


@pytest.mark.parametrize("hot", [None, 0, 48, 1024])
def test_hybrid_lockstep_and_convergence(oracle, monkeypatch, hot):
    """hot = entries kept in LDS (None: as many as fit).  4000 groups with log-normal sizes: several thousand
    used (size, count) pairs beside 64 KB of group vectors."""
    if hot is not None:
        monkeypatch.setenv("MSWEEP_FORCE_LDS", "10")        # groups in LDS, the slot tables not (even if they would fit)
        monkeypatch.setenv("MSWEEP_HYBRID_HOT", str(hot))
    R, G = (400_000, 4000) if hot is None else (60_000, 500)
    p = diverse(R, G, 21, max_other=12)
    lut = precalc_lls(p["group_sizes"])
    with Core(0) as core:
        res, tr, logc, alpha0 = solve_csr(core, p)
        li = core.layout_info()
        print(li)
        assert li["record_bytes"] == 4
        if hot is None:
            # the natural case: if this shape ever fits LDS entirely the test is not testing the hybrid area
            assert li["index_records"] == 1 and 0 < li["slot_entries_in_lds"] < li["slot_entries"], li
        else:
            assert li["index_records"] == 1 and li["slot_entries_in_lds"] == min(hot, li["slot_entries"]) // 16 * 16
        if hot in (None, 48):
            assert 0 < li["rows_from_memory"] < li["rows"]
        ref = oracle.rcg_optl_csr(p["rowptr"], p["grp"], lutidx_of(p, lut), lut, np.log(0.01), G, logc, alpha0, trace=20)
        lockstep(tr, ref["trace"], 20)
        assert res["iters"] == ref["iters"]
        assert_theta(res["theta"], ref["theta"])
        assert res["theta"].sum() == pytest.approx(1.0, abs=1e-12)
        # gamma blocks: the same columns as the oracle's gamma
        if hot == 48:
            g_ref = oracle.rcg_optl_csr(p["rowptr"], p["grp"], lutidx_of(p, lut), lut, np.log(0.01), G, logc, alpha0,
                                        want_gamma=True)["gamma"]
            blk = core.gamma_block(100, 400)
            np.testing.assert_allclose(blk, g_ref[:, 100:400], rtol=1e-9, atol=1e-9)
            np.testing.assert_array_equal(core.get_dense_logl(), _dense(p, lut))
        # EM and the bootstrap ride on the same layout
        if hot is not None:
            em = core.solve(logc, alpha0, algo=ALGO_EM, tol=1e-8, max_iters=100)     # (the oracle's time: ~0.08 s per iteration)
            em_ref = oracle.em_dense(_dense(p, lut), logc, alpha0, tol=1e-8, max_iters=100)
            assert em["iters"] == em_ref["iters"]
            assert_theta(em["theta"], em_ref["theta"])
        w = p["ec_counts"].astype(np.uint32)
        theta_b, iters_b = core.bootstrap(w, 7, int(w.sum()), 0, 2, alpha0)
        counts = oracle.bootstrap_counts(w, 7, int(w.sum()), 2)
        for b in range(2):
            with np.errstate(divide="ignore"):
                lc = np.log(counts[b].astype(float))
            rb = oracle.rcg_optl_csr(p["rowptr"], p["grp"], lutidx_of(p, lut), lut, np.log(0.01), G, lc, alpha0)
            assert abs(int(iters_b[b]) - rb["iters"]) <= 2
            if int(iters_b[b]) == rb["iters"]:
                assert_theta(theta_b[b], rb["theta"])


def _dense(p, lut):
    from conftest import dense_from_csr
    return dense_from_csr(p, lut)


@pytest.mark.parametrize("hot,sched", [(0, 1), (32, 1), (4096, 1), (32, 0)])
def test_hybrid_device_packer_matches_host_packer(monkeypatch, hot, sched):
    """Hot / cold segments, cold classes in the EC order, rows per hot segment: the device packer
    (pack_kernels.hpp) against the host reference implementation, byte for byte; ragged lengths incl. empty ECs,
    streaming (17..256 cells) and long (> 256) ECs."""
    monkeypatch.setenv("MSWEEP_FORCE_LDS", "10")
    monkeypatch.setenv("MSWEEP_HYBRID_HOT", str(hot))
    rng = np.random.default_rng(5)
    G, E = 600, 6000
    sizes = np.minimum(1 + rng.lognormal(3.0, 1.2, G).astype(np.int64), 400).astype(np.uint64)
    lens = rng.integers(0, 17, E)
    lens[rng.choice(E, 200, replace=False)] = rng.integers(17, 257, 200)
    lens[rng.choice(E, 8, replace=False)] = rng.integers(257, 500, 8)
    lut = precalc_lls(sizes)
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    grp = np.concatenate([np.sort(rng.choice(G, k, replace=False)) for k in lens]).astype(np.uint32)
    cnt = rng.integers(1, sizes[grp] + 1).astype(np.uint32)
    hashes, infos = [], []
    for host in (True, False):
        if host:
            monkeypatch.setenv("MSWEEP_HOST_PACK", "1")
        else:
            monkeypatch.delenv("MSWEEP_HOST_PACK")
        with Core(0) as core:
            core.set_pack_schedule(sched)     # (0: the cells keep their CSR order, hot ones first -- msw_core_set_pack_schedule)
            core.set_csr(rowptr, grp, cnt, lut, np.log(0.01), G)
            hashes.append(core.layout_hash())
            infos.append(core.layout_info())
    assert infos[0] == infos[1] and infos[0]["index_records"] == 1 and infos[0]["bank_scheduled"] == sched
    assert hashes[0] == hashes[1]


def test_hybrid_off_switch_gives_the_same_answer(oracle, monkeypatch):
    """MSWEEP_HYBRID=0: the all-memory tables of the previous layout (and 8-byte records where the byte offsets
    need them) -- same iterations, same abundances."""
    p = diverse(80_000, 3000, 23, max_other=10)
    monkeypatch.setenv("MSWEEP_FORCE_LDS", "10")
    out = []
    for hyb in ("1", "0"):
        monkeypatch.setenv("MSWEEP_HYBRID", hyb)
        with Core(0) as core:
            res, tr, logc, alpha0 = solve_csr(core, p)
            out.append((res, core.layout_info()))
    assert out[0][1]["index_records"] == 1 and out[1][1]["index_records"] == 0
    assert out[0][0]["iters"] == out[1][0]["iters"]
    assert_theta(out[0][0]["theta"], out[1][0]["theta"], rel=1e-9, abs_=1e-12)


def test_hybrid_replanned_when_the_slice_geometry_does_not_fit(oracle, monkeypatch):
    """The hot-segment split of a slice rides above its 27-bit row offset: a likelihood of 2^27 rows or more (34 GB of
    records) is planned again WITHOUT the hybrid area -- all-memory tables -- instead of being refused
    (host_pack.inc).  The limit is lowered by a developer switch to reach the branch at test size."""
    p = diverse(80_000, 3000, 23, max_other=10)
    monkeypatch.setenv("MSWEEP_FORCE_LDS", "10")
    with Core(0) as core:
        ref, _, logc, alpha0 = solve_csr(core, p)
        assert core.layout_info()["index_records"] == 1
    monkeypatch.setenv("MSWEEP_HYBRID_MAX_ROWS", "100")
    with Core(0) as core:
        res, _, _, _ = solve_csr(core, p)
        li = core.layout_info()
        assert li["index_records"] == 0 and li["rows"] > 100, li
    assert res["iters"] == ref["iters"]
    assert_theta(res["theta"], ref["theta"], rel=1e-9, abs_=1e-12)


@pytest.mark.parametrize("G", [12000, 19500])
def test_hybrid_with_many_groups(oracle, G):
    """More groups than the LDS images hold AND more table slots than fit beside what is left: pass B keeps its column
    sums in LDS (mode 3) or sweeps once per range of groups (mode 4, two runs at 19 500 groups), {e, w} / e_g come
    from memory, and the room that remains holds the head of the hybrid area."""
    p = diverse(120_000, G, 29, max_other=10)
    lut = precalc_lls(p["group_sizes"])
    with Core(0) as core:
        res, tr, logc, alpha0 = solve_csr(core, p)
        li = core.layout_info()
        print(li)
        assert li["groups_in_lds"] == 0 and li["passB_mode"] in (3, 4)
        assert li["index_records"] == 1 and li["record_bytes"] == 4, li
        ref = oracle.rcg_optl_csr(p["rowptr"], p["grp"], lutidx_of(p, lut), lut, np.log(0.01), G, logc, alpha0, trace=20)
        lockstep(tr, ref["trace"], 20)
        assert res["iters"] == ref["iters"]
        assert_theta(res["theta"], ref["theta"])


@pytest.mark.parametrize("multilane", ["1", "0"])
def test_hybrid_with_mid_length_ecs(oracle, monkeypatch, multilane):
    """Index records AND ECs of 17..256 cells: slices of 2..16 lanes per EC cut into hot and cold segments like any
    other (MSWEEP_MULTILANE=0: the streaming branch, every table entry from memory), a few ECs beyond 256 cells."""
    monkeypatch.setenv("MSWEEP_FORCE_LDS", "10")
    monkeypatch.setenv("MSWEEP_HYBRID_HOT", "256")
    monkeypatch.setenv("MSWEEP_MULTILANE", multilane)
    rng = np.random.default_rng(8)
    G, E = 600, 4000
    sizes = np.minimum(1 + rng.lognormal(3.0, 1.2, G).astype(np.int64), 400).astype(np.uint64)
    lens = rng.integers(0, 17, E)
    lens[rng.choice(E, 1500, replace=False)] = rng.integers(17, 257, 1500)
    lens[rng.choice(E, 6, replace=False)] = rng.integers(257, 500, 6)
    lut = precalc_lls(sizes)
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    grp = np.concatenate([np.sort(rng.choice(G, k, replace=False)) for k in lens]).astype(np.uint32)
    cnt = rng.integers(1, sizes[grp] + 1).astype(np.uint32)
    logc = np.log(rng.integers(1, 40, E).astype(float))
    alpha0 = rng.uniform(0.5, 2.0, G)
    with Core(0) as core:
        core.set_csr(rowptr, grp, cnt, lut, np.log(0.01), G)
        li = core.layout_info()
        assert li["index_records"] == 1 and li["slot_entries_in_lds"] == 256, li
        assert (sum(li["slices_by_lanes"][:-1]) > 0) == (multilane == "1")
        core.set_trace_theta(15)
        res = core.solve(logc, alpha0)
        tr = core.trace(15, with_theta=True)
        lutidx = (grp * lut.shape[1] + cnt).astype(np.uint32)
        ref = oracle.rcg_optl_csr(rowptr, grp, lutidx, lut, np.log(0.01), G, logc, alpha0, trace=15)
        lockstep(tr, ref["trace"], 15)
        assert res["iters"] == ref["iters"]
        assert_theta(res["theta"], ref["theta"])
