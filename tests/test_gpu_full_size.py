"""GPU: BASELINE.json's configurations AT FULL SIZE, in lock-step with the CPU oracle.

  cfg2  synthetic 1M ECs x 500 groups, dense likelihood          (rcg_optl's own dense boundary)
  cfg3  synthetic 10M reads x 5k groups, CSR-of-ECs               (the bench workload)
  cfg4  cfg3 + bootstrap: replicates of the one mt19937_64 stream (counts bit-exact, every solve checked)
  cfg5  sparse 50M reads x 20k groups, --min-hits 1, device likelihood build on ONE GPU (288 GB)

Protocol: the structured oracle (oracle/rcg_oracle.cpp, fixed EC chunks under
OpenMP) runs the same number of iterations as the HIP path with a theta snapshot per iteration:
  (i)   k <= 20: theta / bound at rel 1e-9, |g|^2 at 1e-7, identical reset decisions;
  (ii)  EVERY iteration up to the last one both sides ran: theta at the north-star tolerance
        (rel 1e-6 on weights >= 1e-4, abs 1e-8 below) -- the worst component is printed;
  (iii) the stop.  ROOT CAUSE of the one-to-eight-iteration differences seen at this size (round 2: 209 / 208 ...,
        round 3: 229 / 221 on the diverse-size input, with theta equal to 2e-10 at every iteration): the bound is
        ~1e8 and is the sum of per-group terms lgamma(N_g) and (M - u_g) Nc_g of magnitude 1e7 each; evaluated in
        fp64 from fp64 N_g and u_g they carry ~1e-9 of rounding each, ~1e-7 in total, in the oracle (whose long
        double accumulators do not help: the noise is in the terms) as in the kernels -- measured |bound_hip -
        bound_oracle| <= 3.2e-7 with theta equal to 1e-13 (gpurun_out r3_stop_debug.log, tools/stop_debug.py) -- while
        the rule compares a gain with 1e-6 and the gains sit within 2e-7 of it for the last five to ten
        iterations.  Which of those iterations stops is decided by that noise on either side (the reference's own
        count moves by ten with -t, docs/gpubenchmarks.md:15-17).  The gate is therefore what the cause allows
        and no more: with NOISE = 5e-7, the HIP path must not stop while the oracle's gain is still above
        tol + NOISE, and must have stopped once it has fallen below tol - NOISE.
  (iv)  the ANSWERS: the theta the HIP path returns at ITS stop against the theta the oracle returns at ITS OWN stop
        (rcgpar's rule on the oracle's own bounds) -- the north star is worded on the result, and a differing stop
        is harmless only if this holds.  Always compared, worst component printed.
  (v)   the bench's own layout (5000 groups: group vectors + slot table + replicas in LDS, pass-B mode 2) against the
        reference-SHAPED algorithm -- the dense-state oracle with rcgpar's four G x E matrices -- to convergence, on
        as many of cfg3's ECs as 16 GB of dense state hold; once more on a bootstrap replicate (-inf log counts).
"""
import os
import time

import numpy as np
import pytest

from conftest import dense_from_csr, lutidx_of
from msweep_amd import synth
from msweep_amd.likelihood import from_alignment, from_dense, from_grouped_counts, precalc_lls
from test_gpu_rcg import FLOOR, REL, ABS, lockstep

pytestmark = pytest.mark.gpu
TOL = 1e-6
NOISE = 5e-7          # rounding noise of a bound difference at ~1e8 (measured <= 3.2e-7): module docstring, (iii)
INIT_BOUND = -100000.0


def worst(got, ref):
    """(worst rel err on weights >= FLOOR, its index, worst abs err below)"""
    got, ref = np.asarray(got), np.asarray(ref)
    big = ref >= FLOOR
    rel = np.where(big, np.abs(got - ref) / np.where(big, ref, 1.0), 0.0)
    ab = np.where(~big, np.abs(got - ref), 0.0)
    return float(rel.max(initial=0.0)), int(rel.argmax()) if rel.size else -1, float(ab.max(initial=0.0))


def oracle_stop(tr, tol=TOL):
    """rcgpar's stop rule (bound - oldbound < tol and no reset) on a recorded bound trace: the
    number of iterations the oracle itself would have run (None: not within the trace)."""
    old = INIT_BOUND
    for k, (b, rs) in enumerate(zip(tr["bound"], tr["didreset"])):
        if rs < 0 or np.isnan(b):
            return None
        if b - old < tol and not rs:
            return k + 1
        old = b
    return None


def assert_stop_within_noise(k_gpu, ref_tr, tag=""):
    """(iii): the HIP path stopped after k_gpu iterations -- allowed iff the oracle's own gains say that the stop
    rule was within its rounding noise of firing there: no earlier iteration with a gain below tol - NOISE (it
    should have stopped before), and a gain below tol + NOISE at k_gpu itself."""
    earliest = oracle_stop(ref_tr, TOL + NOISE)     # nobody may stop before this
    latest = oracle_stop(ref_tr, TOL - NOISE)       # everybody has stopped by then (None: beyond the trace)
    print(f"{tag}: stop window allowed by the oracle's gains +- {NOISE:.0e}: [{earliest}, {latest}], hip {k_gpu}")
    assert earliest is not None and k_gpu >= earliest, (k_gpu, earliest)
    assert latest is None or k_gpu <= latest, (k_gpu, latest)


def check_against_oracle(tag, res, tr, ref_tr):
    """(i)-(iii) of the module docstring; tr / ref_tr: per-iteration traces with theta."""
    lockstep(tr, ref_tr, 20)
    k_gpu = res["iters"]
    k_orc = oracle_stop(ref_tr)
    n = min(k_gpu, tr["n"], int(np.sum(ref_tr["didreset"] >= 0)))
    assert tr["didreset"][:n].tolist() == ref_tr["didreset"][:n].tolist()
    w_rel = w_abs = 0.0
    w_at = (-1, -1)
    for k in range(n):
        r, g, a = worst(tr["theta"][k], ref_tr["theta"][k])
        if r > w_rel:
            w_rel, w_at = r, (k, g)
        w_abs = max(w_abs, a)
    print(f"{tag}: iterations hip {k_gpu} / oracle {k_orc}; over all {n} common iterations worst rel err "
          f"{w_rel:.2e} (iteration {w_at[0]}, group {w_at[1]}, theta {ref_tr['theta'][w_at[0]][w_at[1]]:.3e}), "
          f"worst abs err below the floor {w_abs:.2e}; bound rel diff at the end "
          f"{abs(tr['bound'][n - 1] - ref_tr['bound'][n - 1]) / abs(ref_tr['bound'][n - 1]):.1e}")
    assert w_rel <= REL and w_abs <= ABS
    assert_stop_within_noise(k_gpu, ref_tr, tag)
    assert_returned_theta(tag, res["theta"], k_gpu, ref_tr)
    assert res["theta"].sum() == pytest.approx(1.0, abs=1e-11)


def assert_returned_theta(tag, theta_gpu, k_gpu, ref_tr):
    """(iv): what each side RETURNS -- hip's theta after its k_gpu iterations against the oracle's after ITS OWN stop.
    Same stop: the north-star tolerance.  Different stops (inside the noise window of (iii)): the difference is
    printed and must be EXPLAINED, component by component, by the oracle's own movement between the two iterations
    (every common iteration already agrees at the north-star tolerance) -- and that movement is reported as what it
    is, a finding about the stop rule: `bound - oldbound < 1e-6` on a bound of 1e8 fires while the smallest weights
    above the floor still move by 1e-5..1e-4 per iteration, so the rule itself -- in rcgpar as here -- determines
    them to 1e-4..1e-3, not to 1e-6 (the reference's own count moves 205 -> 215 with -t, docs/gpubenchmarks.md:15-17)."""
    k_orc = oracle_stop(ref_tr)
    assert k_orc is not None, f"{tag}: the oracle's own stop lies beyond its trace ({len(ref_tr['bound'])} iterations)"
    th_o = ref_tr["theta"][k_orc - 1]
    r, g, a = worst(theta_gpu, th_o)
    print(f"{tag}: RETURNED theta, hip at its stop ({k_gpu}) vs oracle at its own ({k_orc}): worst rel err {r:.2e} "
          f"(group {g}, theta {th_o[g]:.3e}), worst abs err below the floor {a:.2e}")
    if k_gpu == k_orc:
        assert r <= REL and a <= ABS, (tag, k_gpu, k_orc, r, a)
        return
    th_same = ref_tr["theta"][k_gpu - 1]           # the oracle's iterate where hip stopped
    rd, gd, ad = worst(th_same, th_o)
    print(f"{tag}: FINDING -- the stops differ ({k_gpu} / {k_orc}); the oracle's OWN theta moves by rel {rd:.2e} "
          f"(group {gd}, theta {th_o[gd]:.3e}) / abs {ad:.2e} below the floor between those two iterations: at this "
          f"--tol the stop rule leaves the weights determined to that")
    allowed = np.where(th_o >= FLOOR, REL * th_o, ABS) + np.abs(th_same - th_o)
    excess = np.abs(theta_gpu - th_o) - allowed
    assert np.all(excess <= 0), (tag, int(excess.argmax()), float(excess.max()))
    assert r <= 2e-3, (tag, r)                      # and nothing beyond what such a stop difference has ever shown


@pytest.fixture(scope="module")
def cfg3():
    t0 = time.time()
    p = synth.make_csr_problem(10_000_000, 5000, seed=2)
    lut = precalc_lls(p["group_sizes"])
    p["lut"], p["lutidx"] = lut, lutidx_of(p, lut)
    print(f"cfg3 generated in {time.time() - t0:.0f} s: E = {len(p['rowptr']) - 1}, nnz = {len(p['grp'])}")
    return p


@pytest.fixture()
def oracle_mt(oracle):
    """The structured oracle's EC loops are chunked for OpenMP: use the box's CPU share."""
    from conftest import cpu_share
    n = oracle.num_threads()
    oracle.set_num_threads(cpu_share())
    yield oracle
    oracle.set_num_threads(min(4, n))


def oracle_csr_trace(oracle, p, logc, alpha0, n_iters):
    """The structured oracle in EXTENDED precision (orc_rcg_opts::extended, round 5): the judge at the sizes the
    dense-state oracle cannot run.  MSWEEP_TEST_ORACLE_FP64=1: the fp64 arithmetic of rounds 1-4."""
    G = len(p["group_sizes"])
    return oracle.rcg_optl_csr(p["rowptr"], p["grp"], p["lutidx"], p["lut"], np.log(0.01), G, logc, alpha0,
                               tol=-1.0, max_iters=n_iters, trace=n_iters,
                               extended=os.environ.get("MSWEEP_TEST_ORACLE_FP64", "0") != "1")["trace"]


def test_cfg3_csr_10M_x_5k_lockstep(gpu_core, oracle_mt, cfg3):
    p, G = cfg3, 5000
    alpha0 = np.ones(G)
    lik = from_grouped_counts(gpu_core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
    gpu_core.set_trace_theta(1024)
    res = gpu_core.solve(lik.log_counts(), alpha0)
    assert res["iters"] < 1000
    tr = gpu_core.trace(res["iters"], with_theta=True)
    gpu_core.set_trace_theta(0)
    t0 = time.time()
    ref_tr = oracle_csr_trace(oracle_mt, p, lik.log_counts(), alpha0, res["iters"] + 16)
    print(f"oracle: {res['iters'] + 2} iterations in {time.time() - t0:.0f} s")
    check_against_oracle("cfg3", res, tr, ref_tr)
    ok = tr["didreset"] == 0
    assert np.all(np.diff(tr["bound"])[ok[1:]] > -1e-5)            # ELBO monotone over accepted steps
    # run-to-run reproducibility at this size: the column sums are integer (fixed-point) sums, so a second
    # and a third solve stop at the same iteration with the same bits (fp64 atomics: 209 / 210 / 211 and
    # theta 2e-4 apart, profiles/r02_fixedpoint_vs_fp64_atomics.txt)
    for _ in range(2):
        again = gpu_core.solve(lik.log_counts(), alpha0)
        assert again["iters"] == res["iters"] and again["bound"] == res["bound"]
        np.testing.assert_array_equal(again["theta"], res["theta"])


def test_diverse_group_sizes_10M_x_5k_lockstep(gpu_core, oracle_mt):
    """cfg3's reads and groups with the group sizes of real groupings (log-normal, up to 400 sequences: 12 910 used
    (size, count) pairs, 207 KB of slot table per sweep): the hybrid slot area -- index records, the most-used
    entries in LDS, cold segments from memory -- at full size, every iteration against the structured oracle."""
    G = 5000
    p = synth.make_csr_problem(10_000_000, G, seed=2, group_sizes=synth.diverse_group_sizes)
    lut = precalc_lls(p["group_sizes"])
    p["lut"], p["lutidx"] = lut, lutidx_of(p, lut)
    alpha0 = np.ones(G)
    lik = from_grouped_counts(gpu_core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
    li = gpu_core.layout_info()
    print(li)
    assert li["index_records"] == 1 and li["record_bytes"] == 4 and 0 < li["slot_entries_in_lds"] < li["slot_entries"]
    assert 0 < li["rows_from_memory"] < li["rows"] // 10
    gpu_core.set_trace_theta(1024)
    res = gpu_core.solve(lik.log_counts(), alpha0)
    assert res["iters"] < 1000
    tr = gpu_core.trace(res["iters"], with_theta=True)
    gpu_core.set_trace_theta(0)
    ref_tr = oracle_csr_trace(oracle_mt, p, lik.log_counts(), alpha0, res["iters"] + 16)
    check_against_oracle("diverse sizes", res, tr, ref_tr)
    again = gpu_core.solve(lik.log_counts(), alpha0)
    assert again["iters"] == res["iters"] and again["bound"] == res["bound"]
    np.testing.assert_array_equal(again["theta"], res["theta"])


def test_cfg4_bootstrap_10M_x_5k(gpu_core, oracle_mt, cfg3):
    """cfg4 on one GPU: msw_core_bootstrap over the resident cfg3 likelihood.  The resampled counts of
    every replicate are bit-exact with libstdc++'s mt19937_64 + discrete_distribution (the types
    src/BootstrapSample.cpp:33-73 instantiates); every replicate's abundances against an oracle solve
    on those counts."""
    p, G, B = cfg3, 5000, 3      # (three full-size oracle traces: 100 s of the suite; replicate 999: the test below)
    alpha0 = np.ones(G)
    from_grouped_counts(gpu_core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
    w = p["ec_counts"].astype(np.uint32)
    draws = int(w.sum())
    assert draws == 10_000_000
    t0 = time.time()
    counts = oracle_mt.bootstrap_counts(w, 42, draws, B)
    t1 = time.time()
    got = gpu_core.resample_counts(w, 42, draws, 0, B)
    np.testing.assert_array_equal(got, counts)
    # a later slice of the stream on its own: what rank 1 of 2 would draw
    np.testing.assert_array_equal(gpu_core.resample_counts(w, 42, draws, 1, B), counts[1:])
    t2 = time.time()
    theta, iters = gpu_core.bootstrap(w, 42, draws, 0, B, alpha0)
    t3 = time.time()
    print(f"cfg4: libstdc++ resampling {t1 - t0:.1f} s, device resampling + download {t2 - t1:.1f} s, "
          f"{B} replicates solved in {t3 - t2:.2f} s ({iters.tolist()} iterations)")
    for b in range(B):
        with np.errstate(divide="ignore"):
            logc = np.log(counts[b].astype(float))                   # -inf for ECs drawn zero times (:70)
        k = int(iters[b])
        ref_tr = oracle_csr_trace(oracle_mt, p, logc, alpha0, k + 16)
        k_orc = oracle_stop(ref_tr)
        r, g, a = worst(theta[b], ref_tr["theta"][k - 1])
        print(f"cfg4 replicate {b}: iterations hip {k} / oracle {k_orc}; theta after {k} iterations: worst rel err "
              f"{r:.2e} (group {g}, theta {ref_tr['theta'][k - 1][g]:.3e}), worst abs err below the floor {a:.2e}")
        assert r <= REL and a <= ABS
        assert_stop_within_noise(k, ref_tr, f"cfg4 replicate {b}")      # see the module docstring, (iii)
        assert_returned_theta(f"cfg4 replicate {b}", theta[b], k, ref_tr)  # (iv)
        assert theta[b].sum() == pytest.approx(1.0, abs=1e-11)       # normalised by the resampled total (:513)


def test_cfg4_last_replicate_of_1000_is_bit_exact(gpu_core, oracle_mt, cfg3):
    """BASELINE config 4 is `--iters 1000` (src/mSWEEP.cpp:132,498): replicate 999 of cfg3's counts starts
    999 * 10^7 = 9.99e9 words into the ONE mt19937_64(--seed 42) stream.  The device gets there by the host's GF(2)
    jump-ahead; the reference by drawing the 999 replicates before it.  tests/golden/mt_deep_state.json holds
    libstdc++'s OWN generator state at that word (84 s of std::mt19937_64::discard, gen_mt_deep_state.cpp): the
    oracle resumes libstdc++'s mt19937_64 + discrete_distribution<uint32_t> from it (10^7 draws over cfg3's 9.4 M
    ECs) and the device's counts must be the same bits.  Then the cumulative table: a second bootstrap call with the
    same EC counts must reuse the resident one (33.6 ms per call in round 4)."""
    import json
    p, G = cfg3, 5000
    from_grouped_counts(gpu_core, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
    w = p["ec_counts"].astype(np.uint32)
    draws = int(w.sum())
    assert draws == 10_000_000
    case = next(c for c in json.load(open(os.path.join(os.path.dirname(__file__), "golden", "mt_deep_state.json")))["cases"]
                if c["seed"] == 42 and c["skip"] == 999 * draws)
    t0 = time.time()
    ref = oracle_mt.bootstrap_counts_from_state(w, np.array(case["state"], np.uint64), case["pos"], draws, 1)
    t1 = time.time()
    got = gpu_core.resample_counts(w, 42, draws, 999, 1000)
    t2 = time.time()
    print(f"replicate 999: libstdc++ from its own state at word {case['skip']}: {t1 - t0:.1f} s; device (jump-ahead + "
          f"resampling + download) {t2 - t1:.2f} s; {int((got[0] == 0).sum())} ECs drawn zero times")
    np.testing.assert_array_equal(got, ref)
    assert int(got.sum()) == draws
    # ... and a block that STEPS over a replicate boundary after the jump (998 -> 999), as a rank's block does
    np.testing.assert_array_equal(gpu_core.resample_counts(w, 42, draws, 998, 1000)[1], ref[0])
    # the cumulative table stays resident across calls that bring the same counts
    alpha0 = np.ones(G)
    th1, it1 = gpu_core.bootstrap(w, 42, draws, 999, 1000, alpha0)
    bt1 = gpu_core.last_bootstrap_timing()
    th2, it2 = gpu_core.bootstrap(w, 42, draws, 999, 1000, alpha0)
    bt2 = gpu_core.last_bootstrap_timing()
    print(f"table_ms: first bootstrap call after resample_counts {bt1['table_ms']:.2f} (reused {bt1['table_reused']}), "
          f"second {bt2['table_ms']:.2f} (reused {bt2['table_reused']})")
    assert bt1["table_reused"] == 1 and bt2["table_reused"] == 1 and bt2["table_ms"] < 5.0
    np.testing.assert_array_equal(th1, th2)
    assert it1.tolist() == it2.tolist()
    # (the solve of a replicate against the oracle: test_cfg4_bootstrap_10M_x_5k; here the replicate's abundances only
    # have to be a distribution -- a full-size oracle trace costs 90 s of the suite)
    assert np.all(np.isfinite(th1)) and abs(th1[0].sum() - 1.0) < 1e-11 and 150 < int(it1[0]) < 300
    # changed counts are NOT served from the kept table
    w2 = w.copy()
    w2[len(w2) // 2] += 1
    gpu_core.bootstrap(w2, 42, draws, 0, 1, alpha0)
    assert gpu_core.last_bootstrap_timing()["table_reused"] == 0


def test_cfg2_dense_1M_x_500_lockstep(gpu_core, oracle_mt):
    """cfg2's own criterion ("match CPU abundances to 1e-6") at its own size, through the dense
    boundary rcg_optl hands over (msw_core_set_dense_logl), against the structured oracle on the dense
    matrix; the dense-state oracle (rcgpar's four G x E matrices, 16 GB) as soft cross-check."""
    E, G = 1_000_000, 500
    t0 = time.time()
    p = synth.make_dense_problem(E, G, seed=1)
    alpha0 = np.ones(G)
    print(f"cfg2 generated in {time.time() - t0:.0f} s")
    from_dense(gpu_core, p["logl"], p["logc"])
    gpu_core.set_trace_theta(1024)
    res = gpu_core.solve(p["logc"], alpha0)
    tr = gpu_core.trace(res["iters"], with_theta=True)
    gpu_core.set_trace_theta(0)
    assert res["iters"] < 1000
    t0 = time.time()
    n = res["iters"] + 16
    ref_tr = oracle_mt.rcg_optl_dense_structured(p["logl"], p["logc"], alpha0, tol=-1.0, max_iters=n, trace=n)["trace"]
    print(f"structured oracle: {n} iterations in {time.time() - t0:.0f} s")
    check_against_oracle("cfg2", res, tr, ref_tr)
    # soft cross-check: the first iterations of the dense-state algorithm as the reference structures it
    t0 = time.time()
    d = oracle_mt.rcg_optl_dense(p["logl"], p["logc"], alpha0, tol=-1.0, max_iters=8, trace=8)
    print(f"dense-state oracle: 8 iterations in {time.time() - t0:.0f} s")
    lockstep(tr, d["trace"], 8, rel=1e-8)


def theta_gates(tag, got, ref):
    r, g, a = worst(got, ref)
    print(f"{tag}: worst rel err {r:.2e} (group {g}, theta {ref[g]:.3e}), worst abs err below the floor {a:.2e}")
    assert r <= REL and a <= ABS, (tag, r, a)


def test_cfg3_layout_vs_dense_state_oracle_to_convergence(gpu_core, oracle_mt, cfg3):
    """(v) of the module docstring.  The first N_DENSE ECs of cfg3 with all 5000 groups: the HIP CSR path in the layout
    the bench runs (asserted) against rcgpar's algorithm AS THE REFERENCE STRUCTURES IT (oracle.rcg_optl_dense: gamma,
    step, oldstep and L as G x E fp64 matrices, log domain throughout, no background trick, no guard) --
    lock-step over the first 20 iterations, the converged abundances each side returns, and the same once more for a
    bootstrap replicate of those ECs (a third of the log counts -inf, src/BootstrapSample.cpp:70).
    N_DENSE: 30 000 ECs by default (1.2 GB per matrix, ~80 s of oracle time: the driver gives the whole GPU suite 15
    minutes); MSWEEP_TEST_DENSE_ECS=100000 is the size the round-3 review asked for -- its output of round 4 is kept in
    profiles/r04_parity_full_size.txt (4 GB per matrix, 270 s)."""
    N_DENSE, G = int(os.environ.get("MSWEEP_TEST_DENSE_ECS", "30000")), 5000
    p = cfg3
    nz = int(p["rowptr"][N_DENSE])
    q = dict(rowptr=np.ascontiguousarray(p["rowptr"][:N_DENSE + 1]), grp=np.ascontiguousarray(p["grp"][:nz]),
             cnt=np.ascontiguousarray(p["cnt"][:nz]), ec_counts=np.ascontiguousarray(p["ec_counts"][:N_DENSE]),
             group_sizes=p["group_sizes"])
    alpha0 = np.ones(G)
    lik = from_grouped_counts(gpu_core, q["rowptr"], q["grp"], q["cnt"], q["ec_counts"], q["group_sizes"])
    li = gpu_core.layout_info()
    print(li)
    assert li["record_bytes"] == 4 and li["index_records"] == 0 and li["groups_in_lds"] == 1
    assert li["table_in_lds"] == 1 and li["passB_mode"] == 2          # the bench's layout (BENCH_r03 "layout")
    L = dense_from_csr(q, p["lut"])                                   # 4 GB
    w = q["ec_counts"].astype(np.uint32)
    counts = gpu_core.resample_counts(w, 42, int(w.sum()), 0, 1)[0]
    np.testing.assert_array_equal(counts, oracle_mt.bootstrap_counts(w, 42, int(w.sum()), 1)[0])
    with np.errstate(divide="ignore"):
        cases = [("cfg3 slice", lik.log_counts()), ("cfg3 slice, bootstrap replicate", np.log(counts.astype(float)))]
    from conftest import cpu_share
    noise_r = noise_a = 0.0
    for tag, logc in cases:
        gpu_core.set_trace_theta(32)
        res = gpu_core.solve(logc, alpha0)
        tr = gpu_core.trace(min(res["iters"], 32), with_theta=True)
        gpu_core.set_trace_theta(0)
        t0 = time.time()
        d = oracle_mt.rcg_optl_dense(L, logc, alpha0, trace=32)
        th_d = oracle_mt.mixture_components(d["gamma"], logc)
        print(f"{tag}: {N_DENSE} ECs x {G} groups, iterations hip {res['iters']} / dense-state oracle {d['iters']} "
              f"({time.time() - t0:.0f} s); zero-count ECs: {int(np.sum(np.isinf(logc)))}")
        k = min(20, res["iters"], d["iters"])
        assert tr["didreset"][:k].tolist() == d["trace"]["didreset"][:k].tolist()
        np.testing.assert_allclose(tr["bound"][:k], d["trace"]["bound"][:k], rtol=1e-9)
        w_lock = 0.0
        for i in range(k):
            r, g, a = worst(tr["theta"][i], d["trace"]["theta"][i])
            w_lock = max(w_lock, r)
            assert r <= REL and a <= ABS, (tag, i, r, a)
        print(f"{tag}: lock-step with the dense-state oracle over {k} iterations: worst rel err {w_lock:.2e}")
        assert abs(res["iters"] - d["iters"]) <= 3
        r, g, a = worst(res["theta"], th_d)
        print(f"{tag}: RETURNED theta, hip at its stop vs the dense-state oracle at its own: worst rel err {r:.2e} "
              f"(group {g}, theta {th_d[g]:.3e}), worst abs err below the floor {a:.2e}")
        self_r = self_a = 0.0
        if tag == "cfg3 slice":
            # How far the reference-shaped algorithm is from ITSELF when only the order of its additions changes
            # (another OpenMP thread count: other partial sums in update_N_k / ELBO / |g|^2 -- what `-t` does to the
            # reference, docs/gpubenchmarks.md:15-17): the recursion amplifies rounding by ~10 x per 10-20 iterations
            # (SURVEY.md 7.3b), so two arithmetically different evaluations of the same mathematics cannot agree
            # better than this after ~80 iterations, whichever two they are.
            n_thr = oracle_mt.num_threads()
            oracle_mt.set_num_threads(max(1, cpu_share() * 2 // 3))
            t0 = time.time()
            d2 = oracle_mt.rcg_optl_dense(L, logc, alpha0)
            oracle_mt.set_num_threads(n_thr)
            self_r, g2, self_a = worst(oracle_mt.mixture_components(d2["gamma"], logc), th_d)
            print(f"{tag}: the dense-state oracle against ITSELF on {max(1, cpu_share() * 2 // 3)} instead of {n_thr} threads "
                  f"({time.time() - t0:.0f} s): iterations {d2['iters']} / {d['iters']}, worst rel diff {self_r:.2e} (group {g2}), "
                  f"worst abs diff below the floor {self_a:.2e}")
            del d2
        noise_r, noise_a = max(noise_r, self_r), max(noise_a, self_a)   # (the replicate is gated by the first case's)
        # the gate: the north-star tolerance, widened by no more than the reference-shaped algorithm's own
        # sensitivity to the order of its additions, measured above on this very input (x 3: one sample of a noise)
        assert r <= max(REL, 3.0 * noise_r) and a <= max(ABS, 3.0 * noise_a), (tag, r, a, noise_r)
        assert r <= 2e-5, (tag, r)         # 1e-7 .. 1e-5 expected; beyond that it is a bug
        del d


def test_cfg5_sparse_50M_x_20k_min_hits(gpu_core, oracle_mt):
    """cfg5 on one GPU: 50M reads x 20k groups, theta supported on 2 000 groups, <= 8 listed groups per
    read, through msw_core_build_likelihood with --min-hits 1 (include/Likelihood.hpp:141-171): mask
    equality, the compacted likelihood against the host's own compaction in lock-step with the oracle."""
    G = 20_000
    t0 = time.time()
    p = synth.make_csr_problem(50_000_000, G, seed=3, max_other=7, theta_support=2000, chunk=2_000_000)
    t1 = time.time()
    aln = synth.csr_to_targets(p, shuffle=False)
    t2 = time.time()
    E, nnz = len(p["rowptr"]) - 1, len(p["grp"])
    print(f"cfg5 generated in {t1 - t0:.0f} s (+ {t2 - t1:.0f} s for {len(aln['ec_targets'])} target hits): "
          f"E = {E}, nnz = {nnz}")
    lik = from_alignment(gpu_core, aln["ec_tptr"], aln["ec_targets"], aln["target_group"], p["group_sizes"],
                         p["ec_counts"], min_hits=1)
    t3 = time.time()
    del aln
    hit = np.zeros(G, bool)
    hit[p["grp"]] = True
    np.testing.assert_array_equal(lik.groups_considered(), hit)        # groups with >= 1 aligned read
    G2 = lik.n_groups
    assert G2 == int(hit.sum()) and 1900 <= G2 <= 2000                # ~90 % of the groups pruned
    assert gpu_core.shape() == (G2, E, nnz)
    np.testing.assert_array_equal(lik.log_counts(), np.log(p["ec_counts"].astype(np.float64)))
    alpha0 = np.ones(G2)
    gpu_core.set_trace_theta(64)
    res = gpu_core.solve(None, alpha0)                                # log counts left on the device by the build
    t4 = time.time()
    tr = gpu_core.trace(min(res["iters"], 64), with_theta=True)
    gpu_core.set_trace_theta(0)
    print(f"cfg5: build {t3 - t2:.1f} s, {G2} of {G} groups kept, {res['iters']} iterations in {t4 - t3:.2f} s")
    assert res["iters"] < 5000 and res["theta"].sum() == pytest.approx(1.0, abs=1e-11)
    kept = np.nonzero(hit)[0]
    assert res["theta"][p["theta_true"][kept] > 0].sum() > 0.99        # mass on the true support
    # the same problem compacted on the host (original group order among the kept groups, :168-171)
    newid = np.cumsum(hit) - 1
    q = dict(rowptr=p["rowptr"], grp=newid[p["grp"]].astype(np.uint32), cnt=p["cnt"],
             group_sizes=p["group_sizes"][kept])
    q["lut"] = precalc_lls(q["group_sizes"])
    q["lutidx"] = lutidx_of(q, q["lut"])
    n = min(res["iters"], 12)
    t0 = time.time()
    ref_tr = oracle_csr_trace(oracle_mt, q, lik.log_counts(), alpha0, n)
    print(f"oracle: {n} iterations in {time.time() - t0:.0f} s")
    lockstep(tr, ref_tr, n)
