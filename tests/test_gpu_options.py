"""GPU: the solver options of msw_core_set_option -- the knobs of rcgpar's loops that SURVEY.md 3.2 flags as
restated from memory -- each non-default value against the oracle run with the same knob
(oracle/msweep_oracle.h orc_rcg_opts.check_every / init_bound, orc_em_opts)."""
import numpy as np
import pytest

from conftest import dense_from_csr, lutidx_of
from msweep_amd import synth
from msweep_amd.core import ALGO_EM, MswError
from msweep_amd.likelihood import from_grouped_counts, precalc_lls
from test_gpu_rcg import assert_theta, lockstep

pytestmark = pytest.mark.gpu


@pytest.fixture()
def core_opts(gpu_core):
    """the session's handle with its options restored afterwards"""
    yield gpu_core
    for k, v in (("check_every", 1), ("init_bound", -100000.0), ("em_prior", 0), ("em_stop", 0)):
        gpu_core.set_option(k, v)


@pytest.fixture(scope="module")
def prob():
    p = synth.make_csr_problem(30000, 80, seed=61, max_other=6)
    p["lut"] = precalc_lls(p["group_sizes"])
    p["lutidx"] = lutidx_of(p, p["lut"])
    return p


def test_defaults_and_bad_values(core_opts):
    c = core_opts
    assert c.get_option("check_every") == 1 and c.get_option("init_bound") == -100000.0
    assert c.get_option("em_prior") == 0 and c.get_option("em_stop") == 0
    for k, v in (("check_every", 0), ("check_every", 2.5), ("em_prior", 2), ("em_stop", -1), ("init_bound", np.nan)):
        with pytest.raises(MswError):
            c.set_option(k, v)
    with pytest.raises(MswError):
        c.set_option("nonsense", 1)


def test_check_every_5_stops_on_the_grid(core_opts, oracle, prob):
    """MSW_OPT_CHECK_EVERY 5: the trajectory is the default's, the stop lands on the next multiple of 5 at or after the
    default's -- the shape of every iteration count the reference publishes (docs/gpubenchmarks.md:15-25)."""
    p, G = prob, 80
    alpha0 = np.ones(G)
    lik = from_grouped_counts(core_opts, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
    core_opts.set_trace_theta(512)
    d = core_opts.solve(lik.log_counts(), alpha0)
    core_opts.set_option("check_every", 5)
    r = core_opts.solve(lik.log_counts(), alpha0)
    tr = core_opts.trace(r["iters"], with_theta=True)
    core_opts.set_trace_theta(0)
    ref = oracle.rcg_optl_csr(p["rowptr"], p["grp"], p["lutidx"], p["lut"], np.log(0.01), G, lik.log_counts(), alpha0,
                              check_every=5, trace=r["iters"] + 5)
    print(f"check_every: default stops at {d['iters']}, on the 5-grid hip {r['iters']} / oracle {ref['iters']}")
    assert r["iters"] % 5 == 0 and d["iters"] <= r["iters"]
    assert r["iters"] == ref["iters"]
    lockstep(tr, ref["trace"], 20)
    assert_theta(r["theta"], ref["theta"])
    # the default trajectory is a prefix: bit-identical bounds up to the default's stop
    core_opts.set_option("check_every", 1)
    core_opts.set_trace_theta(0)
    d2 = core_opts.solve(lik.log_counts(), alpha0)
    assert d2["iters"] == d["iters"] and d2["bound"] == d["bound"]
    assert tr["bound"][d["iters"] - 1] == d["bound"]


def test_init_bound_minus_inf_changes_the_first_step_like_the_oracle(core_opts, oracle, prob):
    """MSW_OPT_INIT_BOUND: with -inf iteration 0 can never take the steepest-descent retry (SURVEY.md 3.2 dagger)."""
    p, G = prob, 80
    alpha0 = np.ones(G)
    lik = from_grouped_counts(core_opts, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
    core_opts.set_option("init_bound", -np.inf)
    core_opts.set_trace_theta(64)
    r = core_opts.solve(lik.log_counts(), alpha0)
    tr = core_opts.trace(min(r["iters"], 64), with_theta=True)
    core_opts.set_trace_theta(0)
    ref = oracle.rcg_optl_csr(p["rowptr"], p["grp"], p["lutidx"], p["lut"], np.log(0.01), G, lik.log_counts(), alpha0,
                              init_bound=-np.inf, trace=64)
    assert tr["didreset"][0] == 0 and ref["trace"]["didreset"][0] == 0
    lockstep(tr, ref["trace"], 20)
    assert abs(r["iters"] - ref["iters"]) <= 1
    assert_theta(r["theta"], ref["theta"])


@pytest.mark.parametrize("prior,stop", [("ml", "gain"), ("map", "theta"), ("ml", "theta")])
def test_em_variants_match_the_oracles(core_opts, oracle, prob, prior, stop):
    """em_torch is absent from the reference tree; ML / MAP and the two stop rules are explicit on both sides."""
    p, G = prob, 80
    alpha0 = np.full(G, 1.5)            # a prior that MAP and ML disagree on
    lik = from_grouped_counts(core_opts, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
    core_opts.set_option("em_prior", 1 if prior == "ml" else 0)
    core_opts.set_option("em_stop", 1 if stop == "theta" else 0)
    tol = 1e-7
    r = core_opts.solve(lik.log_counts(), alpha0, tol=tol, max_iters=20000, algo=ALGO_EM)
    L = dense_from_csr(p, p["lut"])
    ref = oracle.em_dense(L, lik.log_counts(), alpha0, tol=tol, max_iters=20000, prior=prior, stop=stop)
    dflt = oracle.em_dense(L, lik.log_counts(), alpha0, tol=tol, max_iters=20000)
    print(f"EM {prior}/{stop}: iterations hip {r['iters']} / oracle {ref['iters']} (default variant: {dflt['iters']})")
    assert abs(r["iters"] - ref["iters"]) <= 2
    np.testing.assert_allclose(r["theta"], ref["theta"], rtol=1e-6, atol=1e-9)
    if prior == "ml":   # and it IS a different estimate from the default's
        assert np.max(np.abs(ref["theta"] - dflt["theta"])) > 1e-5


def test_em_check_every(core_opts, oracle, prob):
    p, G = prob, 80
    alpha0 = np.ones(G)
    lik = from_grouped_counts(core_opts, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
    core_opts.set_option("check_every", 5)
    r = core_opts.solve(lik.log_counts(), alpha0, tol=1e-6, max_iters=20000, algo=ALGO_EM)
    ref = oracle.em_dense(dense_from_csr(p, p["lut"]), lik.log_counts(), alpha0, tol=1e-6, max_iters=20000, check_every=5)
    assert r["iters"] % 5 == 0 and abs(r["iters"] - ref["iters"]) <= 10
    np.testing.assert_allclose(r["theta"], ref["theta"], rtol=1e-6, atol=1e-9)


def test_options_reach_the_bootstrap_replicates(core_opts, oracle, prob):
    """the replicates run on solver states of their own (host_bootstrap.inc share_likelihood): they inherit the options"""
    p, G = prob, 80
    alpha0 = np.ones(G)
    from_grouped_counts(core_opts, p["rowptr"], p["grp"], p["cnt"], p["ec_counts"], p["group_sizes"])
    w = p["ec_counts"].astype(np.uint32)
    core_opts.set_option("check_every", 5)
    theta, iters = core_opts.bootstrap(w, 3, int(w.sum()), 0, 3, alpha0)
    assert np.all(iters % 5 == 0)
