"""`python -m msweep_amd` -- the estimation part of mSWEEP's command line (src/mSWEEP.cpp:68-148)
over the MI355X core: Themisto plaintext in, `<prefix>_abundances.txt` out.  Only the flags that
reach the hot path are accepted; binning, likelihood dumps and compression are out of scope."""
import argparse
import sys

import numpy as np

from . import parallel
from .core import ALGO_EM, ALGO_RCG, PREC_DOUBLE, PREC_FLOAT, Core, MswError
from .likelihood import from_device_alignment, from_dense
from .reference import read_reference
from .sample import BootstrapSample, PlainSample


def parse(argv):
    ap = argparse.ArgumentParser(prog="python -m msweep_amd")
    ap.add_argument("--themisto-1")
    ap.add_argument("--themisto-2")
    ap.add_argument("--themisto", help="comma separated list of alignment files")
    ap.add_argument("--themisto-mode", default="intersection")
    ap.add_argument("-i", required=True, dest="indicators")
    ap.add_argument("-o", default="", dest="prefix")
    ap.add_argument("-t", type=int, default=1, help="accepted for compatibility (the GPU core ignores it)")
    ap.add_argument("--max-iters", type=int, default=5000)
    ap.add_argument("--tol", type=float, default=0.000001)
    ap.add_argument("--algorithm", default="rcgcpu")   # the reference's default (src/mSWEEP.cpp:127); served by the GPU RCG kernels
    ap.add_argument("--emprecision", default="double")
    ap.add_argument("--iters", type=int, default=0)
    ap.add_argument("--seed", type=int, default=26012023)
    ap.add_argument("--bootstrap-count", type=int, default=0)
    ap.add_argument("-q", type=float, default=0.65)
    ap.add_argument("-e", type=float, default=0.01)
    ap.add_argument("--alphas")
    ap.add_argument("--zero-inflation", type=float, default=0.01)
    ap.add_argument("--min-hits", type=int, default=0)
    ap.add_argument("--run-rate", action="store_true")
    ap.add_argument("--write-probs", action="store_true")
    ap.add_argument("--print-probs", action="store_true")
    ap.add_argument("--write-likelihood", action="store_true")
    ap.add_argument("--read-likelihood")
    ap.add_argument("--no-fit-model", action="store_true")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--verbose", action="store_true")
    return ap.parse_args(argv)


def read_likelihood_file(path, n_groups):
    counts, cols = [], []
    with open(path) as f:
        for line in f:
            parts = line.rstrip("\n").split("\t")
            if len(parts) != n_groups + 1:
                raise RuntimeError("Could not read from the likelihoods file.")
            counts.append(int(parts[0]))
            cols.append([float(x) for x in parts[1:]])
    return np.array(counts, np.uint64), np.ascontiguousarray(np.array(cols).T)


def write_likelihood_file(f, ec_counts, L):
    for j in range(L.shape[1]):
        f.write(str(int(ec_counts[j])) + "\t" + "\t".join("%g" % x for x in L[:, j]) + "\n")


def write_probs(of, names, zero_names, core, block=8192):
    """Sample::write_probs[2] (src/Sample.cpp:63-85,154-186), one line per EC, streamed from the device in
    blocks of ECs (msw_core_gamma_block): the G x E matrix is never held, here or there."""
    of.write("ec_id\t" + "\t".join(list(names) + list(zero_names)) + "\n")
    E = core.shape()[1]
    for e0 in range(0, E, block):
        probs = np.exp(core.gamma_block(e0, min(E, e0 + block)))
        for jj in range(probs.shape[1]):
            of.write(str(e0 + jj) + "\t" + "\t".join(["%g" % x for x in probs[:, jj]] + ["0"] * len(zero_names)) + "\n")
    of.write("\n")
    of.flush()


def _digamma(x):      # src/Sample.cpp:87-97
    r = 0.0
    while x < 7:
        r -= 1 / x
        x += 1
    x -= 0.5
    xx = 1.0 / x
    xx2 = xx * xx
    xx4 = xx2 * xx2
    return r + np.log(x) + (1. / 24.) * xx2 - (7.0 / 960.0) * xx4 + (31.0 / 8064.0) * xx4 * xx2 - (127.0 / 30720.0) * xx4 * xx4


def dirichlet_kld_rate(alphas):
    """Sample::dirichlet_kld + get_rates (src/Sample.cpp:99-152).  alphas_i = sum_j c_j exp(gamma_ij) is the
    column sum the solve already reduced on the device (theta_i * sum c)."""
    from math import lgamma
    a0 = float(np.sum(alphas))
    log_kld = np.array([np.log(max(lgamma(a0) - lgamma(a0 - aj) - lgamma(aj) + aj * (_digamma(aj) - _digamma(a0)), 1e-16))
                        for aj in map(float, alphas)])
    mx = max(0.0, float(log_kld.max()))
    lsum = np.log(np.exp(log_kld - mx).sum()) + mx
    return np.exp(log_kld), np.exp(log_kld - lsum)


def main(argv=None):
    a = parse(sys.argv[1:] if argv is None else argv)
    aln = None
    try:
        core = Core(a.device)
    except MswError as ex:
        sys.stderr.write(f"Initialising the GPU failed:\n  {ex}\nexiting\n")
        return 1
    try:
        with open(a.indicators) as f:
            grouping = read_reference(f)
        if not a.read_likelihood:
            files = a.themisto.split(",") if a.themisto else [x for x in (a.themisto_1, a.themisto_2) if x]
            if not files:
                raise RuntimeError("no pseudoalignment files given")
            # the reader on the device (msw_alignment_read_device): text -> equivalence classes in HBM, consumed there
            # by the likelihood build; the reference's messages for text it does not take
            aln = core.read_alignment(files, len(grouping.group_indicators), a.themisto_mode)
    except (RuntimeError, OSError, MswError) as ex:
        sys.stderr.write(f"Reading the pseudoalignments failed:\n  {ex}\nexiting\n")
        return 1
    if a.algorithm == "rcgcpu":
        # the reference's default: the same RCG algorithm on the host (rcgpar::rcg_optl_omp); this core runs it
        # on the GPU -- there is no CPU path here
        if a.verbose:
            sys.stderr.write("note: --algorithm rcgcpu is served by the GPU RCG kernels (same algorithm as rcggpu)\n")
    algo = ALGO_RCG if a.algorithm in ("rcggpu", "rcgcpu") else ALGO_EM   # anything else -> em (src/mSWEEP.cpp:200)
    prec = PREC_FLOAT if a.emprecision == "float" else PREC_DOUBLE
    # (--emprecision float: fp32 kernels where the layout allows, msweep_amd/csrc/em_f32_kernels.hpp; the library
    # reports which through msw_timing::em_float_kernels)
    try:
        # ordering the cells for the LDS banks pays from about the 1 000th iteration on: bootstrap runs (msw_core_set_pack_schedule)
        core.set_pack_schedule(a.iters >= 5)
        if a.read_likelihood:
            # --read-likelihood (include/Likelihood.hpp:224-253): "count \t L(0,j) ... L(G-1,j)" per EC
            ec_counts, L = read_likelihood_file(a.read_likelihood, grouping.get_n_groups())
            lik = from_dense(core, L, np.log(ec_counts.astype(np.float64)))
            n_reads = total_reads = int(ec_counts.sum())
        else:
            if aln.n_ecs == 0:
                raise RuntimeError("no read aligned against the reference")
            lik = from_device_alignment(core, aln, grouping.group_indicators, grouping.get_sizes(), a.q, a.e,
                                        a.zero_inflation, a.min_hits)
            ec_counts = aln.ec_counts()
            n_reads = aln.n_reads
    except (MswError, RuntimeError, OSError, ValueError) as ex:
        sys.stderr.write(f"Building the log-likelihood array failed:\n  {ex}\nexiting\n")
        return 1
    if a.write_likelihood:
        # --write-likelihood (include/Likelihood.hpp:255-273), default ostream precision; the file is
        # <prefix>_likelihoods.tsv (src/OutfileDesignator.cpp:67-74; the flag's help text says .txt)
        with open(f"{a.prefix}_likelihoods.tsv" if a.prefix else "likelihoods.tsv", "w") as f:
            write_likelihood_file(f, ec_counts, lik.log_mat())
    if a.no_fit_model:
        core.close()
        return 0
    G = lik.n_groups
    prior = np.ones(G)
    if a.alphas:
        prior = np.array([float(x) for x in a.alphas.split(",")])
        if len(prior) != G:
            sys.stderr.write("Error: --alphas must have the same number of values as there are groups.")
            return 1
    total = int(ec_counts.sum())
    sample = BootstrapSample(n_reads, total, a.iters) if a.iters > 0 else PlainSample(n_reads, total)
    try:
        # a likelihood built on the device keeps its log counts there: nothing to upload per solve
        res = core.solve(None if not a.read_likelihood else lik.log_counts(), prior, a.tol, a.max_iters, algo, prec)
        if a.verbose:
            t = core.trace(min(res["iters"], 4096))
            for k in range(0, t["n"], 5):
                sys.stderr.write(f"  iter: {k}, bound: {t['bound'][k]:g}, |g|: {t['newnorm'][k]:g}\n")
        sample.store_abundances(res["theta"])
        if a.iters > 0:
            if a.seed == 26012023:      # the reference's "random seed" sentinel (src/BootstrapSample.cpp:48-50)
                seed = int(np.random.SeedSequence().generate_state(1)[0] & 0x7fffffff)
            else:
                seed = ((a.seed + 2**31) % 2**32) - 2**31            # size_t -> int32 narrowing (Sample.hpp:169)
            # ConstructSample quirk (src/Sample.cpp:38-39): --bootstrap-count without --bin-reads
            # passes the number of ITERATIONS as the count
            draws = a.iters if a.bootstrap_count > 0 else total
            w = ec_counts.astype(np.uint32)
            thetas, _ = core.bootstrap(w, seed, draws, 0, a.iters, prior, a.tol, a.max_iters, algo, prec)
            for row in thetas:
                sample.store_abundances(row)
    except MswError as ex:
        sys.stderr.write(f"Estimating relative abundances failed:\n  {ex}\nexiting\n")
        return 1
    names = grouping.get_names()
    mask = lik.groups_considered()
    est = [n for n, m in zip(names, mask) if m]
    zero = [n for n, m in zip(names, mask) if not m]
    if a.write_probs or a.print_probs:
        # Sample::write_probs (src/Sample.cpp:63-85): header ec_id + group names, one row per EC of exp(gamma)
        for dst in ([open(f"{a.prefix}_probs.tsv", "w")] if a.write_probs and a.prefix else []) + \
                   ([sys.stdout] if a.print_probs or (a.write_probs and not a.prefix) else []):
            write_probs(dst, est, zero if a.min_hits > 0 else [], core)
            if dst is not sys.stdout:
                dst.close()
    out = open(f"{a.prefix}_abundances.txt", "w") if a.prefix else sys.stdout
    if a.run_rate:
        # experimental RATE / KLD (src/Sample.cpp:99-152, table written at src/mSWEEP.cpp:529-545)
        kld, rate = dirichlet_kld_rate(np.asarray(sample.get_abundances()) * total)
        sample._header(out)
        out.write("#c_id\tmean_theta\tRATE\tKLD\n")
        for n, t, r, k in zip(est, sample.get_abundances(), rate, kld):
            out.write(f"{n}\t{t:g}\t{r:g}\t{k:g}\n")
        for n in zero:
            out.write(f"{n}\t0\t0\t0\n")
        out.flush()
    elif a.min_hits > 0:
        sample.write_abundances2(est, zero, out)
    else:
        sample.write_abundances(est, out)
    if a.prefix:
        out.close()
    core.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
