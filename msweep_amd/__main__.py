"""`python -m msweep_amd` -- the estimation part of mSWEEP's command line (src/mSWEEP.cpp:68-148)
over the MI355X core: Themisto plaintext in, `<prefix>_abundances.txt` out.  Only the flags that
reach the hot path are accepted; binning, likelihood dumps and compression are out of scope."""
import argparse
import sys

import numpy as np

from . import parallel
from .alignment import Alignment
from .core import ALGO_EM, ALGO_RCG, PREC_DOUBLE, PREC_FLOAT, Core, MswError
from .likelihood import from_alignment
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
    ap.add_argument("--algorithm", default="rcggpu")
    ap.add_argument("--emprecision", default="double")
    ap.add_argument("--iters", type=int, default=0)
    ap.add_argument("--seed", type=int, default=26012023)
    ap.add_argument("--bootstrap-count", type=int, default=0)
    ap.add_argument("-q", type=float, default=0.65)
    ap.add_argument("-e", type=float, default=0.01)
    ap.add_argument("--alphas")
    ap.add_argument("--zero-inflation", type=float, default=0.01)
    ap.add_argument("--min-hits", type=int, default=0)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--verbose", action="store_true")
    return ap.parse_args(argv)


def main(argv=None):
    a = parse(sys.argv[1:] if argv is None else argv)
    try:
        with open(a.indicators) as f:
            grouping = read_reference(f)
        files = a.themisto.split(",") if a.themisto else [x for x in (a.themisto_1, a.themisto_2) if x]
        if not files:
            raise RuntimeError("no pseudoalignment files given")
        aln = Alignment(len(grouping.group_indicators))
        streams = [open(p) for p in files]
        aln.read(a.themisto_mode, streams)
        for s in streams:
            s.close()
        aln.collapse()
    except (RuntimeError, OSError) as ex:
        sys.stderr.write(f"Reading the pseudoalignments failed:\n  {ex}\nexiting\n")
        return 1
    if a.algorithm == "rcgcpu":
        sys.stderr.write("rcgcpu is the reference's CPU path; use rcggpu or emgpu with this core\n")
        return 1
    algo = ALGO_RCG if a.algorithm == "rcggpu" else ALGO_EM      # anything else -> em (src/mSWEEP.cpp:200)
    prec = PREC_FLOAT if a.emprecision == "float" else PREC_DOUBLE
    try:
        core = Core(a.device)
        lik = from_alignment(core, aln.ec_tptr, aln.ec_targets, grouping.group_indicators, grouping.get_sizes(),
                             aln.ec_counts, a.q, a.e, a.zero_inflation, a.min_hits)
    except MswError as ex:
        sys.stderr.write(f"Building the log-likelihood array failed:\n  {ex}\nexiting\n")
        return 1
    G = lik.n_groups
    prior = np.ones(G)
    if a.alphas:
        prior = np.array([float(x) for x in a.alphas.split(",")])
        if len(prior) != G:
            sys.stderr.write("Error: --alphas must have the same number of values as there are groups.")
            return 1
    total = int(aln.ec_counts.sum())
    sample = BootstrapSample(aln.n_reads(), total, a.iters) if a.iters > 0 else PlainSample(aln.n_reads(), total)
    try:
        res = core.solve(lik.log_counts(), prior, a.tol, a.max_iters, algo, prec)
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
            w = aln.ec_counts.astype(np.uint32)
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
    out = open(f"{a.prefix}_abundances.txt", "w") if a.prefix else sys.stdout
    if a.min_hits > 0:
        sample.write_abundances2(est, zero, out)
    else:
        sample.write_abundances(est, out)
    if a.prefix:
        out.close()
    core.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
