"""Developer tool: randomised sweep of the device bootstrap resampler (msw_core_resample_counts: the reference's ONE
sequential mt19937_64 stream, generate_canonical, discrete_distribution's partial sums and lower_bound --
src/BootstrapSample.cpp:33-73 over libstdc++) against the oracle, which calls the same <random> types: bit-exact
counts for random weights (zeros, one-hot, heavy tails, equal), seeds, draw counts and replicate windows.
usage: fuzz_bootstrap.py [n_cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from msweep_amd.core import Core  # noqa: E402
from oracle import Oracle  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
O = Oracle()
core = Core(0)
for case in range(n_cases):
    E = int(rng.choice([1, 2, 3, 63, 64, 65, 1000, 4097, 100_000, 1_000_000]))
    kind = str(rng.choice(["ones", "small", "heavy", "zeros", "onehot", "huge"]))
    if kind == "ones":
        w = np.ones(E, np.uint32)
    elif kind == "small":
        w = rng.integers(1, 4, E).astype(np.uint32)
    elif kind == "heavy":
        w = np.minimum(rng.pareto(0.8, E) + 1, 4e6).astype(np.uint32)
    elif kind == "zeros":
        w = (rng.integers(0, 3, E) * (rng.random(E) < 0.3)).astype(np.uint32)
        w[rng.integers(0, E)] = 1
    elif kind == "onehot":
        w = np.zeros(E, np.uint32)
        w[rng.integers(0, E)] = int(rng.integers(1, 1000))
    else:
        w = rng.integers(1, 2**31, E).astype(np.uint32) if E <= 2 else rng.integers(1, (2**32 - 1) // E, E).astype(np.uint32)
    seed = int(rng.choice([0, 1, 42, 2**31 - 1, 2**32 - 1, int(rng.integers(0, 2**31))]))
    draws = int(rng.choice([1, 2, 100, 10_000, 300_000]))
    if rng.random() < 0.3:
        draws = int(min(int(w.astype(np.uint64).sum()), 2_000_000))      # the reference's default: as many as reads
    draws = max(draws, 1)
    r0 = int(rng.integers(0, 4))
    r1 = r0 + int(rng.integers(1, 4))
    tag = f"case {case}: E {E} weights {kind} seed {seed} draws {draws} replicates [{r0}, {r1})"
    try:
        got = core.resample_counts(w, seed, draws, r0, r1)
        want = O.bootstrap_counts(w, seed, draws, r1)[r0:r1]
        np.testing.assert_array_equal(got, want)
        assert np.all(got.sum(1) == draws)
    except Exception as ex:  # noqa: BLE001
        print("FAILED", tag, "::", str(ex)[:600], flush=True)
        sys.exit(1)
    if case % 10 == 0:
        print("ok", tag, flush=True)
print(f"{n_cases} cases passed")
core.close()
