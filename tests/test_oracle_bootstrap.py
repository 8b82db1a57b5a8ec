"""CPU: bootstrap resampling stream.  Golden vectors come from libstdc++'s own std::mt19937_64 +
std::discrete_distribution<uint32_t> (tests/golden/gen_bootstrap.cpp), the exact types the
reference uses (src/BootstrapSample.cpp:33-73)."""
import numpy as np

from conftest import load_golden


def test_mt19937_64_words(oracle):
    g = load_golden("bootstrap_golden.json")
    assert oracle.mt_words(42, 0, 8).tolist() == g["mt_first_words_seed42"]
    assert int(oracle.mt_words(5489, 9999, 1)[0]) == g["mt_word_10000_seed5489"]  # C++ standard check value


def test_stdlib_and_restated_match_golden(oracle):
    g = load_golden("bootstrap_golden.json")
    for c in g["cases"]:
        exp = np.array(c["counts"], np.uint32)
        for restated in (False, True):
            got = oracle.bootstrap_counts(c["weights"], c["seed"], c["draws"], len(exp), restated=restated)
            np.testing.assert_array_equal(got, exp)
        assert np.all(exp.sum(1) == c["draws"])


def test_restated_equals_stdlib_on_larger_inputs(oracle):
    rng = np.random.default_rng(9)
    w = rng.integers(0, 1000, 5000).astype(np.uint32)
    w[rng.random(5000) < 0.2] = 0
    a = oracle.bootstrap_counts(w, -123456, 20000, 3)
    b = oracle.bootstrap_counts(w, -123456, 20000, 3, restated=True)
    np.testing.assert_array_equal(a, b)
    assert np.all(a[:, w == 0] == 0)
    cp = oracle.discrete_cp(w)
    assert cp[-1] == 1.0 and np.all(np.diff(cp) >= 0)


def test_from_state_resumes_libstdcxx_deep_in_the_stream(oracle):
    """tests/golden/mt_deep_state.json: libstdc++'s own mt19937_64 state 9.99e9 words in (the start of replicate 999
    of BASELINE config 4) and the 16 words it produced next.  The oracle's from-state replicate loop must consume
    exactly those words: counts = lower_bound of double(word) * 2^-64 on the normalised partial sums."""
    g = load_golden("mt_deep_state.json")
    rng = np.random.default_rng(4)
    w = rng.integers(1, 50, 37).astype(np.uint32)
    cp = oracle.discrete_cp(w)
    for c in g["cases"]:
        words = np.array(c["next_words"], np.uint64)
        p = np.minimum(words.astype(np.float64) * 2.0 ** -64, np.nextafter(1.0, 0.0))
        exp = np.bincount(np.searchsorted(cp, p, side="left"), minlength=len(w)).astype(np.uint32)
        got = oracle.bootstrap_counts_from_state(w, np.array(c["state"], np.uint64), c["pos"], len(words), 1)
        np.testing.assert_array_equal(got[0], exp)
    # and at a depth the restated generator can step to: the seed-42 stream's words [1.5e9, 1.5e9 + 16)
    c = next(c for c in g["cases"] if c["skip"] == 1_500_000_000)
    assert oracle.mt_words(42, c["skip"], 16).tolist() == c["next_words"]
