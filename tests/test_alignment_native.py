"""Native Themisto reader + EC collapse (msw_alignment_*, host code of libmsweep_core.so) against the
pure-Python mirror of mSWEEP::Alignment (msweep_amd/alignment.py, include/mSWEEP_alignment.hpp:54-215)
on random single- and paired-end inputs, and the reference's error messages.  No GPU needed."""
import io

import numpy as np
import pytest

from msweep_amd.alignment import Alignment, ec_hash
from msweep_amd.core import MswError, read_alignment


def _write_strand(path, rng, n_reads, n_targets, p_unaligned=0.2, shuffle=True, dup_lines=True):
    lines = []
    for r in range(n_reads):
        if rng.random() < p_unaligned:
            lines.append(f"{r}")
            continue
        k = int(rng.integers(1, 7))
        t = rng.choice(n_targets, k, replace=False)
        lines.append(f"{r} " + " ".join(str(int(x)) for x in t))
    if dup_lines:  # the same read id on a second line: the sets are united (bits OR-ed, :60-66)
        for r in rng.choice(n_reads, max(1, n_reads // 20), replace=False):
            lines.append(f"{int(r)} {int(rng.integers(0, n_targets))}")
    if shuffle:
        rng.shuffle(lines)
    path.write_text("\n".join(lines) + "\n")
    return len(lines)


def _python_alignment(paths, n_targets, mode):
    a = Alignment(n_targets)
    streams = [open(p) for p in paths]
    a.read(mode, streams)
    for s in streams:
        s.close()
    # the reference only visits read ids below the last strand's line count (:148)
    a._reads = {r: t for r, t in a._reads.items() if r < a.n_queries}
    a.collapse()
    return a


@pytest.mark.parametrize("mode,n_strands", [("intersection", 1), ("intersection", 2), ("union", 2)])
def test_native_reader_matches_python_mirror(tmp_path, mode, n_strands):
    rng = np.random.default_rng(7 + n_strands + len(mode))
    n_targets, n_reads = 37, 600
    paths = []
    for s in range(n_strands):
        p = tmp_path / f"strand_{s}.txt"
        _write_strand(p, rng, n_reads, n_targets, dup_lines=(s == 0))
        paths.append(str(p))
    ref = _python_alignment(paths, n_targets, mode)
    got = read_alignment(paths, n_targets, mode)
    assert got["n_reads"] == ref.n_queries
    np.testing.assert_array_equal(got["ec_counts"], ref.ec_counts)
    np.testing.assert_array_equal(got["ec_tptr"], ref.ec_tptr)
    np.testing.assert_array_equal(got["ec_targets"], ref.ec_targets)
    rp = got["ec_rptr"].astype(int)
    assert [got["ec_reads"][rp[i]:rp[i + 1]].tolist() for i in range(len(rp) - 1)] == ref.ec_read_ids
    # ECs in ascending order of the reference's hash (std::map, :186)
    tp = got["ec_tptr"].astype(int)
    hashes = [ec_hash(got["ec_targets"][tp[i]:tp[i + 1]].tolist()) for i in range(len(tp) - 1)]
    assert hashes == sorted(hashes) and len(set(hashes)) == len(hashes)
    # Alignment.read_files() is the same thing behind the mirror's interface
    a = Alignment(n_targets)
    a.read_files(mode, paths)
    a.collapse()
    assert a.n_reads() == ref.n_reads() and a.n_ecs() == ref.n_ecs()
    np.testing.assert_array_equal(a.ec_targets, ref.ec_targets)


def test_native_reader_errors(tmp_path):
    good = tmp_path / "good.txt"
    good.write_text("0 1 2\n1 3\n")
    bad = tmp_path / "bad.txt"
    bad.write_text("0 1 2\n1 x3\n")
    with pytest.raises(MswError, match="File format not supported on line 2 with content: 1 x3"):
        read_alignment([str(bad)], 10)
    with pytest.raises(MswError, match="more target sequences than expected"):
        read_alignment([str(good)], 3)
    with pytest.raises(MswError, match="themisto-mode"):
        read_alignment([str(good)], 10, "xor")
    with pytest.raises(MswError, match="cannot open"):
        read_alignment([str(tmp_path / "missing.txt")], 10)
    compact = tmp_path / "compact.txt"
    compact.write_text("10,4\n")
    with pytest.raises(MswError, match="compact"):
        read_alignment([str(compact)], 10)
    # only unaligned reads: zero ECs, n_reads = number of lines
    empty = tmp_path / "unaligned.txt"
    empty.write_text("0\n1\n2\n")
    r = read_alignment([str(empty)], 10)
    assert r["n_reads"] == 3 and len(r["ec_counts"]) == 0 and r["ec_tptr"].tolist() == [0]


def test_native_reader_huge_read_id_does_not_size_the_tables(tmp_path):
    """A malformed line with read id 4e9 must not allocate (max id + 1) table entries: ids at or beyond the
    line count never take part in the equivalence classes (include/mSWEEP_alignment.hpp:148) and are dropped."""
    import resource
    f = tmp_path / "huge.txt"
    f.write_text("0 1 2\n1 3\n4000000000 5\n2 1 2\n")
    before = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    r = read_alignment([str(f)], 10)
    after = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    assert after - before < 1_000_000                     # KiB: nowhere near 3 x 32 GB
    g = tmp_path / "ok.txt"
    g.write_text("0 1 2\n1 3\n3\n2 1 2\n")
    ok = read_alignment([str(g)], 10)
    for k in ("ec_tptr", "ec_targets", "ec_counts"):
        np.testing.assert_array_equal(r[k], ok[k])
    assert r["n_reads"] == ok["n_reads"] == 4


def test_native_reader_multithreaded_chunks(tmp_path):
    """Files above 1 MB are parsed in per-thread chunks cut at line boundaries: same result as the
    mirror, and the line number of a late format error is that of the whole file."""
    rng = np.random.default_rng(99)
    n_targets, n_reads = 5000, 150_000
    p = tmp_path / "big.txt"
    n_lines = _write_strand(p, rng, n_reads, n_targets, dup_lines=True)
    assert p.stat().st_size > 2 * (1 << 20)   # at least three chunks
    ref = _python_alignment([str(p)], n_targets, "intersection")
    got = read_alignment([str(p)], n_targets)
    assert got["n_reads"] == ref.n_queries == n_lines
    np.testing.assert_array_equal(got["ec_counts"], ref.ec_counts)
    np.testing.assert_array_equal(got["ec_tptr"], ref.ec_tptr)
    np.testing.assert_array_equal(got["ec_targets"], ref.ec_targets)
    lines = p.read_text().splitlines()
    bad_line = len(lines) - 7
    lines[bad_line - 1] = "12 34 x"
    p.write_text("\n".join(lines) + "\n")
    with pytest.raises(MswError, match=f"File format not supported on line {bad_line} with content: 12 34 x"):
        read_alignment([str(p)], n_targets)


def test_gzip_input_equals_plaintext(tmp_path):
    """Themisto's --gzip-output / the `.aln.gz` inputs of docs/gpubenchmarks.md:3: the reference opens every
    input through bxzstr (magic-byte detection); the native reader inflates gzip with zlib and gives the same
    equivalence classes as on the plain text; bzip2 / xz input is named as unsupported."""
    import bz2
    import gzip
    rng = np.random.default_rng(91)
    n_targets, n_reads = 53, 5000
    plain = tmp_path / "s.aln"
    _write_strand(plain, rng, n_reads, n_targets)
    gz = tmp_path / "s.aln.gz"
    with gzip.open(gz, "wb", compresslevel=6) as f:
        f.write(plain.read_bytes())
    a, b = read_alignment([str(plain)], n_targets), read_alignment([str(gz)], n_targets)
    assert a["n_reads"] == b["n_reads"]
    for k in ("ec_tptr", "ec_targets", "ec_counts", "ec_rptr", "ec_reads"):
        np.testing.assert_array_equal(a[k], b[k])
    # multi-member gzip (concatenated streams, what `cat a.gz b.gz` gives) reads as one text
    two = tmp_path / "two.aln.gz"
    raw = plain.read_bytes()
    cut = raw.index(b"\n", len(raw) // 2) + 1
    two.write_bytes(gzip.compress(raw[:cut]) + gzip.compress(raw[cut:]))
    c = read_alignment([str(two)], n_targets)
    np.testing.assert_array_equal(a["ec_targets"], c["ec_targets"])
    np.testing.assert_array_equal(a["ec_counts"], c["ec_counts"])
    bz = tmp_path / "s.aln.bz2"
    bz.write_bytes(bz2.compress(raw))
    with pytest.raises(MswError, match="bzip2-compressed"):
        read_alignment([str(bz)], n_targets)
    # a truncated gzip file is an error, not a short alignment
    bad = tmp_path / "bad.aln.gz"
    bad.write_bytes(gz.read_bytes()[:-20])
    with pytest.raises(MswError, match="gzip"):
        read_alignment([str(bad)], n_targets)


def test_views_outlive_the_call_and_equal_the_copy(tmp_path):
    """read_alignment returns read-only VIEWS of the native handle's storage (msw_alignment_view: no copy of the EC ->
    target lists); the handle lives as long as any of the arrays; copy=True (msw_alignment_export, on the reader's
    threads) gives ordinary writable arrays with the same contents."""
    import gc
    rng = np.random.default_rng(8)
    p = tmp_path / "s.txt"
    _write_strand(p, rng, 5000, 300)
    v = read_alignment([str(p)], 300)
    c = read_alignment([str(p)], 300, copy=True)
    for k in ("ec_tptr", "ec_targets", "ec_counts", "ec_rptr", "ec_reads"):
        np.testing.assert_array_equal(v[k], c[k])
        assert not v[k].flags.writeable and c[k].flags.writeable
    assert v["n_reads"] == c["n_reads"]
    targets = v["ec_targets"]
    want = c["ec_targets"].copy()
    del v
    gc.collect()
    junk = [np.full(100_000, 7, np.uint32) for _ in range(20)]   # (memory churn: a freed handle would be overwritten)
    np.testing.assert_array_equal(targets, want)
    del junk
    with pytest.raises(ValueError):
        targets[0] = 1
