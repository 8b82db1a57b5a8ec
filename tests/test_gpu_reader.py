"""The Themisto reader on the device (msw_alignment_read_device, msweep_amd/csrc/host_reader.inc + reader_kernels.hpp)
against the host reader (msw_alignment_read, itself held against the Python mirror of mSWEEP::Alignment in
test_alignment_native.py): the five arrays equal, element for element, on single- and paired-end inputs, lines in any
order, targets in any order and repeated, read ids repeated and out of range, CRLF, trailing blanks, no final line
feed, empty files, gzip; text the kernels do not judge goes to the host parser and carries its messages; the likelihood
built from the device-resident arrays is the one built from the host arrays."""
import gzip

import numpy as np
import pytest

from msweep_amd.core import Core, MswError, read_alignment

pytestmark = pytest.mark.gpu

KEYS = ("ec_tptr", "ec_targets", "ec_counts", "ec_rptr", "ec_reads")


def _equal(dev, host):
    assert dev["n_reads"] == host["n_reads"]
    for k in KEYS:
        np.testing.assert_array_equal(dev[k], host[k], err_msg=k)


def _lines(rng, n_reads, n_targets, p_unaligned=0.2, max_k=7, sort_targets=False, dup_targets=False, dup_lines=True,
           shuffle=True, big_row_every=0):
    lines = []
    for r in range(n_reads):
        if rng.random() < p_unaligned:
            lines.append(f"{r}")
            continue
        k = int(rng.integers(1, max_k))
        if big_row_every and r % big_row_every == 0:
            k = min(n_targets, 900)
        t = rng.choice(n_targets, k, replace=False)
        if sort_targets:
            t = np.sort(t)
        t = [int(x) for x in t]
        if dup_targets and rng.random() < 0.3:
            t = t + [t[0]] + t[-1:]
        lines.append(f"{r} " + " ".join(map(str, t)))
    if dup_lines:
        for r in rng.choice(n_reads, max(1, n_reads // 20), replace=False):
            lines.append(f"{int(r)} {int(rng.integers(0, n_targets))}")
    if shuffle:
        rng.shuffle(lines)
    return lines


@pytest.fixture(scope="module")
def core():
    with Core(0) as c:
        yield c


@pytest.mark.parametrize("mode,n_strands", [("intersection", 1), ("intersection", 2), ("union", 2), ("union", 3)])
@pytest.mark.parametrize("flavour", ["sorted", "any-order", "dup-targets"])
def test_device_reader_equals_host_reader(core, tmp_path, mode, n_strands, flavour):
    rng = np.random.default_rng([len(mode), n_strands, len(flavour)])
    n_targets, n_reads = 211, 5000
    paths = []
    for s in range(n_strands):
        p = tmp_path / f"s{s}.txt"
        lines = _lines(rng, n_reads + 17 * s, n_targets, sort_targets=flavour == "sorted", dup_targets=flavour == "dup-targets",
                       dup_lines=flavour != "sorted", shuffle=flavour != "sorted", big_row_every=1500 if flavour == "any-order" else 0)
        p.write_text("\n".join(lines) + "\n")
        paths.append(str(p))
    host = read_alignment(paths, n_targets, mode)
    dev = core.read_alignment(paths, n_targets, mode)
    assert dev.on_device   # served by the kernels, not by the host parser behind the entry
    assert (dev.n_ecs, dev.n_hits, dev.n_aligned, dev.n_reads) == (len(host["ec_counts"]), len(host["ec_targets"]),
                                                                  len(host["ec_reads"]), host["n_reads"])
    _equal(dev.arrays(), host)


def test_text_shapes_the_parser_takes(core, tmp_path):
    """CRLF, blanks at line ends, several blanks, no final line feed, a read id at and beyond the line count, tokens that
    straddle segment (16-byte) and tile (4096-byte) boundaries."""
    n_targets = 100000
    cases = {
        "crlf": "0 1 2\r\n1 3\r\n2\r\n",
        "trailing-blank": "0 1 2 \n1 3 \n",
        "two-blanks": "0 1  2\n1   3\n",
        "no-final-lf": "0 1 2\n1 3",
        "no-final-lf-cr": "0 1 2\n1 3\r",
        "id-at-line-count": "0 5\n2 6\n1 7\n7 8\n",
        "one-line": "0 99999",
        "only-ids": "0\n1\n2\n",
        "straddle": "".join(f"{i} " + " ".join(str(99000 + (i * 7 + j) % 999) for j in range(1 + i % 9)) + "\n" for i in range(3000)),
    }
    for name, text in cases.items():
        p = tmp_path / f"{name}.txt"
        p.write_bytes(text.encode())
        host = read_alignment([str(p)], n_targets)
        dev = core.read_alignment([str(p)], n_targets)
        assert dev.on_device, name
        _equal(dev.arrays(), host)


def test_empty_and_unaligned_files(core, tmp_path):
    e = tmp_path / "empty.txt"
    e.write_text("")
    u = tmp_path / "unaligned.txt"
    u.write_text("0\n1\n2\n")
    a = tmp_path / "a.txt"
    a.write_text("0 1\n1 2\n2 1\n")
    for paths, mode in (([e], "intersection"), ([u], "intersection"), ([a, u], "intersection"), ([a, u], "union"),
                        ([a, e], "union"), ([e, a], "union"), ([a, e], "intersection")):
        host = read_alignment([str(x) for x in paths], 5, mode)
        dev = core.read_alignment([str(x) for x in paths], 5, mode)
        assert dev.on_device
        _equal(dev.arrays(), host)


def test_text_the_kernels_do_not_judge_goes_to_the_host_parser(core, tmp_path):
    good = tmp_path / "good.txt"
    good.write_text("0 1 2\n1 3\n")
    bad = tmp_path / "bad.txt"
    bad.write_text("0 1 2\n1 x3\n")
    with pytest.raises(MswError, match="File format not supported on line 2 with content: 1 x3"):
        core.read_alignment([str(bad)], 10)
    with pytest.raises(MswError, match="more target sequences than expected"):
        core.read_alignment([str(good)], 3)
    empty_line = tmp_path / "empty_line.txt"
    empty_line.write_text("0 1\n\n1 2\n")
    with pytest.raises(MswError, match="File format not supported on line 2"):
        core.read_alignment([str(empty_line)], 10)
    compact = tmp_path / "compact.txt"
    compact.write_text("3,5,query,target\n0 1\n")
    with pytest.raises(MswError, match="compact"):
        core.read_alignment([str(compact)], 10)
    with pytest.raises(MswError, match="cannot open"):
        core.read_alignment([str(tmp_path / "missing.txt")], 10)
    # accepted by the host parser although the kernels hand it over: leading zeros beyond ten digits
    zeros = tmp_path / "zeros.txt"
    zeros.write_text("0 00000000003 1\n1 2\n")
    z = core.read_alignment([str(zeros)], 10)
    assert not z.on_device
    _equal(z.arrays(), read_alignment([str(zeros)], 10))
    with pytest.raises(MswError, match="themisto-mode"):
        core.read_alignment([str(good)], 10, "both")


def test_gzip_input(core, tmp_path):
    rng = np.random.default_rng(5)
    lines = _lines(rng, 3000, 97)
    plain = tmp_path / "p.txt"
    plain.write_text("\n".join(lines) + "\n")
    gz = tmp_path / "p.txt.gz"
    with gzip.open(gz, "wt") as f:
        f.write("\n".join(lines) + "\n")
    dev = core.read_alignment([str(gz)], 97)   # (inflated on the host, parsed on the device)
    assert dev.on_device
    _equal(dev.arrays(), read_alignment([str(plain)], 97))


def test_more_than_one_staging_chunk_and_likelihood_from_device_arrays(core, tmp_path):
    """~150 MB of text per strand (three 64 MB staging chunks), then msw_core_build_likelihood_aln on the resident
    arrays against msw_core_build_likelihood on the host reader's: the same layout, bit for bit."""
    rng = np.random.default_rng(11)
    n_targets, n_groups, n_reads = 3000, 60, 1_500_000
    k = rng.integers(6, 22, n_reads)
    base = rng.integers(0, n_targets - 50, n_reads)
    paths = []
    for s in range(2):
        p = tmp_path / f"big{s}.txt"
        with open(p, "w") as f:
            step = 100000
            for r0 in range(0, n_reads, step):
                rows = []
                for r in range(r0, min(n_reads, r0 + step)):
                    t = base[r] + np.arange(k[r]) * 2 + s * (r % 3 == 0)
                    rows.append(f"{r} " + " ".join(map(str, t.tolist())))
                f.write("\n".join(rows) + "\n")
        paths.append(str(p))
    host = read_alignment(paths, n_targets, "intersection")
    aln = core.read_alignment(paths, n_targets, "intersection")
    assert aln.on_device
    _equal(aln.arrays(), host)
    target_group = (np.arange(n_targets) % n_groups).astype(np.uint32)
    group_sizes = np.bincount(target_group, minlength=n_groups).astype(np.uint64)
    g1, m1, logc1 = core.build_likelihood_aln(aln, target_group, group_sizes, min_hits=1)
    h1 = core.layout_hash()
    shape1 = core.shape()
    g2, m2, logc2 = core.build_likelihood(host["ec_tptr"], host["ec_targets"], target_group, group_sizes, host["ec_counts"], min_hits=1)
    assert (g1, shape1, h1) == (g2, core.shape(), core.layout_hash())
    np.testing.assert_array_equal(m1, m2)
    np.testing.assert_array_equal(logc1, logc2)


def test_repeated_reads_reuse_the_handles_memory_and_trim_gives_it_back(core, tmp_path):
    """The reader keeps its device temporaries on the handle (ReaderPool): a second and third read -- of the same files, of
    smaller and of larger ones -- and a read after msw_core_trim all give the host reader's arrays."""
    rng = np.random.default_rng(3)
    files = {}
    for name, n in (("mid", 4000), ("small", 300), ("large", 20000)):
        p = tmp_path / f"{name}.txt"
        p.write_text("\n".join(_lines(rng, n, 150)) + "\n")
        files[name] = str(p)
    host = {k: read_alignment([v], 150) for k, v in files.items()}
    for name in ("mid", "mid", "small", "large", "mid"):
        _equal(core.read_alignment([files[name]], 150).arrays(), host[name])
    core.trim()
    _equal(core.read_alignment([files["large"]], 150).arrays(), host["large"])
    both = read_alignment([files["mid"], files["large"]], 150, "union")
    _equal(core.read_alignment([files["mid"], files["large"]], 150, "union").arrays(), both)


def test_bench_e2e_leg_small():
    """`bench.py --config e2e` at a small size: both readers run, the device reader's abundances.txt equals the host
    reader's (asserted inside the leg), the line carries both sets of stages."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "e2e", "--reads", "200000", "--groups", "300"],
                         capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["unit"] == "reads/s" and line["value"] > 0
    for key in ("stages_s", "stages_first_pass_s"):
        assert line[key]["ecs"] == line["host_reader"]["stages_s"]["ecs"] > 0
        assert line[key]["iters"] == line["host_reader"]["stages_s"]["iters"]


def test_host_resident_alignment_behind_the_device_entry(core, tmp_path, monkeypatch):
    """A text that would not fit the device goes to the host reader behind msw_alignment_read_device (forced here by the
    developer switch); msw_core_build_likelihood_aln then takes the handle's HOST arrays: same likelihood."""
    rng = np.random.default_rng(21)
    n_targets, n_groups = 400, 20
    p = tmp_path / "a.txt"
    p.write_text("\n".join(_lines(rng, 6000, n_targets)) + "\n")
    target_group = (np.arange(n_targets) % n_groups).astype(np.uint32)
    group_sizes = np.bincount(target_group, minlength=n_groups).astype(np.uint64)
    dev = core.read_alignment([str(p)], n_targets)
    g1, m1, l1 = core.build_likelihood_aln(dev, target_group, group_sizes)
    h1 = core.layout_hash()
    monkeypatch.setenv("MSWEEP_READER_FORCE_HOST", "1")
    host = core.read_alignment([str(p)], n_targets)
    monkeypatch.delenv("MSWEEP_READER_FORCE_HOST")
    assert dev.on_device and not host.on_device
    _equal(host.arrays(), dev.arrays())
    g2, m2, l2 = core.build_likelihood_aln(host, target_group, group_sizes)
    assert (g1, h1) == (g2, core.layout_hash())
    np.testing.assert_array_equal(l1, l2)
    np.testing.assert_array_equal(host.ec_counts(), dev.ec_counts())


def test_default_bench_line_carries_text_to_abundances():
    """The default (cfg3) line at a small size: `text_to_abundances` = the reads as Themisto text through the device
    reader, the host reader beside it, the same abundances.txt."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--reads", "200000", "--groups", "300", "--steps", "5",
                          "--warmup", "2", "--no-cpu-baseline", "--bootstrap-per-rank", "0"],
                         capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    t = line["text_to_abundances"]
    assert t["same_abundances_txt"] is True and t["seconds"] > 0
    assert t["stages_s"]["ecs"] == t["host_reader"]["ecs"] > 0 and t["stages_s"]["iters"] == t["host_reader"]["iters"]


def test_fuzz_mutated_text(core, tmp_path):
    """Random byte-level damage to small files: the device entry and the host reader end the same way -- the same five
    arrays, or the same error message (the kernels hand anything they do not judge to the host parser; what they DO take
    they must parse as the host does)."""
    import os
    # (MSWEEP_READER_FUZZ_SEED / _CASES: longer campaigns with other seeds, tools/README.md)
    rng = np.random.default_rng(int(os.environ.get("MSWEEP_READER_FUZZ_SEED", "2024")))
    n_cases = int(os.environ.get("MSWEEP_READER_FUZZ_CASES", "3000"))
    palette = [b" ", b"  ", b"\n", b"\r\n", b"\r", b"\t", b"x", b"-", b"+", b",", b"0", b"7", b"99", b"4294967295", b"4294967296",
               b"00000000000", b"\n\n", b" \n", b"\x00", b"12345678901"]
    n_targets = 64
    n_ok = n_err = 0
    served = {True: 0, False: 0}   # texts the kernels took / handed to the host parser (and it accepted)
    p = tmp_path / "f.txt"
    for case in range(n_cases):
        # (one case in ten spans several 4 KB tiles of the text kernels)
        lines = _lines(rng, int(rng.integers(200, 700)) if case % 10 == 0 else int(rng.integers(1, 40)), n_targets, dup_lines=bool(rng.integers(0, 2)), shuffle=bool(rng.integers(0, 2)))
        text = bytearray(("\n".join(lines) + ("\n" if rng.random() < 0.8 else "")).encode())
        for _ in range(int(rng.integers(0, 4))):
            at = int(rng.integers(0, len(text) + 1))
            what = int(rng.integers(0, 3))
            if what == 0:
                text[at:at] = palette[int(rng.integers(0, len(palette)))]
            elif what == 1 and len(text):
                del text[min(at, len(text) - 1)]
            elif len(text):
                text[min(at, len(text) - 1)] = palette[int(rng.integers(0, len(palette)))][0]
        p.write_bytes(bytes(text))
        results = []
        def on_device_entry():
            al = core.read_alignment([str(p)], n_targets)
            served[al.on_device] += 1
            return al.arrays()
        for fn in (lambda: read_alignment([str(p)], n_targets), on_device_entry):
            try:
                results.append(("ok", fn()))
            except MswError as ex:
                results.append(("err", str(ex)))
        assert results[0][0] == results[1][0], (case, bytes(text), results[0], results[1])
        if results[0][0] == "ok":
            _equal(results[1][1], results[0][1])
            n_ok += 1
        else:
            assert results[0][1] == results[1][1], (case, bytes(text))
            n_err += 1
    assert n_ok > 300 and n_err > 300 and served[True] > 300, (n_ok, n_err, served)


@pytest.mark.parametrize("mode,n_strands", [("intersection", 1), ("intersection", 2), ("union", 2)])
def test_device_reader_against_the_python_mirror_of_the_reference(core, tmp_path, mode, n_strands):
    """The device reader held directly against msweep_amd/alignment.py -- the line-by-line mirror of
    include/mSWEEP_alignment.hpp:54-215 -- not only against the host reader: classes in ascending order of the
    reference's hash, their reads, counts and target sets."""
    from msweep_amd.alignment import Alignment, ec_hash
    rng = np.random.default_rng(70 + n_strands + len(mode))
    n_targets, n_reads = 37, 700
    paths = []
    for s in range(n_strands):
        p = tmp_path / f"strand_{s}.txt"
        p.write_text("\n".join(_lines(rng, n_reads, n_targets, dup_lines=(s == 0))) + "\n")
        paths.append(str(p))
    ref = Alignment(n_targets)
    streams = [open(p) for p in paths]
    ref.read(mode, streams)
    for s in streams:
        s.close()
    ref._reads = {r: t for r, t in ref._reads.items() if r < ref.n_queries}   # ids below the last strand's line count (:148)
    ref.collapse()
    dev = core.read_alignment(paths, n_targets, mode)
    assert dev.on_device
    got = dev.arrays()
    assert got["n_reads"] == ref.n_queries
    np.testing.assert_array_equal(got["ec_counts"], ref.ec_counts)
    np.testing.assert_array_equal(got["ec_tptr"], ref.ec_tptr)
    np.testing.assert_array_equal(got["ec_targets"], ref.ec_targets)
    rp = got["ec_rptr"].astype(int)
    assert [got["ec_reads"][rp[i]:rp[i + 1]].tolist() for i in range(len(rp) - 1)] == ref.ec_read_ids
    tp = got["ec_tptr"].astype(int)
    hashes = [ec_hash(got["ec_targets"][tp[i]:tp[i + 1]].tolist()) for i in range(len(tp) - 1)]
    assert hashes == sorted(hashes) and len(set(hashes)) == len(hashes)


@pytest.mark.parametrize("mode", ["intersection", "union"])
def test_merge_of_rows_beyond_the_staging_area(core, tmp_path, mode):
    """Paired-end merge: wavefronts whose 64 reads fit the LDS staging area (kMergeCap targets per strand) beside
    wavefronts whose rows do not (reads of ~2000 targets: merged in memory), in one input."""
    rng = np.random.default_rng(31 + len(mode))
    n_targets, n_reads = 3000, 1000
    paths = []
    for s in range(2):
        lines = []
        for r in range(n_reads):
            if r % 97 == 5:
                t = rng.choice(n_targets, int(rng.integers(1500, 2500)), replace=False)
            elif r % 13 == 0:
                t = rng.choice(n_targets, int(rng.integers(20, 60)), replace=False)
            else:
                t = rng.choice(n_targets, int(rng.integers(0, 9)), replace=False)
            lines.append(f"{r}" + "".join(f" {int(x)}" for x in t))
        p = tmp_path / f"s{s}.txt"
        p.write_text("\n".join(lines) + "\n")
        paths.append(str(p))
    dev = core.read_alignment(paths, n_targets, mode)
    assert dev.on_device
    _equal(dev.arrays(), read_alignment(paths, n_targets, mode))
