"""CPU: host plumbing around the path -- Themisto plaintext reader, paired-end merge, EC collapse
(include/mSWEEP_alignment.hpp:54-215), group indicators (include/Grouping.hpp, src/Reference.cpp)
and the abundances.txt byte format (src/PlainSample.cpp:32-71, src/BootstrapSample.cpp:75-130)."""
import io

import numpy as np

from msweep_amd.alignment import Alignment, ec_hash
from msweep_amd.reference import read_reference
from msweep_amd.sample import VERSION, BootstrapSample, PlainSample


def test_reference_grouping_order_and_sizes():
    g = read_reference(io.StringIO("b\na\nb\nc\na\nb\n"))
    assert g.get_names() == ["b", "a", "c"]                  # ids in order of first appearance
    assert g.get_sizes().tolist() == [3, 2, 1]
    assert g.group_indicators.tolist() == [0, 1, 0, 2, 1, 0]
    assert g.max_group_size() == 3 and g.get_n_groups() == 3


def test_hash_matches_reference_formula():
    # hand evaluation of hash ^= j + 0x517cc1b727220a95 + (hash << 6) + (hash >> 2) in 64 bits
    h = 0
    for j in (2, 5):
        h ^= (j + 0x517cc1b727220a95 + ((h << 6) % 2**64) + (h >> 2)) % 2**64
    assert ec_hash((2, 5)) == h
    assert ec_hash(()) == 0 and ec_hash((0,)) == 0x517cc1b727220a95


def test_plaintext_read_merge_collapse():
    s1 = "0 1 2\n1 0\n2\n3 1 2\n4 5\n"
    s2 = "0 2 1 3\n1 0 4\n2 3\n3 2 1\n4 0\n"
    a = Alignment(6)
    a.read("intersection", [io.StringIO(s1), io.StringIO(s2)])
    a.collapse()
    # reads 0 and 3 -> {1,2}; read 1 -> {0}; reads 2 and 4 unaligned after the intersection
    assert a.n_reads() == 5 and a.n_ecs() == 2
    pats = {tuple(a.ec_targets[a.ec_tptr[i]:a.ec_tptr[i + 1]].tolist()): a.reads_in_ec(i) for i in range(2)}
    assert pats == {(1, 2): 2, (0,): 1}
    hs = [ec_hash(tuple(a.ec_targets[a.ec_tptr[i]:a.ec_tptr[i + 1]].tolist())) for i in range(2)]
    assert hs == sorted(hs)                                   # EC order = ascending hash
    u = Alignment(6)
    u.read("union", [io.StringIO(s1), io.StringIO(s2)])
    u.collapse()
    assert int(u.ec_counts.sum()) == 5                        # every read aligns somewhere in the union
    single = Alignment(6)
    single.read("intersection", [io.StringIO(s1)])
    single.collapse()
    assert int(single.ec_counts.sum()) == 4 and single.n_reads() == 5


def test_abundances_format_plain_and_minhits():
    s = PlainSample(1000, 987)
    s.store_abundances([0.5, 0.25, 1.23456789e-7])
    out = io.StringIO()
    s.write_abundances(["g1", "g2", "g3"], out)
    assert out.getvalue() == (f"#mSWEEP_version:\t{VERSION}\n#num_reads:\t1000\n#num_aligned:\t987\n"
                              "#c_id\tmean_theta\ng1\t0.5\ng2\t0.25\ng3\t1.23457e-07\n")
    out = io.StringIO()
    s.store_abundances([0.75, 0.25])
    s.write_abundances2(["a", "c"], ["b"], out)
    assert out.getvalue().splitlines()[-3:] == ["a\t0.75", "c\t0.25", "b\t0"]


def test_abundances_format_bootstrap():
    s = BootstrapSample(10, 9, 2)
    for th in ([0.6, 0.4], [0.5, 0.5], [0.7, 0.3]):
        s.store_abundances(th)
    out = io.StringIO()
    s.write_abundances(["x", "y"], out)
    lines = out.getvalue().splitlines()
    assert lines[3] == "#bootstrap_iters:\t2"
    assert lines[4] == "#c_id\tmean_theta\tbootstrap_mean_thetas"
    assert lines[5:] == ["x\t0.6\t0.5\t0.7", "y\t0.4\t0.5\t0.3"]
    out = io.StringIO()
    s.write_abundances2(["x", "y"], ["z"], out)
    assert out.getvalue().splitlines()[-1] == "z\t0\t0\t0"
