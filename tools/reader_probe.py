"""Developer probe: where the time between msw_alignment_read_device and msw_core_build_likelihood_aln goes, and -- with
a fourth argument `check` -- the device reader against the host reader at that size (inputs beyond 2^32 bytes of
text per strand: 45 M reads).
usage: python tools/reader_probe.py [reads] [groups] [sleep_s] [check]"""
import os, sys, time, tempfile, shutil
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msweep_amd import synth
from msweep_amd.core import Core
from msweep_amd.likelihood import from_device_alignment

R = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
nap = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
check = len(sys.argv) > 4 and sys.argv[4] == "check"
# MSWEEP_PROBE_DIR: keep the generated strands there (several runs of the probe -- e.g. over MSWEEP_READER_BLOCK_MB -- in
# one GPU job without generating the text again)
keep = os.environ.get("MSWEEP_PROBE_DIR")
tmp = keep or tempfile.mkdtemp(prefix="msweep_probe_", dir=os.environ.get("TMPDIR", "/tmp"))
os.makedirs(tmp, exist_ok=True)
try:
    f = [os.path.join(tmp, "r1.txt"), os.path.join(tmp, "r2.txt")]
    meta = os.path.join(tmp, f"meta_{R}_{G}.npz")
    if keep and os.path.exists(meta):
        m = np.load(meta)
        aln = {"n_targets": int(m["n_targets"]), "target_group": m["target_group"]}
        prob = {"group_sizes": m["group_sizes"]}
    else:
        prob = synth.make_csr_problem(R, G, seed=2)
        aln = synth.csr_to_targets(prob, shuffle=False)
        E = len(prob["ec_counts"])
        rng = np.random.default_rng(11)
        ec_of = rng.permutation(np.repeat(np.arange(E, dtype=np.int64), prob["ec_counts"].astype(np.int64)))
        for k, path in enumerate(f):
            synth.write_themisto(path, ec_of, aln["ec_tptr"], aln["ec_targets"], chunk=1_000_000,
                                 extra=(rng, 0.1, aln["n_targets"]) if k else None)
        if keep:
            np.savez(meta, n_targets=aln["n_targets"], target_group=aln["target_group"], group_sizes=prob["group_sizes"])
    nt = int(aln["n_targets"])
    print("text bytes per strand:", [os.path.getsize(x) for x in f], flush=True)
    core = Core(0)
    if check:
        from msweep_amd.core import read_alignment
        t0 = time.perf_counter()
        host = read_alignment(f, nt, "intersection")
        t1 = time.perf_counter()
        dev = core.read_alignment(f, nt, "intersection")
        t2 = time.perf_counter()
        d = dev.arrays()
        t3 = time.perf_counter()
        for k in ("ec_tptr", "ec_targets", "ec_counts", "ec_rptr", "ec_reads"):
            assert np.array_equal(d[k], host[k]), k
        print(f"check: host reader {t1 - t0:.2f} s, device reader {t2 - t1:.3f} s (+ {t3 - t2:.2f} s to copy the arrays out); "
              f"{len(host['ec_counts'])} classes, {len(host['ec_targets'])} hits: equal", flush=True)
        del host, dev, d
    for rep in range(4):
        t0 = time.perf_counter()
        al = core.read_alignment(f, nt, "intersection")
        t1 = time.perf_counter()
        if nap:
            time.sleep(nap)
        t2 = time.perf_counter()
        lik = from_device_alignment(core, al, aln["target_group"], prob["group_sizes"])
        t3 = time.perf_counter()
        res = core.solve(None, np.ones(lik.n_groups))
        t4 = time.perf_counter()
        del lik, al
        t5 = time.perf_counter()
        print(f"rep {rep}: read {t1 - t0:.3f} build {t3 - t2:.3f} solve {t4 - t3:.3f} del {t5 - t4:.3f}", flush=True)
finally:
    if not keep:
        shutil.rmtree(tmp, ignore_errors=True)
