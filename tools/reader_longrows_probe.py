"""Developer probe: the device reader on reads that hit HUNDREDS of targets each (real reference collections: a read
aligns to most genomes of its species) -- rows far beyond the 64 targets a wavefront sorts and the staging area of the
paired-end merge.  Device against host reader: arrays equal, seconds of both.
usage: python tools/reader_longrows_probe.py [reads] [targets_per_read] [n_targets]"""
import os, sys, time, tempfile, shutil
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msweep_amd.core import Core, read_alignment

R = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 300
NT = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
tmp = tempfile.mkdtemp(prefix="msweep_long_", dir=os.environ.get("TMPDIR", "/tmp"))
try:
    rng = np.random.default_rng(1)
    names = [str(i) for i in range(NT)]
    files = []
    start = rng.integers(0, NT - 2 * L, R)
    length = rng.integers(L // 2, L * 3 // 2, R)
    for s in range(2):
        p = os.path.join(tmp, f"r{s}.txt")
        with open(p, "w") as f:
            buf = []
            for r in range(R):
                a = int(start[r]) + (7 * s if r % 3 == 0 else 0)
                buf.append(f"{r} " + " ".join(names[a:a + int(length[r])]))
                if len(buf) == 20000:
                    f.write("\n".join(buf) + "\n")
                    buf = []
            if buf:
                f.write("\n".join(buf) + "\n")
        files.append(p)
    print("text bytes per strand:", [os.path.getsize(x) for x in files], flush=True)
    core = Core(0)
    for mode in ("intersection", "union"):
        t0 = time.perf_counter()
        host = read_alignment(files, NT, mode)
        t1 = time.perf_counter()
        for rep in range(2):
            t2 = time.perf_counter()
            dev = core.read_alignment(files, NT, mode)
            t3 = time.perf_counter()
        assert dev.on_device
        d = dev.arrays()
        for k in ("ec_tptr", "ec_targets", "ec_counts", "ec_rptr", "ec_reads"):
            assert np.array_equal(d[k], host[k]), k
        from msweep_amd.likelihood import from_device_alignment
        G = 100
        tg = (np.arange(NT) * G // NT).astype(np.uint32)   # contiguous targets share a group: ~10 groups per read
        sizes = np.bincount(tg, minlength=G).astype(np.uint64)
        for rep in range(2):
            t4 = time.perf_counter()
            lik = from_device_alignment(core, dev, tg, sizes)
            t5 = time.perf_counter()
            res = core.solve(None, np.ones(lik.n_groups))
            t6 = time.perf_counter()
        print(f"{mode}: build {t5 - t4:.3f} s, solve {t6 - t5:.3f} s ({res['iters']} iterations, nnz {core.shape()[2]})", flush=True)
        print(f"{mode}: host reader {t1 - t0:.3f} s, device reader {t3 - t2:.3f} s; {dev.n_ecs} classes, {dev.n_hits} hits: equal", flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
