"""Wall time of the drivers on cfg3-sized text: msweep_mini (C++) and `python -m msweep_amd`, whole processes -- HIP
start-up, handle, reader on the device, build, solve, abundances.txt -- on two Themisto strands of `reads` reads.
usage: python tools/cli_wall.py [reads] [groups]   (MSWEEP_PROBE_DIR keeps the generated strands, as tools/reader_probe.py)"""
import os, subprocess, sys, tempfile, time, shutil
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from msweep_amd import synth

R = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
keep = os.environ.get("MSWEEP_PROBE_DIR")
tmp = keep or tempfile.mkdtemp(prefix="msweep_cli_", dir=os.environ.get("TMPDIR", "/tmp"))
os.makedirs(tmp, exist_ok=True)
try:
    f = [os.path.join(tmp, "r1.txt"), os.path.join(tmp, "r2.txt")]
    clus = os.path.join(tmp, "clustering.txt")
    if not (os.path.exists(clus) and all(os.path.exists(x) for x in f)):
        prob = synth.make_csr_problem(R, G, seed=2)
        aln = synth.csr_to_targets(prob, shuffle=False)
        E = len(prob["ec_counts"])
        rng = np.random.default_rng(11)
        ec_of = rng.permutation(np.repeat(np.arange(E, dtype=np.int64), prob["ec_counts"].astype(np.int64)))
        for k, path in enumerate(f):
            synth.write_themisto(path, ec_of, aln["ec_tptr"], aln["ec_targets"], chunk=1_000_000,
                                 extra=(rng, 0.1, aln["n_targets"]) if k else None)
        with open(clus, "w") as c:
            c.write("\n".join(f"g{int(g)}" for g in aln["target_group"]) + "\n")
        del prob, aln, ec_of
    lib = os.path.join(ROOT, "msweep_amd")
    mini = os.path.join(tmp, "msweep_mini")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-o", mini, os.path.join(lib, "cpp", "msweep_mini.cpp"), "-L" + lib,
                           "-lmsweep_core", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"])
    args = ["--themisto-1", f[0], "--themisto-2", f[1], "-i", clus]
    outs = {}
    for name, cmd in (("msweep_mini", [mini]), ("python -m msweep_amd", [sys.executable, "-m", "msweep_amd"])):
        for rep in range(3):
            o = os.path.join(tmp, f"out_{name.split()[0]}")
            t = time.perf_counter()
            p = subprocess.run(cmd + args + ["-o", o], capture_output=True, text=True, cwd=ROOT)
            dt = time.perf_counter() - t
            assert p.returncode == 0, p.stderr[-1000:]
            print(f"{name}: pass {rep}: {dt:.3f} s wall (whole process)", flush=True)
        outs[name] = open(o + "_abundances.txt").read()
    a, b = (outs[k].splitlines() for k in outs)
    # (the header lines name the program's own version strings; the abundances must agree)
    same = [x for x in a if not x.startswith("#")] == [x for x in b if not x.startswith("#")]
    print("abundance rows of the two drivers equal:", same)
finally:
    if not keep:
        shutil.rmtree(tmp, ignore_errors=True)
