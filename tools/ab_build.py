#!/usr/bin/env python3
"""Developer A/B tool: build library variants build_ab/lib_<name>.so with extra compiler flags (the build macros of
README.md: MSW_FX, MSW_PASS_THREADS_A/B, MSW_B_KEEPN, MSW_COLD_ROWS, MSW_ODD_SLICES, MSW_FAST_DIV, MSW_LONG_KEEP, the four
scheduler weights MSW_W_AT / MSW_W_EW / MSW_W_XT / MSW_W_E -- all four together ...)
that tools/ab_bench.sh -- or any script through MSWEEP_CORE_LIB -- times in one GPU job next to the default build.
(Rounds 1-2 also patched instrumentation hooks into the sweeps here -- conflict-free LDS addresses, stream only, no
atomics ...: DESIGN.md 5 quotes their results; the hooks did not survive the round-3 record refactor.)
usage: ab_build.py name "-DMACRO=value ..." [name flags]..."""
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if a != "--nopatch"]      # (accepted for old command lines)
if len(args) < 2 or len(args) % 2:
    sys.exit(__doc__)
os.makedirs(os.path.join(R, "build_ab"), exist_ok=True)
for name, flags in zip(args[0::2], args[1::2]):
    out = os.path.join(R, "build_ab", f"lib_{name}.so")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", *flags.split(),
           "-o", out, os.path.join(R, "msweep_amd", "csrc", "msweep_core.hip"), "-L/opt/rocm/lib", "-lrccl", "-lz",
           "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    print(name, "OK" if r.returncode == 0 else "FAILED\n" + "\n".join(l for l in r.stderr.splitlines() if "error" in l)[:2000])
