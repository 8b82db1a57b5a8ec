#!/usr/bin/env python3
"""Developer A/B tool: copy csrc into build_ab/exp, (unless --nopatch) apply the MSW_EXP instrumentation
hooks to the sweeps, and build library variants build_ab/lib_<name>.so that tools/ab_bench.sh times in
one GPU job through MSWEEP_CORE_LIB.  MSW_EXP: 1 near-conflict-free LDS addresses, 2 no cell arithmetic
in pass A, 3 record stream only, 4 no column-sum atomics, 5 no log / division, 6 no log.  The hooks are
text patches against sweep_kernels.hpp and need refreshing when the patched lines change.
usage: ab_build.py [--nopatch] name "-DMSW_EXP=3 " [name flags]..."""
import os, shutil, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
X = os.path.join(R, "build_ab", "exp")
shutil.rmtree(X, ignore_errors=True)
os.makedirs(os.path.join(X, "msweep_amd")); os.makedirs(os.path.join(X, "include"))
shutil.copytree(os.path.join(R, "msweep_amd", "csrc"), os.path.join(X, "msweep_amd", "csrc"))
shutil.copy(os.path.join(R, "include", "msweep_core.h"), os.path.join(X, "include"))
p = os.path.join(X, "msweep_amd", "csrc", "sweep_kernels.hpp"); s = open(p).read()
NOPATCH = "--nopatch" in sys.argv
if NOPATCH: sys.argv.remove("--nopatch")
def rep(a, b, optional=False):
    global s
    if NOPATCH: return
    if optional and a not in s:
        print("ab_build: stale hook skipped:", a.strip()[:50]); return
    assert a in s, a[:60]
    s = s.replace(a, b)
rep('''  auto EW_ = [&](RT r) -> double2 { return tab16<GLDS>(ew_b, R::hi2(r, shift)); };
  auto XT_ = [&](RT r) -> double2 { return tab16<TLDS>(xt_b, R::lo(r, mask)); };''','''#if MSW_EXP == 1
  auto EW_ = [&](RT r) -> double2 { return tab16<GLDS>(ew_b, bhi2 + lane * 16 + (R::hi2(r, shift) & 16)); };
  auto XT_ = [&](RT r) -> double2 { return tab16<TLDS>(xt_b, lane * 16 + (R::lo(r, mask) & 16)); };
#elif MSW_EXP == 3
  auto EW_ = [&](RT r) -> double2 { return make_double2((double)R::hi2(r, shift), 1.0); };
  auto XT_ = [&](RT r) -> double2 { return make_double2((double)R::lo(r, mask), 1.0); };
#else
  auto EW_ = [&](RT r) -> double2 { return tab16<GLDS>(ew_b, R::hi2(r, shift)); };
  auto XT_ = [&](RT r) -> double2 { return tab16<TLDS>(xt_b, R::lo(r, mask)); };
#endif''')
rep('''  const double xm = x - p0;
  const double xD = x * D;
  const double wx = w * xm;
  c.zs = fma(e, xm, c.zs);
  c.t1 = fma(e, xD + wx, c.t1);
  c.t2 = fma(e, fma(xD, D, w * fma(2.0, xD, wx)), c.t2);''','''#if MSW_EXP == 2 || MSW_EXP == 3
  c.zs += e + D;
  c.t1 += x + w;
#else
  const double xm = x - p0;
  const double xD = x * D;
  const double wx = w * xm;
  c.zs = fma(e, xm, c.zs);
  c.t1 = fma(e, xD + wx, c.t1);
  c.t2 = fma(e, fma(xD, D, w * fma(2.0, xD, wx)), c.t2);
#endif''')
rep('''  auto E_ = [&](RT r) -> double { return tab8<GLDS>(e_b, R::hi(r, shift)); };
  auto XT_ = [&](RT r) -> double2 { return tab16<TLDS>(xt_b, R::lo(r, mask)); };
  auto XM_ = [&](RT r) -> double { return tab8<TLDS>(xt_b, R::lo(r, mask)); };''','''#if MSW_EXP == 1
  auto E_ = [&](RT r) -> double { return tab8<GLDS>(e_b, bhi + lane * 8 + (R::hi(r, shift) & 8)); };
  auto XT_ = [&](RT r) -> double2 { return tab16<TLDS>(xt_b, lane * 16 + (R::lo(r, mask) & 16)); };
  auto XM_ = [&](RT r) -> double { return tab8<TLDS>(xt_b, lane * 16 + (R::lo(r, mask) & 16)); };
#elif MSW_EXP == 3
  auto E_ = [&](RT r) -> double { return (double)R::hi(r, shift); };
  auto XT_ = [&](RT r) -> double2 { return make_double2((double)R::lo(r, mask), 1.0); };
  auto XM_ = [&](RT r) -> double { return (double)R::lo(r, mask); };
#else
  auto E_ = [&](RT r) -> double { return tab8<GLDS>(e_b, R::hi(r, shift)); };
  auto XT_ = [&](RT r) -> double2 { return tab16<TLDS>(xt_b, R::lo(r, mask)); };
  auto XM_ = [&](RT r) -> double { return tab8<TLDS>(xt_b, R::lo(r, mask)); };
#endif
  double exp_sink = 0.0;''')
rep('''    const uint32_t off = R::hi(r, shift);
    if constexpr (GMODE == 2)''','''#if MSW_EXP == 1
    const uint32_t off = bhi + lane * 8 + (R::hi(r, shift) & 8);
#else
    const uint32_t off = R::hi(r, shift);
#endif
#if MSW_EXP == 3
    exp_sink += v + (double)off; return;
#endif
#if MSW_EXP == 4
    exp_sink += v + (double)off; return;
#endif
    if constexpr (GMODE == 2)''')

rep("""        const double rj = c / Z;
        s_rH += rj * H;
        s_W += rj;
        // padding records""","""#if MSW_EXP == 5
        const double rj = c * Z;
#else
        const double rj = c / Z;
#endif
        s_rH += rj * H;
        s_W += rj;
        // padding records""", optional=True)
rep("""        if (sb.c8 <= 3u) {""","""#if MSW_EXP == 5 || MSW_EXP == 6
        if (true) { s_clogZ += c * Z; } else
#endif
        if (sb.c8 <= 3u) {""", optional=True)
rep("""  s_clogZ = block_sum(s_clogZ, sh);""","""  s_clogZ = block_sum(s_clogZ + exp_sink * 1e-300, sh);""")
open(p, "w").write(s)
args = sys.argv[1:]
for name, flags in zip(args[0::2], args[1::2]):
    out = os.path.join(R, "build_ab", f"lib_{name}.so")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", *flags.split(),
           "-o", out, os.path.join(X, "msweep_amd", "csrc", "msweep_core.hip"), "-L/opt/rocm/lib", "-lrccl", "-lz",
           "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    print(name, "OK" if r.returncode == 0 else "FAILED\n" + "\n".join(l for l in r.stderr.splitlines() if "error" in l)[:2000])
