"""CPU (hipcc cross-compiles without a GPU): registers and scratch of the sweeps' main instantiations.

The sweeps live at their register limits -- pass B's 16-row kernel at 3 wavefronts per SIMD (<= 168 registers), its 8-row
kernel and pass A at 4 (<= 128), the value-record kernels at 2 (<= 256) -- and a spill shows in no parity test: it only
makes a sweep slower (round 5: priming the first slice cost the value-record sweeps 312 bytes of scratch per lane and 45 %
of their time, found in the last profile run).  This test compiles the instantiations the BASELINE configurations use
(cfg3 / cfg5: offset records; diverse group sizes: index records; cfg2: value records; the fp32 EM sweep; the dense
sweeps) in a translation unit of their own and holds every one to zero scratch and to its occupancy class."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TU = r'''
#include "kernels.hpp"
#include "em_kernels.hpp"
#include "em_f32_kernels.hpp"
using namespace msw;
#define B(ENC, GM, TL, ML, RC) template __global__ void msw::k_passB<ENC, GM, TL, ML, RC>(const Scalars *, SellDev, const double *, const double2 *, double *, double *, double *, RangeB, GuardDev);
#define A(ENC, GL, TL, ML) template __global__ void msw::k_passA<ENC, GL, TL, ML>(const Scalars *, SellDev, const double2 *, const double2 *, double *, const double *, int, GuardDev);
B(kEncNarrow, 2, true, false, 16) B(kEncNarrow, 2, true, true, 16) B(kEncNarrow, 2, true, false, 8) B(kEncNarrow, 1, true, false, 16)
B(kEncIndex, 1, false, false, 16) B(kEncValue, 2, false, true, 16)
A(kEncNarrow, true, true, false) A(kEncNarrow, true, true, true) A(kEncIndex, true, false, false) A(kEncValue, true, false, true)
template __global__ void msw::k_em_passB_f32<false>(const Scalars *, SellDev, const double *, const float *, const float *, double *, double *, GuardDev);
template __global__ void msw::k_dense_passA<8>(const Scalars *, const double *, int, uint32_t, const double *, const double *, double *);
template __global__ void msw::k_dense_passB<8>(const Scalars *, const double *, int, uint32_t, const double *, const double *, double *, double *);
'''
# (mangled-name fragment, register ceiling)
LIMITS = [
    ("k_passBILi0ELi2ELb1ELb0ELi16E", 168), ("k_passBILi0ELi2ELb1ELb1ELi16E", 168), ("k_passBILi0ELi1ELb1ELb0ELi16E", 168),
    ("k_passBILi0ELi2ELb1ELb0ELi8E", 128), ("k_passBILi2ELi1ELb0ELb0ELi16E", 168), ("k_passBILi3ELi2ELb0ELb1ELi16E", 256),
    ("k_passAILi0ELb1ELb1ELb0E", 128), ("k_passAILi0ELb1ELb1ELb1E", 128), ("k_passAILi2ELb1ELb0ELb0E", 168),
    ("k_passAILi3ELb1ELb0ELb1E", 256), ("k_em_passB_f32ILb0E", 128), ("k_dense_passAILi8E", 128), ("k_dense_passBILi8E", 128),
]


def test_sweeps_fit_their_registers_without_scratch(tmp_path):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    src = tmp_path / "mini.hip"
    src.write_text(TU)
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-I", os.path.join(ROOT, "msweep_amd", "csrc"),
                        "-Rpass-analysis=kernel-resource-usage", str(src), "-o", str(tmp_path / "mini.o")],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    res, cur = {}, None
    for ln in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            cur = m.group(1)
            continue
        m = re.search(r"(VGPRs|ScratchSize \[bytes/lane\]): (\d+)", ln)
        if m and cur:
            res.setdefault(cur, {})[m.group(1)[0]] = int(m.group(2))
    for frag, vmax in LIMITS:
        hit = [(k, v) for k, v in res.items() if frag in k]
        assert len(hit) == 1, (frag, [k for k in res if "k_pass" in k or "k_em" in k or "k_dense" in k])
        name, v = hit[0]
        assert v["S"] == 0, f"{frag}: {v['S']} bytes of scratch per lane"
        assert v["V"] <= vmax, f"{frag}: {v['V']} registers (ceiling {vmax})"
