#!/bin/bash
# Registers / scratch / occupancy of a few sweep instantiations compiled ALONE (seconds instead of the library's four minutes).
# usage: mini_tu.sh "<extra hipcc flags, e.g. -DMSW_PASSB_BATCH=2>"   (edit the instantiation list below as needed)
set -e
d=$(mktemp -d)
root=$(cd "$(dirname "$0")/.." && pwd)
cat > $d/mini.hip <<'EOT'
#include "kernels.hpp"
#include "em_kernels.hpp"
#include "em_f32_kernels.hpp"
using namespace msw;
template __global__ void msw::k_passB<kEncNarrow, 2, true, false, 16>(const Scalars *, SellDev, const double *, const double2 *, double *, double *, double *, RangeB, GuardDev);
template __global__ void msw::k_passB<kEncNarrow, 2, true, false, 8>(const Scalars *, SellDev, const double *, const double2 *, double *, double *, double *, RangeB, GuardDev);
template __global__ void msw::k_passA<kEncNarrow, true, true, false>(const Scalars *, SellDev, const double2 *, const double2 *, double *, const double *, int, GuardDev);
template __global__ void msw::k_em_passB_f32<false>(const Scalars *, SellDev, const double *, const float *, const float *, double *, double *, GuardDev);
template __global__ void msw::k_dense_passA<8>(const Scalars *, const double *, int, uint32_t, const double *, const double *, double *);
template __global__ void msw::k_dense_passB<8>(const Scalars *, const double *, int, uint32_t, const double *, const double *, double *, double *);
EOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c $1 -I$root/msweep_amd/csrc -Rpass-analysis=kernel-resource-usage $d/mini.hip -o $d/mini.o 2>&1 | python3 -c "
import re,sys
cur=None
for ln in sys.stdin:
    if 'error' in ln: print(ln.rstrip())
    m=re.search(r'Function Name: (\S+)',ln)
    if m: cur=m.group(1); continue
    if cur and re.search(r'k_(pass|em_pass|dense_pass)',cur):
        m=re.search(r'(VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)',ln)
        if m: print(cur[8:56], m.group(1), m.group(2))
"
rm -rf $d
