#!/usr/bin/env python3
"""Developer tool: registers / scratch / occupancy of the sweep kernels from
`hipcc -Rpass-analysis=kernel-resource-usage` output (stderr of a compile saved to a file).
usage: hipcc ... -Rpass-analysis=kernel-resource-usage -c msweep_core.hip 2> res.txt; kernel_resources.py res.txt [filter]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else "k_pass"
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].strip(" ]")
    if flt not in name:
        continue

    def f(k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else "?"
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r"^void msw::", "", dn).split("(")[0]
    scr, occ, lds = f(r"ScratchSize \[bytes/lane\]"), f(r"Occupancy \[waves/SIMD\]"), f(r"LDS Size \[bytes/block\]")
    print(f"{dn:45s} VGPR {f('VGPRs'):>4s} AGPR {f('AGPRs'):>4s} SGPR {f('SGPRs'):>4s} scratch {scr:>4s} occupancy {occ} LDS {lds}")
