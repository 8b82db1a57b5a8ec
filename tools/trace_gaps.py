"""Developer tool: per-kernel durations and the gaps between consecutive kernels from a rocprofv3
--kernel-trace CSV (the last N dispatches).  usage: trace_gaps.py kernel_trace.csv [N]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
prev_end = None
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void msw::", "")[:40]
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(s - t0) / 1e3:10.1f} us  {name:40s} {(e - s) / 1e3:8.1f} us  gap {gap:6.1f} us")
    prev_end = e
