#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc runs of bench.py into HBM bytes per launch of the sweep kernels.

usage: pmc_traffic.py <dir with *_counter_collection.csv from separate --pmc passes> <out.json>
Unit / gfx950 corrections as prescribed by MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE are in KiB... FETCH_SIZE counts 128-B read requests at 64 B on gfx950, so read bytes =
2 x FETCH_SIZE x 1024; cross-check with TCC_EA0_RDREQ x 128 B when collected."""
import csv, glob, json, os, statistics, sys
from collections import defaultdict

def main(d, out):
    vals = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name") or row.get("Kernel Name")
            if not k or ("pass" not in k and "k_redfin" not in k and "k_finstep" not in k and "k_em_fin" not in k):
                continue
            k = k.split("(")[0]
            vals[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    res = {}
    for k, cs in vals.items():
        r = {c: statistics.median(v) for c, v in cs.items()}
        if "FETCH_SIZE" in r:
            r["hbm_bytes_per_launch"] = 2 * r["FETCH_SIZE"] * 1024 + r.get("WRITE_SIZE", 0.0) * 1024
        if "TCC_EA0_RDREQ_sum" in r:
            r["rdreq_x128B"] = r["TCC_EA0_RDREQ_sum"] * 128
        res[k] = r
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
