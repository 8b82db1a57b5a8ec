"""Developer tool: per-kernel averages of the counters of one `rocprofv3 --pmc ... --output-format csv` run.
usage: pmc_kernel_table.py <dir> [kernel name substring ...]"""
import csv, glob, sys, collections
d = sys.argv[1]
subs = sys.argv[2:] or ["k_pass"]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(s in k for s in subs):
            continue
        k = k.split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        print(f"    {c:28s} {acc[k][c] / n[k][c]:16.1f}  (x{n[k][c]})")
