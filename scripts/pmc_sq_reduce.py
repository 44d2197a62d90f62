"""Per-kernel means of the SQ counters collected by scripts/pmc_sq.sh (rocprofv3 counter_collection csv)."""
import csv, glob, os, sys
from collections import defaultdict
tot = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
for k, c in tot.items():
    n = len(cnt[k])
    if c.get("SQ_WAVE_CYCLES", 0) / max(n, 1) < 1e7:
        continue
    w = c["SQ_WAVE_CYCLES"]
    print(f"{k}  launches {n}")
    for name in sorted(c):
        print(f"   {name:28s} {c[name] / n:16.0f}  {100 * c[name] / w:7.2f} % of WAVE_CYCLES")
