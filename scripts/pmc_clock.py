"""profiles/effective_clock.json from one `rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace` pass of the bench command:
effective clock of a launch = GRBM_GUI_ACTIVE / 8 XCDs / its wall time (MI355X_MICROARCH.md, DVFS give-back), mean over the
launches of a kernel family (names as Engine.kernel_families() / scripts/pmc_traffic.py give them).

    python scripts/pmc_clock.py gpurun_out/<tag>/pmc_c "<how it was collected>"
"""
import csv, glob, json, os, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_traffic.py")).read()
ns = {}
exec(src[src.index("F6 ="):src.index("def collect")], ns)            # the family table of pmc_traffic.py (+ the tn4y split)
FAMILIES, TN4Y, tn4y_split = ns["FAMILIES"], ns["TN4Y"], ns["tn4y_split"]
acc = defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    which = tn4y_split(rows)
    for r in rows:
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
            continue
        dt = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])           # ns
        if dt < 1e6:                                                           # the quotient reads high on launches under ~1 ms
            continue
        for key, fam in FAMILIES.items():
            if key in r["Kernel_Name"]:
                acc[fam].append(float(r["Counter_Value"]) / 8.0 / dt)           # cycles per ns = GHz
        if r["Dispatch_Id"] in which:
            acc[TN4Y[which[r["Dispatch_Id"]]]].append(float(r["Counter_Value"]) / 8.0 / dt)
out = {"effective_clock_ghz": {k: round(sum(v) / len(v), 3) for k, v in acc.items() if 1.0 < sum(v) / len(v) < 2.6},
       "launches_averaged": {k: len(v) for k, v in acc.items()},
       "source": "one rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace pass of `bench.py --steps 3 --warmup 1 --no-cpu-baseline "
                 "--no-kernel-timers --no-extras` (" + (sys.argv[2] if len(sys.argv) > 2 else "scripts/collect_r05_profiles.sh c3clock")
                 + "): GRBM_GUI_ACTIVE / 8 XCDs / dispatch wall time, mean over the launches (MI355X_MICROARCH.md, DVFS give-back); "
                 "families whose quotient falls outside 1.0 - 2.6 GHz (a wrapped counter) are left out",
       "nominal_clock_ghz": 2.4}
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "effective_clock.json")
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
