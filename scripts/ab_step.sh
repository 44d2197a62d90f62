#!/bin/bash
# same-box A/B of the train step under rocprofv3 kernel statistics: scripts/ab_step.sh <tag> [lib.so] ...
# prints the step time and the kernels above 1 ms per step for every library given (no argument / "prod": the in-tree one)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04/ab; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  if [ "$lib" = prod ]; then unset TONAL_HIP_LIB; else export TONAL_HIP_LIB=$R/$lib; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o s -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-extras > $O/$name.log 2>&1
  python3 - <<PY
import csv, glob, json
rows = list(csv.DictReader(open(glob.glob("$O/$name/**/*kernel_stats.csv", recursive=True)[0])))
line = [l for l in open("$O/$name.log") if l.startswith("{")]
print("== $name", "ms/step (profiled):", json.loads(line[-1])["ms_per_step"] if line else "?")
for r in rows:
    ms = float(r["TotalDurationNs"]) / 1e6 / 4
    if ms > 0.9:
        print(f'   {r["Name"][:70]:70s} calls {r["Calls"]:>4s}  ms/step {ms:7.2f}  avg {float(r["AverageNs"])/1e6:7.3f}')
PY
done
