#!/bin/bash
# per-rank batch sweep of the weight gradient's split-K target (TONAL_TN_TARGET = tiles x splits aimed at); same box, same call
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tnt; mkdir -p $O
B=${B:-32}
for v in "$@"; do
  TONAL_TN_TARGET=$v timeout -k 10 200 python $R/bench.py --batch $B --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/t_${B}_$v.log 2>&1 || exit 1
  python - <<PY
import json
for l in open("$O/t_${B}_$v.log"):
    if l.startswith("{"):
        d = json.loads(l); pl = d["roofline"]["per_launch"]
        print("B $B target $v: step %.2f ms  conv2_wgrad %.3f  conv3_wgrad %.3f" % (d["ms_per_step"], pl["conv2_wgrad"]["ms"], pl["conv3_wgrad"]["ms"]), flush=True)
PY
done
