#!/bin/bash
# Texture-addresser / L1 counters of the BA kernel, two or three per pass (the TA / TCP blocks expose few counter slots)
set -u
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/lmta_$TAG; mkdir -p $OUT
CMD="python tools/bench_ba.py --windows 256 --reps 1"
i=0
for set in "TA_TA_BUSY_sum TA_BUSY_avr" "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "TCP_TCP_LATENCY_sum TCP_GATE_EN1_sum"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/set$i -o run -- $CMD > $OUT/set$i.log 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
acc = defaultdict(float); n = defaultdict(int)
for f in glob.glob(os.path.join(root, "set*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "lm_window_kernel" in r["Kernel_Name"] and "Lb1" not in r["Kernel_Name"] and "<true>" not in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
lines = ["%-40s %.4e  (%d dispatches)" % (k, v, n[k]) for k, v in sorted(acc.items())]
open(os.path.join(root, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
