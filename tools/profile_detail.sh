#!/bin/bash
# Issue / stall / texture-addresser counters of the SGBM kernels (run through gpurun), a few counters per pass
#   tools/profile_detail.sh <tag> [batch]          SGBM kernels of tools/bench_sgbm.py
#   FILTER=orb_ CMD="python tools/bench_orb.py --batch 256 --reps 2" tools/profile_detail.sh <tag>     any other kernel family
set -u
TAG=${1:-r03}; B=${2:-32}; FILTER=${FILTER:-sgbm}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/sgbmd_$TAG; mkdir -p $OUT
CMD=${CMD:-"python tools/bench_sgbm.py --batch $B --reps 1"}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
           "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/set$i -o run -- $CMD > $OUT/set$i.log 2>&1
done
python - "$OUT" "$FILTER" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(root, "set*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("vslam::", "")
        if sys.argv[2] not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
lines = []
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0)):
    lines.append(k)
    for c, v in sorted(acc[k].items()):
        lines.append("    %-40s %.4e per dispatch (%d dispatches)" % (c, v / max(n[k][c], 1), n[k][c]))
open(os.path.join(root, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:90]))
PY
