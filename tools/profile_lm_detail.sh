#!/bin/bash
# Detailed SQ / TA / LDS counters of the BA kernel (run through gpurun): tools/profile_lm_detail.sh <tag>
set -u
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/lmdetail_$TAG; mkdir -p $OUT
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
CMD="python tools/bench_ba.py --windows 256 --reps 1"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_WAVES SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU" \
           "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum" \
           "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/set$i -o run -- $CMD > $OUT/set$i.log 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
acc = defaultdict(float)
for f in glob.glob(os.path.join(root, "set*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "lm_window_kernel" in r["Kernel_Name"] and "Lb0" not in r["Kernel_Name"] and "<false>" in r["Kernel_Name"].replace("Lb0", "<false>"):
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
for f in glob.glob(os.path.join(root, "set*", "*counter_collection.csv")):
    pass
lines = ["%-40s %.4e" % (k, v) for k, v in sorted(acc.items())]
open(os.path.join(root, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
grep -c . $OUT/avail.txt; tail -3 $OUT/set4.log
