#!/bin/bash
# SQ issue/stall counters for one command (run through gpurun): where do the wave cycles of each kernel go?
# usage: tools/profile_sq.sh <tag> <command...>
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/sq_$TAG; mkdir -p $OUT
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES \
  --kernel-trace --output-format csv -d $OUT/pmc -o run -- "$@" > $OUT/run.log 2>&1
python - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
f = glob.glob(os.path.join(root, "pmc", "*counter_collection.csv"))
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("vslam::", "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
lines = []
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    lines.append("%-34s disp %4d  wave_cycles %.3e  active_any %5.1f%%  active_valu %5.1f%%  wait_any %5.1f%%  wait_inst %5.1f%%  valu_insts %.3e" % (
        k[:34], n[k], wc, 100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_VALU", 0) / wc, 100 * c.get("SQ_WAIT_ANY", 0) / wc,
        100 * c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_INSTS_VALU", 0)))
open(os.path.join(root, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:20]))
PY
