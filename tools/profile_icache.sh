#!/bin/bash
# instruction-cache counters of a command (run through gpurun): does a kernel's code fit the 64 KB cache two CUs share?
# usage: tools/profile_icache.sh <tag> <command...>
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/icache_$TAG; mkdir -p $OUT
timeout 600 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES \
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
    if r["Counter_Name"] == "SQC_ICACHE_REQ": n[k] += 1
lines = []
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    if "at::" in k or "rocclr" in k: continue
    req = c.get("SQC_ICACHE_REQ", 0) or 1
    lines.append("%-36s disp %4d  icache_req %.3e  hit %5.1f%%  miss %5.1f%%  miss_dup %5.1f%%  ifetch %.3e  wait_inst_any %5.1f%% of wave cycles" % (
        k[:36], n[k], req, 100 * c.get("SQC_ICACHE_HITS", 0) / req, 100 * c.get("SQC_ICACHE_MISSES", 0) / req, 100 * c.get("SQC_ICACHE_MISSES_DUPLICATE", 0) / req,
        c.get("SQ_IFETCH", 0), 100 * c.get("SQ_WAIT_INST_ANY", 0) / (c.get("SQ_WAVE_CYCLES", 0) or 1)))
open(os.path.join(root, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:16]))
PY
