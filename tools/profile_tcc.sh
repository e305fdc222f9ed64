#!/bin/bash
# L2 / fabric request counters for one command (run through gpurun): HBM-side bytes and L2 hit rate per kernel.
# usage: tools/profile_tcc.sh <tag> <command...>
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/tcc_$TAG; mkdir -p $OUT
timeout 900 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/rd -o run -- "$@" > $OUT/rd.log 2>&1
timeout 900 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $OUT/wr -o run -- "$@" > $OUT/wr.log 2>&1
python - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for sub in ("rd", "wr"):
    for f in glob.glob(os.path.join(root, sub, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("vslam::", "")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "TCC_HIT_sum": n[k] += 1
lines = []
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("TCC_EA0_RDREQ_sum", 0)):
    rd = c.get("TCC_EA0_RDREQ_sum", 0); rd32 = c.get("TCC_EA0_RDREQ_32B_sum", 0)
    wr = c.get("TCC_EA0_WRREQ_sum", 0); wr64 = c.get("TCC_EA0_WRREQ_64B_sum", 0)
    hit = c.get("TCC_HIT_sum", 0); miss = c.get("TCC_MISS_sum", 0)
    rb = (rd - rd32) * 64 + rd32 * 32; wb = wr64 * 64 + (wr - wr64) * 32
    lines.append("%-34s disp %4d  rd %.3f GB (req %.3e, 32B %.3e)  wr %.3f GB (req %.3e, 64B %.3e)  L2 hit %.1f%%" % (
        k[:34], n[k], rb / 1e9, rd, rd32, wb / 1e9, wr, wr64, 100 * hit / max(hit + miss, 1)))
open(os.path.join(root, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:12]))
PY
