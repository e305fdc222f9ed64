#!/bin/bash
# Instruction mix (VALU / SALU / LDS / VMEM / SMEM wave-instructions, LDS activity) per kernel for one command (run through gpurun).
# usage: tools/profile_mix.sh <tag> <command...>
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/mix_$TAG; mkdir -p $OUT
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM \
  --kernel-trace --output-format csv -d $OUT/pmc1 -o run -- "$@" > $OUT/run1.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAVES \
  --kernel-trace --output-format csv -d $OUT/pmc2 -o run -- "$@" > $OUT/run2.log 2>&1
python - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float))
for sub in ("pmc1", "pmc2"):
    for f in glob.glob(os.path.join(root, sub, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("vslam::", "")
            acc[k][sub + ":" + r["Counter_Name"]] += float(r["Counter_Value"])
lines = []
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("pmc1:SQ_WAVE_CYCLES", 0)):
    g = lambda n, s="pmc1": c.get(s + ":" + n, 0.0)
    wc2 = g("SQ_WAVE_CYCLES", "pmc2") or 1
    lines.append("%-30s waves %.3e valu %.3e salu %.3e lds %.3e vmem_rd %.3e vmem_wr %.3e smem %.3e | busy_cycles %.3e | of wave cycles: valu %4.1f%% scalar %4.1f%% lds %4.1f%% wait_lds %4.1f%% | lds bank conflict %4.1f%% of lds active" % (
        k[:30], g("SQ_WAVES", "pmc2"), g("SQ_INSTS_VALU"), g("SQ_INSTS_SALU"), g("SQ_INSTS_LDS"), g("SQ_INSTS_VMEM_RD"), g("SQ_INSTS_VMEM_WR"), g("SQ_INSTS_SMEM"), g("SQ_BUSY_CYCLES"),
        100 * g("SQ_ACTIVE_INST_VALU", "pmc2") / wc2, 100 * g("SQ_ACTIVE_INST_SCA", "pmc2") / wc2, 100 * g("SQ_ACTIVE_INST_LDS", "pmc2") / wc2, 100 * g("SQ_WAIT_INST_LDS", "pmc2") / wc2,
        100 * g("SQ_LDS_BANK_CONFLICT", "pmc2") / (g("SQ_LDS_IDX_ACTIVE", "pmc2") or 1)))
open(os.path.join(root, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:12]))
PY
