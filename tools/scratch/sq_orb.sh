cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/sq_orb_r6; mkdir -p $OUT
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/sq -o b -- python tools/bench_orb.py --batch 512 --reps 3 > $OUT/log.txt 2>&1
python - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
f = glob.glob(os.path.join(root, "sq", "**", "*counter_collection.csv"), recursive=True)
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("vslam::", "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    if "at::" in k or "rocclr" in k: continue
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    print("%-36s disp %4d wave_cyc %.3e busy_cyc %.3e active_any %5.1f%% active_valu %5.1f%% wait_any %5.1f%% wait_inst %5.1f%% valu_insts %.3e waves %.3e" % (
        k[:36], n[k], wc, c.get("SQ_BUSY_CYCLES",0), 100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_VALU", 0) / wc, 100 * c.get("SQ_WAIT_ANY", 0) / wc, 100*c.get("SQ_WAIT_INST_ANY",0)/wc, c.get("SQ_INSTS_VALU", 0), c.get("SQ_WAVES",0)))
PY
