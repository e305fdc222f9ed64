#!/usr/bin/env python3
"""Summarise rocprofv3 outputs of tools/profile_round.sh: per-kernel time stats and HBM traffic from the PMC passes.
FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B (rocprofv3); on gfx950 FETCH_SIZE under-reports wide coalesced
reads by 2x (MI355X_MICROARCH.md, HBM section) -- both the raw and the x2-corrected read bytes are printed."""
import csv, glob, json, os, sys
from collections import defaultdict

root = sys.argv[1]
def short(n):
    n = n.split("(")[0]
    return n.replace("void ", "").replace("vslam::", "")

stats = glob.glob(os.path.join(root, "trace", "*kernel_stats.csv"))
if stats:
    print("== kernel stats ==")
    for r in csv.DictReader(open(stats[0])):
        if "vslam" in r["Name"]:
            print("%-38s calls %5s avg %10.1f us total %9.3f ms  %5s%%" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
out = {}
for name, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    files = glob.glob(os.path.join(root, sub, "*counter_collection.csv"))
    if not files:
        print("no counter file for", name); continue
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(files[0])):
        if r.get("Counter_Name") != name: continue
        k = short(r["Kernel_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    print("== %s (sum over dispatches, units of 1 KiB) ==" % name)
    for k, (v, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
        if "_kernel" in k and "at::" not in k and "rocclr" not in k:
            print("%-38s dispatches %5d  total %12.1f KiB  per dispatch %10.1f KiB" % (k, n, v, v / max(n, 1)))
        out.setdefault(k, {})[name] = dict(total_kib=v, dispatches=n)
json.dump(out, open(os.path.join(root, "pmc_summary.json"), "w"), indent=1)
