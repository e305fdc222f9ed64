#!/usr/bin/env python3
"""profiles/traffic.json from the counter passes of tools/profile_all.sh (gpurun_out/prof_<tag>/pmc_summary.json,
gpurun_out/sq_<tag>_ba/summary.txt, gpurun_out/tcc_<tag>_ba/summary.txt): HBM-side bytes of the dominant kernel per launch set,
corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE counts 64 B per 128-B request: x2; WRITE_SIZE exact -- calibrated on
known byte counts by tools/scratch/fetch_calib.hip in round 1).  usage: tools/make_traffic_json.py <tag> <batch> [dispatches_per_set=4]"""
import json, os, re, sys
tag, batch = sys.argv[1], int(sys.argv[2]); per_set = int(sys.argv[3]) if len(sys.argv) > 3 else 4
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pm = json.load(open(os.path.join(root, "gpurun_out", "prof_" + tag, "pmc_summary.json")))
k = [n for n in pm if n.startswith("lm_window_kernel<false>")][0]
f = pm[k]["FETCH_SIZE"]; w = pm[k]["WRITE_SIZE"]
fetch_kib = f["total_kib"] / f["dispatches"]; write_kib = w["total_kib"] / w["dispatches"]
out = dict(batch=batch, kernel="lm_window_kernel", tag=tag, fetch_kib_per_dispatch=round(fetch_kib, 1), write_kib_per_dispatch=round(write_kib, 1),
           fetch_size_correction=2.0, hbm_bytes_per_launch_set_raw_counters=(fetch_kib + write_kib) * 1024 * per_set,
           hbm_bytes_per_launch_set=(2 * fetch_kib + write_kib) * 1024 * per_set,
           note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/profile_round.sh), per dispatch x %d dispatches per schedule; "
                "HBM-side bytes = 2 x FETCH_SIZE + WRITE_SIZE (FETCH_SIZE tallies 64 B per 128-B request on gfx950; calibrated in round 1 with "
                "tools/scratch/fetch_calib.hip)" % per_set)
for name, key in (("sq_%s_ba" % tag, "sq"), ("tcc_%s_ba" % tag, "tcc")):
    p = os.path.join(root, "gpurun_out", name, "summary.txt")
    if os.path.exists(p):
        line = [l for l in open(p) if l.startswith("lm_window_kernel<false>")]
        if line:
            out[key + "_summary"] = " ".join(line[0].split())
            m = re.search(r"valu_insts ([0-9.e+]+)", line[0]); d = re.search(r"disp\s+(\d+)", line[0])
            if m and d:
                out["valu_wave_insts_per_window_schedule"] = float(m.group(1)) / (int(d.group(1)) / per_set) / batch
json.dump(out, open(os.path.join(root, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
