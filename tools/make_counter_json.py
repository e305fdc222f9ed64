#!/usr/bin/env python3
"""profiles/<out>.json from the counter passes of tools/profile_r04.sh: HBM-side bytes of a kernel family per launch set and its SQ figures,
stamped with the sha256 of the kernel sources they were measured on (bench.py reports `traffic: null` when a source has changed since).
HBM-side bytes = 2 x FETCH_SIZE + WRITE_SIZE (FETCH_SIZE tallies 64 B per 128-B request on gfx950: MI355X_MICROARCH.md HBM section, calibrated
in round 1 with tools/scratch/fetch_calib.hip; WRITE_SIZE exact), separate --pmc passes.
usage: make_counter_json.py <prof dir> <out name> <batch> <steps in the profiled run> <kernel name prefix>[,prefix...] <source.hip>[,...] [label]"""
import hashlib, json, os, re, sys

prof, out_name, batch, steps, prefixes, sources = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5].split(","), sys.argv[6].split(",")
label = sys.argv[7] if len(sys.argv) > 7 else prefixes[0]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pm = json.load(open(os.path.join(prof, "pmc_summary.json")))
fetch = write = 0.0; per = {}
for k, v in pm.items():
    if not any(k.startswith(p) for p in prefixes) or "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    f, w = v["FETCH_SIZE"]["total_kib"] * 1024 / steps, v["WRITE_SIZE"]["total_kib"] * 1024 / steps
    per[k] = {"fetch_bytes_raw_per_set": round(f), "write_bytes_per_set": round(w), "dispatches_per_set": v["FETCH_SIZE"]["dispatches"] / steps}
    fetch += f; write += w
sha = {s: hashlib.sha256(open(os.path.join(root, "stereo-visual-slam_amd", "csrc", s), "rb").read()).hexdigest()[:16] for s in sources}
out = dict(kernel=label, batch=batch, launch_set="one bench step's launches of: " + ", ".join(prefixes), source_sha16=sha,
           fetch_bytes_raw_per_launch_set=round(fetch), write_bytes_per_launch_set=round(write), fetch_size_correction=2.0,
           hbm_bytes_per_launch_set=round(2 * fetch + write), per_kernel=per,
           source="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/profile_r06.sh) over %d steps; HBM-side bytes = 2 x FETCH_SIZE + WRITE_SIZE" % steps)
sq = os.path.join(prof, "sq_summary.txt")
if os.path.exists(sq):
    rows = {}
    for line in open(sq):
        name = line.split()[0]
        if any(name.startswith(p) for p in prefixes):
            m = {k: float(v) for k, v in re.findall(r"(\w+)\s+([0-9][0-9.e+]*)%?", line)}
            rows[name] = line.strip()
            if name.startswith("lm_window_kernel<false") or name.startswith("ba_resident_kernel"):
                d = int(m.get("disp", 0)) or 1
                # one launch per schedule: lm_window_kernel<false, true> (in-kernel adaptive schedule) and ba_resident_kernel<true>; lm_window_kernel<false, false>: three
                per_sched = 1.0 if ("<false, true>" in line or name.startswith("ba_resident_kernel")) else 3.0
                # (with ba_resident_kernel in the set, lm_window_kernel only takes the deferred windows -- usually none: its launches return at once)
                key = "resident_" if name.startswith("ba_resident_kernel") else ""
                out[key + "valu_wave_insts_per_window_schedule"] = m["valu_insts"] / (d / per_sched) / batch
                out[key + "valu_active_pct_of_wave_cycles"] = m.get("active_valu")
    out["sq"] = rows
json.dump(out, open(os.path.join(root, "profiles", out_name), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k not in ("per_kernel", "sq")}, indent=1))
