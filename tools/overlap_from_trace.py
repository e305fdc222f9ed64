"""How much do the kernels of two batches in flight overlap?  Reads a rocprofv3 --kernel-trace CSV (Start_Timestamp / End_Timestamp per
dispatch, ns) and prints, for the busiest contiguous part of the run: the wall span, the union of the kernel intervals (time with at
least one kernel running), the time with two or more running, and the sum of the durations.
usage: python tools/overlap_from_trace.py <dir with *kernel_trace.csv> [skip_fraction]"""
import csv, glob, os, sys

root = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
f = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for r in csv.DictReader(open(f[0])):
    name = r["Kernel_Name"]
    if "at::" in name or "rocclr" in name or "hbm_copy_probe" in name:
        continue
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name.split("(")[0].replace("void ", "").replace("vslam::", ""), r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t0 + skip * (t1 - t0)
rows = [r for r in rows if r[0] >= lo]
ev = []
for s, e, _, _ in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth, last, busy1, busy2, big_gaps = 0, ev[0][0], 0, 0, 0
for t, d in ev:
    if depth >= 1: busy1 += t - last
    if depth >= 2: busy2 += t - last
    if depth == 0 and t - last >= 500000: big_gaps += t - last   # >= 0.5 ms with nothing running: between the timed regions (synchronise, copy probe, downloads)
    depth += d; last = t
span = max(r[1] for r in rows) - rows[0][0] - big_gaps
tot = sum(e - s for s, e, _, _ in rows)
queues = sorted(set(r[3] for r in rows))
print("dispatches %d on queues/streams %s" % (len(rows), ",".join(queues)))
print("(idle gaps >= 0.5 ms between the regions, %.3f ms, are not part of the span)" % (big_gaps / 1e6))
print("span %.3f ms | >=1 kernel running %.3f ms (%.1f%%) | >=2 running %.3f ms (%.1f%%) | sum of kernel durations %.3f ms (%.2fx the span)" % (
    span / 1e6, busy1 / 1e6, 100.0 * busy1 / span, busy2 / 1e6, 100.0 * busy2 / span, tot / 1e6, tot / span))
per = {}
for s, e, n, _ in rows:
    a = per.setdefault(n, [0, 0]); a[0] += e - s; a[1] += 1
for n, (d, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:12]:
    print("  %-40s calls %5d  avg %9.1f us  total %9.3f ms" % (n[:40], c, d / c / 1e3, d / 1e6))
