"""Compact timeline of a rocprofv3 --kernel-trace CSV: start (us from the first launch of the LAST `--last` ms), duration, queue,
short kernel name -- to read how the batches in flight interleave (r05).

    rocprofv3 --kernel-trace --output-format csv -d OUT -o t -- python bench.py ...
    python tools/trace_timeline.py OUT/.../t_kernel_trace.csv [--last-ms 12] > timeline.txt
"""
import csv
import sys

path = sys.argv[1]
last_ms = float(sys.argv[sys.argv.index("--last-ms") + 1]) if "--last-ms" in sys.argv else None
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        nm = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        nm = nm.split("<")[0].split("(")[0][-40:] + ("<" + nm.split("<", 1)[1][:24] if "<" in nm else "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), nm,
                     r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?"))))
rows.sort()
t_end = max(r[1] for r in rows)
if last_ms is not None:
    rows = [r for r in rows if r[0] >= t_end - last_ms * 1e6]
t0 = rows[0][0]
queues = sorted(set(r[2] for r in rows))
print("# %d kernels, queues %s, span %.3f ms" % (len(rows), queues, (t_end - t0) / 1e6))
for s, e, q, nm, g, w in rows:
    print("%10.1f %8.1f  q%-3s %-66s grid %s x %s" % ((s - t0) / 1e3, (e - s) / 1e3, queues.index(q), nm, g, w))
