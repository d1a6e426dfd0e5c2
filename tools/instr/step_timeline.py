"""One step of the bench from a rocprofv3 kernel trace: start / end / duration / queue of every dispatch between two
consecutive k_skin_fwd24x8 launches.   python tools/instr/step_timeline.py <run_kernel_trace.csv> [which=-3]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
which = int(sys.argv[2]) if len(sys.argv) > 2 else -3
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_skin_fwd24x8")]
a, b = idx[which], idx[which + 1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = {}
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    q = r.get("Queue_Id", "?")
    gap = s - prev_end.get(q, s)
    prev_end[q] = e
    print("%8.1f %8.1f %7.1f  gap %5.1f  q%s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, gap / 1e3, q, r["Kernel_Name"][:60]))
