"""rocprofv3 counter CSVs of traffic_calib.py -> the calibration table (bytes the counters report against the bytes moved)."""
import csv, glob, os, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from traffic_calib import N_REC, N_ROWS, STREAM

acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = defaultdict(float)
        names, order = {}, []
        for r in csv.DictReader(open(f)):
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
            per[(r["Dispatch_Id"], "duration_ns")] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            if r["Dispatch_Id"] not in names:
                order.append(r["Dispatch_Id"])
            names[r["Dispatch_Id"]] = r["Kernel_Name"].split("(")[0]
        seen = defaultdict(int)
        for disp in sorted(order, key=int):
            k = names[disp]
            if not k.startswith("cal_"):
                continue
            if k == "cal_scatter48":          # alternates: without / with the tag store
                k = "cal_scatter48" if seen[k] % 2 == 0 else "cal_scatter48+tag4"
                seen["cal_scatter48"] += 1
            for (dd, c), v in per.items():
                if dd == disp:
                    acc[k][c].append(v)
useful = {"cal_stream_read": (STREAM, 0), "cal_stream_write": (0, STREAM), "cal_gather48": (48 * N_REC, 0),
          "cal_gather48_indexed": (52 * N_REC, 0), "cal_rows16": (1024 * N_ROWS, 0), "cal_scatter48": (0, 48 * N_REC),
          "cal_scatter48+tag4": (0, 52 * N_REC), "cal_scatter4": (0, 4 * N_REC)}
med = lambda v: sorted(v)[len(v) // 2]
print("%-22s %10s %12s %12s %10s %10s %9s %9s" % ("pattern", "ms", "useful rd MB", "useful wr MB", "FETCH MB", "WRITE MB", "FETCH/rd", "WRITE/wr"))
for k in useful:
    if k not in acc:
        continue
    c = {n: med(v) for n, v in acc[k].items()}
    rd, wr = useful[k]
    f, w = c.get("FETCH_SIZE", float("nan")) * 1024.0, c.get("WRITE_SIZE", float("nan")) * 1024.0
    extra = "  ".join("%s=%.3g" % (n, v) for n, v in sorted(c.items()) if n not in ("FETCH_SIZE", "WRITE_SIZE", "duration_ns"))
    print("%-22s %10.3f %12.1f %12.1f %10.1f %10.1f %9s %9s  %s" % (k, c.get("duration_ns", 0) / 1e6, rd / 1e6, wr / 1e6, f / 1e6, w / 1e6,
          "%.3f" % (f / rd) if rd else "-", "%.3f" % (w / wr) if wr else "-", extra))
