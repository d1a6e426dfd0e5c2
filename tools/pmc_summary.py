"""Aggregate rocprofv3 counter_collection CSVs: median counter value per launch for each kernel.

Usage: python tools/pmc_summary.py gpurun_out/pmc_TAG_*  [--json out.json]
FETCH_SIZE / WRITE_SIZE are reported in KB by rocprofv3; on gfx950 FETCH_SIZE under-reports by 2x
(MI355X_MICROARCH.md, HBM section), which `hbm_bytes` below corrects."""
import csv, glob, json, os, sys
from collections import defaultdict

def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    if out_json in args:
        args.remove(out_json)
    if "--workload" in sys.argv:
        w = sys.argv[sys.argv.index("--workload") + 1]
        if w in args:
            args.remove(w)
    acc = defaultdict(lambda: defaultdict(list))
    for d in args:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per_dispatch = defaultdict(float)
            names = {}
            for r in csv.DictReader(open(f)):
                key = (r["Dispatch_Id"], r["Counter_Name"])
                per_dispatch[key] += float(r["Counter_Value"])
                per_dispatch[(r["Dispatch_Id"], "duration_ns")] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                # (template arguments kept: the instantiations of one kernel are different launches of a step -- the three
                #  tiers of k_bin_scatter, of which a view set uses one and the others return at once)
                names[r["Dispatch_Id"]] = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            for (disp, cname), val in per_dispatch.items():
                acc[names[disp]][cname].append(val)
    res = {}
    for k in sorted(acc):
        if not k.startswith("k_"):
            continue
        # (every dispatch of a kernel in the bench process has the same shape: the target images are rendered by one
        # fused launch over all views, bench.py)
        # median over the launches: the first two or three steps of the process run with host synchronisation and without the
        # depth cut (longer lists), the rest is the steady state the bench times
        res[k] = {c: sorted(v)[len(v) // 2] for c, v in acc[k].items()}
        res[k]["launches"] = max(len(v) for v in acc[k].values())
    cols = sorted({c for k in res for c in res[k] if c != "launches"})
    print("%-22s %8s " % ("kernel", "launches") + " ".join("%16s" % c[:16] for c in cols))
    for k, d in res.items():
        print("%-22s %8d " % (k, d["launches"]) + " ".join("%16.1f" % d.get(c, float("nan")) for c in cols))
    if out_json:
        for k, d in res.items():
            if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
                d["hbm_bytes_per_launch"] = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
        wl = None
        if "--workload" in sys.argv:
            wl = [int(x) for x in sys.argv[sys.argv.index("--workload") + 1].split(",")]
        # one step: every kernel's mean per launch x its launches per step (the dominant backward blend runs once per step)
        steps = max(1, res.get("k_blend_bwd", {}).get("launches", 1))
        per_step = lambda d: round(d["launches"] / steps) if d["launches"] * 2 >= steps else 0   # (set-up kernels run a few times per process)
        step_bytes = sum(d.get("hbm_bytes_per_launch", 0.0) * per_step(d) for d in res.values())
        step_ns = sum(d.get("duration_ns", 0.0) * per_step(d) for d in res.values())
        json.dump({"workload": wl, "step_hbm_bytes": step_bytes, "step_kernel_ns": step_ns, "source": "rocprofv3 --pmc passes of `python bench.py --steps 4 --warmup 2 --no-cpu-baseline` "
                   "(tools/pmc_collect.sh: one pass per counter group, never combined with API traces), median per launch, "
                   "summed over XCDs / SEs; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) KB (gfx950 FETCH_SIZE correction, "
                   "MI355X_MICROARCH.md HBM section; checked on this path's gathers and scatters: profiles/r06_counter_calibration.txt)",
                   "kernels": res}, open(out_json, "w"), indent=1)

if __name__ == "__main__":
    main()
