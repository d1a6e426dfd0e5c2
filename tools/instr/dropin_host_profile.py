"""Operator route (bench.py --route dropin): where the HOST spends the step (cProfile over the bench's own loop).

The route is host-bound once nothing in it waits for the device (bench line: host_issue_ms_per_step ~ ms_per_step), so the
Python time per call of the library's entry points is what is left to trim.

    python tools/instr/dropin_host_profile.py [steps]
"""
import argparse
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 100
sys.argv = sys.argv[:1]

import bench  # noqa: E402

bench.REPEATS = 1
ns = argparse.Namespace(views=8, gaussians=300000, width=1920, height=1080, kind="hand", warmup=12, steps=STEPS, dropin_fenced=False, profile_all=False)
pr = cProfile.Profile()
pr.enable()
bench.dropin_main(ns)
pr.disable()
n = ns.warmup + 2 * ns.steps
st = pstats.Stats(pr)
st.sort_stats("cumulative")
rows = []
for (fn, line, name), (cc, nc, tt, ct, callers) in st.stats.items():
    rows.append((ct / n * 1e6, tt / n * 1e6, nc / n, "%s:%d %s" % (os.path.basename(fn), line, name)))
rows.sort(reverse=True)
print("host time per step (us): cumulative, own, calls/step   [%d steps incl. warm-up and the breakdown loop]" % n)
for ct, tt, nc, nm in rows[:70]:
    print("%9.1f %9.1f %7.2f  %s" % (ct, tt, nc, nm))
