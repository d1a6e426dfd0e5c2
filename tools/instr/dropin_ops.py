"""Operator route (bench.py --route dropin): which torch ops the step launches besides the library's kernels.

torch.profiler over a few steps, grouped by op name and input shapes, device time per step.  The glue -- torch.cat of the
features, zeros_like + 0, permutes, gradient accumulation -- is what the reference's own module code does around the operators
(INTEGRATION.md "glue kernels").

    python tools/instr/dropin_ops.py [steps]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.argv = [sys.argv[0]] + sys.argv[1:]
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 10

import bench  # noqa: E402


def main():
    import argparse
    captured = {}
    orig_sync = torch.cuda.synchronize

    # run bench.dropin_main's set-up by borrowing its step function: re-create it here (same code path as the bench)
    ns = argparse.Namespace(views=8, gaussians=300000, width=1920, height=1080, kind="hand", warmup=12, steps=STEPS, dropin_fenced=False,
                            profile_all=False)
    src = bench.dropin_main
    # the bench times inside dropin_main; for the op table the step function is needed: rebuild through a tiny hook
    import types
    code = src.__code__
    # simplest: monkey-patch time.perf_counter's first use -- instead run the bench function under the profiler with few steps
    bench.REPEATS = 1
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        bench.dropin_main(ns)
    n = ns.warmup + 2 * ns.steps            # warm-up + timed loop + breakdown loop
    rows = []
    for e in prof.key_averages(group_by_input_shape=True):
        dt = getattr(e, "self_device_time_total", None)
        if dt is None:
            dt = e.self_cuda_time_total
        if dt > 0:
            rows.append((dt / n, e.count / n, e.key, str(e.input_shapes)[:110]))
    rows.sort(reverse=True)
    print("device time per step by op and input shapes (%d steps in the profile incl. warm-up)" % n)
    for dt, c, k, shp in rows[:60]:
        print("%8.1f us  %5.2f calls/step  %-38s %s" % (dt, c, k[:38], shp))
    print("sum %.1f us/step" % sum(r[0] for r in rows))


main()
