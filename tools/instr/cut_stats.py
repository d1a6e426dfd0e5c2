"""Depth-cut accounting on the bench scene: pairs in the tile lists without / with the cut, what the blend consumed, how
many tiles saturate, where the kept-but-unconsumed entries sit.   python tools/instr/cut_stats.py [views] [gaussians]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from manus_amd import rasterizer  # noqa: E402
from manus_amd.engine import HipViewCompute  # noqa: E402
from manus_amd.synthetic import camera_table, make_scene  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300000
W, H = 1920, 1080
dev = torch.device("cuda", 0)
scene = make_scene(n_gaussians=N, kind="hand", seed=0, n_cameras=V, width=W, height=H, device=dev)
ct = camera_table(scene["cameras"], dev)
targets = torch.rand((V, 3, H, W), device=dev)
c = HipViewCompute(scene, targets, ct, loss="l1+ssim")
views = list(range(V))
c(views)
rasterizer.check_overflow(dev)
rasterizer.set_sync_policy(False, dev)
T = ((W + 15) // 16) * ((H + 15) // 16)


def region(ws, off, idx, n):
    return ws.buf[off[idx]: off[idx] + 4 * n].view(torch.int32).cpu().numpy().astype(np.int64) & 0xFFFFFFFF


def snap():
    ws = rasterizer.context(dev).last_ws
    off = c._layout(ws, V, N, W, H)
    ts = region(ws, off, 7, V * T + 1)
    return dict(count=np.diff(ts), done=region(ws, off, 9, V * T), qend=region(ws, off, 28, V * T),
                zcut=region(ws, off, 26, V * T), zused=region(ws, off, 27, V * T))


c(views)          # fenced, no hints: full lists
torch.cuda.synchronize()
full = snap()
c(views)          # hints of the step before
torch.cuda.synchronize()
cut = snap()
rasterizer.check_overflow(dev)
ne = full["count"] > 0
sat = ne & (full["qend"] != 0xFFFFFFFF)
print("tiles: %d non-empty of %d, %d saturate (%.1f %%)" % (ne.sum(), V * T, sat.sum(), 100.0 * sat.sum() / ne.sum()))
print("pairs in lists: full %d, with cut %d (%.3f)" % (full["count"].sum(), cut["count"].sum(), cut["count"].sum() / full["count"].sum()))
print("consumed (sum of deepest contributor): %d (%.3f of full)" % (full["done"].sum(), full["done"].sum() / full["count"].sum()))
print("  in saturating tiles: full %d, consumed %d, end-of-walk %d, kept by the cut %d" % (
    full["count"][sat].sum(), full["done"][sat].sum(), full["qend"][sat].sum(), cut["count"][sat].sum()))
print("  in unsaturated tiles: full %d, consumed %d" % (full["count"][ne & ~sat].sum(), full["done"][ne & ~sat].sum()))
hinted = full["zcut"] != 0
print("tiles with a hint after the full forward: %d; tiles cut in the next: %d" % (hinted.sum(), (cut["zused"] != 0).sum()))
for lo, hi in ((1, 64), (64, 256), (256, 1024), (1024, 4096), (4096, 1 << 30)):
    m = sat & (full["count"] >= lo) & (full["count"] < hi)
    mu = ne & ~sat & (full["count"] >= lo) & (full["count"] < hi)
    print("  lists [%5d, %5d): sat tiles %6d full %9d end %9d kept %9d | unsat tiles %6d full %9d" % (
        lo, min(hi, 99999), m.sum(), full["count"][m].sum(), full["qend"][m].sum(), cut["count"][m].sum(), mu.sum(), full["count"][mu].sum()))
