"""Depth cut with the on-device repair under a moving model (bench scene, fused Adam in the loop): per step the quadrants
registered, the repaired tiles, the entries found behind their cuts (inside the depth window; largest tile), those left
behind the window, the appended list entries / checkpoints, and -- for a flagged forward -- why (MgrHeader::rep_why).

    python tools/instr/repair_stats.py [steps] [margin scale] [penalty]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from manus_amd import rasterizer  # noqa: E402
from manus_amd.engine import HipViewCompute  # noqa: E402
from manus_amd.optim import GaussianOptimizer  # noqa: E402
from manus_amd.synthetic import camera_table, make_scene  # noqa: E402

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 150
MARGIN = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
PEN = int(sys.argv[3]) if len(sys.argv) > 3 else 16
V, N, W, H = 8, 300000, 1920, 1080
dev = torch.device("cuda", 0)
views = list(range(V))
scene = make_scene(n_gaussians=N, kind="hand", seed=0, n_cameras=V, width=W, height=H, device=dev)
ct = camera_table(scene["cameras"], dev)
g = torch.Generator(device="cpu").manual_seed(123)
pert = dict(scene)
pert["params"] = {k: (v + 0.01 * v.abs().mean() * torch.randn(v.shape, generator=g).to(dev)) for k, v in scene["params"].items()}
with torch.no_grad():
    targets = HipViewCompute(pert, torch.zeros((V, 3, H, W), device=dev), ct).forward_views_fused(views)[0].contiguous()
rasterizer.context(dev).clear()
c = HipViewCompute(scene, targets, ct, loss="l1+ssim", depth_cut=True)
c.cut_margin, c.cut_penalty = MARGIN, PEN
opt = GaussianOptimizer(c.params, adopt=True)
ctx = rasterizer.context(dev)
c(views, 1.0 / V)
rasterizer.check_overflow(dev)
rasterizer.set_sync_policy(False, dev)
WHY = {1: "units", 2: "view tiles", 4: "candidates", 8: "list / checkpoints", 16: "hinted tile empty", 32: "depth window"}
T = ((W + 15) // 16) * ((H + 15) // 16)
rows, flagged, why_count = [], 0, {}
for step in range(STEPS):
    o = c(views, 1.0 / V)
    bad = False
    try:
        rasterizer.poll(dev)
    except RuntimeError:
        bad = True
    torch.cuda.synchronize()
    ws = ctx.last_ws
    off = c._layout(ws, V, N, W, H)
    hdr = ws.buf[:4096].view(torch.int32).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    nu, lst, ck, why = int(hdr[32]), int(hdr[33]), int(hdr[34]), int(hdr[35])
    pairs = int(ws.buf[off[7] + 4 * V * T: off[7] + 4 * V * T + 4].view(torch.int32).item())
    if c._cut_bit and nu:
        nuc = min(nu, 1024)
        units = ws.buf[off[30]: off[30] + 64 * nuc].view(torch.int32).cpu().numpy().reshape(nuc, 16).astype(np.int64) & 0xFFFFFFFF
        cnt = ws.buf[off[31]: off[31] + 4 * nuc].view(torch.int32).cpu().numpy().astype(np.int64)
        trep = ws.buf[off[29]: off[29] + 4 * V * T].view(torch.int32).cpu().numpy()
        owners = np.array([u for u in range(nuc) if trep[units[u, 0]] == u + 1])
        rows.append((nu, len(owners), int(cnt[owners].sum()), int(cnt[owners].max()), int(units[owners, 12].sum()), lst, ck, pairs))
    elif c._cut_bit:
        rows.append((0, 0, 0, 0, 0, 0, 0, pairs))
    if bad:
        flagged += 1
        for b, name in WHY.items():
            if why & b:
                why_count[name] = why_count.get(name, 0) + 1
        o = c(views, 1.0 / V)
        rasterizer.poll(dev)
    opt.update_learning_rate(opt.state_step + 1)
    opt.step(o["grads"])
    c.mark_params_changed()
a = np.array(rows, dtype=np.float64)
q = lambda col, p: float(np.percentile(a[:, col], p)) if len(a) else 0.0
print("margins x%.2f, countdown %d: %d steps, %d forwards with the cut, %d flagged %s" % (MARGIN, PEN, STEPS, len(a), flagged, why_count))
for name, col in (("quadrants registered", 0), ("tiles repaired", 1), ("entries found (all tiles)", 2), ("entries found (largest tile)", 3),
                  ("instances behind the windows", 4), ("appended list entries", 5), ("appended checkpoints", 6), ("pairs binned", 7)):
    print("  %-30s median %9.0f  p90 %9.0f  max %9.0f" % (name, q(col, 50), q(col, 90), q(col, 100)))
print("quadrants repaired in all: %d" % ctx.cut_repairs)
