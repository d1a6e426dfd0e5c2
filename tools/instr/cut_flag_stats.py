"""Depth cut under a MOVING model: how many tiles a step flags, and how much list they lack.

The bench scene with the fused Adam step in the loop (the reference's learning rates), the cut applied on EVERY step at fixed
margins (no back-off: a flagged step is run again on full lists, which also renews the hints).  Per configuration of the
margins: steps flagged, flagged tiles per flagged step (median / max), the entries those tiles' walks needed beyond their cut
lists (from the full-list re-run), pairs binned with the cut against full lists.  This is the workload an on-device repair of
flagged tiles would see (LAB.md, round 6).

    python tools/instr/cut_flag_stats.py [steps] [views]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from manus_amd import rasterizer  # noqa: E402
from manus_amd._lib import lib  # noqa: E402
from manus_amd.engine import HipViewCompute  # noqa: E402
from manus_amd.optim import GaussianOptimizer  # noqa: E402
from manus_amd.synthetic import camera_table, make_scene  # noqa: E402

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 120
V = int(sys.argv[2]) if len(sys.argv) > 2 else 8
N, W, H = 300000, 1920, 1080
dev = torch.device("cuda", 0)
T = ((W + 15) // 16) * ((H + 15) // 16)
views = list(range(V))


def run(scale, interior):
    scene = make_scene(n_gaussians=N, kind="hand", seed=0, n_cameras=V, width=W, height=H, device=dev)
    ct = camera_table(scene["cameras"], dev)
    g = torch.Generator(device="cpu").manual_seed(123)
    pert = dict(scene)
    pert["params"] = {k: (v + 0.01 * v.abs().mean() * torch.randn(v.shape, generator=g).to(dev)) for k, v in scene["params"].items()}
    rasterizer.set_sync_policy(True, dev)
    with torch.no_grad():
        targets = HipViewCompute(pert, torch.zeros((V, 3, H, W), device=dev), ct).forward_views_fused(views)[0].contiguous()
    rasterizer.context(dev).clear()
    c = HipViewCompute(scene, targets, ct, loss="l1+ssim", depth_cut=True)
    opt = GaussianOptimizer(c.params, adopt=True)
    ctx = rasterizer.context(dev)
    orig = c._cut_flag

    def fixed_margins(ws, view_ids, V_, N_, W_, H_):
        c._cut_pause, c._cut_scale, c._cut_seen = 0, 1.0, ctx.cut_retries      # no back-off, no widening
        bit = orig(ws, view_ids, V_, N_, W_, H_)
        lib().mgr_raster_set_cut_margin(0.125 * scale, int(64 * scale), 0.0625 * scale, 2.0e-4 * scale, interior)
        return bit

    c._cut_flag = fixed_margins
    c(views, 1.0 / V)
    rasterizer.check_overflow(dev)
    rasterizer.set_sync_policy(False, dev)

    def reg(ws, off, i, n):
        return ws.buf[off[i]: off[i] + 4 * n].view(torch.int32).cpu().numpy().astype(np.int64) & 0xFFFFFFFF

    flagged_steps, tiles, lack, lack_max, pairs_cut, pairs_full, cut_steps = 0, [], [], [], [], [], 0
    for step in range(STEPS):
        o = c(views, 1.0 / V)
        used_cut = c._cut_bit != 0
        try:
            rasterizer.poll(dev)
            if used_cut:
                cut_steps += 1
                torch.cuda.synchronize()
                ws = ctx.last_ws
                off = c._layout(ws, V, N, W, H)
                pairs_cut.append(int(reg(ws, off, 7, V * T + 1)[-1]))
        except RuntimeError:
            torch.cuda.synchronize()
            ws = ctx.last_ws
            off = c._layout(ws, V, N, W, H)
            cnt = np.diff(reg(ws, off, 7, V * T + 1))
            qend, zused = reg(ws, off, 28, V * T), reg(ws, off, 27, V * T)
            bad = np.nonzero((zused != 0) & ((qend == 0xFFFFFFFF) | (cnt == 0)))[0]
            o = c(views, 1.0 / V)             # full lists (cut_block)
            rasterizer.poll(dev)
            torch.cuda.synchronize()
            cnt2, done2 = np.diff(reg(ws, off, 7, V * T + 1)), reg(ws, off, 9, V * T)
            flagged_steps += 1
            tiles.append(len(bad))
            need = np.maximum(done2[bad] - cnt[bad], 0)      # entries the deepest walk of the tile consumed beyond the cut list
            lack.append(int(need.sum()))
            lack_max.append(int(need.max()) if len(bad) else 0)
            pairs_full.append(int(cnt2.sum()))
        opt.update_learning_rate(opt.state_step + 1)
        opt.step(o["grads"])
        c.mark_params_changed()
    q = lambda a, p: (float(np.percentile(a, p)) if len(a) else 0.0)
    print("margins x%.2f interior=%d: %d steps, %d with the cut, %d flagged | flagged tiles per flagged step: median %.0f, p90 %.0f, max %.0f | "
          "entries lacking per flagged step: median %.0f, max %.0f (deepest single tile %d) | pairs binned: cut %.2f M, full %.2f M"
          % (scale, interior, STEPS, cut_steps + flagged_steps, flagged_steps, q(tiles, 50), q(tiles, 90), q(tiles, 100), q(lack, 50), q(lack, 100),
             max(lack_max) if lack_max else 0, np.mean(pairs_cut) / 1e6 if pairs_cut else 0.0, np.mean(pairs_full) / 1e6 if pairs_full else 0.0))
    sys.stdout.flush()
    ctx.clear()


for scale, interior in ((1.0, 0), (1.0, 1), (0.5, 0), (0.25, 0), (2.0, 0)):
    run(scale, interior)
