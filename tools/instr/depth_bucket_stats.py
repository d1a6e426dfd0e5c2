"""Instance sort: how the depth buckets fill in the trained state (the Trainer's model after some hundred Adam steps).

Per view: the depth range of the visible instances, its 0.1 / 99.9 percentiles, the largest of B uniform buckets over the range
(what k_dbin_rank's items are built from: an item is ~MGR_DB_ITEM keys plus its last bucket) for B = 1024 / 4096, linear in z
and linear in log z.

    python tools/instr/depth_bucket_stats.py [adam steps]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from manus_amd import rasterizer  # noqa: E402
from manus_amd.engine import HipViewCompute, Trainer  # noqa: E402
from manus_amd.synthetic import camera_table, make_scene  # noqa: E402
from parity import layout  # noqa: E402

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 600
dev = torch.device("cuda", 0)
V, N, W, H = 8, 300000, 1920, 1080
scene = make_scene(n_gaussians=N, kind="hand", seed=0, n_cameras=V, width=W, height=H, device=dev)
ct = camera_table(scene["cameras"], dev)
g = torch.Generator(device="cpu").manual_seed(123)
pert = dict(scene)
pert["params"] = {k: (v + 0.01 * v.abs().mean() * torch.randn(v.shape, generator=g).to(dev)) for k, v in scene["params"].items()}
with torch.no_grad():
    targets = HipViewCompute(pert, torch.zeros((V, 3, H, W), device=dev), ct).forward_views_fused(list(range(V)))[0].contiguous()
compute = HipViewCompute(scene, targets, ct, loss="l1+ssim")
tr = Trainer(compute, V, extent=0.3, opts=dict(remove_seg_end=0, densify_from_step=10 ** 9, opacity_reset_interval=10 ** 9))
tr.global_step = 1


def report(tag):
    torch.cuda.synchronize()
    ws = rasterizer.context(dev).last_ws
    L = layout(V, N, W, H, ws.cap)
    depth = ws.buf[L["depth"]: L["depth"] + 4 * V * N].view(torch.float32).reshape(V, N).cpu().numpy()
    rect = ws.buf[L["rect"]: L["rect"] + 8 * V * N].view(torch.int16).reshape(V, N, 4).cpu().numpy().astype(np.int64) & 0xFFFF
    vis = ((rect[..., 2] - rect[..., 0]) * (rect[..., 3] - rect[..., 1]) > 0)
    t = ws.tiers or 0
    print("%s: sort items of the last forward beyond 13/16 of k_dbin_rank's small capacity: %d, of its large one: %d; beyond the small capacity: %d; "
          "debug bits of the next forward: %d" % (tag, (t >> 8) & 0xFF, (t >> 16) & 0xFF, (t >> 24) & 0x7F, ws.skip_bits()))
    for v in range(V):
        z = depth[v][vis[v]]
        lo, hi = z.min(), z.max()
        p = np.percentile(z, [0.1, 99.9])
        row = []
        for B in (1024, 4096):
            row.append(int(np.bincount(np.minimum(((z - lo) * ((B - 1) / (hi - lo))).astype(np.int64), B - 1), minlength=B).max()))
            lz = np.log(z)
            row.append(int(np.bincount(np.minimum(((lz - lz.min()) * ((B - 1) / (lz.max() - lz.min()))).astype(np.int64), B - 1), minlength=B).max()))
        # the sort items of this view (1024 linear buckets, 768 keys per item: an item = the buckets whose first key lies in its range)
        cnt = np.bincount(np.minimum(((z - lo) * (1023 / (hi - lo))).astype(np.int64), 1023), minlength=1024)
        start = np.concatenate([[0], np.cumsum(cnt)])
        first_item = start[:-1] // 768
        sizes = np.bincount(first_item[cnt > 0], weights=cnt[cnt > 0]).astype(np.int64)
        top = np.sort(sizes)[::-1][:4]
        print("  view %d: %6d visible, z in [%.4f, %.4f], 0.1 / 99.9 %%: [%.4f, %.4f] | largest bucket: 1024 lin %5d log %5d, 4096 lin %5d log %5d | largest items %s"
              % (v, z.size, lo, hi, p[0], p[1], row[0], row[1], row[2], row[3], top.tolist()))


for _ in range(3):
    tr.train_step()
report("start")
for k in range(STEPS):
    if tr.global_step % 100 == 0:
        tr.global_step += 1
    tr.train_step()
report("after %d Adam steps" % STEPS)
