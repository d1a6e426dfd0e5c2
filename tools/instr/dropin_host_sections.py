"""Operator route: host time per section of the step (perf_counter, no profiler, no synchronisation inside the loop).

    python tools/instr/dropin_host_sections.py [steps]
"""
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from manus_amd import _lib, losses, rasterizer  # noqa: E402
from manus_amd.engine import HipViewCompute  # noqa: E402
from manus_amd.modules import hand_forward  # noqa: E402
from manus_amd.render import render_gaussians  # noqa: E402
from manus_amd.structures import Bones  # noqa: E402
from manus_amd.synthetic import camera_table, make_scene  # noqa: E402

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda", 0)
V, N, W, H = 8, 300000, 1280, 720
scene = make_scene(n_gaussians=N, kind="hand", seed=0, n_cameras=V, width=W, height=H, device=dev)
ct = camera_table(scene["cameras"], dev)
with torch.no_grad():
    targets = HipViewCompute(scene, torch.zeros((V, 3, H, W), device=dev), ct).forward_views_fused(list(range(V)))[0]
    targets_hwc = targets.permute(0, 2, 3, 1).contiguous()
rasterizer.context(dev).clear()
P = {k: v.detach().clone().requires_grad_(True) for k, v in scene["params"].items()}


class Model:
    _xyz, _scaling, _rotation = P["_xyz"], P["_scaling"], P["_rotation"]
    grid_center, grid_scale, grid_weights = scene.get("grid_center"), scene.get("grid_scale"), scene.get("grid")

    @property
    def get_features(self):
        return torch.cat([P["_features_dc"], P["_features_rest"]], dim=1)

    @property
    def get_opacity(self):
        return torch.sigmoid(P["_opacity"])


model = Model()
cams = [SimpleNamespace(fovx=c["fovx"], fovy=c["fovy"], height=c["height"], width=c["width"],
                        world_view_transform=torch.tensor(c["world_view_transform"], dtype=torch.float32, device=dev)[None],
                        full_proj_transform=torch.tensor(c["full_proj_transform"], dtype=torch.float32, device=dev)[None],
                        camera_center=torch.tensor(c["camera_center"], dtype=torch.float32, device=dev)[None]) for c in scene["cameras"]]
batches = [dict(bones_posed=Bones(None, None, None, scene["posed"][v]), bones_rest=Bones(None, None, None, scene["rest"])) for v in range(V)]
bg = scene["bg"]
acc = {k: [] for k in ("zero_grad", "hand_forward", "render_gaussians", "loss", "backward")}


def step(k, rec):
    v = k % V
    t0 = time.perf_counter()
    for t in P.values():
        t.grad = None
    t1 = time.perf_counter()
    pred = hand_forward(model, batches[v])
    t2 = time.perf_counter()
    out = render_gaussians(pred.posed_xyz, pred.posed_cov, pred.cano_xyz, pred.cano_features, pred.cano_opacity, cams[v], bg, sh_degree=3,
                           tf=pred.get("tf"), device=dev)
    t3 = time.perf_counter()
    gt = targets_hwc[v]
    loss = 0.8 * losses.l1_loss(out["render"], gt) + 0.2 * (1.0 - losses.ssim(out["render"], gt))
    t4 = time.perf_counter()
    loss.backward()
    t5 = time.perf_counter()
    if rec:
        for name, d in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            acc[name].append(d)


for k in range(16):
    step(k, False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(STEPS):
    step(k, True)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("host issue %.4f ms/step, wall %.4f ms/step (%d steps, auto fences %d)" % (1e3 * t_issue / STEPS, 1e3 * t_all / STEPS, STEPS, rasterizer.context(dev).auto_fenced))
for name, v in acc.items():
    print("%-18s median %7.1f us   mean %7.1f us" % (name, 1e6 * float(np.median(v)), 1e6 * float(np.mean(v))))
