"""How full are the lane groups of k_inst_bwd?  (view, Gaussian) instances with pair records against 8 lanes per active Gaussian."""
import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from manus_amd import rasterizer
from manus_amd.engine import HipViewCompute
from manus_amd.synthetic import camera_table, make_scene
V = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = 300000; W, H = 1920, 1080
dev = torch.device("cuda", 0)
scene = make_scene(n_gaussians=N, kind="hand", seed=0, n_cameras=V, width=W, height=H, device=dev)
ct = camera_table(scene["cameras"], dev)
c = HipViewCompute(scene, torch.rand((V, 3, H, W), device=dev), ct, loss="l1+ssim")
views = list(range(V))
for _ in range(4): c(views, 1.0 / V)
torch.cuda.synchronize()
ws = rasterizer.context(dev).last_ws
off = c._layout(ws, V, N, W, H)
k = 22   # "inst_tag" in the order of mgr_raster_layout (tools/parity.py: layout)
tag = ws.buf[off[k]: off[k] + 4 * V * N].view(torch.int32).cpu().numpy().reshape(V, N)
ep = np.bincount(tag.ravel().astype(np.int64) & 0xFFFF).argmax() if False else tag.max()
has = tag == ep
per_g = has.sum(0)
act = (per_g > 0).sum()
print("views", V, "instances with records", has.sum(), "active Gaussians", act, "lanes", act * V, "useful share %.3f" % (has.sum() / max(1, act * V)))
print("views-with-records histogram over active Gaussians:", np.bincount(per_g)[1:].tolist())
p2 = np.where(per_g > 4, 8, np.where(per_g > 2, 4, np.where(per_g > 1, 2, per_g)))
print("lanes with runs padded to powers of two:", p2.sum(), "share of now %.3f" % (p2.sum() / max(1, act * V)))
