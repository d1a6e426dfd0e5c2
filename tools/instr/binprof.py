import os, sys, ctypes, torch, numpy as np
_R = __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, _R + '/tests')
from manus_amd import rasterizer as rz, _lib
from manus_amd.engine import HipViewCompute
from manus_amd.synthetic import camera_table, make_scene
DEV = "cuda:0"
n, views, W, H = 300000, 8, 1920, 1080
sc = make_scene(n_gaussians=n, kind="hand", seed=0, n_cameras=views, width=W, height=H, device=DEV)
ct = camera_table(sc["cameras"], DEV)
hc = HipViewCompute(sc, torch.zeros((views, 3, H, W), device=DEV), ct, loss="l1")
rz.set_sync_policy(True)
L = ctypes.CDLL(_lib.LIB_PATH)
out = (ctypes.c_ulonglong * (8 * 4096))()
for it in range(3):
    with torch.no_grad():
        hc.forward_views_fused(list(range(views)))
    torch.cuda.synchronize()
    L.mgr_binprof(out)
    a = np.array(list(out), dtype=np.float64).reshape(4096, 8)[:2344]
    a = a[a[:, 5] > 0]
    t0 = a[:, 6].min()
    print("waves", len(a), "ticks (100 MHz) mean per phase [producer: wait for records, scan, expansion; consumer: ranked steps, by-instance]:", a[:, :5].mean(0).round(1).tolist(),
          "loop total", a[:, 5].mean().round(1), "start min/mean/max", (a[:, 6] - t0).min(), (a[:, 6] - t0).mean().round(0), (a[:, 6] - t0).max(),
          "end max", (a[:, 7] - t0).max())
