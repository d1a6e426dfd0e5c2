"""Bounding box (in 16 x 16 tiles) of the non-background tiles of every view of the bench scene: what sizes k_bin_scatter's LDS."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from manus_amd.engine import HipViewCompute
from manus_amd.synthetic import camera_table, make_scene
DEV = "cuda:0"; V = int(os.environ.get("V", 8)); N = int(os.environ.get("N", 300000)); W, H = 1920, 1080
kind = os.environ.get("KIND", "hand")
sc = make_scene(n_gaussians=N, kind=kind, seed=0, n_cameras=V, width=W, height=H, device=DEV)
hc = HipViewCompute(sc, torch.zeros((V, 3, H, W), device=DEV), camera_table(sc["cameras"], DEV))
img = hc.forward_views_fused(list(range(V)))[0]
for v in range(V):
    m = (img[v] < 0.9999).any(0)
    pad = torch.zeros((1088, 1920), dtype=torch.bool, device=DEV); pad[:H] = m
    t = pad.view(68, 16, 120, 16).any(3).any(1)
    ys, xs = t.any(1).nonzero(), t.any(0).nonzero()
    bw, bh = int(xs.max() - xs.min() + 1), int(ys.max() - ys.min() + 1)
    print("view %d: %d non-empty tiles, box %d x %d = %d tiles" % (v, int(t.sum()), bw, bh, bw * bh))
