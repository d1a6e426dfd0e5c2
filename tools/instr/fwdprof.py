"""-DFWD_PROF build: where the waves of k_blend_fwd spend their time (wall_clock64 per wave and phase)."""
import sys, os, ctypes, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from manus_amd import _lib
from manus_amd.engine import HipViewCompute
from manus_amd.synthetic import camera_table, make_scene
DEV = 'cuda:0'; V = int(os.environ.get("V", 8)); N = 300000; W, H = 1920, 1080
sc = make_scene(n_gaussians=N, kind="hand", seed=0, n_cameras=V, width=W, height=H, device=DEV)
ct = camera_table(sc["cameras"], DEV)
hc = HipViewCompute(sc, torch.zeros((V, 3, H, W), device=DEV) + 0.5, ct, loss="l1+ssim")
ids = list(range(V))
dll = ctypes.CDLL(_lib.LIB_PATH)
def grab():
    z = (ctypes.c_ulonglong * 8)(); dll.mgr_debug_fprof(z); return np.array(list(z), dtype=np.int64)
for _ in range(3): hc(ids, 1.0 / V)
torch.cuda.synchronize(); a = grab()
hc(ids, 1.0 / V); torch.cuda.synchronize(); d = grab() - a
names = ["ticket + tile prologue", "box test / compaction", "pair loop", "checkpoint + loop", "wait for other quadrants", "epilogue"]
tot = d[:6].sum()
print("waves", d[7], "tiles (per wave sum)", d[6], "-> tiles", d[6] // 4, "; wave time in the tile loop %.1f us mean" % (tot / d[7] / 100.0))
for k in range(6):
    print("%-26s %6.1f %%   %.2f us per tile" % (names[k], 100.0 * d[k] / tot, d[k] / max(1, d[6]) / 100.0))
