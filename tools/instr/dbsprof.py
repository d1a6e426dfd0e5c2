"""-DDBS_PROF build: where the workgroups of k_dbin_sort spend their time (wall_clock64, 10 ns ticks, thread 0 per item)."""
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
    z = (ctypes.c_ulonglong * 16)(); dll.mgr_debug_dbsprof(z); return np.array(list(z), dtype=np.int64)
for _ in range(4): hc(ids, 1.0 / V)
torch.cuda.synchronize(); a = grab()
hc(ids, 1.0 / V); torch.cuda.synchronize(); b = grab(); d = b - a
print("views", V, "items", d[2], "keys", d[3], "parts", d[4], "keys/item %.0f" % (d[3] / max(1, d[2])))
print("mean prologue %.2f us, mean sorts %.2f us per item" % (d[0] / max(1, d[2]) / 100.0, d[1] / max(1, d[2]) / 100.0))
print("max over all steps so far: slowest item %.2f us, largest item %d keys, latest end after workgroup start %.2f us" % (b[5] / 100.0, b[6], b[7] / 100.0))
print("items by size <=1024 / <=2048 / <=3072 / <=4096 / more:", d[8:13])
