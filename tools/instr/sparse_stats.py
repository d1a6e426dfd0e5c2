"""-DMGR_STATS build: how sparse are the blend kernels' steps?  Histograms over the number of pixels of a wave's 8x8
quadrant that are still active in the forward blend, weighted by batches and pair steps, and over the number of pixels for
which an evaluated list entry is valid."""
import sys, os, ctypes, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from manus_amd import _lib
from manus_amd.engine import HipViewCompute
from manus_amd.synthetic import camera_table, make_scene
DEV = 'cuda:0'; V = int(os.environ.get("V", 8)); N = 300000; W, H = 1920, 1080
sc = make_scene(n_gaussians=N, kind="hand", seed=0, n_cameras=V, width=W, height=H, device=DEV)
ct = camera_table(sc["cameras"], DEV)
hc = HipViewCompute(sc, torch.rand((V, 3, H, W), device=DEV) * 0 + 0.5, ct, loss="l1+ssim")
ids = list(range(V))
dll = ctypes.CDLL(_lib.LIB_PATH)
def grab(fn, rows):
    z = (ctypes.c_ulonglong * (rows * 65))()
    fn(z)
    return np.array(list(z), dtype=np.int64).reshape(rows, 65)
hc(ids, 1.0 / V); torch.cuda.synchronize()
f0 = grab(dll.mgr_debug_fhist, 5)
hc(ids, 1.0 / V); torch.cuda.synchronize()
f = grab(dll.mgr_debug_fhist, 5) - f0
def show(name, h):
    tot = h.sum()
    cum = np.cumsum(h) / max(1, tot)
    print(name, "total", tot)
    print("   bins 0..8:", h[:9].tolist(), " cum share at <=1,2,4,8,16,32: ", ["%.3f" % cum[k] for k in (1, 2, 4, 8, 16, 32)])
show("fwd batches by active pixels        ", f[0])
show("fwd pair steps by active pixels     ", f[1])
show("fwd pair steps (pos>=2048) by active", f[2])
show("fwd evaluated entries by valid px   ", f[3])
show("fwd survivors by active pixels      ", f[4])
np.save("gpurun_out/sparse_fhist.npy", f)
