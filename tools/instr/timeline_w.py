"""-DMGR_TIMELINE build: per-(tile, quadrant) timeline of the wave-granular k_blend_fwd_w (wall_clock64, 10 ns ticks)."""
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
L = ctypes.CDLL(_lib.LIB_PATH)
NREC = 6144 * 4 * 24   # MGR_FWD_GRID workgroups x 4 waves x TLW_PER_WAVE
buf = (ctypes.c_ulonglong * (NREC * 4))(); n = ctypes.c_uint(0)
for _ in range(3): hc(ids, 1.0 / V)
torch.cuda.synchronize(); L.mgr_debug_timeline_w(buf, ctypes.byref(n))
hc(ids, 1.0 / V); torch.cuda.synchronize(); L.mgr_debug_timeline_w(buf, ctypes.byref(n))
a = np.frombuffer(buf, dtype=np.int64).reshape(NREC, 4)
a = a[a[:, 1] > 0]
a = a[a[:, 0] > a[:, 0].max() - 200000]   # the most recent launch only (a wave's rows persist when a later launch gives it fewer units)
hole = (a[:, 3] >> 40) & 1; print("holes", int(hole.sum())); a = a[hole == 0]
t0 = a[:, 0].min(); st = a[:, 0] - t0; en = a[:, 1] - t0; dur = en - st
nl = a[:, 2] & 0xFFFFFFF; quad = (a[:, 2] >> 28) & 3; wg = a[:, 2] >> 32; mx = a[:, 3]
print("units", len(a), "span", en.max(), "ticks (10 ns)")
o = np.argsort(-en)[:16]
print("last-finishing units: start dur nlist consumed ticks/consumed wg")
for i in o: print(int(st[i]), int(dur[i]), int(nl[i]), int(mx[i]), "%.2f" % (dur[i] / max(1, mx[i])), int(wg[i]))
o = np.argsort(-dur)[:16]
print("longest units: start dur nlist consumed ticks/consumed quad")
for i in o: print(int(st[i]), int(dur[i]), int(nl[i]), int(mx[i]), "%.2f" % (dur[i] / max(1, mx[i])), int(quad[i]))
h, _ = np.histogram(en, bins=10, range=(0, en.max())); print("finish histogram", h)
h, _ = np.histogram(st, bins=10, range=(0, en.max())); print("start histogram ", h)
busy = np.zeros(20)
for k in range(20):
    lo, hi = en.max() * k / 20, en.max() * (k + 1) / 20
    busy[k] = (np.minimum(en, hi) - np.maximum(st, lo)).clip(0).sum() / (hi - lo)
print("waves busy per 5% of the span:", busy.astype(int).tolist())
for lo, hi in ((0, 256), (256, 1024), (1024, 4096), (4096, 1 << 30)):
    m = (mx >= lo) & (mx < hi)
    print("consumed %5d..%-6d: %6d units, wave-time %.1f ms-wave, ticks/consumed entry median %.2f, fixed ~%.0f ticks" % (lo, hi, m.sum(), dur[m].sum() / 1e5, np.median(dur[m] / np.maximum(1, mx[m])), np.median(dur[m & (mx < 8)]) if (m & (mx < 8)).any() else -1))
