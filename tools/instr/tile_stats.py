import sys, ctypes, torch, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from manus_amd import rasterizer as rz, _lib
from manus_amd.engine import HipViewCompute
from manus_amd.synthetic import camera_table, make_scene
DEV='cuda:0'; V=8; N=300000; W,H=1920,1080
sc = make_scene(n_gaussians=N, kind="hand", seed=0, n_cameras=V, width=W, height=H, device=DEV)
ct = camera_table(sc["cameras"], DEV)
hc = HipViewCompute(sc, torch.rand((V,3,H,W), device=DEV)*0+0.5, ct, loss="l1+ssim")
L=_lib.lib(); ids=list(range(V))
hc(ids, 1.0/V); torch.cuda.synchronize()
ws = rz.context().last_ws
arr=(ctypes.c_size_t*32)(); L.mgr_raster_layout(V,N,W,H,ws.cap,arr,32)
T=120*68; VT=V*T
ts = ws.buf[int(arr[7]):int(arr[7])+4*(VT+1)].view(torch.int32).cpu().numpy().astype(np.int64)
done = ws.buf[int(arr[9]):int(arr[9])+4*VT].view(torch.int32).cpu().numpy().astype(np.int64)
n = np.diff(ts)
o = np.argsort(-done)[:40]
print("top tiles by consumed depth: (n, done)", [(int(n[i]), int(done[i])) for i in o])
for thr in (3000, 4000, 5000, 6000, 8000):
    m = done > thr
    print("done > %d: %d tiles; of them n>=16384: %d, n>=8192: %d" % (thr, m.sum(), (m & (n>=16384)).sum(), (m & (n>=8192)).sum()))
print("n>=16384:", (n>=16384).sum(), "n>=8192:", (n>=8192).sum(), "sum done for n>=8192:", done[n>=8192].sum(), "total", done.sum())
