import sys, ctypes, torch, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from manus_amd import rasterizer as rz, _lib
from manus_amd.engine import HipViewCompute
from manus_amd.synthetic import camera_table, make_scene
DEV='cuda:0'; V=8; N=300000; W,H=1920,1080
sc = make_scene(n_gaussians=N, kind="hand", seed=0, n_cameras=V, width=W, height=H, device=DEV)
ct = camera_table(sc["cameras"], DEV)
hc = HipViewCompute(sc, torch.zeros((V,3,H,W), device=DEV)+0.5, ct, loss="l1+ssim")
ids=list(range(V))
for _ in range(3): hc(ids, 1.0/V)
torch.cuda.synchronize()
L=ctypes.CDLL(_lib.LIB_PATH)
buf=(ctypes.c_ulonglong*(8192*4))()
L.mgr_debug_timeline3(buf)
a=np.array(buf[:],dtype=np.int64).reshape(8192,4)
a=a[a[:,1]>0]
t0=a[:,0].min(); dur=(a[:,1]-a[:,0]); start=a[:,0]-t0; end=a[:,1]-t0
print("items", len(a), "kernel span ticks", end.max(), "(wall_clock64 ticks; 100MHz => 10 ns/tick)")
o=np.argsort(-end)[:25]
print("last-finishing tiles: start, dur, nlist, tmax, ticks/consumed entry")
for i in o: print(int(start[i]), int(dur[i]), int(a[i,2]), int(a[i,3]), "%.3f" % (dur[i]/max(1,a[i,3])))
# fit: dur ~ a + b*tmax for tiles tmax>1000
m=a[:,3]>1000
print("ticks per consumed entry (median, tmax>1000):", np.median(dur[m]/a[m,3]), " for tmax>5000:", np.median(dur[a[:,3]>5000]/a[a[:,3]>5000,3]))
print("ticks per LIST entry for tiles where tmax==nlist (unsaturated):", np.median(dur[(a[:,3]>=a[:,2]-2)&(a[:,2]>3000)]/a[(a[:,3]>=a[:,2]-2)&(a[:,2]>3000),2]))
h,_=np.histogram(end, bins=10, range=(0,end.max())); print("finish-time histogram", h)
b=np.array(buf[:],dtype=np.int64).reshape(8192,4)
idx=np.nonzero(b[:,1]>0)[0]
st=b[idx,0]-t0
late=idx[(b[idx,2]>4096)&(st>5000)]
print("large tiles (nlist>4096) starting late: count", len(late))
for i in late[:15]: print("queue idx", int(i), "start", int(b[i,0]-t0), "dur", int(b[i,1]-b[i,0]), "nlist", int(b[i,2]), "tmax", int(b[i,3]))
print("start times of queue idx 0..20:", [int(b[i,0]-t0) for i in range(20)])
print("nlist of queue idx 1500..1560:", [int(b[i,2]) for i in range(1500,1560,6)])
big=idx[b[idx,2]>4096]; print("queue idx range of nlist>4096:", big.min(), big.max(), len(big))
print("number of tiles with start<2000:", (st<2000).sum())
# per-workgroup view (-DMGR_TIMELINE=2): when each workgroup entered the tile loop, left it, ended; tiles per workgroup
L.mgr_debug_timeline(buf2 := (ctypes.c_ulonglong*(2048*4))())
g=np.array(buf2[:],dtype=np.int64).reshape(2048,4); g=g[g[:,2]>0]
print("workgroups", len(g), "enter loop (min/median/max ticks)", int((g[:,0]-t0).min()), int(np.median(g[:,0]-t0)), int((g[:,0]-t0).max()),
      "leave loop", int((g[:,1]-t0).min()), int(np.median(g[:,1]-t0)), int((g[:,1]-t0).max()), "end", int((g[:,2]-t0).max()),
      "tiles per workgroup min/median/max", int(g[:,3].min()), int(np.median(g[:,3])), int(g[:,3].max()))
h,_=np.histogram(g[:,0]-t0, bins=10, range=(0, (g[:,2]-t0).max())); print("workgroup entry histogram", h)
h,_=np.histogram(g[:,1]-t0, bins=10, range=(0, (g[:,2]-t0).max())); print("workgroup loop-exit histogram", h)
h,_=np.histogram(start, bins=10, range=(0,end.max())); print("tile start histogram", h)
