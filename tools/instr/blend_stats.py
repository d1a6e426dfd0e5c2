import sys, ctypes, torch, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from manus_amd import rasterizer as rz, _lib
from manus_amd.engine import HipViewCompute
from manus_amd.synthetic import camera_table, make_scene
DEV='cuda:0'; V=8; N=300000; W,H=1920,1080
sc = make_scene(n_gaussians=N, kind="hand", seed=0, n_cameras=V, width=W, height=H, device=DEV)
ct = camera_table(sc["cameras"], DEV)
hc = HipViewCompute(sc, torch.rand((V,3,H,W), device=DEV)*0+0.5, ct, loss="l1+ssim")
L=_lib.lib()
ids=list(range(V))
hc(ids, 1.0/V); torch.cuda.synchronize()
z=(ctypes.c_ulonglong*16)()
fn=ctypes.CDLL(_lib.LIB_PATH).mgr_debug_stats
fn(z); base=list(z)
hc(ids, 1.0/V); torch.cuda.synchronize()
fn(z); d=[a-b for a,b in zip(z,base)]
print("box tests", d[0], "survivors", d[1], "pair iters", d[2], "valid lane evals", d[3], "entries w/ any", d[4], "full pair iters", d[5])
print("survivor rate %.3f; lane utilisation over executed pair iters: %.3f; entries with a valid pixel %.3f of the evaluated; valid pixels per such entry %.1f" % (d[1]/d[0], d[3]/(d[2]*128), d[4]/(2*d[2]), d[3]/d[4]))
if d[6]:
    print("(entry, 4x4 block) combinations with a valid pixel %d: lane utilisation at block granularity %.3f" % (d[6], d[3] / (d[6] * 16)))
if d[9]:
    print("active pixels (last > first) of the visited quadrants, weighted by their pair steps: %.3f of the lanes; valid / active %.3f" % (d[8] / (d[9] * 64), d[3] / (2 * d[8])))
    print("pair steps now %d; with one entry list per 4x4 block (validity-exact lists): max over the blocks %d (%.3f), perfectly balanced %d (%.3f)"
          % (d[9], d[10], d[10] / d[9], d[11], d[11] / d[9]))
# tile statistics
ws = rz.context().last_ws
arr=(ctypes.c_size_t*32)(); L.mgr_raster_layout(V,N,W,H,ws.cap,arr,32)
T=120*68; VT=V*T
ts = ws.buf[int(arr[7]):int(arr[7])+4*(VT+1)].view(torch.int32).cpu().numpy().astype(np.int64)
done = ws.buf[int(arr[9]):int(arr[9])+4*VT].view(torch.int32).cpu().numpy().astype(np.int64)
n = np.diff(ts)
print("keys", n.sum(), "consumed", done.sum(), "tiles nonempty", (n>0).sum())
for lo,hi in ((1,64),(64,512),(512,2048),(2048,4096),(4096,16384),(16384,1<<30)):
    m=(n>=lo)&(n<hi)
    print("tiles %6d..%-8d: %6d tiles, keys %9d (%.1f%%), consumed %9d (%.1f%% of their keys)" % (lo,hi,m.sum(), n[m].sum(), 100*n[m].sum()/n.sum(), done[m].sum(), 100*done[m].sum()/max(1,n[m].sum())))
big = n >= 2048
for thr in (512, 1024, 2048, 4096):
    print("tiles>=2048 with done <= %d: %.3f of tiles, holding %.3f of their keys" % (thr, (done[big] <= thr).mean(), n[big][done[big] <= thr].sum()/n[big].sum()))
for fr in (0.125, 0.25, 0.5):
    ok = done[big] <= fr*n[big]
    print("tiles>=2048 with done <= %.3f n: %.3f of tiles, %.3f of keys" % (fr, ok.mean(), n[big][ok].sum()/n[big].sum()))
# pixel-level: n_contrib vs tile_done is max; check how many pixels unsaturated in big tiles
nc = ws.buf[int(arr[17]):int(arr[17])+4*V*W*H].view(torch.int32).reshape(V,H,W)
img = hc.last_image
# a pixel is 'saturated' if rendering stopped early: approximate via final colour not containing bg... skip
sp = ws.buf[int(arr[16]):int(arr[16])+4*V*W*H].view(torch.int32).reshape(V,H,W)
# per tile: all pixels saturated? and the max stop position
sp_t = sp[:, :1072].reshape(V, 67, 16, 120, 16).permute(0,1,3,2,4).reshape(V, 67*120, 256)
allsat = (sp_t > 0).all(-1).cpu().numpy()
maxstop = sp_t.max(-1).values.cpu().numpy()
nn = n.reshape(V, 68*120)[:, :67*120]; dd = done.reshape(V,68*120)[:, :67*120]
bigm = nn >= 2048
print("big tiles: %d, all-saturated: %.3f; keys in all-saturated big tiles: %.3f" % (bigm.sum(), allsat[bigm].mean(), nn[bigm & allsat].sum()/nn[bigm].sum()))
for K in (1024, 2048, 4096):
    ok = bigm & allsat & (maxstop <= K)
    print("K=%d: big tiles finished within K: %.3f of big tiles, %.3f of big-tile keys" % (K, ok.sum()/bigm.sum(), nn[ok].sum()/nn[bigm].sum()))
for fr in (0.25, 0.5):
    ok = bigm & allsat & (maxstop <= fr*nn)
    print("K=%.2f n: finished: %.3f of big tiles, %.3f of keys" % (fr, ok.sum()/bigm.sum(), nn[ok].sum()/nn[bigm].sum()))
uns = bigm & ~allsat
cnt_uns = (sp_t == 0).sum(-1).cpu().numpy()
print("unsaturated big tiles: %d; median unsaturated pixels per such tile: %s; keys there %.3f" % (uns.sum(), np.median(cnt_uns[uns]) if uns.any() else None, nn[uns].sum()/nn[bigm].sum()))
