"""engine.Trainer on the bench workload (300 k hand, 8 views of 1080p): optimisation steps per second of the product's own
training step -- render + loss + backward, fence poll, on_after_backward bookkeeping, learning-rate schedule, fused Adam --
next to `bench.py --optimizer` (the same kernels without the Trainer's host logic).  Density control is parked (its steps
are every-100-steps events): python tools/time_trainer.py [steps] [--depth-cut | --no-depth-cut]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manus_amd.engine import HipViewCompute, Trainer
from manus_amd.synthetic import camera_table, make_scene

args = [a for a in sys.argv[1:] if not a.startswith("--")]
steps = int(args[0]) if args else 200
cut = True if "--depth-cut" in sys.argv else False if "--no-depth-cut" in sys.argv else None      # None: the Trainer's default
dev = torch.device("cuda", 0)
V, N, W, H = 8, 300000, 1920, 1080
scene = make_scene(n_gaussians=N, kind="hand", seed=0, n_cameras=V, width=W, height=H, device=dev)
ct = camera_table(scene["cameras"], dev)
g = torch.Generator(device="cpu").manual_seed(123)
pert = dict(scene)
pert["params"] = {k: (v + 0.01 * v.abs().mean() * torch.randn(v.shape, generator=g).to(dev)) for k, v in scene["params"].items()}
with torch.no_grad():
    targets = HipViewCompute(pert, torch.zeros((V, 3, H, W), device=dev), ct).forward_views_fused(list(range(V)))[0].contiguous()
compute = HipViewCompute(scene, targets, ct, loss="l1+ssim")
kw = {} if cut is None else {"depth_cut": cut}
tr = Trainer(compute, V, extent=0.3, opts=dict(remove_seg_end=0, densify_from_step=10 ** 9, opacity_reset_interval=10 ** 9), **kw)
tr.global_step = 1          # (step 0 and every 100th run the keypoint test of the hand module: an every-100-steps event)
for _ in range(10):
    tr.train_step()
torch.cuda.synchronize()
ts, issue = [], []
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(steps):
        if tr.global_step % 100 == 0:
            tr.global_step += 1
        out = tr.train_step()
    issue.append((time.perf_counter() - t0) / steps)      # (the host is done issuing; the rest is the device's backlog)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / steps)
ms = sorted(ts)[1] * 1e3
print("host issue time %.4f ms / step (median of 3)" % (sorted(issue)[1] * 1e3))
from manus_amd import rasterizer
ctx = rasterizer.context(dev)
print("Trainer.train_step: %.4f ms / step = %.1f steps/s (median of 3 x %d steps; depth cut %s; re-run steps %d, flagged forwards %d, quadrants repaired on the device %d; loss %.6f)"
      % (ms, 1e3 / ms, steps, "on" if compute.depth_cut else "off", tr.retries, ctx.cut_retries, ctx.cut_repairs, float(out["loss"])))
