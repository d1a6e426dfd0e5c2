#!/usr/bin/env python3
"""Benchmark of the MANUS hot path on MI355X: training iterations per second, one iteration =
V views (forward + image loss (0.8 L1 + 0.2 (1-SSIM), or L1 with --loss l1) + backward to the six leaf-parameter gradients, + gradient
all-reduce when N > 1), 300k articulated Gaussians, 1920x1080, 8 views, synthetic data.

    python bench.py [--gpus N] [--steps K] [--warmup W]

For N > 1 the driver launches it under torch.distributed.run (one rank per GPU, RCCL); the
8 views are sharded across the ranks (strong scaling).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)
DOMINANT = "k_blend_bwd"
PMC_FILE = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")  # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes


def measured_traffic(kernel, N, V, W, H):
    """HBM bytes per launch of `kernel` from the committed PMC passes (same workload only)."""
    try:
        d = json.load(open(PMC_FILE))
        if d.get("workload") != [N, V, W, H]:
            return None
        return d["kernels"][kernel]["hbm_bytes_per_launch"]
    except Exception:
        return None


def algorithmic_bytes(N, V, R, P, n_poses):
    """B_iter = n_poses*1496*N + V*(936*N + 188*R + 40*P); R, P per view."""
    return n_poses * 1496 * N + V * (936 * N + 188 * R + 40 * P)


def kernel_algorithmic_bytes(name, N, V, R, P):
    """Per-launch algorithmic bytes of one kernel (all V views are in one launch), from the
    per-unit figures of SURVEY.md 8(d)."""
    per = {
        "k_blend_bwd": V * (P * 20 + R * 112),      # K7: pixel state + per pair gather 40 + 9-float RMW 72
        "k_blend_fwd": V * (R * 40 + P * 20),       # K6
        "k_tile_sort": V * R * 24,                  # K4-K5: one read + write of key + payload
        "k_emit": V * (N * 8 + R * 12),             # K2-K3
        "k_preprocess": V * N * 76,                 # K1
        "k_preprocess_bwd": V * N * 100,            # K8-K9
    }
    return per.get(name)


def cpu_baseline(scene_cpu, cams, sample_views=4, n_views=8, loss="l1+ssim"):
    """Oracle ("port") timed on the host cores on `sample_views` of the workload's views (each with its own
    pose): the torch restatement of LBS/cov/SH forward+backward and of the image loss (all cores) and the scalar
    C rasterizer oracle forward+backward (one core), scaled to iterations/s for n_views views."""
    from oracle import RasterOracle
    from oracle import torch_ref as tr
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count()
    threads = max(1, min(avail, 32))   # more threads only add scheduling overhead to these elementwise ops
    torch.set_num_threads(threads)
    K = max(1, min(sample_views, len(cams)))
    tt = np.zeros(5)   # lbs+sh fwd, raster fwd, loss, raster bwd, lbs+sh bwd
    num_rendered = 0
    for k in range(K):
        cam0 = cams[k]
        P = {n: v.clone().requires_grad_(True) for n, v in scene_cpu["params"].items()}
        cc = torch.tensor(np.asarray(cam0["camera_center"], np.float32))
        t0 = time.time()
        o = tr.hand_forward(P, scene_cpu["grid"], scene_cpu["grid_center"], scene_cpu["grid_scale"],
                            scene_cpu["posed"][k], scene_cpu["rest"], cc)
        t1 = time.time()
        ro = RasterOracle(cam0["width"], cam0["height"], math.tan(cam0["fovx"] / 2), math.tan(cam0["fovy"] / 2),
                          np.asarray(cam0["world_view_transform"], np.float32).reshape(-1),
                          np.asarray(cam0["full_proj_transform"], np.float32).reshape(-1),
                          o["posed_xyz"].detach().numpy(), o["posed_cov"].detach().numpy(),
                          o["colors"].detach().numpy(), o["opacity"].detach().numpy()[:, 0], np.ones(3, np.float32))
        t2 = time.time()
        if loss == "l1+ssim":   # image loss of the step on the oracle's image (torch restatement, all threads)
            img = torch.tensor(np.ascontiguousarray(ro.color)).permute(1, 2, 0).clone().requires_grad_(True)
            tgt = torch.full_like(img, 0.5)
            (gi,) = torch.autograd.grad(tr.rgb_ssim_loss(img, tgt), img)
            g = np.ascontiguousarray(gi.permute(2, 0, 1).numpy())
        else:
            g = np.sign(ro.color - 0.5).astype(np.float32) / ro.color.size
        t2b = time.time()
        b = ro.backward(g)
        t3 = time.time()
        chain = ((o["posed_xyz"] * torch.tensor(b["means3D"])).sum() + (o["posed_cov"] * torch.tensor(b["cov3D"])).sum()
                 + (o["colors"] * torch.tensor(b["colors"])).sum() + (o["opacity"][:, 0] * torch.tensor(b["opacity"])).sum())
        chain.backward()
        t4 = time.time()
        tt += np.array([t1 - t0, t2 - t1, t2b - t2, t3 - t2b, t4 - t3])
        num_rendered += int(ro.num_rendered)
    tt /= K
    t_view = float(tt.sum())
    return {"value": 1.0 / (t_view * n_views), "unit": "iters/s", "cores": threads, "kind": "port",
            "sample": "%d of %d views (%.1f s of CPU work), N=%d, 1920x1080, per view: torch LBS+cov+SH fwd %.2fs + bwd %.2fs "
                      "(%d threads), scalar C rasterizer fwd %.2fs + bwd %.2fs (1 thread), image loss %s %.2fs; "
                      "value = 1/(%d x %.2fs)"
                      % (K, n_views, t_view * K, scene_cpu["params"]["_xyz"].shape[0], tt[0], tt[4], threads, tt[1], tt[3],
                         loss, tt[2], n_views, t_view),
            "num_rendered": num_rendered // K}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gaussians", type=int, default=300000)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--kind", default="hand")
    ap.add_argument("--loss", default="l1+ssim", choices=["l1", "l1+ssim"],
                    help="image loss of the step: 0.8 L1 + 0.2 (1 - SSIM) (HAND_GAUSSIAN.yaml:22-23) or L1 alone")
    ap.add_argument("--optimizer", action="store_true",
                    help="also run the fused Adam step inside every timed step (outside the headline metric, "
                         "which SURVEY 8d defines without the optimizer)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-all", action="store_true", help="print a per-kernel HIP-event breakdown to stderr")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm; MANUS_BENCH_BACKEND=gloo lets the N>1 path be exercised on a
        # single-GPU box (all ranks then share device 0)
        dist.init_process_group(os.environ.get("MANUS_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
    local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from manus_amd import _lib, rasterizer
    from manus_amd.engine import HipViewCompute, ViewShardedStep
    from manus_amd.synthetic import camera_table, make_scene

    V, N, W, H = args.views, args.gaussians, args.width, args.height
    scene = make_scene(n_gaussians=N, kind=args.kind, seed=0, n_cameras=V, width=W, height=H, device=dev)
    ct = camera_table(scene["cameras"], dev)

    # target images: the same scene with parameters perturbed by 1 % (non-trivial dL/dimage)
    g = torch.Generator(device="cpu").manual_seed(123)
    pert = dict(scene)
    pert["params"] = {k: (v + 0.01 * v.abs().mean() * torch.randn(v.shape, generator=g).to(dev))
                      for k, v in scene["params"].items()}
    with torch.no_grad():
        hp = HipViewCompute(pert, torch.zeros((V, 3, H, W), device=dev), ct)
        targets = torch.cat([hp.forward_views([v])[0] for v in range(V)]).contiguous()
        del hp
    compute = HipViewCompute(scene, targets, ct, loss=args.loss)
    shapes = {k: v.shape for k, v in compute.params.items()}
    step = ViewShardedStep(N, shapes, compute, V, rank=rank, world_size=world)
    V_local = len(step.local_views)
    opt = None
    if args.optimizer:
        from manus_amd.optim import GaussianOptimizer
        opt = GaussianOptimizer(compute.params, adopt=True)
        base_step = step.step

        def step_with_adam():
            o = base_step()
            opt.update_learning_rate(opt.state_step + 1)
            opt.step(o["grads"])
            return o

        step.step = step_with_adam

    for _ in range(max(1, args.warmup)):     # also learns the pair capacity (with host syncs)
        out = step.step()
    npairs_local = rasterizer.check_overflow()
    rasterizer.set_sync_policy(False)         # timed region: no host synchronisation
    out = step.step()
    rasterizer.check_overflow()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    _lib.profile_enable(True)                 # HIP events around every library kernel, on its stream
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step.step()
    barrier()
    dt = time.perf_counter() - t0
    prof = _lib.profile_report()
    _lib.profile_enable(False)
    rasterizer.check_overflow()
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # With the optimizer in the loop a Gaussian can walk out of the skin-weight grid; its weights are then 0/0 = NaN
    # exactly like the reference (gaussian_utils.py:193-195), which drops such rows when it loads a checkpoint.
    nonfinite = sum(int((~torch.isfinite(x)).sum()) for x in out["grads"].values())
    assert args.optimizer or nonfinite == 0, "non-finite gradients"

    if rank == 0:
        R_view = npairs_local / max(1, V_local)          # measured pairs per view (num_rendered)
        P_px = W * H
        n_poses = V if args.kind == "hand" else 0
        b_iter = algorithmic_bytes(N, V, R_view, P_px, n_poses)
        dom = prof.get(DOMINANT)
        roof = None
        if dom:
            avg_ms = dom[1] / dom[0]
            kb = kernel_algorithmic_bytes(DOMINANT, N, V_local, R_view, P_px)
            ach = kb / (avg_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": DOMINANT, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                    "traffic": measured_traffic(DOMINANT, N, V_local, W, H) if world == 1 else None,
                    "avg_kernel_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": int(kb),
                    "iter_algorithmic_GBps": round(b_iter * args.steps / dt / 1e9, 2)}
        if args.profile_all:
            tot = sum(v[1] for v in prof.values())
            for k, (c, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
                print("%-20s %6d launches %9.3f ms avg %8.4f ms  %5.1f%%" % (k, c, ms, ms / c, 100 * ms / tot),
                      file=sys.stderr)
            print("library kernels %.3f ms/iter of %.3f ms/iter wall" % (tot / args.steps, 1e3 * dt / args.steps),
                  file=sys.stderr)
        cpu = None
        if not args.no_cpu_baseline and world == 1 and args.kind == "hand":
            sc_cpu = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in scene.items() if k != "params"}
            sc_cpu["params"] = {k: v.detach().cpu() for k, v in scene["params"].items()}
            cpu = cpu_baseline(sc_cpu, scene["cameras"], n_views=V, loss=args.loss)
        line = {
            "metric": "train iters/sec (fwd+bwd) 300k Gaussians @1080p, 8 views; PSNR parity",
            "value": round(args.steps / dt, 4), "unit": "iters/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "HAND_GAUSSIAN: %d Gaussians, 21-transform LBS, %d views %dx%d, one pose per view "
                                   "(n_poses=%d), image loss %s, fwd+bwd to leaf grads" % (N, V, W, H, n_poses, "0.8*L1 + 0.2*(1-SSIM)" if args.loss == "l1+ssim" else "L1"),
                       "gaussians": N, "views": V, "width": W, "height": H, "views_per_gpu": V_local,
                       "pairs_per_view": int(R_view), "parallelism": "views/%d" % world,
                       "optimizer_in_step": bool(args.optimizer), "nonfinite_grad_values": nonfinite},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
