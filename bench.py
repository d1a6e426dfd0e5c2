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
PMC_FILE = os.path.join(ROOT, "profiles", "r06_pmc.json")  # rocprofv3 --pmc passes of this same command (tools/measure_round.sh)
REPEATS = 5   # the --steps loop is timed this many times; the median goes into the line, min / max beside it


def pmc_counters(kernel, N, V, W, H):
    """Per-launch counter means of `kernel` from the committed PMC passes (same workload only): HBM bytes
    ((2*FETCH_SIZE + WRITE_SIZE) KB, the guide's gfx950 FETCH_SIZE correction) and the VALU issue fraction
    4 * SQ_INSTS_VALU / (1024 SIMDs * cycles): a wave64 VALU instruction occupies its SIMD for 4 cycles (transcendentals
    longer: a lower bound); cycles = the dispatch's duration in the same pass x 2.4 GHz (the peak engine clock; profiled
    passes run slower, MI355X_MICROARCH.md: again a lower bound); and the share of cycles the CUs' LDS arrays were busy
    (SQ_LDS_IDX_ACTIVE)."""
    try:
        d = json.load(open(PMC_FILE))
        if d.get("workload") != [N, V, W, H]:
            return {}
        k = d["kernels"][kernel]
        out = {"traffic": k.get("hbm_bytes_per_launch"), "duration_ns": k.get("duration_ns"),
               "step_hbm_bytes": d.get("step_hbm_bytes"), "step_kernel_ns": d.get("step_kernel_ns")}
        if "SQ_INSTS_VALU" in k and k.get("duration_ns"):
            out["valu_issue_frac"] = round(4.0 * k["SQ_INSTS_VALU"] / (1024.0 * k["duration_ns"] * 2.4), 4)
        if "SQ_LDS_IDX_ACTIVE" in k and k.get("duration_ns"):
            # LDS-array cycles summed over the 256 CUs / (256 x cycles of the dispatch)
            out["lds_active_frac"] = round(k["SQ_LDS_IDX_ACTIVE"] / (256.0 * k["duration_ns"] * 2.4), 4)
        return out
    except Exception:
        return {}


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


def kernel_unit_bytes(name, V, consumed, P):
    """Per-launch bytes of the units the launch PROCESSES (what `roofline.frac` prices since round 5).  The blend kernels
    work on the list entries the forward's walks consumed (sum over tiles of the deepest contributor), not on the
    rectangle pairs of SURVEY 8(d) -- exact null-pair culling and early termination leave most of those unread:
      k_blend_bwd   112 B per consumed entry (gather 40 + 9-float record 72) + 20 B per pixel
      k_blend_fwd    40 B per consumed entry                                 + 20 B per pixel
    Other kernels: None (their units are the survey's)."""
    if consumed is None:
        return None
    per = {"k_blend_bwd": 112 * consumed + 20 * P * V, "k_blend_fwd": 40 * consumed + 20 * P * V}
    return per.get(name)


# DESIGN 7's model of the N-GPU step (hand 300 k, 8 views of 1080p sharded): compute of a rank by its number of views,
# measured on one GPU (profiles/r05_other_configs/hand_v{1,2,4}.json and the headline), + the dense exchange of
# 61 N + 2 floats between "every peer link at once" (direct reduce-scatter + all-gather: 2 (S / n) / 76.8 GB/s) and "a ring
# over one link at a time" ((n - 1) times that).
COMPUTE_MS_BY_VIEWS = {8: 1.34, 4: 0.87, 2: 0.70, 1: 0.54}
# ... and of BASELINE config 4 (COMPOSITE, 500 k Gaussians, 53 cameras over 8 ranks: seven views on the fullest rank),
# profiles/r05_other_configs/composite_500k_v7.json; six views interpolated between that and the hand's per-view slope
COMPOSITE_MS_BY_VIEWS = {7: 1.81, 6: 1.62}


def predicted_ms(world, N, V, kind, W, H):
    """One-GPU compute of the fullest rank + the dense exchange between "every peer link at once" and "a ring over one link"
    (DESIGN 8).  Only for the two multi-GPU configurations of BASELINE.json measured on one GPU: cfg3 (hand, 300 k, 8 views) and
    cfg4 (composite, 500 k, 53 cameras)."""
    if world <= 1 or (W, H) != (1920, 1080):
        return None
    per_rank = -(-V // world)          # views on the fullest rank
    if kind == "hand" and V == 8 and N == 300000 and V % world == 0:
        comp = COMPUTE_MS_BY_VIEWS.get(per_rank)
    elif kind == "composite" and V == 53 and N == 500000 and world == 8:
        comp = COMPOSITE_MS_BY_VIEWS.get(per_rank)
    else:
        comp = None
    if comp is None:
        return None
    S = (61 * N + 2) * 4
    direct = 2.0 * (S / world) / 76.8e9 * 1e3
    ring = direct * (world - 1)
    return {"compute_ms": comp, "views_on_the_fullest_rank": per_rank, "exchange_bytes_dense": S,
            "exchange_dense_ms_direct": round(direct, 3), "exchange_dense_ms_ring": round(ring, 3),
            "step_ms_low": round(comp + direct, 3), "step_ms_high": round(comp + ring, 3),
            "model": "DESIGN 8: one-GPU compute of the fullest rank's views + 61N+2 floats over 7 xGMI links x 76.8 GB/s per direction"}


def cpu_model_string():
    """The host CPU's model name (BASELINE.md 3 asks for it next to os.cpu_count())."""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def _median_time(fn, warmup=2, reps=5):
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def torch_chain_timing(scene_cpu, cam0, sizes=(10000, 300000)):
    """SURVEY.md 8(d) "CPU baseline": the pure-PyTorch restatement of a1-a5 + a11 (einsum LBS, grid_sample,
    linalg.inv, eval_sh, project_points: oracle/torch_ref.py, pinned to the reference by tests/golden) on the host
    cores, torch.set_num_threads(os.cpu_count()), 2 warm-ups, median of 5; forward and forward+backward; ms and
    Gaussians/s; N = 10k (BASELINE config 1) and the bench size."""
    from oracle import torch_ref as tr
    cc = torch.tensor(np.asarray(cam0["camera_center"], np.float32))
    K = torch.tensor(np.asarray(cam0["K"], np.float32))
    E = torch.tensor(np.asarray(cam0["extr"], np.float32))[:3, :4]
    out = {}
    n_all = scene_cpu["params"]["_xyz"].shape[0]
    for n in sizes:
        n = min(n, n_all)
        P = {k: v[:n].clone().requires_grad_(True) for k, v in scene_cpu["params"].items()}

        def fwd():
            o = tr.hand_forward(P, scene_cpu["grid"], scene_cpu["grid_center"], scene_cpu["grid_scale"],
                                scene_cpu["posed"][0], scene_cpu["rest"], cc)
            o["uv"] = tr.project_points(o["posed_xyz"][None], K, E)
            return o

        def fwd_bwd():
            o = fwd()
            (o["posed_xyz"].sum() + o["posed_cov"].sum() + o["colors"].sum() + o["opacity"].sum()).backward()
            for v in P.values():
                v.grad = None

        with torch.no_grad():
            t_f = _median_time(fwd)
        t_fb = _median_time(fwd_bwd)
        out[str(n)] = {"fwd_ms": round(1e3 * t_f, 2), "fwd_bwd_ms": round(1e3 * t_fb, 2),
                       "gaussians_per_s_fwd": round(n / t_f), "gaussians_per_s_fwd_bwd": round(n / t_fb)}
    return out


def cpu_baseline_and_parity(scene_cpu, cams, targets_cpu, gpu, sample_views=2, n_views=8, loss="l1+ssim"):
    """The oracle ("port") on the host cores for `sample_views` of the workload's views with the step's own targets
    and loss, which gives (i) the reported CPU baseline and (ii) the parity of the benchmarked GPU step:

      end to end       torch LBS/cov/SH chain -> scalar C rasterizer -> image loss -> backward, independent of the GPU;
                       compared with the GPU step's images and its leaf gradients for the same views
      identical inputs the GPU kernels' own per-(view, Gaussian) records (pixel centre, conic, opacity, colour, depth,
                       radius) blended by the scalar oracle forward + backward, the blend sums pushed through the torch
                       chain: the blend decisions (alpha threshold, early stop) are taken on bit-identical inputs

    gpu: dict with img (K,3,H,W), g_img (K,3,H,W) = dL/dimage the GPU step used, grads {leaf: tensor} of the step over
    the sampled views, grec (K,N,12), depth (K,N), radii (K,N) as numpy / cpu tensors."""
    from oracle import BlendOracle, RasterOracle
    from oracle import torch_ref as tr
    from tools.parity import align_threshold_decisions, kernel_last_gaussian   # (the -m gpu tests assert the same comparison)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count()
    # SURVEY 8(d) asks for set_num_threads(os.cpu_count()); on the 256-thread GPU host that makes these elementwise
    # ops 200x SLOWER (measured: 22 s instead of 0.1 s per forward), so the thread count is capped at 32 and stated
    threads = max(1, min(avail, os.cpu_count() or 1, 32))
    torch.set_num_threads(threads)
    K = max(1, min(sample_views, len(cams)))
    H, W = cams[0]["height"], cams[0]["width"]
    N = scene_cpu["params"]["_xyz"].shape[0]
    scale = 1.0 / n_views
    tt = np.zeros(5)   # lbs+sh fwd, raster fwd, loss, raster bwd, lbs+sh bwd
    num_rendered = 0
    P_e2e = {n: v.clone().requires_grad_(True) for n, v in scene_cpu["params"].items()}
    P_idn = {n: v.clone().requires_grad_(True) for n, v in scene_cpu["params"].items()}
    psnr = lambda a, b: -10.0 * math.log10(float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)))
    par = {"views": K, "psnr_delta_db": 0.0, "img_max_abs": 0.0, "img_mean_abs": 0.0,
           "identical_inputs": {"psnr_delta_db": 0.0, "img_max_abs": 0.0, "pairs_within_2e-4_of_alpha_threshold": 0,
                                "alpha_decisions_aligned": 0, "stop_decisions_aligned": 0, "stop_decisions_not_at_threshold": 0}}
    wts = torch.tensor(tr.CONIC_GRAD_WEIGHTS)
    for k in range(K):
        cam0 = cams[k]
        cc = torch.tensor(np.asarray(cam0["camera_center"], np.float32))
        view = np.asarray(cam0["world_view_transform"], np.float32).reshape(-1)
        proj = np.asarray(cam0["full_proj_transform"], np.float32).reshape(-1)
        tfx, tfy = math.tan(cam0["fovx"] / 2), math.tan(cam0["fovy"] / 2)
        tgt = targets_cpu[k]
        # ---- end to end, timed (the CPU baseline)
        t0 = time.time()
        o = tr.hand_forward(P_e2e, scene_cpu["grid"], scene_cpu["grid_center"], scene_cpu["grid_scale"],
                            scene_cpu["posed"][k], scene_cpu["rest"], cc)
        t1 = time.time()
        ro = RasterOracle(W, H, tfx, tfy, view, proj, o["posed_xyz"].detach().numpy(), o["posed_cov"].detach().numpy(),
                          o["colors"].detach().numpy(), o["opacity"].detach().numpy()[:, 0], np.ones(3, np.float32))
        t2 = time.time()
        img = torch.tensor(np.ascontiguousarray(ro.color)).permute(1, 2, 0).clone().requires_grad_(True)
        tg_hwc = tgt.permute(1, 2, 0)
        if loss == "l1+ssim":   # the step's image loss on the oracle's image (torch restatement of loss_utils)
            lv = tr.rgb_ssim_loss(img, tg_hwc) * scale
        else:
            lv = (img - tg_hwc).abs().mean() * scale
        (gi,) = torch.autograd.grad(lv, img)
        g = np.ascontiguousarray(gi.permute(2, 0, 1).numpy())
        t2b = time.time()
        b = ro.backward(g)
        t3 = time.time()
        ((o["posed_xyz"] * torch.tensor(b["means3D"])).sum() + (o["posed_cov"] * torch.tensor(b["cov3D"])).sum()
         + (o["colors"] * torch.tensor(b["colors"])).sum() + (o["opacity"][:, 0] * torch.tensor(b["opacity"])).sum()).backward()
        t4 = time.time()
        tt += np.array([t1 - t0, t2 - t1, t2b - t2, t3 - t2b, t4 - t3])
        num_rendered += int(ro.num_rendered)
        gi_np = gpu["img"][k]
        d = np.abs(gi_np - ro.color)
        par["psnr_delta_db"] = max(par["psnr_delta_db"], abs(psnr(gi_np, tgt.numpy()) - psnr(ro.color, tgt.numpy())))
        par["img_max_abs"] = max(par["img_max_abs"], float(d.max()))
        par["img_mean_abs"] = max(par["img_mean_abs"], float(d.mean()))
        del ro
        # ---- identical blend inputs: the kernels' own records through the oracle blend, then the torch chain
        r = gpu["grec"][k]
        bo = BlendOracle(W, H, r[:, 0:2], gpu["depth"][k], r[:, 2:5], r[:, 5], gpu["radii"][k], r[:, 6:9],
                         np.ones(3, np.float32))
        ii = par["identical_inputs"]
        # pairs within rounding of the alpha / transmittance thresholds take the side the KERNELS took (their own device
        # function is asked: tools/parity.py); what remains is arithmetic
        amb, fl, sfl, svi = align_threshold_decisions(bo, r, r[:, 6:9], np.ones(3, np.float32), W,
                                                      kernel_last_gaussian(k, K, N, W, H, gpu["n_contrib"][k]))
        ii["pairs_within_2e-4_of_alpha_threshold"] += amb
        ii["alpha_decisions_aligned"] += fl
        ii["stop_decisions_aligned"] += sfl
        ii["stop_decisions_not_at_threshold"] += svi
        d = np.abs(gi_np - bo.color)
        ii["psnr_delta_db"] = max(ii["psnr_delta_db"], abs(psnr(gi_np, tgt.numpy()) - psnr(bo.color, tgt.numpy())))
        ii["img_max_abs"] = max(ii["img_max_abs"], float(d.max()))
        bb = bo.backward(gpu["g_img"][k])
        o2 = tr.hand_forward(P_idn, scene_cpu["grid"], scene_cpu["grid_center"], scene_cpu["grid_scale"],
                             scene_cpu["posed"][k], scene_cpu["rest"], cc)
        ndc, conic = tr.project_ewa(o2["posed_xyz"], o2["posed_cov"], W, H, tfx, tfy, torch.tensor(view), torch.tensor(proj))
        tv = torch.tensor((gpu["radii"][k] > 0)[:, None].astype(np.float32))
        ((ndc * torch.tensor(bb["means2D"][:, :2])).mul(tv).sum() + (conic * wts * torch.tensor(bb["conic"])).mul(tv).sum()
         + (o2["colors"] * torch.tensor(bb["colors"])).mul(tv).sum()
         + (o2["opacity"][:, 0] * torch.tensor(bb["opacity"])).mul(tv[:, 0]).sum()).backward()
        del bo
    def grad_err(Pref):
        out, rows = {}, {}
        for n, v in Pref.items():
            a_ = gpu["grads"][n].reshape(v.shape).numpy().astype(np.float64)
            b_ = v.grad.numpy().astype(np.float64)
            den = float(np.abs(b_).max()) or 1.0
            out[n] = float("%.3g" % (float(np.abs(a_ - b_).max()) / den))
            rows[n] = float("%.3g" % float((np.abs(a_ - b_).reshape(N, -1).max(1) > 2e-5 * den).mean()))
        return out, rows
    par["grad_max_rel_err"], par["rows_over_2e-5"] = grad_err(P_e2e)
    par["identical_inputs"]["grad_max_rel_err"], _ = grad_err(P_idn)
    ii = par["identical_inputs"]
    ii["pass"] = bool(ii["psnr_delta_db"] < 0.01 and max(ii["grad_max_rel_err"].values()) < 1e-4 and ii["stop_decisions_not_at_threshold"] == 0)
    par["note"] = ("GPU step vs the CPU oracle on %d of the %d views at the bench size, same targets and loss; "
                   "grad_max_rel_err = max|a-b| / max|b| per leaf.  'identical_inputs' feeds the kernels' own per-instance "
                   "records to the oracle blend, with the (pixel, Gaussian) pairs that sit within rounding of the alpha >= 1/255 / "
                   "T < 1e-4 thresholds forced to the side the kernels took (bars: PSNR delta < 0.01 dB, grad < 1e-4; the run exits "
                   "non-zero when they fail; the same comparison is asserted by tests/test_gpu_fullsize.py); the end-to-end figures "
                   "compare two independent fp32 chains whose 1e-7 input differences flip isolated threshold decisions "
                   "(rows_over_2e-5 = fraction of Gaussians affected)" % (K, n_views))
    for key in ("psnr_delta_db", "img_max_abs", "img_mean_abs"):
        par[key] = float("%.3g" % par[key])
    for key in ("psnr_delta_db", "img_max_abs"):
        par["identical_inputs"][key] = float("%.3g" % par["identical_inputs"][key])
    tt /= K
    t_view = float(tt.sum())
    chain = torch_chain_timing(scene_cpu, cams[0])
    cpu = {"value": round(1.0 / (t_view * n_views), 5), "unit": "iters/s", "cores": threads, "kind": "port",
           "cpu_count": os.cpu_count(), "cpu_model": cpu_model_string(),
           "sample": "oracle port, %d of %d views (%.1f s of CPU work), N=%d, %dx%d; per view: torch LBS+cov+SH fwd %.2fs + "
                     "bwd %.2fs (%d threads), scalar C rasterizer fwd %.2fs + bwd %.2fs (1 thread), image loss %s %.2fs; "
                     "value = 1/(%d x %.2fs).  Not comparable with the reference's CUDA path; the torch chain alone "
                     "(SURVEY 8d) is under torch_chain"
                     % (K, n_views, t_view * K, N, W, H, tt[0], tt[4], threads, tt[1], tt[3], loss, tt[2], n_views, t_view),
           "torch_chain": chain, "raster_oracle_s_per_view": {"fwd": round(float(tt[1]), 3), "bwd": round(float(tt[3]), 3)},
           "num_rendered": num_rendered // K}
    return cpu, par


def dropin_main(args):
    """`--route dropin`: the zero-change route.  What an unmodified MANUS gets with only PYTHONPATH set is the two shim
    packages (`diff_gaussian_rasterization.GaussianRasterizer`, `simple_knn._C.distCUDA2`) under torch autograd, ONE
    (frame, view) per step (config/trainer/trainer.yaml:5) at the training resolution 1280 x 720
    (config/datasets/train/brics_dynamic.yaml:7-8), with the reference-shaped module code around the operator:
        TrainingModule.forward    -> manus_amd.modules.hand_forward      (hand_dynamic.py:86-137; skin weights + LBS operators)
        render_gaussians          -> manus_amd.render.render_gaussians   (gaussian_utils.py:349-428; SH operator + GaussianRasterizer,
                                     one host read of the pair count per forward, like upstream)
        loss_func's image terms   -> manus_amd.losses.l1_loss / ssim     (base.py:323-365)
        loss.backward()              torch autograd through the five operators' backward kernels
    No fused kernels, no engine, no cross-step state.  Not the headline: one view per step, another resolution."""
    from types import SimpleNamespace
    from manus_amd import _lib, losses, rasterizer
    from manus_amd.modules import hand_forward, object_forward
    from manus_amd.render import render_gaussians
    from manus_amd.structures import Bones
    from manus_amd.synthetic import camera_table, make_scene
    from manus_amd.engine import HipViewCompute
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    V, N = args.views, args.gaussians
    W, H = (1280, 720) if (args.width, args.height) == (1920, 1080) else (args.width, args.height)
    scene = make_scene(n_gaussians=N, kind=args.kind, seed=0, n_cameras=V, width=W, height=H, device=dev)
    ct = camera_table(scene["cameras"], dev)
    g = torch.Generator(device="cpu").manual_seed(123)
    pert = dict(scene)
    pert["params"] = {k: (v + 0.01 * v.abs().mean() * torch.randn(v.shape, generator=g).to(dev)) for k, v in scene["params"].items()}
    with torch.no_grad():
        targets = HipViewCompute(pert, torch.zeros((V, 3, H, W), device=dev), ct).forward_views_fused(list(range(V)))[0]
        targets_hwc = targets.permute(0, 2, 3, 1).contiguous()
    rasterizer.context(dev).clear()
    P = {k: v.detach().clone().requires_grad_(True) for k, v in scene["params"].items()}
    is_hand = args.kind == "hand"

    class Model:      # the attributes of GaussianModel the module code reads (gaussian.py:55-82)
        _xyz, _scaling, _rotation = P["_xyz"], P["_scaling"], P["_rotation"]
        grid_center, grid_scale, grid_weights = scene.get("grid_center"), scene.get("grid_scale"), scene.get("grid")

        @property
        def get_features(self):
            return torch.cat([P["_features_dc"], P["_features_rest"]], dim=1)

        @property
        def get_opacity(self):
            return torch.sigmoid(P["_opacity"])

    model = Model()
    cams = [SimpleNamespace(fovx=c["fovx"], fovy=c["fovy"], height=c["height"], width=c["width"],
                            world_view_transform=torch.tensor(c["world_view_transform"], dtype=torch.float32, device=dev)[None],
                            full_proj_transform=torch.tensor(c["full_proj_transform"], dtype=torch.float32, device=dev)[None],
                            camera_center=torch.tensor(c["camera_center"], dtype=torch.float32, device=dev)[None]) for c in scene["cameras"]]
    batches = [dict(bones_posed=Bones(None, None, None, scene["posed"][v]), bones_rest=Bones(None, None, None, scene["rest"]))
               for v in range(V)] if is_hand else [None] * V
    bg = scene["bg"]
    state = {"k": 0, "radii": None}

    def one_step():
        v = state["k"] % V
        state["k"] += 1
        for t in P.values():
            t.grad = None
        pred = hand_forward(model, batches[v]) if is_hand else object_forward(model)
        out = render_gaussians(pred.posed_xyz, pred.posed_cov, pred.cano_xyz, pred.cano_features, pred.cano_opacity, cams[v], bg,
                               sh_degree=3, tf=pred.get("tf"), device=dev)
        gt = targets_hwc[v]
        loss = 0.8 * losses.l1_loss(out["render"], gt) + 0.2 * (1.0 - losses.ssim(out["render"], gt))
        loss.backward()
        state["radii"] = out["radii"]
        return loss

    for _ in range(max(1, args.warmup)):
        loss = one_step()
    torch.cuda.synchronize()
    if args.dropin_fenced:
        rasterizer.check_overflow()
        rasterizer.set_sync_policy(False)
    dts, issue = [], []
    _lib.profile_enable(False)
    for _ in range(REPEATS):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = one_step()
        issue.append(time.perf_counter() - t0)      # (the host is done issuing; what is left is the device's backlog)
        torch.cuda.synchronize()
        dts.append(time.perf_counter() - t0)
    dt = float(np.median(dts))
    host_issue_ms = 1e3 * float(np.median(issue)) / args.steps
    if args.dropin_fenced:
        rasterizer.check_overflow()      # (raises if a forward of the timed loops ran out of pair capacity)
    # per-kernel breakdown of the library launches (HIP events around each: slows the step, taken in a separate loop)
    _lib.profile_enable(True, only=None)
    for _ in range(args.steps):
        one_step()
    prof = _lib.profile_report()
    _lib.profile_enable(False)
    tot = sum(v[1] for v in prof.values())
    lib_ms = tot / args.steps
    if args.profile_all:
        for k, (c, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
            print("%-22s %6d launches %9.3f ms avg %8.4f ms  %5.1f%%" % (k, c, ms, ms / c, 100 * ms / tot), file=sys.stderr)
        print("library kernels %.3f ms/step of %.3f ms/step wall (the rest: torch glue kernels -- cat, sigmoid, permute, the loss's "
              "arithmetic, gradient accumulation -- and the host: ctypes calls; the pair count is read back for the first forwards only)"
              % (lib_ms, 1e3 * dt / args.steps), file=sys.stderr)
    finite = all(bool(torch.isfinite(t.grad).all()) for t in P.values())
    dom = max(prof, key=lambda k_: prof[k_][1]) if prof else None
    line = {"metric": "train iters/sec (fwd+bwd), drop-in operator route: 1 view per step, %dk Gaussians @%dx%d "
                      "[not the headline configuration: GaussianRasterizer + reference-shaped modules under autograd]" % (N // 1000, W, H),
            "headline": False, "value": round(args.steps / dt, 4), "unit": "iters/s", "n_gpus": 1, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4), "repeats": REPEATS,
            "ms_per_step_min": round(1e3 * min(dts) / args.steps, 4), "ms_per_step_max": round(1e3 * max(dts) / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s, %d Gaussians, ONE view per step (cycling over %d cameras) at %dx%d, modules.hand_forward + "
                                   "render.render_gaussians + losses (0.8 L1 + 0.2 (1 - SSIM)) under torch autograd" % (args.kind, N, V, W, H),
                       "route": "dropin", "gaussians": N, "views_per_step": 1, "width": W, "height": H,
                       "host_syncs_per_step": 0 if (args.dropin_fenced or rasterizer.context(dev).auto_fenced > 0) else 1,
                       "library_kernel_ms_per_step": round(lib_ms, 4),
                       "host_issue_ms_per_step": round(host_issue_ms, 4),
                       "dominant_library_kernel": dom, "dominant_kernel_ms": round(prof[dom][1] / prof[dom][0], 4) if dom else None,
                       "finite_grads": finite, "loss": float(loss.detach())},
            "roofline": None, "cpu_baseline": None, "parity": None}
    print(json.dumps(line))
    sys.stdout.flush()


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
    ap.add_argument("--sh-storage", default="fp32", choices=["fp32", "fp16"],
                    help="storage of _features_rest read by the render kernels (fp16 = BASELINE config 5's option; the "
                         "headline number and the parity block use fp32, like the reference)")
    ap.add_argument("--allreduce", default="auto", choices=["auto", "dense", "compact"],
                    help="N > 1: 'dense' all-reduces all 61 N floats, 'compact' only the rows that received a gradient on some "
                         "rank; 'auto' times both for a few steps after the warm-up and keeps the faster one")
    ap.add_argument("--dense-allreduce", action="store_true", help="(same as --allreduce dense)")
    ap.add_argument("--round-robin-views", action="store_true",
                    help="N > 1: assign views to ranks round-robin instead of balancing them by measured pairs per view")
    ap.add_argument("--sharded-adam", action="store_true",
                    help="N > 1 with --optimizer: reduce-scatter the gradients, Adam on the owned 1/N of the elements, "
                         "all-gather the parameters (instead of all-reduce + the full Adam step on every rank)")
    ap.add_argument("--dense-loss-scan", action="store_true",
                    help="image loss without the rasterizer's tile occupancy: compares rendered and target image everywhere "
                         "to find the spans that need work (the default settles spans under empty tiles from the target alone)")
    ap.add_argument("--fresh-grads", action="store_true",
                    help="a new, fully zeroed set of gradient tensors every step instead of buffers the compute object keeps "
                         "(like .grad tensors) and the backward zeroing only the rows that need it")
    ap.add_argument("--gaussian-order", default="given", choices=["given", "morton"],
                    help="morton: the model's rows sorted along a Z-order curve first (a what-if for spatially coherent rows; not the headline)")
    ap.add_argument("--depth-cut", action="store_true",
                    help="time the step WITH the depth-cut hints (the previous forward's per-tile saturation depth bounds the "
                         "binning; exact).  Off by default, like engine.Trainer: it only pays while the model stands still "
                         "between steps.  The default line reports the hinted figure as `value_with_hints`")
    ap.add_argument("--no-depth-cut", action="store_true", help="(the default since round 5; accepted for old scripts)")
    ap.add_argument("--dropin-fenced", action="store_true",
                    help="--route dropin with rasterizer.set_sync_policy(False): the one line a MANUS user can add -- no blocking read of the "
                         "pair count per forward (the capacity learnt during warm-up is used; overflows are reported by rasterizer.poll())")
    ap.add_argument("--no-hints-variant", action="store_true", help="skip the extra timed region with the depth-cut hints on")
    ap.add_argument("--route", default="fused", choices=["fused", "dropin"],
                    help="dropin: what an unmodified MANUS gets with only PYTHONPATH set -- one (frame, view) per step through "
                         "modules.hand_forward + render.render_gaussians (GaussianRasterizer under autograd) + losses at the "
                         "reference's 1280x720 training resolution; not the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle leg (cpu_baseline and parity)")
    ap.add_argument("--cam-radius", type=float, default=1.2,
                    help="radius of the camera sphere in metres: 1.2 = the capture-like set of SURVEY 8(d) (the headline), 0.45 = its "
                         "'close-up' set (the hand fills the frame, deep tile lists); anything but 1.2 is labelled, not the headline")
    ap.add_argument("--trained-steps", type=int, default=300,
                    help="Adam steps taken before the `value_trained_state` figure is timed (the same fwd+bwd loop on the model those "
                         "steps leave; 0 = skip it)")
    ap.add_argument("--parity-views", type=int, default=2, help="views of the step run through the CPU oracle")
    ap.add_argument("--profile-all", action="store_true", help="print a per-kernel HIP-event breakdown to stderr")
    args = ap.parse_args()
    if args.route == "dropin":
        return dropin_main(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)                     # before the process group: RCCL binds its communicator to the current device
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm; MANUS_BENCH_BACKEND=gloo lets the N>1 path be exercised on a
        # single-GPU box (all ranks then share device 0)
        dist.init_process_group(os.environ.get("MANUS_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)

    from manus_amd import _lib, rasterizer
    from manus_amd.engine import HipViewCompute, ViewShardedStep
    from manus_amd.synthetic import camera_table, make_scene

    # an instrumented / knock-out build of the library (tools/instr; MANUS_HIP_VARIANT names the .so) labels its line and is
    # never the headline; one whose knock-outs change results (bit 0) also skips the parity block
    variant_bits = int(_lib.lib().mgr_build_variant())
    variant_name = os.environ.get("MANUS_HIP_VARIANT", "")
    V, N, W, H = args.views, args.gaussians, args.width, args.height
    scene = make_scene(n_gaussians=N, kind=args.kind, seed=0, n_cameras=V, width=W, height=H, device=dev, cam_radius=args.cam_radius)
    if args.gaussian_order == "morton":   # (an experiment, not the headline workload: the rows of a spatially sorted model)
        xyz = scene["params"]["_xyz"]
        q = ((xyz - xyz.min(0).values) / (xyz.max(0).values - xyz.min(0).values + 1e-12) * 1023.0).long().clamp_(0, 1023)
        code = torch.zeros(xyz.shape[0], dtype=torch.long, device=xyz.device)
        for bit in range(10):
            for ax in range(3):
                code |= ((q[:, ax] >> bit) & 1) << (3 * bit + ax)
        nh = scene["n_hand"] if scene["kind"] == "composite" else 0
        if nh:   # the articulated rows stay in front of the static ones
            code = code + (torch.arange(xyz.shape[0], device=xyz.device) >= nh).long() * (1 << 31)
        perm = torch.argsort(code)
        scene["params"] = {k: v[perm].contiguous() for k, v in scene["params"].items()}
    ct = camera_table(scene["cameras"], dev)

    # target images: the same scene with parameters perturbed by 1 % (non-trivial dL/dimage)
    g = torch.Generator(device="cpu").manual_seed(123)
    pert = dict(scene)
    pert["params"] = {k: (v + 0.01 * v.abs().mean() * torch.randn(v.shape, generator=g).to(dev))
                      for k, v in scene["params"].items()}
    with torch.no_grad():   # one fused launch over all views: every dispatch of this process has the bench's shape
        hp = HipViewCompute(pert, torch.zeros((V, 3, H, W), device=dev), ct)
        targets = hp.forward_views_fused(list(range(V)))[0].contiguous()
        del hp
    rasterizer.context(dev).clear()
    compute = HipViewCompute(scene, targets, ct, loss=args.loss, sh_storage=args.sh_storage, sparse_loss=not args.dense_loss_scan,
                             depth_cut=args.depth_cut and not args.no_depth_cut, persistent_grads=not args.fresh_grads)
    shapes = {k: v.shape for k, v in compute.params.items()}
    sharded = args.sharded_adam and args.optimizer and world > 1
    # N > 1: the views go to the ranks by measured cost (pairs per view from one forward of every view -- deterministic,
    # so every rank computes the same assignment), heaviest first to the least loaded rank
    weights = None
    if world > 1 and not args.round_robin_views:
        from manus_amd.engine import view_costs
        weights = view_costs(compute.pairs_per_view(), N)
        rasterizer.context(dev).clear()
    mode = "dense" if (args.dense_allreduce or args.allreduce == "dense") else args.allreduce
    mode_timings = None

    def make_step(m):
        return ViewShardedStep(N, shapes, compute, V, rank=rank, world_size=world, compact=(m == "compact") and not sharded,
                               scatter=sharded, view_weights=weights)

    def timed(st, k):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            st.step()
        torch.cuda.synchronize()
        tt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)      # every rank sees the same number and takes the same decision
        return float(tt.item()) / k

    if world > 1 and not sharded and mode == "auto":
        # only rows with a gradient on some rank need to travel (exact; 43 % of the rows in this scene), but the
        # row-compacted exchange costs a host synchronisation, a second small collective and ~25 torch launches: which of
        # the two wins depends on the fabric, so both are timed here (after their own warm-up) and the faster one is kept
        mode_timings = {}
        st = make_step("dense")
        for _ in range(max(1, args.warmup)):     # learns the pair capacity (with host syncs)
            st.step()
        rasterizer.check_overflow()
        rasterizer.set_sync_policy(False)
        for m in ("dense", "compact"):
            st = make_step(m)
            for _ in range(2):
                st.step()
            mode_timings[m] = round(1e3 * timed(st, 5), 4)
            compute.grad_arena = None
        rasterizer.check_overflow()
        rasterizer.set_sync_policy(True)
        mode = min(mode_timings, key=mode_timings.get)
    elif mode == "auto":
        mode = "dense"
    step = make_step(mode)
    V_local = len(step.local_views)
    opt = None
    if args.optimizer:
        from manus_amd.optim import GaussianOptimizer
        opt = GaussianOptimizer(compute.params, adopt=True)
        base_step = step.step
        if sharded:
            opt.flatten(step.padded_g)
            compute.set_params(opt.parameters())
            opt.p = {k: v.detach() for k, v in compute.params.items()}

        def step_with_adam():
            o = base_step()
            if compute.depth_cut and not rasterizer.context(dev).sync_every_forward:
                # the model moves under the depth-cut hints: wait for the forward's fence (the backward queued behind it
                # keeps the GPU busy) and run the step again if it was flagged -- what engine.Trainer does every step
                try:
                    rasterizer.poll(dev)
                except RuntimeError:
                    o = base_step()
                    rasterizer.poll(dev)
            opt.update_learning_rate(opt.state_step + 1)
            if sharded:
                opt.step_range(step._store, *step.owned)
                step.all_gather_params(opt.pflat)
            else:
                opt.step(o["grads"])
            compute.mark_params_changed()
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

    # The dominant kernel is the one THIS run spends most of its time in (k_blend_bwd with eight views on the GPU, the
    # forward blend with one or two): three untimed steps with HIP events around every library kernel decide it, so that
    # every line -- also the one-view-per-GPU shares of a multi-GPU run -- carries the roofline of its own dominant kernel.
    _lib.profile_enable(True, only=None)
    for _ in range(3):
        step.step()
    pre = _lib.profile_report()
    _lib.profile_enable(False)
    pre.pop("k_image_loss_list", None)      # (runs on the side stream next to the forward blend: its bracket spans the overlap)
    DOMINANT = max(pre, key=lambda k_: pre[k_][1]) if pre else "k_blend_bwd"
    if world > 1:   # every rank brackets the same kernel (rank 0's choice)
        names = sorted(pre) if pre else ["k_blend_bwd"]
        pick = torch.tensor([names.index(DOMINANT)], device=dev)
        dist.broadcast(pick, src=0)
        DOMINANT = names[int(pick.item())] if int(pick.item()) < len(names) else DOMINANT
    # HIP events on the kernel's own stream: around the dominant kernel only (the roofline's duration), around every
    # library kernel with --profile-all (26 bracketed launches per step cost ~0.1 ms of the step).
    # The loop of exactly --steps steps (barrier + synchronize on both sides, MAX over the ranks) is timed REPEATS times:
    # the line carries the median, the spread beside it -- one hiccup in a 40 ms region no longer moves the headline.
    # (the two events around a launch cost ~6 us of idle GPU each: the dominant kernel is bracketed on every fourth step of
    # the timed region -- 5 x steps / 4 samples of its duration, all taken inside it)
    def timed_region():
        _lib.profile_enable(True, only=None if args.profile_all else DOMINANT, every=1 if args.profile_all else 4)
        dts_ = []
        for _ in range(REPEATS):
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                o_ = step.step()
            barrier()
            d_ = time.perf_counter() - t0
            if world > 1:
                tt = torch.tensor([d_], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                d_ = float(tt.item())
            dts_.append(d_)
        p_ = _lib.profile_report()
        _lib.profile_enable(False)
        return dts_, p_, o_

    dts, prof, out = timed_region()
    # A flagged forward inside the timed region (depth-cut hints that did not fit: the model stands still in this loop, so none
    # is expected) would have left an incomplete step in the measurement: if that ever happens the region is measured again
    # without the depth cut, and the line says so.
    remeasured = False
    try:
        rasterizer.check_overflow()
    except _lib.ManusHipError:
        if args.optimizer or world > 1:
            raise
        compute.depth_cut = False
        rasterizer.set_sync_policy(True)
        step.step()
        rasterizer.check_overflow()
        rasterizer.set_sync_policy(False)
        step.step()
        dts, prof, out = timed_region()
        rasterizer.check_overflow()
        remeasured = True
    dt = float(np.median(dts))
    # With the optimizer in the loop a Gaussian can walk out of the skin-weight grid; its weights are then 0/0 = NaN
    # exactly like the reference (gaussian_utils.py:193-195), which drops such rows when it loads a checkpoint.
    nonfinite = sum(int((~torch.isfinite(x)).sum()) for x in out["grads"].values())
    assert args.optimizer or nonfinite == 0 or (variant_bits & 1), "non-finite gradients"   # (only a knock-out build may produce them: tools/instr)
    # list entries the blend actually consumed (sum over tiles of the deepest contributor): the units k_blend_fwd / _bwd process
    consumed = None
    import ctypes
    ws = rasterizer.context(dev).last_ws
    if ws is not None:
        arr = (ctypes.c_size_t * 32)()
        _lib.lib().mgr_raster_layout(V_local, N, W, H, ws.cap, arr, 32)
        VT = V_local * ((W + 15) // 16) * ((H + 15) // 16)
        consumed = int(ws.buf[int(arr[9]): int(arr[9]) + 4 * VT].view(torch.int32).sum().item())

    def keep(o_):
        return {k: ({n: g.clone() for n, g in v.items()} if isinstance(v, dict) else (v.clone() if torch.is_tensor(v) else v))
                for k, v in o_.items()}

    def same_step(a_, b_, img_a, img_b):
        """Bit-equality of two steps' outputs (the loss scalar is summed with float atomics: 1e-6)."""
        bad = [k for k in a_["grads"] if not torch.equal(a_["grads"][k], b_["grads"][k])]
        bad += [k for k in ("grad2d", "vis", "radii") if not torch.equal(a_[k], b_[k])]
        if not torch.equal(img_a, img_b):
            bad.append("image")
        if abs(float(a_["loss"]) - float(b_["loss"])) > 1e-6 * max(1.0, abs(float(b_["loss"]))):
            bad.append("loss")
        return bad

    # The headline is timed in the configuration training runs in (engine.Trainer's and HipViewCompute's defaults: depth cut
    # off, kept buffers).  The depth-cut hints pay only while the model stands still between steps -- as it does in this
    # fwd+bwd loop -- so their figure is an extra key, `value_with_hints`, measured here on the same box and process.
    static_model = world == 1 and not args.optimizer
    timed_out = timed_img = None
    if static_model:
        timed_out, timed_img = keep(out), compute.last_image.clone()
    hints = None
    if static_model and not compute.depth_cut and not args.no_hints_variant and compute.fused:
        compute.depth_cut = os.environ.get("MANUS_DEPTH_CUT", "1") != "0"
        if compute.depth_cut:
            for _ in range(3):      # the first forward leaves the hints, the following ones bin against them
                step.step()
            rasterizer.check_overflow()
            dts_h, _, out_h = timed_region()
            try:
                rasterizer.check_overflow()
                flagged = False
            except _lib.ManusHipError:
                flagged = True
            dt_h = float(np.median(dts_h))
            hints = {"value": round(args.steps / dt_h, 4), "ms_per_step": round(1e3 * dt_h / args.steps, 4),
                     "flagged_forwards": rasterizer.context(dev).cut_retries, "remeasure_needed": flagged,
                     "differs_from_headline_step": same_step(out_h, timed_out, compute.last_image, timed_img)}
        compute.depth_cut = False
        step.step()                 # (back on full lists: the parity block below runs the headline's configuration)
        rasterizer.check_overflow()

    # What the same loop costs on the state TRAINING leaves: a few hundred Adam steps at the reference's learning rates lengthen
    # the walks (2.9 M -> ~4 M consumed list entries on this scene); the headline above is timed on the initial state.  Same
    # process, same compute object and mode; the parameters are put back bit for bit afterwards (the parity block follows).
    trained = None
    if static_model and args.trained_steps > 0 and compute.fused:
        from manus_amd.optim import GaussianOptimizer
        saved = {k: v.detach().clone() for k, v in compute.params.items()}
        try:
            opt_t = GaussianOptimizer(compute.params, adopt=True)
            for it_ in range(args.trained_steps):
                o_t = step.step()
                if it_ % 50 == 49:
                    try:
                        rasterizer.poll(dev)
                    except RuntimeError:       # the pair capacity was outgrown: the hint is enlarged, the step repeated
                        o_t = step.step()
                opt_t.update_learning_rate(opt_t.state_step + 1)
                opt_t.step(o_t["grads"])
                compute.mark_params_changed()
            rasterizer.set_sync_policy(True)
            step.step()
            rasterizer.check_overflow()
            rasterizer.set_sync_policy(False)
            step.step()
            dts_t, prof_t, out_t = timed_region()
            rasterizer.check_overflow()
            dt_t = float(np.median(dts_t))
            ws_t = rasterizer.context(dev).last_ws
            arr_t = (ctypes.c_size_t * 32)()
            _lib.lib().mgr_raster_layout(V_local, N, W, H, ws_t.cap, arr_t, 32)
            consumed_t = int(ws_t.buf[int(arr_t[9]): int(arr_t[9]) + 4 * V_local * ((W + 15) // 16) * ((H + 15) // 16)].view(torch.int32).sum().item())
            dom_t = prof_t.get(DOMINANT)
            trained = {"value": round(args.steps / dt_t, 4), "ms_per_step": round(1e3 * dt_t / args.steps, 4),
                       "adam_steps_before": args.trained_steps, "consumed_pairs": consumed_t,
                       "dominant_kernel_ms": round(dom_t[1] / dom_t[0], 4) if dom_t else None,
                       "nonfinite_grad_values": sum(int((~torch.isfinite(x)).sum()) for x in out_t["grads"].values()),
                       "what": "the headline's fwd+bwd loop (no optimizer inside the timed steps) on the parameters %d fused Adam steps "
                               "at the reference's learning rates leave: the state a training run is in" % args.trained_steps}
        except _lib.ManusHipError as exc:
            trained = {"error": str(exc)[:200]}
        with torch.no_grad():
            for k, v in compute.params.items():
                v.copy_(saved[k])
        compute.mark_params_changed()
        del saved
        rasterizer.set_sync_policy(True)
        step.step()
        rasterizer.check_overflow()
        rasterizer.set_sync_policy(False)
        step.step()
        rasterizer.check_overflow()

    if world > 1 and out.get("overflow") is not None and float(out["overflow"]) > 0.0:
        # (summed over the ranks by the step's collective: a rasterizer overflow on some rank, or -- compact mode -- a union of rows
        # beyond the capacity the exchange was sized for; engine.Trainer runs such a step again, a benchmark must not count it)
        raise RuntimeError("bench: the last timed step raised the step's overflow word (%g)" % float(out["overflow"]))
    views_by_rank = None
    if world > 1:
        views_by_rank = [None] * world
        dist.all_gather_object(views_by_rank, [int(v) for v in step.local_views])
    # digest of the step's (reduced) leaf gradients: lets an N-rank line be compared with the one-rank line of the same size
    digest = {k: [float(g.double().sum()), float(g.double().abs().sum())] for k, g in out["grads"].items()}
    digest["loss"] = float(out["loss"])

    if rank == 0:
        R_view = npairs_local / max(1, V_local)          # measured pairs per view (num_rendered)
        P_px = W * H
        n_poses = V if args.kind == "hand" else 0
        b_iter = algorithmic_bytes(N, V, R_view, P_px, n_poses)
        dom = prof.get(DOMINANT)
        roof = None
        if dom:
            avg_ms = dom[1] / dom[0]
            kb_survey = kernel_algorithmic_bytes(DOMINANT, N, V_local, R_view, P_px) or 0
            kb = kernel_unit_bytes(DOMINANT, V_local, consumed, P_px) or kb_survey
            ach = kb / (avg_ms * 1e-3) / 1e9
            pmc = pmc_counters(DOMINANT, N, V_local, W, H) if world == 1 else {}
            # the committed counter passes describe THIS kernel only if their dispatch took as long as it does now
            stale = bool(pmc.get("duration_ns")) and abs(pmc["duration_ns"] * 1e-6 - avg_ms) > 0.05 * avg_ms
            if stale or not pmc.get("duration_ns"):
                pmc = {"stale": bool(stale)}
            roof = {"bound": "hbm", "kernel": DOMINANT, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5), "frac_units": round(ach / HBM_PEAK_GBS, 5), "traffic": pmc.get("traffic"),
                    "avg_kernel_ms": round(avg_ms, 4), "kernel_duration_samples": int(dom[0]), "algorithmic_bytes_per_launch": int(kb),
                    "units": ("list entries the forward's walks consumed x %d B + pixels x 20 B (the units this launch processes)"
                              % (112 if DOMINANT == "k_blend_bwd" else 40)) if kb != kb_survey else "SURVEY 8(d) per-unit figures",
                    "consumed_pairs": consumed,
                    "survey_formula_bytes_per_launch": int(kb_survey),
                    "frac_survey_formula": round(kb_survey / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                    "iter_algorithmic_GBps": round(b_iter * args.steps / dt / 1e9, 2),
                    "note": "peak / unit / achieved / frac price the HBM roofline on the bytes of the units the launch processes "
                            "(consumed list entries, not SURVEY 8(d)'s rectangle pairs, most of which exact culling and early "
                            "termination never read: that figure is frac_survey_formula); this path has no dense contraction; "
                            "`limiter` says what the counters say bounds the kernel"}
            if pmc.get("stale"):
                roof["traffic_stale"] = True    # counters on file are of an older build of the kernel: not reused
            if pmc.get("step_hbm_bytes") and pmc.get("step_kernel_ns"):
                # HBM bytes of ALL kernels of one step by the counters (the sum over the kernels of their mean per launch x
                # their launches per step) over this run's step time -- what the step as a whole moves, next to the
                # algorithmic figure above; only when the counter pass's kernels took as long as this run's step
                step_ms = 1e3 * dt / args.steps
                if abs(pmc["step_kernel_ns"] * 1e-6 - step_ms) < 0.08 * step_ms:
                    roof["step_traffic_bytes"] = int(pmc["step_hbm_bytes"])
                    roof["step_traffic_GBps"] = round(pmc["step_hbm_bytes"] / (step_ms * 1e-3) / 1e9, 1)
                    roof["step_traffic_frac"] = round(pmc["step_hbm_bytes"] / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            if pmc.get("traffic"):
                roof["frac_traffic"] = round(pmc["traffic"] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
            if pmc.get("valu_issue_frac") is not None:
                roof["valu_issue_frac"] = pmc["valu_issue_frac"]
                ft = roof.get("frac_traffic") or 0.0
                lf = pmc.get("lds_active_frac") or 0.0
                if lf:
                    roof["lds_active_frac"] = lf
                vf = pmc["valu_issue_frac"]
                roof["limiter"] = ("hbm" if ft > 0.6 else "lds + valu-issue" if (lf > 0.55 and vf > 0.45) else "valu-issue" if vf > 0.5
                                   else "lds" if lf > 0.55 else "latency/occupancy (VALU + LDS, neither saturated)")
        if args.profile_all:
            tot = sum(v[1] for v in prof.values())
            for k, (c, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
                print("%-20s %6d launches %9.3f ms avg %8.4f ms  %5.1f%%" % (k, c, ms, ms / c, 100 * ms / tot),
                      file=sys.stderr)
            print("library kernels %.3f ms/iter of %.3f ms/iter wall" % (tot / (args.steps * REPEATS), 1e3 * dt / args.steps),
                  file=sys.stderr)
        cpu = parity = None
        if not args.no_cpu_baseline and world == 1 and args.kind == "hand" and not args.optimizer and args.sh_storage == "fp32" and not (variant_bits & 1):
            # The GPU step once more on the sampled views alone (same targets, same loss, same 1/V scale), ON THE TIMED
            # REGION'S compute object and IN ITS MODE (no host sync per forward: fenced; kept buffers; no depth cut) -- run
            # twice, so that the second one has the row- / tile-selective fills of the kept buffers in use like every timed
            # step had -- keeping the images, dL/dimage and the kernels' per-instance records for the oracle.
            Ks = min(args.parity_views, V)
            ids = list(range(Ks))
            compute.grad_arena = None
            for attempt in range(3):
                compute(ids, 1.0 / V)
                o = compute(ids, 1.0 / V)
                try:
                    rasterizer.check_overflow()      # (blocking; raises if either forward ran out of pair capacity: the
                    break                            #  hint was enlarged, the two steps are run again)
                except _lib.ManusHipError:
                    if attempt == 2:
                        raise
            gpu_mode = {"sync_per_forward": bool(rasterizer.context(dev).sync_every_forward), "depth_cut": bool(compute.depth_cut),
                    "kept_buffers": bool(compute.persistent_grads),
                    "selective_fills_in_use": bool(compute.persistent_grads and compute._pg_ws is not None and compute._pimg_ws is not None)}
            img_s = compute.last_image
            _, g_img = compute._image_loss(img_s, compute._select(ids)["targets"], 1.0 / V)
            torch.cuda.synchronize()
            ws2 = rasterizer.context(dev).last_ws
            import ctypes
            arr = (ctypes.c_size_t * 32)()
            _lib.lib().mgr_raster_layout(Ks, N, W, H, ws2.cap, arr, 32)
            raw = ws2.buf
            from tools.parity import read_records      # (test infrastructure: only the parity leg reads the kernels' records back)
            gpu = {"n_contrib": raw[int(arr[17]): int(arr[17]) + Ks * W * H * 4].view(torch.int32).reshape(Ks, H, W).cpu().numpy(),
                   "img": img_s.cpu().numpy(), "g_img": g_img.cpu().numpy(),
                   "grads": {k: v.detach().cpu() for k, v in o["grads"].items()},
                   "grec": read_records(raw, int(arr[1]), Ks * N).reshape(Ks, N, 12),
                   "depth": raw[int(arr[2]): int(arr[2]) + Ks * N * 4].view(torch.float32).reshape(Ks, N).cpu().numpy(),
                   "radii": compute.last_radii.cpu().numpy()}
            sc_cpu = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in scene.items() if k != "params"}
            sc_cpu["params"] = {k: v.detach().cpu() for k, v in scene["params"].items()}
            cpu, parity = cpu_baseline_and_parity(sc_cpu, scene["cameras"], targets[:Ks].cpu(), gpu, sample_views=Ks,
                                                  n_views=V, loss=args.loss)
            # ... and the bridge from the timed step itself: the LAST TIMED STEP's outputs (all V views, cloned right after
            # the timed region) against a plain step of a second compute object -- host sync per forward, fresh zero-filled
            # tensors, no hints, no skipped launches: bit for bit
            rasterizer.set_sync_policy(True)
            plain = HipViewCompute(scene, targets, ct, loss=args.loss, sh_storage=args.sh_storage, sparse_loss=not args.dense_loss_scan,
                                   depth_cut=False, persistent_grads=False)
            o_plain = plain(list(range(V)), 1.0 / V)
            torch.cuda.synchronize()
            differs = same_step(timed_out, o_plain, timed_img, plain.last_image)
            del plain, o_plain
            parity["gpu_step_mode"] = gpu_mode
            parity["timed_step_vs_plain_step"] = {"differs": differs, "equal": not differs,
                                                  "what": "last step of the timed region (all %d views; fenced, kept buffers) vs a step of a second "
                                                          "compute object with a host sync per forward and fresh zeroed tensors: torch.equal on the six "
                                                          "leaf gradients, grad2d, vis, radii and the image; loss to 1e-6" % V}
            parity["identical_inputs"]["pass"] = bool(parity["identical_inputs"]["pass"] and not differs)
        # the headline is BASELINE.json's metric on BASELINE.json's configuration, timed the way training runs it (no depth-cut
        # hints); anything else is labelled by its arguments
        headline = (args.kind == "hand" and N == 300000 and V == 8 and (W, H) == (1920, 1080) and args.loss == "l1+ssim"
                    and not args.optimizer and args.sh_storage == "fp32" and not args.dense_loss_scan and not compute.depth_cut
                    and not args.depth_cut and not args.fresh_grads and args.cam_radius == 1.2 and not variant_bits and not variant_name)
        k_str = "%dk" % (N // 1000) if N % 1000 == 0 else str(N)
        res_str = "1080p" if (W, H) == (1920, 1080) else "%dx%d" % (W, H)
        metric = ("train iters/sec (fwd+bwd) %s Gaussians @%s, %d views; PSNR parity" % (k_str, res_str, V))
        if not headline:
            metric += " [not the headline configuration: %s%s%s%s%s%s]" % (args.kind, ", optimizer in the step" if args.optimizer else "",
                                                                          ", fp16 SH storage" if args.sh_storage == "fp16" else "",
                                                                          ", depth-cut hints" if args.depth_cut else "",
                                                                          ", cameras at %.2f m" % args.cam_radius if args.cam_radius != 1.2 else "",
                                                                          ", library variant '%s' (bits %d)" % (variant_name, variant_bits) if (variant_bits or variant_name) else "")
        kind_str = {"hand": "HAND_GAUSSIAN: %d Gaussians, 21-transform LBS" % N, "object": "OBJ_GAUSSIAN: %d static Gaussians" % N,
                    "composite": "COMPOSITE: %d Gaussians (hand, 21-transform LBS + static object)" % N}.get(args.kind, args.kind)
        ms = [1e3 * d_ / args.steps for d_ in dts]
        line = {
            "metric": metric, "headline": bool(headline and args.gaussian_order == "given"),
            "value": round(args.steps / dt, 4), "unit": "iters/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4), "repeats": REPEATS,
            "ms_per_step_min": round(min(ms), 4), "ms_per_step_max": round(max(ms), 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "value_with_hints": hints["value"] if hints else None,
            "ms_per_step_with_hints": hints["ms_per_step"] if hints else None,
            "value_trained_state": trained.get("value") if trained else None,
            "ms_per_step_trained_state": trained.get("ms_per_step") if trained else None,
            "variant": ({"name": variant_name, "bits": variant_bits} if (variant_bits or variant_name) else None),
            "config": {"workload": "%s, %d views %dx%d, one pose per view "
                                   "(n_poses=%d), image loss %s, fwd+bwd to leaf grads" % (kind_str, V, W, H, n_poses, "0.8*L1 + 0.2*(1-SSIM)" if args.loss == "l1+ssim" else "L1"),
                       "gaussians": N, "views": V, "width": W, "height": H, "views_per_gpu": V_local,
                       "cameras": ("capture-like: %d cameras on a %.2f m sphere (SURVEY 8d)" % (V, args.cam_radius) if args.cam_radius == 1.2 else
                                   "close-up: %d cameras on a %.2f m sphere (SURVEY 8d's deep-list set at 0.45 m)" % (V, args.cam_radius)),
                       "trained_state": trained,
                       "pairs_per_view": int(R_view), "parallelism": "views/%d" % world,
                       "view_assignment": (None if world == 1 else "round-robin" if weights is None else "balanced by measured pairs per view (LPT)"),
                       "views_by_rank": views_by_rank, "grad_digest": digest,
                       "allreduce": (None if world == 1 else "reduce-scatter + sharded Adam + all-gather" if sharded else
                                     {"mode": mode, "ms_per_step_by_mode": mode_timings,
                                      "detail": "dense 61N floats" if mode == "dense" else
                                                "rows with a gradient (%s of %d; the collective is sized for %s) x 60 floats + 2N bytes"
                                                % (step.exchanged_rows(), N, getattr(step, "last_cap_rows", None))}),
                       "predicted_ms": predicted_ms(world, N, V, args.kind, W, H),
                       "optimizer_in_step": bool(args.optimizer), "sh_storage": args.sh_storage,
                       "gaussian_order": args.gaussian_order,
                       "gradient_buffers": ("fresh tensors, every row zeroed every step" if (args.fresh_grads or world > 1) else
                                            "gradients and image kept by the compute object (like .grad; engine.Trainer's and HipViewCompute's default); rows without a gradient are zeroed, and empty tiles written with the background, only where the previous step left something else"),
                       "remeasured_without_hints": remeasured,
                       "depth_cut": ("off" if not (compute.depth_cut or args.depth_cut) else
                                     "per-tile saturation depth of the previous forward of the same views bounds the binning; exact "
                                     "(flagged and re-run without it when a cut list runs out): %d flagged forwards in this run"
                                     % rasterizer.context(dev).cut_retries),
                       "with_hints": (None if not hints else
                                      {"what": "the same loop with the depth-cut hints on (HipViewCompute(depth_cut=True)): the previous forward's "
                                               "per-tile saturation depth bounds the binning; exact; pays only while the model stands still "
                                               "between steps, which is why training (engine.Trainer) and the headline run without it",
                                       **hints}),
                       "loss_span_list": ("full comparison of rendered and target image" if args.dense_loss_scan else
                                          "tile occupancy of the forward + per-view target-vs-background column masks (computed once per view)"
                                          if compute.target_map else "tile occupancy of the forward + target background"),
                       "nonfinite_grad_values": nonfinite},
            "roofline": roof, "cpu_baseline": cpu, "parity": parity,
        }
        print(json.dumps(line))
        sys.stdout.flush()
        if parity is not None and not parity["identical_inputs"]["pass"]:
            print("bench: PARITY FAILED on identical blend inputs: %s / timed step vs plain step: %s"
                  % (json.dumps(parity["identical_inputs"]), json.dumps(parity["timed_step_vs_plain_step"])), file=sys.stderr)
            sys.exit(3)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
