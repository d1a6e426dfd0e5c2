"""Synthetic MANUS-shaped scenes: a 20-bone hand skeleton taken from
the reference's bundled pose data, Gaussians sampled on the bones the way the
reference initialises them, a MANO-like skin-weight voxel grid, and capture-like
cameras.  Pure data generation (torch/numpy); used by tests, smoke and bench.

Reference rules followed (brown-ivl/manus):
  sampling on bones   src/utils/train_utils.py:104-139
  voxel grid geometry src/datasets/brics_dynamic.py:99-144 (+ extra.py:258 grid order)
  cameras             src/utils/cam_utils.py:50-78, data/camera_paths/real.pkl intrinsics
"""
import math
import os

import numpy as np
import torch

from .cam_utils import get_opengl_camera_attributes
from .transforms import bone_transforms

_HERE = os.path.dirname(os.path.abspath(__file__))


def load_skeleton():
    d = np.load(os.path.join(_HERE, "data", "skeleton.npz"))
    return {k: d[k] for k in d.files}


def grid_geometry(heads, tails, res=128, ratio=(1.1, 0.9, 0.65), offset=(0.0, 0.0, -0.03)):
    """(D,H,W), center (3,), scale (3,) exactly as build_voxel_grid derives them."""
    keypts = np.concatenate([heads[:1], tails], 0)
    cmin, cmax = keypts.min(0), keypts.max(0)
    center = (cmax + cmin) / 2 + np.asarray(offset)
    xr, yr, zr = ratio
    rs = (res / np.array([xr, yr, zr])).astype(np.int32)
    d, h, w = int(rs[2]), int(rs[1]), int(rs[0])
    s = np.linalg.norm(cmax - cmin) / 2
    scale = np.array([s * zr, s * yr, s * xr], dtype=np.float32)
    return (d, h, w), center.astype(np.float32), scale


def make_skin_grid(heads, tails, dims, center, scale, device="cpu", chunk=1 << 20):
    """(D,H,W,21) fp32: softmax(-|p - bone_mid|/0.01) over 20 bones, background
    channel = 1 (others 0) farther than 2 cm from every bone mid-segment."""
    D, H, W = dims
    dev = torch.device(device)
    hd = torch.as_tensor(heads, dtype=torch.float32, device=dev)
    tl = torch.as_tensor(tails, dtype=torch.float32, device=dev)
    c = torch.as_tensor(center, dtype=torch.float32, device=dev)
    s = torch.as_tensor(scale, dtype=torch.float32, device=dev)
    zs = torch.linspace(-1, 1, D, device=dev)
    ys = torch.linspace(-1, 1, H, device=dev)
    xs = torch.linspace(-1, 1, W, device=dev)
    out = torch.empty((D * H * W, 21), dtype=torch.float32, device=dev)
    seg = tl - hd
    seg_l2 = (seg * seg).sum(-1).clamp_min(1e-12)
    total = D * H * W
    for b in range(0, total, chunk):
        idx = torch.arange(b, min(total, b + chunk), device=dev)
        iz = idx // (H * W)
        iy = (idx // W) % H
        ix = idx % W
        p = torch.stack([xs[ix], ys[iy], zs[iz]], -1) * s + c  # u=(x,y,z) indexes (W,H,D)
        rel = p[:, None, :] - hd[None]
        t = ((rel * seg[None]).sum(-1) / seg_l2[None]).clamp(0, 1)
        dist = (rel - t[..., None] * seg[None]).norm(dim=-1)  # distance to the bone segment
        w = torch.softmax(-dist / 0.01, dim=-1)
        far = dist.min(dim=-1).values > 0.02
        w = torch.where(far[:, None], torch.zeros_like(w), w)
        bgc = far.float()[:, None]
        out[idx] = torch.cat([w, bgc], -1)
    return out.reshape(D, H, W, 21)


def sample_on_bones(heads, tails, rest_tf, n_per_bone, gen):
    """MVN around bone mids (std len/5,len/4,len/4 in the bone frame) + half as many
    around the heads (std len/6,len/4,len/6): train_utils.py:104-139."""
    hd = torch.as_tensor(heads, dtype=torch.float32)
    tl = torch.as_tensor(tails, dtype=torch.float32)
    R = torch.as_tensor(rest_tf, dtype=torch.float32)[:, :3, :3]
    ln = (tl - hd).norm(dim=1, keepdim=True)
    out = []
    for mid, sc, n in (((hd + tl) / 2, torch.cat([ln / 5, ln / 4, ln / 4], -1), n_per_bone),
                       (hd, torch.cat([ln / 6, ln / 4, ln / 6], -1), n_per_bone // 2)):
        z = torch.randn((n, hd.shape[0], 3), generator=gen)
        local = z * sc[None]
        pts = torch.einsum("bij,nbj->nbi", R, local) + mid[None]
        out.append(pts.reshape(-1, 3))
    return torch.cat(out, 0)


def look_at_extrinsics(cam_pos, target, up=(0.0, 0.0, 1.0)):
    """World->camera (3,4), OpenCV axes (x right, y down, z forward)."""
    c, t = np.asarray(cam_pos, np.float64), np.asarray(target, np.float64)
    z = t - c
    z /= np.linalg.norm(z)
    upv = np.asarray(up, np.float64)
    if abs(np.dot(upv, z)) > 0.99:
        upv = np.array([0.0, 1.0, 0.0])
    x = np.cross(z, upv)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    R = np.stack([x, y, z], 0)
    return np.concatenate([R, (-R @ c)[:, None]], 1)


def make_cameras(n, target, radius=1.2, width=1920, height=1080, focal=2666.6666666667):
    """n cameras on a Fibonacci lattice over the upper 3/4 of a sphere, looking at `target`."""
    cams = []
    ga = math.pi * (3.0 - math.sqrt(5.0))
    for k in range(n):
        zc = 1.0 - (k + 0.5) / n * 1.5  # z from +1 down to -0.5
        r = math.sqrt(max(0.0, 1.0 - zc * zc))
        th = ga * k
        pos = np.asarray(target) + radius * np.array([r * math.cos(th), r * math.sin(th), zc])
        E = look_at_extrinsics(pos, target)
        K = np.array([[focal * width / 1920.0, 0, (width - 1) / 2.0],
                      [0, focal * width / 1920.0, (height - 1) / 2.0], [0, 0, 1.0]])
        cams.append(get_opengl_camera_attributes(K, E, width, height))
    return cams


def camera_table(cams, device):
    """(V,40) fp32 device table for the C ABI from camera dicts."""
    rows = []
    for c in cams:
        row = np.zeros(40, np.float32)
        row[0] = math.tan(c["fovx"] * 0.5)
        row[1] = math.tan(c["fovy"] * 0.5)
        row[2:18] = np.asarray(c["world_view_transform"], np.float32).reshape(-1)
        row[18:34] = np.asarray(c["full_proj_transform"], np.float32).reshape(-1)
        row[34:37] = np.asarray(c["camera_center"], np.float32)
        rows.append(row)
    return torch.from_numpy(np.stack(rows)).to(device)


def make_scene(n_gaussians=300000, kind="hand", seed=0, grid_res=128, n_cameras=8, width=1920, height=1080,
               cam_radius=1.2, device="cpu", n_poses=None, sigma_range=(5e-4, 4e-3)):
    """Returns a dict: params (leaf tensors, reference names/shapes), grid, grid_center,
    grid_scale, rest/posed transforms (P,20,4,4), transforms (P,21,4,4), cameras (list of
    dicts), bg.  kind = "hand" | "object" | "composite"."""
    gen = torch.Generator().manual_seed(seed)
    sk = load_skeleton()
    heads, tails, rest = sk["rest_heads"], sk["rest_tails"], sk["rest_matrixs"]
    dims, center, scale = grid_geometry(heads, tails, res=grid_res)
    n_hand = n_gaussians if kind == "hand" else (0 if kind == "object" else int(n_gaussians * 0.6))
    n_obj = n_gaussians - n_hand
    parts = []
    if n_hand:
        per = max(2, int(math.ceil(n_hand / 30.0)))
        pts = sample_on_bones(heads, tails, rest, per, gen)
        pts = pts[torch.randperm(pts.shape[0], generator=gen)[:n_hand]]
        lo = torch.tensor(center - 0.95 * scale)
        hi = torch.tensor(center + 0.95 * scale)
        pts = torch.max(torch.min(pts, hi), lo)  # keep inside the skin grid (reference prunes outliers)
        parts.append(pts)
    if n_obj:
        ctr = torch.tensor([0.06, 0.0, 0.10])
        pts = (torch.rand((n_obj, 3), generator=gen) - 0.5) * 0.15 + ctr + 0.03 * 0.1 * torch.randn((n_obj, 3), generator=gen)
        parts.append(pts)
    xyz = torch.cat(parts, 0).float()
    N = xyz.shape[0]
    lo_s, hi_s = math.log(sigma_range[0]), math.log(sigma_range[1])
    params = {
        "_xyz": xyz,
        "_scaling": (torch.rand((N, 3), generator=gen) * (hi_s - lo_s) + lo_s).float(),
        "_rotation": torch.randn((N, 4), generator=gen),
        "_opacity": 1.5 * torch.randn((N, 1), generator=gen),
        "_features_dc": torch.randn((N, 1, 3), generator=gen),
        "_features_rest": 0.1 * torch.randn((N, 15, 3), generator=gen),
    }
    params = {k: v.to(device) for k, v in params.items()}
    P = n_poses if n_poses is not None else n_cameras
    frames = [2, 1, 3, 0]
    posed = np.stack([sk["pose_matrixs"][frames[p % 4]] for p in range(P)])
    rest_t = torch.as_tensor(rest, dtype=torch.float32)
    T = torch.stack([bone_transforms(torch.as_tensor(posed[p], dtype=torch.float32), rest_t) for p in range(P)])
    centroid = (0.5 * (sk["pose_heads"][2] + sk["pose_tails"][2])).mean(0)
    cams = make_cameras(n_cameras, centroid, radius=cam_radius, width=width, height=height)
    # keypoints of every pose: cat(heads[:1], tails) (hand_dynamic.py:199-201)
    keyp = np.stack([np.concatenate([sk["pose_heads"][frames[p % 4]][:1], sk["pose_tails"][frames[p % 4]]], 0) for p in range(P)])
    scene = dict(params=params, N=N, n_hand=n_hand, kind=kind, grid_dims=dims,
                 keypoints=torch.as_tensor(keyp, dtype=torch.float32).to(device),
                 grid_center=torch.as_tensor(center).to(device), grid_scale=torch.as_tensor(scale).to(device),
                 rest=rest_t.to(device), posed=torch.as_tensor(posed, dtype=torch.float32).to(device),
                 transforms=T.to(device), cameras=cams, bg=torch.ones(3, device=device), width=width,
                 height=height, heads=heads, tails=tails)
    if n_hand:
        scene["grid"] = make_skin_grid(heads, tails, dims, center, scale, device=device)
    return scene


def make_masks(scene, posed_xyz, margin=9):
    """Synthetic segmentation masks (V,H,W,1) uint8, one per camera: the pixels within `margin` of the projection of
    any posed Gaussian mean (posed_xyz: (V,N,3) or (N,3)).  Pure data generation for the pruning-path tests."""
    W, H = scene["width"], scene["height"]
    out = []
    for v, cam in enumerate(scene["cameras"]):
        pts = posed_xyz[v] if posed_xyz.dim() == 3 else posed_xyz
        pts = pts.detach().cpu().double()
        K = torch.as_tensor(cam["K"], dtype=torch.float64)
        E = torch.as_tensor(cam["extr"], dtype=torch.float64)[:3, :4]
        q = (K @ E @ torch.cat([pts, torch.ones(pts.shape[0], 1, dtype=torch.float64)], 1).T).T
        uv = q[:, :2] / q[:, 2:]
        ok = (q[:, 2] > 0) & (uv[:, 0] >= 0) & (uv[:, 0] <= W - 1) & (uv[:, 1] >= 0) & (uv[:, 1] <= H - 1)
        img = torch.zeros((H, W), dtype=torch.float32)
        ij = uv[ok].long()
        img[ij[:, 1], ij[:, 0]] = 1.0
        k = 2 * margin + 1
        # (a square window: the maximum over rows, then over columns -- 2k instead of k*k comparisons per pixel)
        img = torch.nn.functional.max_pool2d(img[None, None], (k, 1), stride=1, padding=(margin, 0))
        img = torch.nn.functional.max_pool2d(img, (1, k), stride=1, padding=(0, margin))[0, 0]
        out.append(img.to(torch.uint8)[..., None])
    return torch.stack(out)
