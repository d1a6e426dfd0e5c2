"""Shared helpers for the test-suite (scene builders, error metrics)."""
import math

import numpy as np
import torch

from manus_amd.cam_utils import get_opengl_camera_attributes
from manus_amd.synthetic import look_at_extrinsics


def psnr(a, b):
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))
    return float("inf") if mse == 0 else -10.0 * math.log10(mse)


def max_rel_err(a, b):
    """max|a-b| / max|b|  (per tensor; the definition used for 'grad max-rel-err')."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    den = float(np.max(np.abs(b)))
    if den == 0:
        return float(np.max(np.abs(a)))
    return float(np.max(np.abs(a - b))) / den


def make_camera(W, H, pos=(0.3, -0.2, -1.5), target=(0, 0, 0), focal=None):
    focal = focal if focal is not None else 1.2 * W
    K = np.array([[focal, 0, (W - 1) / 2.0], [0, focal, (H - 1) / 2.0], [0, 0, 1.0]])
    E = look_at_extrinsics(pos, target, up=(0, 1, 0))
    return get_opengl_camera_attributes(K, E, W, H)


def cam_args(cam):
    return dict(W=cam["width"], H=cam["height"], tanfovx=math.tan(cam["fovx"] / 2),
                tanfovy=math.tan(cam["fovy"] / 2),
                view=np.asarray(cam["world_view_transform"], np.float32).reshape(-1),
                proj=np.asarray(cam["full_proj_transform"], np.float32).reshape(-1))


def random_gaussians(n, seed=0, spread=0.35, sigma=(0.01, 0.06), opacity=(0.05, 0.95)):
    """Random anisotropic Gaussians around the origin: means (n,3), cov (n,6), colors, opacity."""
    g = np.random.default_rng(seed)
    means = (g.normal(size=(n, 3)) * spread).astype(np.float32)
    s = g.uniform(sigma[0], sigma[1], size=(n, 3))
    q = g.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    r, x, y, z = q.T
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                  2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                  2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(n, 3, 3)
    L = R * s[:, None, :]
    S = L @ L.transpose(0, 2, 1)
    cov = np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], -1).astype(np.float32)
    colors = g.uniform(0, 1, size=(n, 3)).astype(np.float32)
    op = g.uniform(opacity[0], opacity[1], size=(n,)).astype(np.float32)
    return means, cov, colors, op


def cam_table_np(cams):
    rows = []
    for c in cams:
        row = np.zeros(40, np.float32)
        row[0] = math.tan(c["fovx"] * 0.5)
        row[1] = math.tan(c["fovy"] * 0.5)
        row[2:18] = np.asarray(c["world_view_transform"], np.float32).reshape(-1)
        row[18:34] = np.asarray(c["full_proj_transform"], np.float32).reshape(-1)
        row[34:37] = np.asarray(c["camera_center"], np.float32)
        rows.append(row)
    return np.stack(rows)


def t(a, device="cpu", grad=False):
    x = torch.as_tensor(np.asarray(a)).to(device)
    if grad:
        x = x.clone().requires_grad_(True)
    return x


def keep(out):
    """Deep copy of a step's output dict.  HipViewCompute keeps its gradient / statistics buffers by default (the tensors
    of an output dict are the same storage every step, like .grad): what is compared across calls of ONE compute object
    must be cloned first."""
    import torch
    return {k: ({n: g.clone() for n, g in v.items()} if isinstance(v, dict) else (v.clone() if torch.is_tensor(v) else v))
            for k, v in out.items()}
