"""Skin-weight initialisation from the MANO rest mesh over the HIP kernels of `csrc/mesh.hip` (the
model-initialisation side of the dataloader).

Mirrors brown-ivl/manus:

    create_skinning_grid(d, h, w)                          src/utils/extra.py:258-278
    init_mano_weights(points, data, neighbors, filter)     src/utils/train_utils.py:48-89
    build_voxel_grid(bones_rest, mano_data, ...)           src/datasets/brics_dynamic.py:99-144  (Dataset method)
    sample_gaussians_on_bones(bones_rest, mano_data, ...)  src/datasets/brics_dynamic.py:69-97   (Dataset method; the .ply
                                                           dumps and the colour visualisation stay with the caller)

The nearest-vertex search and the mesh signed distance run on the GPU (3.2 M grid points x 778 vertices / 1538 faces for
the default grid); the few array operations around them (bone remap, the "outside" channel, the normalisation) are the
reference's numpy lines, in the reference's dtypes (float64 once the extra channel is concatenated).  There is no CPU
fallback.  The signed distance replaces the external `pysdf` package, which is not in this image: its sign convention
(positive inside) is kept, its numbers are unpinned -- see `include/manus_hip.h`.
"""
import math

import numpy as np
import torch

from ._lib import ManusHipError, check, f32c, lib, ptr, stream

# MANO's 16 joints -> the 20 bones of the capture skeleton (train_utils.py:68)
MANO_TO_OURS = [13, 14, 14, 15, 0, 1, 2, 3, 0, 4, 5, 6, 0, 10, 11, 12, 0, 7, 8, 9]
SDF_THRESHOLD = -0.02   # points more than 2 cm outside the mesh belong to the background channel (train_utils.py:57)


def _dev(device=None):
    if not torch.cuda.is_available():
        raise ManusHipError("manus_amd.mano_init needs a GPU; there is no CPU fallback")
    return torch.device(device or "cuda:0")


def create_skinning_grid(d, h, w, device="cpu"):
    """(d*h*w, 3) points of the [-1, 1]^3 lattice, x fastest, as (x, y, z) (extra.py:258-278)."""
    x = torch.linspace(-1, 1, steps=w, device=device).view(1, 1, 1, w).expand(1, d, h, w)
    y = torch.linspace(-1, 1, steps=h, device=device).view(1, 1, h, 1).expand(1, d, h, w)
    z = torch.linspace(-1, 1, steps=d, device=device).view(1, d, 1, 1).expand(1, d, h, w)
    return torch.cat((x, y, z), dim=0).reshape(1, 3, -1).permute(0, 2, 1)[0]


def knn_mean_rows(points, refs, rows, k, want_idx=False):
    """Mean of the `rows` (m, C) of the k nearest `refs` (m, 3) of every point (n, 3); optionally the indices (n, k)."""
    points, refs, rows = f32c(points), f32c(refs), f32c(rows)
    if not points.is_cuda:
        raise ManusHipError("knn_mean_rows needs GPU tensors; there is no CPU fallback")
    n, m, C = points.shape[0], refs.shape[0], rows.shape[1]
    if rows.shape[0] != m:
        raise ManusHipError("knn_mean_rows: one row per reference point")
    out = torch.empty((n, C), dtype=torch.float32, device=points.device)
    idx = torch.empty((n, k), dtype=torch.int32, device=points.device) if want_idx else None
    check(lib().mgr_knn_mean_rows(n, ptr(points), m, ptr(refs), ptr(rows), C, int(k), ptr(out), ptr(idx), stream()),
          "mgr_knn_mean_rows")
    return (out, idx) if want_idx else out


def mesh_sdf(points, verts, faces, want_winding=False):
    """Signed distance (positive inside) of the points (n, 3) to the mesh verts (nv, 3) / faces (nf, 3)."""
    points, verts = f32c(points), f32c(verts)
    if not points.is_cuda:
        raise ManusHipError("mesh_sdf needs GPU tensors; there is no CPU fallback")
    faces = faces.to(device=points.device, dtype=torch.int32).contiguous()
    if faces.numel() and (int(faces.min()) < 0 or int(faces.max()) >= verts.shape[0]):
        raise ManusHipError("mesh_sdf: face index out of range")
    n = points.shape[0]
    sdf = torch.empty((n,), dtype=torch.float32, device=points.device)
    wind = torch.empty((n,), dtype=torch.float32, device=points.device) if want_winding else None
    check(lib().mgr_mesh_sdf(n, ptr(points), verts.shape[0], ptr(verts), faces.shape[0], ptr(faces), ptr(sdf), ptr(wind),
                             stream()), "mgr_mesh_sdf")
    return (sdf, wind) if want_winding else sdf


def init_mano_weights(points, data, neighbors=20, filter_grid=True, device=None):
    """train_utils.py:48-89.  `data`: {"verts" (778,3), "weights" (778,16), "face" (1538,3)}.  Returns (weights, mask) as
    numpy: weights (n, 20) float32, or (n, 21) float64 with the background channel when `filter_grid` (mask = points not
    farther than 2 cm outside the mesh; None otherwise) -- the reference's dtypes.  The mean over the k rows is formed in
    fp32 on the device and cast; the reference averages in the dtype of the MANO weights (float32 in `mano_rest.pkl`; with
    float64 weights its last bits would differ from this by fp32 rounding, ~6e-8 relative)."""
    dev = _dev(device)
    verts = torch.as_tensor(np.asarray(data["verts"]), dtype=torch.float32, device=dev)
    init_weights = np.asarray(data["weights"])[..., MANO_TO_OURS]
    pts = torch.as_tensor(points, dtype=torch.float32).to(dev)
    mean = knn_mean_rows(pts, verts, torch.as_tensor(np.ascontiguousarray(init_weights), dtype=torch.float32, device=dev),
                         neighbors)
    weights = mean.cpu().numpy().astype(init_weights.dtype, copy=False)
    mask = None
    if filter_grid:
        faces = torch.as_tensor(np.asarray(data["face"]).astype(np.int64), device=dev)
        sdf_value = mesh_sdf(pts, verts, faces).cpu().numpy()
        mask = sdf_value > SDF_THRESHOLD
        weights = np.concatenate([weights, np.zeros((weights.shape[0], 1))], axis=-1)
        outside = sdf_value < SDF_THRESHOLD
        weights[outside] = 0
        weights[outside, -1] = 1
    weights = weights / np.sum(weights, axis=-1, keepdims=True)
    return weights, mask


def build_voxel_grid(bones_rest, mano_data, grid_boundary=(-1, 1), res=128, ratio=(1, 1, 1), offset=(0, 0, 0), device=None):
    """brics_dynamic.py:99-144: the skin-weight voxel grid around the rest skeleton.  Returns
    (scale (1,3), center (3,), grid_points (D,H,W,3), weights (D,H,W,21), mask (D,H,W)) as float32 / bool tensors on the
    CPU, D,H,W = floor(res / ratio) reversed -- what HandGaussianModel stores and `ops.SkinGrid` uploads."""
    heads = np.asarray(bones_rest.heads, dtype=np.float32)
    tails = np.asarray(bones_rest.tails, dtype=np.float32)
    keypts = np.concatenate([heads[:1], tails], axis=0)
    cano_min, cano_max = np.min(keypts, axis=0), np.max(keypts, axis=0)
    center = (cano_max + cano_min) / 2
    center += np.array(offset)                          # (in place: stays float32, like the reference)
    x_ratio, y_ratio, z_ratio = ratio
    res_scaled = (res / np.array([x_ratio, y_ratio, z_ratio])).astype(np.int32)
    d, h, w = math.floor(res_scaled[2]), math.floor(res_scaled[1]), math.floor(res_scaled[0])
    grid_points = create_skinning_grid(d, h, w)
    scale = np.linalg.norm(cano_max - cano_min) / 2
    scale = np.array([[scale * z_ratio, scale * y_ratio, scale * x_ratio]]).astype(np.float32)
    grid_points = grid_points * torch.tensor(scale) + torch.tensor(center)
    weights, mask = init_mano_weights(grid_points, mano_data, neighbors=4, device=device)
    shape = (int(res_scaled[2]), int(res_scaled[1]), int(res_scaled[0]))
    return (torch.tensor(scale, dtype=torch.float32), torch.tensor(center, dtype=torch.float32),
            grid_points.reshape(shape + (3,)), torch.tensor(weights, dtype=torch.float32).reshape(shape + (-1,)),
            None if mask is None else torch.tensor(mask).reshape(shape))
