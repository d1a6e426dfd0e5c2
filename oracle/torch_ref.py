"""CPU oracle (test infrastructure, NOT product code) for the torch half of the
MANUS hot path: skin-weight sampling, linear-blend skinning of means and
covariances, SH colour, camera matrices, FK and pin-hole projection.

This is a plain-PyTorch restatement of the reference's op sequence; every
function cites the reference lines it follows (paths relative to
/root/reference).  It is pinned to the reference by tests/golden/*.npz, which
were produced by importing the reference itself (tests/golden/make_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product path (manus_amd/) never does.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# ---------------------------------------------------------------------------
# SH basis constants (src/utils/sh_utils.py:26-43)
# ---------------------------------------------------------------------------
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
         -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658,
         0.3731763325901154, -0.4570457994644658, 1.445305721320277,
         -0.5900435899266435)


def eval_sh(deg, sh, dirs):
    """Real SH of degree <=3.  sh: (..., C, K), dirs: (..., 3) unit.
    Follows src/utils/sh_utils.py:57-104 (sign pattern :74-103)."""
    res = SH_C0 * sh[..., 0]
    if deg > 0:
        x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
        res = res - SH_C1 * y * sh[..., 1] + SH_C1 * z * sh[..., 2] - SH_C1 * x * sh[..., 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * sh[..., 4] + SH_C2[1] * yz * sh[..., 5]
                   + SH_C2[2] * (2.0 * zz - xx - yy) * sh[..., 6]
                   + SH_C2[3] * xz * sh[..., 7] + SH_C2[4] * (xx - yy) * sh[..., 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3 * xx - yy) * sh[..., 9]
                       + SH_C3[1] * xy * z * sh[..., 10]
                       + SH_C3[2] * y * (4 * zz - xx - yy) * sh[..., 11]
                       + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[..., 12]
                       + SH_C3[4] * x * (4 * zz - xx - yy) * sh[..., 13]
                       + SH_C3[5] * z * (xx - yy) * sh[..., 14]
                       + SH_C3[6] * x * (xx - 3 * yy) * sh[..., 15])
    return res


def rgb2sh(rgb):
    """src/utils/sh_utils.py:123."""
    return (rgb - 0.5) / SH_C0


# ---------------------------------------------------------------------------
# covariance from scale / rotation (src/models/gaussian.py:49-53,84-93;
# src/utils/gaussian_utils.py:279-314, 248-276)
# ---------------------------------------------------------------------------
def quat_to_rotmat(q_raw):
    """Normalise (r,x,y,z) and build R.  src/utils/gaussian_utils.py:279-302."""
    q = q_raw / torch.sqrt((q_raw * q_raw).sum(-1, keepdim=True))
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    rows = [1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
            2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
            2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]
    return torch.stack(rows, -1).reshape(-1, 3, 3)


def covariance_3x3(log_scale, q_raw, scale_mod=1.0):
    """Sigma = (R S)(R S)^T with S = diag(exp(log_scale)*mod).
    src/models/gaussian.py:49-53 + :62-64 (exp activation)."""
    s = torch.exp(log_scale) * scale_mod
    L = quat_to_rotmat(q_raw) * s[:, None, :]
    return L @ L.transpose(1, 2)


def pack_sym6(m):
    """[xx,xy,xz,yy,yz,zz] upper triangle.  src/utils/gaussian_utils.py:248-261."""
    return torch.stack([m[:, 0, 0], m[:, 0, 1], m[:, 0, 2], m[:, 1, 1], m[:, 1, 2], m[:, 2, 2]], -1)


# ---------------------------------------------------------------------------
# skin weights from the voxel grid (src/utils/gaussian_utils.py:167-196)
# ---------------------------------------------------------------------------
def skin_weights_from_grid(xyz, grid_center, grid_scale, grid_weights):
    """grid_weights: (D,H,W,B) channel-last, sampled trilinearly at
    u = (xyz - center)/scale with align_corners=True, zero padding, then
    renormalised to sum 1 (no epsilon, :183)."""
    g = grid_weights.permute(3, 0, 1, 2).unsqueeze(0)
    u = ((xyz - grid_center) / grid_scale).reshape(1, -1, 1, 1, 3)
    w = F.grid_sample(g, u, mode="bilinear", padding_mode="zeros", align_corners=True)
    w = w.reshape(g.shape[1], -1).T
    return w / w.sum(-1, keepdim=True)


def bone_transforms(posed, rest, background=True):
    """T_b = posed_b @ inv(rest_b), plus identity background transform.
    src/modules/hand_dynamic.py:93-102."""
    T = torch.einsum("nij,njk->nik", posed, torch.linalg.inv(rest))
    if background:
        T = torch.cat([T, torch.eye(4, dtype=T.dtype, device=T.device)[None]], 0)
    return T


def lbs_forward(xyz, log_scale, q_raw, skin_wts, transforms):
    """Blend transforms, skin means and covariances.
    src/modules/hand_dynamic.py:106-127.  Returns posed_xyz (N,3),
    posed_cov (N,6), tf (N,4,4)."""
    tf = torch.einsum("nb,bij->nij", skin_wts, transforms)
    xyz_h = F.pad(xyz, (0, 1), value=1.0)
    posed = torch.einsum("nij,nj->ni", tf, xyz_h)[:, :3]
    cov = covariance_3x3(log_scale, q_raw)
    R = tf[:, :3, :3]
    cov = R @ cov @ R.transpose(1, 2)
    return posed, pack_sym6(cov), tf


def sh_colors(posed_xyz, features, cano_xyz, cam_center, sh_degree=3, tf=None):
    """View-dependent colour.  features (N,16,3).  With tf the camera is pulled
    back to canonical space through inv(tf).  src/utils/gaussian_utils.py:431-449."""
    shs = features.transpose(1, 2).reshape(-1, 3, (sh_degree + 1) ** 2)
    cam = cam_center.reshape(1, 3).expand(features.shape[0], 3)
    if tf is not None:
        cam_h = F.pad(cam, (0, 1), value=1.0)
        cam_inv = torch.einsum("nij,nj->ni", torch.linalg.inv(tf), cam_h)[:, :3]
        d = cano_xyz - cam_inv
    else:
        d = posed_xyz - cam
    d = d / d.norm(dim=1, keepdim=True)
    return torch.clamp_min(eval_sh(sh_degree, shs, d) + 0.5, 0.0)


def hand_forward(params, grid, grid_center, grid_scale, posed, rest, cam_center):
    """Whole a1-a5 chain for the hand.  params: dict with _xyz,_scaling,_rotation,
    _features_dc,_features_rest,_opacity.  Returns dict of tensors."""
    w = skin_weights_from_grid(params["_xyz"], grid_center, grid_scale, grid)
    T = bone_transforms(posed, rest)
    pxyz, pcov, tf = lbs_forward(params["_xyz"], params["_scaling"], params["_rotation"], w, T)
    feats = torch.cat([params["_features_dc"], params["_features_rest"]], 1)
    col = sh_colors(pxyz, feats, params["_xyz"], cam_center, 3, tf)
    return dict(posed_xyz=pxyz, posed_cov=pcov, tf=tf, skin_wts=w, colors=col,
                opacity=torch.sigmoid(params["_opacity"]))


def object_forward(params, cam_center):
    """src/modules/object.py:32-41 (no LBS, tf=None)."""
    pxyz = params["_xyz"]
    pcov = pack_sym6(covariance_3x3(params["_scaling"], params["_rotation"]))
    feats = torch.cat([params["_features_dc"], params["_features_rest"]], 1)
    col = sh_colors(pxyz, feats, pxyz, cam_center, 3, None)
    return dict(posed_xyz=pxyz, posed_cov=pcov, colors=col,
                opacity=torch.sigmoid(params["_opacity"]))


def composite_forward(params, n_hand, grid, grid_center, grid_scale, posed, rest, cam_center):
    """src/modules/composite.py:50-78 followed by the colour step of render_gaussians (gaussian_utils.py:431-449):
    hand Gaussians (rows < n_hand) through the hand module, object Gaussians through the object module, everything
    concatenated, tf = the blended transforms for the hand and eye(4) for the object (:58-59), colours from the
    concatenated tensors with that tf."""
    ph = {k: v[:n_hand] for k, v in params.items()}
    po = {k: v[n_hand:] for k, v in params.items()}
    w = skin_weights_from_grid(ph["_xyz"], grid_center, grid_scale, grid)
    hx, hc, htf = lbs_forward(ph["_xyz"], ph["_scaling"], ph["_rotation"], w, bone_transforms(posed, rest))
    ox, oc = po["_xyz"], pack_sym6(covariance_3x3(po["_scaling"], po["_rotation"]))
    eye = torch.eye(4, dtype=htf.dtype)[None].repeat(ox.shape[0], 1, 1)
    pxyz, pcov, tf = torch.cat([hx, ox]), torch.cat([hc, oc]), torch.cat([htf, eye])
    cano = torch.cat([ph["_xyz"], po["_xyz"]])
    feats = torch.cat([torch.cat([ph["_features_dc"], ph["_features_rest"]], 1),
                       torch.cat([po["_features_dc"], po["_features_rest"]], 1)])
    col = sh_colors(pxyz, feats, cano, cam_center, 3, tf)
    return dict(posed_xyz=pxyz, posed_cov=pcov, tf=tf, skin_wts=w, colors=col,
                opacity=torch.sigmoid(torch.cat([ph["_opacity"], po["_opacity"]])))


# ---------------------------------------------------------------------------
# cameras (src/utils/cam_utils.py:19-78)
# ---------------------------------------------------------------------------
def projection_matrix(znear, zfar, fovx, fovy):
    """src/utils/cam_utils.py:19-39 (principal point dropped: l=-r, b=-t)."""
    ty, tx = math.tan(fovy / 2), math.tan(fovx / 2)
    top, right = ty * znear, tx * znear
    P = np.zeros((4, 4))
    P[0, 0] = 2.0 * znear / (2 * right)
    P[1, 1] = 2.0 * znear / (2 * top)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def camera_attributes(K, extr, width, height, zfar=100.0, znear=0.01):
    """src/utils/cam_utils.py:50-78.  extr (3,4).  float64 numpy out."""
    fovx = 2 * math.atan(width / (2 * K[0, 0]))
    fovy = 2 * math.atan(height / (2 * K[1, 1]))
    E = np.concatenate([extr, np.array([[0, 0, 0, 1.0]])], 0)
    wvt = E.T
    proj = projection_matrix(znear, zfar, fovx, fovy).T
    full = wvt @ proj
    center = np.linalg.inv(wvt)[3, :3]
    return dict(width=int(width), height=int(height), fovx=fovx, fovy=fovy, K=K, extr=E,
                world_view_transform=wvt, projection_matrix=proj,
                full_proj_transform=full, camera_center=center)


# ---------------------------------------------------------------------------
# FK and projection (src/utils/transforms.py)
# ---------------------------------------------------------------------------
def _axis_rot(axis, a):
    """src/utils/transforms.py:533-558."""
    c, s, o, z = torch.cos(a), torch.sin(a), torch.ones_like(a), torch.zeros_like(a)
    if axis == "X":
        flat = (o, z, z, z, c, -s, z, s, c)
    elif axis == "Y":
        flat = (c, z, s, z, o, z, -s, z, c)
    else:
        flat = (c, -s, z, s, c, z, z, z, o)
    return torch.stack(flat, -1).reshape(a.shape + (3, 3))


def euler_to_matrix(euler, convention="XYZ", intrinsic=False):
    """src/utils/transforms.py:489-530: intrinsic = reversed convention on flipped angles."""
    if intrinsic:
        convention = convention[::-1]
        euler = euler.flip(-1)
    m = [_axis_rot(c, e) for c, e in zip(convention, torch.unbind(euler, -1))]
    return m[0] @ m[1] @ m[2]


def fk_pose_wrt_root(rest, pose_rot, global_R, global_t, parents):
    """Kinematic-tree FK.  rest (20,4,4), pose_rot (B,20,3,3), global_R (B,3,3),
    global_t (B,3), parents int array (-1 = root).  src/utils/transforms.py:233-261:
    root: G @ rest_i @ pose_i ; child: M_parent @ inv(rest_parent) @ rest_i @ pose_i."""
    B = pose_rot.shape[0]
    pose = torch.zeros(B, pose_rot.shape[1], 4, 4, dtype=rest.dtype)
    pose[:, :, :3, :3] = pose_rot
    pose[:, :, 3, 3] = 1.0
    G = torch.zeros(B, 4, 4, dtype=rest.dtype)
    G[:, :3, :3] = global_R
    G[:, :3, 3] = global_t
    G[:, 3, 3] = 1.0
    M = [None] * len(parents)
    for i, p in enumerate(parents):
        if p == -1:
            M[i] = G @ rest[i][None] @ pose[:, i]
    for i, p in enumerate(parents):
        if p == -1:
            continue
        local = torch.linalg.inv(rest[p]) @ rest[i]
        M[i] = M[p] @ (local[None] @ pose[:, i])
    return torch.stack(M, 1)


def project_points(points, K, extr):
    """points (B,N,3), K (3,3), extr (3,4) -> (B,N,2).  src/utils/transforms.py:304-311."""
    P = K @ extr
    ph = F.pad(points, (0, 1), value=1.0)
    q = torch.einsum("ij,bnj->bni", P, ph)
    return (q / q[..., 2:])[..., :2]


# ---------------------------------------------------------------------------
# EWA projection of the external rasterizer's preprocess ("Forward per Gaussian (K1)") as a
# differentiable torch function: only what carries gradient (no radius / rectangle / culling).  Lets the torch
# chain (LBS, SH) be closed around a blend whose inputs were fixed, so that a kernel fusing all three can be
# checked on identical blend decisions.  Pinned to the scalar C oracle by tests/test_oracle_raster.py.
# ---------------------------------------------------------------------------
def project_ewa(means3D, cov6, W, H, tanfovx, tanfovy, view, proj):
    """means3D (N,3), cov6 (N,6) [xx,xy,xz,yy,yz,zz]; view / proj: the (4,4) world_view_transform /
    full_proj_transform tensors (element (row r, col c) of the math matrix = M.reshape(-1)[4c + r]).
    Returns ndc (N,2) (the quantity `means2D.grad` is taken against: pixel = ((ndc + 1) * size - 1) / 2) and
    conic (N,3) = (A, B, C).  The blend's dL/dB is the derivative w.r.t. ONE of the two symmetric off-diagonal entries
    (K7 accumulates -1/2 gdx dy dG, K8 doubles it again): contract the conic with `conic_grad_weights`."""
    v, P = view.reshape(-1), proj.reshape(-1)
    x, y, z = means3D[:, 0], means3D[:, 1], means3D[:, 2]
    hom = lambda M, r: M[r] * x + M[4 + r] * y + M[8 + r] * z + M[12 + r]
    pw = 1.0 / (hom(P, 3) + 0.0000001)
    ndc = torch.stack([hom(P, 0) * pw, hom(P, 1) * pw], -1)
    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    tx, ty, tz = hom(v, 0), hom(v, 1), hom(v, 2)
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    # Outside the frustum limits t.x becomes +-lim * t.z.  The published backward multiplies dL/dt.x by x_grad_mul = 0
    # there and keeps treating t.x as independent of t.z (Appendix A, K8): the clamped value carries NO gradient.
    rx, ry = tx / tz, ty / tz
    tx = torch.where((rx < -limx) | (rx > limx), (torch.clamp(rx, -limx, limx) * tz).detach(), tx)
    ty = torch.where((ry < -limy) | (ry > limy), (torch.clamp(ry, -limy, limy) * tz).detach(), ty)
    j00, j02 = fx / tz, -(fx * tx) / (tz * tz)
    j11, j12 = fy / tz, -(fy * ty) / (tz * tz)
    M0 = torch.stack([j00 * v[4 * c] + j02 * v[4 * c + 2] for c in range(3)], -1)
    M1 = torch.stack([j11 * v[4 * c + 1] + j12 * v[4 * c + 2] for c in range(3)], -1)
    S = torch.stack([cov6[:, 0], cov6[:, 1], cov6[:, 2], cov6[:, 1], cov6[:, 3], cov6[:, 4], cov6[:, 2], cov6[:, 4],
                     cov6[:, 5]], -1).reshape(-1, 3, 3)
    S0, S1 = torch.einsum("nij,nj->ni", S, M0), torch.einsum("nij,nj->ni", S, M1)
    a = (M0 * S0).sum(-1) + 0.3
    b = (M0 * S1).sum(-1)
    c = (M1 * S1).sum(-1) + 0.3
    det = a * c - b * b
    return ndc, torch.stack([c / det, -b / det, a / det], -1)


CONIC_GRAD_WEIGHTS = (1.0, 2.0, 1.0)   # sum_k w_k * conic_k * dconic_k closes the chain (see project_ewa)


def psnr(a, b):
    """src/utils/loss_utils.py:100-108."""
    mse = torch.mean((a - b) ** 2)
    return -10.0 * torch.log10(mse)


# ---------------------------------------------------------------------------
# image losses
# ---------------------------------------------------------------------------
def ssim_window(window_size=11, sigma=1.5):
    """gaussian(window_size, 1.5) and its outer product (loss_utils.py:39-54)."""
    g = torch.tensor([math.exp(-((x - window_size // 2) ** 2) / float(2 * sigma ** 2)) for x in range(window_size)],
                     dtype=torch.float32)
    g = g / g.sum()
    return g, (g[:, None] @ g[None, :]).float()


def ssim_hwc(img1, img2, window_size=11):
    """ssim() of loss_utils.py:57-97 as the reference calls it: on HWC images (base.py:347), so
    that `channel = img1.size(-3)` = H and the grouped conv slides the window over the (W,3)
    plane of every row (zero padding 5).  img1 (H,W,3), img2 (H,W,3) or (1,H,W,3); returns the
    mean of the SSIM map.  Restated with an explicit per-row convolution (no groups=H trick)."""
    if img2.dim() == 4:
        img2 = img2[0]
    H = img1.shape[0]
    _, w2d = ssim_window(window_size)
    w2d = w2d.to(img1.dtype)[None, None]
    pad = window_size // 2

    def filt(x):  # (H,W,3) -> every row is a 1-channel image of size (W,3)
        return torch.nn.functional.conv2d(x[:, None], w2d, padding=pad)[:, 0]

    mu1, mu2 = filt(img1), filt(img2)
    s11 = filt(img1 * img1) - mu1 * mu1
    s22 = filt(img2 * img2) - mu2 * mu2
    s12 = filt(img1 * img2) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s11 + s22 + C2))
    return m.mean()


def rgb_ssim_loss(pred_hwc, gt_hwc, w_rgb=0.8, w_ssim=0.2):
    """w_rgb * mean|pred-gt| + w_ssim * (1 - ssim): base.py:323-365 with HAND_GAUSSIAN.yaml:22-23."""
    gt = gt_hwc[0] if gt_hwc.dim() == 4 else gt_hwc
    return w_rgb * (pred_hwc - gt).abs().mean() + w_ssim * (1.0 - ssim_hwc(pred_hwc, gt))


# ---------------------------------------------------------------------------
# optimizer step, learning-rate schedule, densification
# ---------------------------------------------------------------------------
LEAVES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")   # group order of training_setup, gaussian.py:133-140


def expon_lr(step, lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """get_expon_lr_func(...)(step), gaussian_utils.py:212-245 (float64 like the numpy original)."""
    if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
        return 0.0
    delay = 1.0
    if lr_delay_steps > 0:
        delay = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
    t = np.clip(step / max_steps, 0, 1)
    return float(delay * np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t))


def group_lrs(opts, spatial_lr_scale, step):
    """Learning rates of the six groups at `step`: training_setup (gaussian.py:133-146) with the xyz rate
    replaced by the schedule (update_learning_rate, gaussian_utils.py:501-508; lr_delay_steps is left at
    its default 0, so position_lr_delay_mult has no effect -- as in the reference)."""
    return [expon_lr(step, opts["position_lr_init"] * spatial_lr_scale, opts["position_lr_final"] * spatial_lr_scale,
                     max_steps=opts["position_lr_max_steps"]),
            opts["feature_lr"], opts["feature_lr"] / 20.0, opts["opacity_lr"], opts["scaling_lr"], opts["rotation_lr"]]


def adam_step(p, g, m, v, lr, t, beta1=0.9, beta2=0.999, eps=1e-15):
    """One torch.optim.Adam update (no weight decay, no amsgrad) of the t-th step (t >= 1), the optimizer of
    gaussian.py:142 (Adam(l, lr=0, eps=1e-15)); returns new (p, m, v)."""
    m = m + (g - m) * (1 - beta1)
    v = v * beta2 + g * g * (1 - beta2)
    bc1, bc2 = 1 - beta1 ** t, 1 - beta2 ** t
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return p - (lr / bc1) * (m / denom), m, v


def _rot_from_raw_quat(r):
    q = r / r.norm(dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=1)
    return R.reshape(-1, 3, 3)


def densify_and_prune(state, accum, denom, max_grad, min_opacity, extent, percent_dense, noise, n_split=2,
                      max_screen_size=None):
    """densify_and_prune of gaussian.py:310-333 (clone :288-308, split :254-286, prune :183-200, optimizer
    surgery :148-252) on a dict of tensors:
        state[name], state[name + "_m"], state[name + "_v"] for name in LEAVES, state["skin"] (N,B) or None.
    noise: standard normals (n_split * n_selected, 3), row c * n_selected + j for copy c of the j-th split
    Gaussian (the layout of `torch.normal(mean, std)` at :264-266 divided by std).
    Returns the new state dict.  max_screen_size falsy: only low opacity (and NaN scales) prunes (`if
    max_screen_size:`, :316); set: also big_points_ws = max scale > 0.1 * extent (:318).  The other half of that
    branch (max_radii2D > max_screen_size, :317) never fires because densification_postfix has just zeroed
    max_radii2D (:249-251); it is therefore absent here.
    New rows get zero Adam moments; the statistics are reset by the caller (they are all zeros afterwards)."""
    N = state["xyz"].shape[0]
    grads = accum / denom
    grads[grads.isnan()] = 0.0
    gn = grads.reshape(N, -1).norm(dim=-1)
    smax = state["scaling"].exp().max(dim=1).values
    clone = (gn >= max_grad) & (smax <= percent_dense * extent)
    split = (grads.reshape(-1) >= max_grad) & (smax > percent_dense * extent)
    sel = torch.nonzero(split)[:, 0]
    ns = sel.shape[0]

    def rows(name, idx):
        return state[name][idx]

    new = {}
    std = state["scaling"].exp()[sel].repeat(n_split, 1)
    samples = noise.reshape(-1, 3)[: n_split * ns] * std
    R = _rot_from_raw_quat(state["rotation"][sel]).repeat(n_split, 1, 1)
    child_xyz = torch.bmm(R, samples[:, :, None])[:, :, 0] + state["xyz"][sel].repeat(n_split, 1)
    child_scaling = torch.log(state["scaling"].exp()[sel].repeat(n_split, 1) / (0.8 * n_split))
    keep_orig = ~split
    for name in LEAVES:
        p = state[name]
        rep = (n_split,) + (1,) * (p.dim() - 1)
        child = p[sel].repeat(*rep)
        if name == "xyz":
            child = child_xyz
        elif name == "scaling":
            child = child_scaling
        new[name] = torch.cat([p[keep_orig], p[clone], child])
        for suf in ("_m", "_v"):
            q = state[name + suf]
            new[name + suf] = torch.cat([q[keep_orig], torch.zeros_like(q[clone]), torch.zeros_like(child)])
    if state.get("skin") is not None:
        sk = state["skin"]
        new["skin"] = torch.cat([sk[keep_orig], sk[clone], sk[sel].repeat(n_split, 1)])
    prune = torch.sigmoid(new["opacity"]).reshape(-1) < min_opacity
    if max_screen_size:
        prune |= new["scaling"].exp().max(dim=1).values > 0.1 * extent
    if torch.isnan(new["scaling"].mean()):
        prune |= torch.isnan(new["scaling"]).any(dim=-1)
    keep = ~prune
    return {k: v[keep] for k, v in new.items()}


def prune_points(state, mask):
    """prune_points of gaussian.py:185-203 on the dict layout of densify_and_prune (every entry keeps the rows where
    mask is False; statistics `accum`, `denom`, `maxrad` included when present)."""
    keep = ~torch.as_tensor(mask).bool().reshape(-1)
    return {k: (v[keep] if v is not None else None) for k, v in state.items()}


def dilate_mask(mask, kernel_size=11):
    """dilate_mask of gaussian_utils.py:35-47: conv2d with a box of ones, zero padding, > 0."""
    k = torch.ones((1, 1, kernel_size, kernel_size), dtype=torch.float32)
    m = torch.nn.functional.conv2d(mask.float()[None, None], k, padding=kernel_size // 2)[0, 0]
    return m > 0


def points_outside_mask(points, K, extr, mask, keypoints=None, dilate=False):
    """get_points_outside_mask of gaussian_utils.py:101-147.  mask (H,W,1); returns (N,1) bool."""
    if dilate:
        mask = dilate_mask(mask[..., 0]).unsqueeze(-1).int()
    p2d = project_points(points[None], K, extr[:3, :4])[0]
    px = torch.clamp(p2d[..., 0], 0, mask.shape[1] - 1).int()
    py = torch.clamp(p2d[..., 1], 0, mask.shape[0] - 1).int()
    inv = ~mask.bool()
    val = inv[py.long(), px.long()]
    if keypoints is not None:
        k2d = project_points(keypoints[None], K, extr[:3, :4])[0]
        kx = torch.clamp(k2d[..., 0], 0, mask.shape[1] - 1).int()
        ky = torch.clamp(k2d[..., 1], 0, mask.shape[0] - 1).int()
        if torch.any(inv[ky.long(), kx.long()]):
            val = torch.zeros_like(val)
    return val


def reset_opacity(state):
    """reset_opacity of gaussian.py:148-151: opacity <- inverse_sigmoid(min(sigmoid(opacity), 0.01)), Adam moments
    of the opacity group zeroed (:153-165)."""
    o = torch.minimum(torch.sigmoid(state["opacity"]), torch.full_like(state["opacity"], 0.01))
    out = dict(state)
    out["opacity"] = torch.log(o / (1 - o))
    out["opacity_m"] = torch.zeros_like(state["opacity"])
    out["opacity_v"] = torch.zeros_like(state["opacity"])
    return out


# ---------------------------------------------------------------------------
# contact distance
# ---------------------------------------------------------------------------
def contact_dist(pt1, pt2, chunk=512):
    """get_contact_dist of gaussian_utils.py:521-549 (the taichi loop) in fp32 numpy: per point of pt1 the
    rooted distance to the nearest point of pt2 and the lowest index attaining it (strict '<' in the loop;
    np.argmin returns the first minimum), min_dist initialised to 1e9.  Sum order dx^2 + dy^2 + dz^2.
    The taichi package is not in this image, so this row is pinned by the reference's other implementation of
    the same quantity, get_contact_map (torch.cdist), through tests/golden/contact.npz (distances only)."""
    a, b = np.asarray(pt1, np.float32), np.asarray(pt2, np.float32)
    n1, n2 = a.shape[0], b.shape[0]
    dist = np.full(n1, 1e9, np.float32)
    idx = np.zeros(n1, np.int64)
    if n2 == 0:
        return dist, idx
    for s in range(0, n1, chunk):
        d = a[s:s + chunk, None, :] - b[None, :, :]
        d2 = d[..., 0] * d[..., 0]
        d2 = d2 + d[..., 1] * d[..., 1]
        d2 = d2 + d[..., 2] * d[..., 2]
        r = np.sqrt(d2)
        j = np.argmin(r, axis=1)
        m = r[np.arange(r.shape[0]), j]
        ok = m < 1e9
        dist[s:s + chunk] = np.where(ok, m, np.float32(1e9))
        idx[s:s + chunk] = np.where(ok, j, 0)
    return dist, idx


def isotropic_reg(log_scale, condition_number=0.4):
    """isotropic_reg of loss_func, base.py:349-356, on the log-scales (get_scaling = exp(_scaling), gaussian.py:62-67)."""
    s = torch.exp(log_scale)
    return torch.mean((s.min(dim=1)[0] / (s.max(dim=1)[0] + 1e-8) - condition_number) ** 2)
