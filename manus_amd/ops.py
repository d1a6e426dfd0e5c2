"""Autograd operators over the articulation kernels of libmanus_hip.so.

Each op mirrors one block of MANUS Python code (reference tree brown-ivl/manus):

    skin_weights   skinning_weights_from_voxel_grid      src/utils/gaussian_utils.py:167-196
    lbs_cov        TrainingModule.forward LBS block      src/modules/hand_dynamic.py:106-127
                   + GaussianModel.get_covariance        src/models/gaussian.py:49-53,84-93
    sh_colors      calculate_colors_from_sh / eval_sh    src/utils/gaussian_utils.py:431-449
    project_points project_points                        src/utils/transforms.py:304-311
    distCUDA2      simple_knn._C.distCUDA2               src/models/gaussian.py:110

All of them require GPU tensors; there is no CPU or PyTorch fallback.
"""
import ctypes

import torch

from ._lib import MGR_MAX_BONES, ManusHipError, check, f32c, lib, ptr, stream


class SkinGrid:
    """Skin-weight voxel grid prepared for the kernels: the reference's channel-last (D,H,W,B)
    tensor, uploaded once and zero-padded to 24 channels (96 B per voxel) so that every trilinear
    corner is six aligned float4 loads."""

    def __init__(self, grid_weights, device=None):
        g = torch.as_tensor(grid_weights, dtype=torch.float32)
        if device is not None:
            g = g.to(device)
        self.D, self.H, self.W, self.B = g.shape
        if self.B > MGR_MAX_BONES:
            raise ManusHipError("skin grid: at most %d transforms" % MGR_MAX_BONES)
        if self.B <= 24:
            self.stride = 24
            p = torch.zeros((self.D, self.H, self.W, 24), dtype=torch.float32, device=g.device)
            p[..., : self.B] = g
            self.data = p.contiguous()
        else:
            self.stride = self.B
            self.data = g.contiguous()


class _SkinWeights(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, sg, center, scale):
        xyz = f32c(xyz)
        center, scale = f32c(center).reshape(-1), f32c(scale).reshape(-1)
        N = xyz.shape[0]
        w = torch.empty((N, sg.B), dtype=torch.float32, device=xyz.device)
        check(lib().mgr_skin_weights_fwd(N, ptr(xyz), ptr(sg.data), sg.D, sg.H, sg.W, sg.B, sg.stride, ptr(center),
                                         ptr(scale), ptr(w), stream()), "mgr_skin_weights_fwd")
        ctx.save_for_backward(xyz, center, scale)
        ctx.sg = sg
        return w

    @staticmethod
    def backward(ctx, g_w):
        xyz, center, scale = ctx.saved_tensors
        sg = ctx.sg
        N = xyz.shape[0]
        g_w = f32c(g_w)
        g_xyz = torch.empty((N, 3), dtype=torch.float32, device=xyz.device)
        check(lib().mgr_skin_weights_bwd(N, ptr(xyz), ptr(sg.data), sg.D, sg.H, sg.W, sg.B, sg.stride, ptr(center),
                                         ptr(scale), ptr(g_w), ptr(g_xyz), 0, stream()), "mgr_skin_weights_bwd")
        return g_xyz, None, None, None


def skin_weights(xyz, grid_weights, grid_center, grid_scale):
    """xyz (N,3); grid_weights: a `SkinGrid` (prepared once) or the reference's (D,H,W,B)
    channel-last tensor (prepared on the fly) -> (N,B), rows sum to 1."""
    if not isinstance(grid_weights, SkinGrid):
        if not xyz.is_cuda:
            raise ManusHipError("manus_amd ops need GPU tensors; there is no CPU fallback")
        grid_weights = SkinGrid(grid_weights, xyz.device)
    return _SkinWeights.apply(xyz, grid_weights, grid_center, grid_scale)


class _LbsCov(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, log_scale, rot, skin_w, transforms, tf44=False):
        xyz, log_scale, rot = f32c(xyz), f32c(log_scale), f32c(rot)
        N = xyz.shape[0]
        if skin_w is not None:
            skin_w = f32c(skin_w)
            transforms = f32c(transforms)
            if transforms.dim() == 3:
                transforms = transforms[None]
            P, B = transforms.shape[0], transforms.shape[1]
            if skin_w.shape[1] != B:
                raise ManusHipError("lbs_cov: skin weights have %d columns, %d transforms given"
                                    % (skin_w.shape[1], B))  # hand_dynamic.py:104
        else:
            P, B = 1, 0
        dev = xyz.device
        pxyz = torch.empty((P, N, 3), dtype=torch.float32, device=dev)
        pcov = torch.empty((P, N, 6), dtype=torch.float32, device=dev)
        rows = 16 if tf44 else 12     # (4x4 rows: the reference's (N,4,4) layout written by the kernel, no cat of the constant row)
        tf = torch.empty((P, N, 4, 4) if tf44 else (P, N, 12), dtype=torch.float32, device=dev)
        check(lib().mgr_lbs_cov_fwd_rows(P, N, B, ptr(xyz), ptr(log_scale), ptr(rot), ptr(skin_w), ptr(transforms),
                                         ptr(pxyz), ptr(pcov), ptr(tf), rows, stream()), "mgr_lbs_cov_fwd")
        ctx.save_for_backward(xyz, log_scale, rot, skin_w, transforms)
        ctx.meta = (P, N, B, rows)
        return pxyz, pcov, tf

    @staticmethod
    def backward(ctx, g_xyz, g_cov, g_tf):
        xyz, log_scale, rot, skin_w, transforms = ctx.saved_tensors
        P, N, B, rows = ctx.meta
        dev = xyz.device
        g_xyz = f32c(g_xyz) if g_xyz is not None else torch.zeros((P, N, 3), dtype=torch.float32, device=dev)
        g_cov = f32c(g_cov) if g_cov is not None else torch.zeros((P, N, 6), dtype=torch.float32, device=dev)
        g_tf = f32c(g_tf) if g_tf is not None else None
        d_xyz = torch.empty((N, 3), dtype=torch.float32, device=dev)
        d_ls = torch.empty((N, 3), dtype=torch.float32, device=dev)
        d_rot = torch.empty((N, 4), dtype=torch.float32, device=dev)
        d_w = torch.empty((N, B), dtype=torch.float32, device=dev) if skin_w is not None else None
        check(lib().mgr_lbs_cov_bwd_rows(P, N, B, ptr(xyz), ptr(log_scale), ptr(rot), ptr(skin_w), ptr(transforms),
                                         ptr(g_xyz), ptr(g_cov), ptr(g_tf), rows, ptr(d_xyz), ptr(d_ls), ptr(d_rot),
                                         ptr(d_w), stream()), "mgr_lbs_cov_bwd")
        return d_xyz, d_ls, d_rot, d_w, None, None


def lbs_cov(xyz, log_scale, rot, skin_w, transforms, tf44=False):
    """Skin means and covariances for P poses.

    xyz (N,3), log_scale (N,3) (`_scaling`), rot (N,4) raw (`_rotation`),
    skin_w (N,B) or None (static object: identity transform),
    transforms (P,B,4,4) / (B,4,4) = posed @ inv(rest) (+ identity background).
    Returns posed_xyz (P,N,3), posed_cov (P,N,6), tf (P,N,12) (rows 0..2 of the
    blended 4x4) -- or, with tf44, (P,N,4,4): the reference's layout, constant last row included, written by the kernel."""
    return _LbsCov.apply(xyz, log_scale, rot, skin_w, transforms, bool(tf44))


class _ShColors(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sh, xyz, tf, cams):
        sh, xyz = f32c(sh), f32c(xyz)
        N = sh.shape[0]
        V = cams.shape[0]
        if sh.shape[1:] != (16, 3):
            raise ManusHipError("sh_colors: features must be (N,16,3) (sh_degree 3)")
        s_xyz = xyz.stride(0) if xyz.dim() == 3 else 0
        s_tf, rows, tf_per_view = 0, 12, False
        if tf is not None:
            tf = f32c(tf)
            if tf.dim() >= 3 and tuple(tf.shape[-2:]) == (4, 4):      # (N,4,4) / (V,N,4,4): the reference's layout, read in place
                rows, tf_per_view = 16, tf.dim() == 4
            else:                                                     # (N,12) / (V,N,12)
                tf_per_view = tf.dim() == 3
            s_tf = tf.stride(0) if tf_per_view else 0
        col = torch.empty((V, N, 3), dtype=torch.float32, device=sh.device)
        check(lib().mgr_sh_color_fwd_rows(V, N, ptr(sh), ptr(xyz), s_xyz, ptr(tf), s_tf, rows, ptr(cams), ptr(col),
                                          stream()), "mgr_sh_color_fwd")
        ctx.save_for_backward(sh, xyz, tf, cams)
        ctx.meta = (V, N, s_xyz, s_tf, rows, tf_per_view)
        return col

    @staticmethod
    def backward(ctx, g_col):
        sh, xyz, tf, cams = ctx.saved_tensors
        V, N, s_xyz, s_tf, rows, tf_per_view = ctx.meta
        dev = sh.device
        g_col = f32c(g_col)
        d_sh = torch.empty((N, 16, 3), dtype=torch.float32, device=dev)
        d_xyz = torch.empty((V, N, 3), dtype=torch.float32, device=dev)
        d_tf = torch.empty((V, N, 4, 4) if rows == 16 else (V, N, 12), dtype=torch.float32, device=dev) if tf is not None else None
        check(lib().mgr_sh_color_bwd_rows(V, N, ptr(sh), ptr(xyz), s_xyz, ptr(tf), s_tf, rows, ptr(cams), ptr(g_col),
                                          ptr(d_sh), ptr(d_xyz), ptr(d_tf), stream()), "mgr_sh_color_bwd")
        g_xyz = d_xyz if xyz.dim() == 3 else (d_xyz.sum(0) if V > 1 else d_xyz[0])
        g_tf = None
        if tf is not None:
            g_tf = d_tf if tf_per_view else (d_tf.sum(0) if V > 1 else d_tf[0])
        return d_sh, g_xyz, g_tf, None


def sh_colors(features, xyz, tf, cams):
    """Degree-3 SH colour for V views: features (N,16,3); xyz (N,3) or (V,N,3)
    (canonical means when tf is given, posed means otherwise); tf (N,12)/(V,N,12)
    or None; cams (V,40).  Returns (V,N,3) = max(sh2rgb + 0.5, 0)."""
    return _ShColors.apply(features, xyz, tf, cams)


def project_points(points, K, extrin):
    """points (B,N,3) or (N,3); K (3,3); extrin (3,4) -> (...,N,2)."""
    p = f32c(points)
    shape = p.shape
    flat = p.reshape(-1, 3)
    K, E = f32c(K).reshape(-1)[:9].contiguous(), f32c(extrin).reshape(-1)[:12].contiguous()
    uv = torch.empty((flat.shape[0], 2), dtype=torch.float32, device=p.device)
    check(lib().mgr_project_points(flat.shape[0], ptr(flat), ptr(K), ptr(E), ptr(uv), stream()),
          "mgr_project_points")
    return uv.reshape(shape[:-1] + (2,))


def distCUDA2(points):
    """Mean squared distance to the 3 nearest other points, (N,3) -> (N,)."""
    p = f32c(points)
    N = p.shape[0]
    out = torch.empty((N,), dtype=torch.float32, device=p.device)
    nbytes = int(lib().mgr_knn3_workspace_bytes(N))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=p.device)
    check(lib().mgr_knn3_mean_dist2(N, ptr(p), ptr(out), ptr(ws), nbytes, stream()), "mgr_knn3_mean_dist2")
    return out


def l1_loss_grad(pred, target, scale=None):
    """sum|pred-target| (device scalar) and d(mean|pred-target|)/dpred * weight.
    `scale` defaults to 1/numel (the reference's torch.mean of the L1 map)."""
    pred, target = f32c(pred), f32c(target)
    n = pred.numel()
    if scale is None:
        scale = 1.0 / n
    g = torch.empty_like(pred)
    s = torch.zeros(1, dtype=torch.float32, device=pred.device)
    check(lib().mgr_l1_loss_grad(n, ptr(pred), ptr(target), float(scale), ptr(g), ptr(s), stream()),
          "mgr_l1_loss_grad")
    return s, g


def image_loss_grad(pred, target, w_l1=0.8, w_ssim=0.2, grad_scale=1.0, loss_offset=0.0, bg=None, tile_start_ptr=None):
    """Fused L1 + SSIM image loss of the training step (loss_utils.py:22-97 as called at
    base.py:323-365), forward and backward, on (V,3,H,W) images.

    Returns (sums, grad): sums[0] = sum|pred-target|, sums[1] = sum of the SSIM map, sums[2] =
    grad_scale * (w_l1 * sums[0] - w_ssim * sums[1]) + loss_offset (device tensor of 3 floats); grad = grad_scale * d/dpred [w_l1 * sum|pred-target| - w_ssim * sum ssim_map].
    The SSIM statistic is the reference's: ssim() called on HWC images, i.e. the 11x11 window
    slides over the (W,3) plane of every row.

    bg (3 floats on the device) + tile_start_ptr (device address of the tile-list offsets of the forward that rendered
    `pred`): `mgr_image_loss_tiles` -- spans under empty tiles are settled from the target alone and their gradient is
    left unwritten (nothing reads it); the sums are the same."""
    pred, target = f32c(pred), f32c(target)
    if pred.dim() == 3:
        pred, target = pred[None], target[None]
    if pred.shape != target.shape or pred.shape[1] != 3:
        raise ManusHipError("image_loss_grad: pred and target must both be (V,3,H,W)")
    V, _, H, W = pred.shape
    g = torch.empty_like(pred)
    sums = torch.empty(3, dtype=torch.float32, device=pred.device)
    nbytes = int(lib().mgr_image_loss_workspace_bytes(V, H, W))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=pred.device)
    if tile_start_ptr is not None and bg is not None:
        check(lib().mgr_image_loss_tiles(V, H, W, ptr(pred), ptr(target), ptr(f32c(bg)), ctypes.c_void_p(int(tile_start_ptr)),
                                         float(w_l1), float(w_ssim), float(grad_scale), float(loss_offset), ptr(g), ptr(sums),
                                         ptr(ws), nbytes, stream()), "mgr_image_loss_tiles")
        return sums, g
    check(lib().mgr_image_loss(V, H, W, ptr(pred), ptr(target), float(w_l1), float(w_ssim), float(grad_scale),
                               float(loss_offset), ptr(g), ptr(sums), ptr(ws), nbytes, stream()), "mgr_image_loss")
    return sums, g


def isotropic_reg_grad(log_scale, condition_number=0.4, weight=1.0, grad_out=None):
    """weight * mean((min s / (max s + 1e-8) - condition_number)^2), s = exp(log_scale), and its gradient
    w.r.t. log_scale (base.py:349-356).  grad_out: optional (N,3) tensor to ADD the gradient to.
    Returns (loss (1,), gradient tensor)."""
    ls = f32c(log_scale)
    N = ls.shape[0]
    acc = grad_out is not None
    g = grad_out if acc else torch.empty_like(ls)
    if acc and (g.dtype != torch.float32 or not g.is_contiguous() or g.shape != ls.shape):
        raise ManusHipError("isotropic_reg_grad: grad_out must be a contiguous fp32 (N,3) tensor")
    loss = torch.empty(1, dtype=torch.float32, device=ls.device)
    nbytes = int(lib().mgr_isotropic_reg_workspace_bytes(N))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=ls.device)
    check(lib().mgr_isotropic_reg(N, ptr(ls), float(condition_number), float(weight), ptr(g), 1 if acc else 0, ptr(loss),
                                  ptr(ws), nbytes, stream()), "mgr_isotropic_reg")
    return loss, g
