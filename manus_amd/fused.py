"""Fused articulated render of V views: canonical MANUS parameters -> images.

`render_views` is one autograd node over `mgr_views_forward/backward`: per (view, Gaussian)
the kernels do LBS of mean and covariance, SH colour with the view direction pulled back to
canonical space, sigmoid opacity, projection, binning, per-tile sort, compositing; the
backward gathers the pair gradients and runs the projection / SH / LBS backward in registers,
summing over views, so that none of the reference's intermediate tensors (posed_xyz,
posed_cov, tf, colors_precomp; src/modules/hand_dynamic.py:128-137, gaussian_utils.py:431-449)
ever exists in HBM.  Results equal the modular ops (`ops.lbs_cov` + `ops.sh_colors` +
`rasterizer.rasterize_views`) — tests/test_gpu_fused.py.
"""
import torch

from . import rasterizer as _rz
from ._lib import check, f32c, lib, ptr, stream


class ViewStats:
    """Densification statistics produced by the backward pass (src/models/gaussian.py:335-338,
    src/utils/gaussian_utils.py:469-471): grad2d = sum_v ||dL_v/dmeans2D[:, :2]||,
    vis = number of views in which the Gaussian is visible, radii = max screen radius."""

    def __init__(self):
        self.grad2d = self.vis = self.radii = None


class _RenderViews(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, log_scale, rot, opacity, f_dc, f_rest, skin_w, transforms, cams, bg, W, H, stats,
                grad2d_scale, grad_arena=None):
        # skin_w may cover only the first rows (hand + object composite): the rest are static
        xyz, log_scale, rot = f32c(xyz), f32c(log_scale), f32c(rot)
        opacity, f_dc, f_rest = f32c(opacity).reshape(-1), f32c(f_dc), f32c(f_rest)
        N, V = xyz.shape[0], cams.shape[0]
        B = n_art = 0
        if skin_w is not None:
            skin_w, transforms = f32c(skin_w), f32c(transforms)
            n_art, B = skin_w.shape
            if n_art > N:
                raise _rz._lib.ManusHipError("render_views: more skin-weight rows than Gaussians")
            if transforms.shape[0] != V or transforms.shape[1] != B:
                raise _rz._lib.ManusHipError("render_views: transforms must be (V,B,4,4), one pose per view")
        bg = f32c(bg).reshape(-1)
        dev = xyz.device
        out = torch.empty((V, 3, H, W), dtype=torch.float32, device=dev)
        radii = torch.empty((V, N), dtype=torch.int32, device=dev)

        sh_half = 0   # the autograd node reads the fp32 leaves (fp16 SH storage is an option of the training engine)

        def launch(ws):
            check(lib().mgr_views_forward(V, N, B, n_art, sh_half, W, H, ptr(cams), ptr(bg), ptr(xyz), ptr(log_scale), ptr(rot),
                                          ptr(opacity), ptr(f_dc), ptr(f_rest), ptr(skin_w), ptr(transforms),
                                          ptr(out), ptr(radii), ptr(ws.buf), ws.nbytes, ws.cap, ws.skip_bits(), stream()),
                  "mgr_views_forward")

        ws, _ = _rz.context(dev).forward(V, N, W, H, launch)
        ctx.lease = _rz._Lease(ws)
        ctx.meta = (V, N, B, n_art, W, H, stats, float(grad2d_scale))
        ctx.arena = grad_arena
        ctx.save_for_backward(xyz, log_scale, rot, opacity, f_dc, f_rest, skin_w, transforms, cams, bg, out, radii)
        ctx.mark_non_differentiable(radii)
        return out, radii

    @staticmethod
    def backward(ctx, g_img, _g_radii):
        xyz, log_scale, rot, opacity, f_dc, f_rest, skin_w, transforms, cams, bg, out, radii = ctx.saved_tensors
        V, N, B, n_art, W, H, stats, g2s = ctx.meta
        ws = ctx.lease.ws
        dev = xyz.device
        g_img = f32c(g_img)
        arena = ctx.arena or {}

        def e(*s, name=None):
            # the caller's arena (e.g. segments of the flat all-reduce buffer): the kernels write there directly
            t = arena.get(name) if name else None
            if t is not None and t.numel() == int(torch.Size(s).numel()) and t.is_contiguous() and t.dtype == torch.float32 \
                    and t.device == dev:
                return t.view(s)
            return torch.empty(s, dtype=torch.float32, device=dev)

        d_xyz, d_ls, d_rot, d_op = e(N, 3), e(N, 3, name="_scaling"), e(N, 4, name="_rotation"), e(N, name="_opacity")
        d_fdc, d_frest = e(N, 1, 3, name="_features_dc"), e(N, 15, 3, name="_features_rest")
        d_w = e(n_art, B) if skin_w is not None else None
        st_g, st_v = (e(N, name="grad2d"), e(N, name="vis")) if stats is not None else (None, None)
        st_r = torch.empty(N, dtype=torch.int32, device=dev) if stats is not None else None
        sh_half = 0
        check(lib().mgr_views_backward(V, N, B, n_art, sh_half, W, H, ptr(cams), ptr(bg), ptr(xyz), ptr(log_scale), ptr(rot),
                                       ptr(opacity), ptr(f_dc), ptr(f_rest), ptr(skin_w), ptr(transforms),
                                       ptr(radii), ptr(out), ptr(g_img), g2s, ptr(d_xyz), ptr(d_ls), ptr(d_rot),
                                       ptr(d_op), ptr(d_fdc), ptr(d_frest), ptr(d_w), ptr(st_g), ptr(st_v),
                                       ptr(st_r), ptr(ws.buf), ws.nbytes, ws.cap, 0, stream()),
              "mgr_views_backward")
        if stats is not None:
            stats.grad2d, stats.vis, stats.radii = st_g, st_v, st_r
        return (d_xyz, d_ls, d_rot, d_op.reshape(-1, 1), d_fdc, d_frest, d_w, None, None, None, None, None, None,
                None, None)


def render_views(xyz, log_scale, rot, opacity_logit, f_dc, f_rest, skin_w, transforms, cams, bg, W, H,
                 stats=None, grad2d_scale=1.0, grad_arena=None):
    """Images (V,3,H,W) and radii (V,N) of V views from the canonical parameters.

    xyz (N,3) `_xyz`; log_scale (N,3) `_scaling`; rot (N,4) `_rotation`; opacity_logit (N,1)
    `_opacity`; f_dc (N,1,3); f_rest (N,15,3); skin_w (N,B) from `ops.skin_weights`, None for a
    static object, or (n_hand,B) with n_hand < N for the hand+object composite (composite.py:50-59:
    the first n_hand Gaussians are skinned, the rest keep the identity transform); transforms (V,B,4,4): posed @ inv(rest) (+ identity) of the pose seen by each
    view; cams (V,40).  `stats` (a ViewStats) receives the densification statistics in backward.
    `grad_arena`: optional {leaf name | "grad2d" | "vis": preallocated fp32 tensor}; the backward kernels write those
    outputs there instead of into fresh tensors (the multi-GPU step passes segments of its all-reduce buffer)."""
    return _RenderViews.apply(xyz, log_scale, rot, opacity_logit, f_dc, f_rest, skin_w, transforms, cams, bg,
                              int(W), int(H), stats, grad2d_scale, grad_arena)
