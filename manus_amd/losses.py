"""Reference-shaped image losses over the fused HIP kernel (`mgr_image_loss`).

Mirrors `src/utils/loss_utils.py` of brown-ivl/manus for the two terms of the training loss that
consume the rendered image (`loss_func`, src/modules/base.py:323-365):

    l1_loss(network_output, gt)           loss_utils.py:22-27   (mean=True form)
    ssim(img1, img2)                      loss_utils.py:57-97   (window 11, size_average=True)

Both take the reference's HWC images (`render` is permuted to (H,W,3) at
src/utils/gaussian_utils.py:418; `gt` may carry a leading batch dimension of 1) and are
differentiable w.r.t. the first argument.  `ssim` reproduces the reference's behaviour on HWC
input exactly: `channel = img1.size(-3)` is the image height, so the window runs over the (W,3)
plane of each row.  GPU tensors only; there is no CPU fallback.
"""
import weakref

import torch

from . import ops
from ._lib import ManusHipError


def _chw(img):
    if img.dim() == 4:
        if img.shape[0] != 1:
            raise ManusHipError("losses: a batched image must have batch size 1 (as in the reference)")
        img = img[0]
    if img.dim() != 3 or img.shape[-1] != 3:
        raise ManusHipError("losses: images are (H,W,3) like the reference's render / gt")
    return img.permute(2, 0, 1).contiguous()


# CHW copy of a ground-truth image: the reference hands the same HWC tensor to both loss terms of a step (loss_func calls
# l1_loss and ssim on one `gt`, base.py:329-347).  Keyed by the tensor OBJECT (weakly: the entry dies with it -- a storage address
# is reused by the allocator for the next step's image) and its version counter.
_GT_CHW = {}      # id(tensor) -> (weak reference to it, version, CHW copy)


def _gt_chw(gt_hwc):
    key = id(gt_hwc)
    ent = _GT_CHW.get(key)
    if ent is None or ent[0]() is not gt_hwc or ent[1] != gt_hwc._version:
        ref = weakref.ref(gt_hwc, lambda _r, k=key: _GT_CHW.pop(k, None))
        ent = _GT_CHW[key] = (ref, gt_hwc._version, _chw(gt_hwc))
    return ent[2]


class _ImageLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred_hwc, gt_hwc, w_l1, w_ssim):
        pred, gt = _chw(pred_hwc), _gt_chw(gt_hwc)
        n = pred.numel()
        ctx.batched = pred_hwc.dim() == 4
        if w_ssim == 0.0:
            # the L1 term alone needs no windows: one streaming pass (mgr_l1_loss_grad) instead of the fused kernel's three
            # launches -- the route is bound by the host's launches (profiles/r05_other_configs/dropin_c_step_timeline.txt)
            s, g = ops.l1_loss_grad(pred, gt, w_l1 / n)
            ctx.save_for_backward(g[None])
            return (s[0] * (w_l1 / n)).clone()
        sums, g = ops.image_loss_grad(pred, gt, w_l1, w_ssim, 1.0 / n)
        ctx.save_for_backward(g)
        # value of w_l1 * mean|d| + w_ssim * (-mean ssim_map); callers add the constant w_ssim
        return sums[2].clone()

    @staticmethod
    def backward(ctx, go):
        (g,) = ctx.saved_tensors
        gh = (g[0] * go).permute(1, 2, 0)
        return (gh[None] if ctx.batched else gh), None, None, None


class _L1Map(torch.Tensor):
    """What `l1_loss(..., mean=False)` returns: the reference's loss_func takes the |pred - gt| map and reduces it
    with torch.mean itself (base.py:329-331).  The map is never materialised: `.mean()` / `torch.mean(x)` runs the
    fused kernel (value + gradient in one pass); any other use falls back to the explicit map of this same kernel's
    inputs being subtracted on the device."""

    @staticmethod
    def __new__(cls, pred, gt):
        t = torch.Tensor._make_subclass(cls, torch.empty(0, device=pred.device))
        t._pred, t._gt = pred, gt
        return t

    def mean(self, *a, **k):
        if a or k:
            return torch.abs(self._pred - self._gt).mean(*a, **k)
        return _ImageLoss.apply(self._pred, self._gt, 1.0, 0.0)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func is torch.mean and len(args) == 1 and not kwargs and isinstance(args[0], _L1Map):
            return args[0].mean()
        if func is torch.Tensor.mean and len(args) == 1 and not kwargs:
            return args[0].mean()
        with torch._C.DisableTorchFunctionSubclass():
            args = tuple(torch.abs(a._pred - a._gt) if isinstance(a, _L1Map) else a for a in args)
            return func(*args, **kwargs)


def l1_loss(network_output, gt, mean=True):
    """loss_utils.py:22-27: torch.abs(network_output - gt).mean() when mean=True; with mean=False the reference
    returns the map and its caller takes torch.mean of it (base.py:329-331) -- that composition runs the same fused
    kernel here."""
    if not mean:
        return _L1Map(network_output, gt)
    return _ImageLoss.apply(network_output, gt, 1.0, 0.0)


def ssim(img1, img2, window_size=11, size_average=True):
    """ssim(img1, img2) of loss_utils.py:57-97 on HWC images (see the module docstring)."""
    if window_size != 11 or not size_average:
        raise ManusHipError("ssim: only window_size=11, size_average=True (the reference's call) is implemented")
    return -_ImageLoss.apply(img1, img2, 0.0, 1.0)


def rgb_ssim_loss(pred_image, gt_image, w_rgb=0.8, w_ssim=0.2):
    """w_rgb * l1_loss + w_ssim * (1 - ssim) in one kernel: the first two terms of
    config/HAND_GAUSSIAN.yaml:22-23 as combined by loss_func (base.py:356-364)."""
    return _ImageLoss.apply(pred_image, gt_image, w_rgb, w_ssim) + w_ssim


class _IsotropicReg(torch.autograd.Function):
    @staticmethod
    def forward(ctx, log_scale, condition_number):
        loss, g = ops.isotropic_reg_grad(log_scale, condition_number, 1.0)
        ctx.save_for_backward(g)
        return loss[0].clone()

    @staticmethod
    def backward(ctx, go):
        (g,) = ctx.saved_tensors
        return g * go, None


def isotropic_reg(log_scale, condition_number=0.4):
    """mean((min_scale / (max_scale + 1e-8) - condition_number)**2) of base.py:349-356, taking the log-scales
    (`model._scaling`; the reference reads `model.get_scaling = exp(_scaling)`)."""
    return _IsotropicReg.apply(log_scale, condition_number)
