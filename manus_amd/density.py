"""Pruning tests and the per-step density control flow of the MANUS training modules on the HIP kernels.

Mirrors (brown-ivl/manus):
    dilate_mask               src/utils/gaussian_utils.py:35-47
    get_points_outside_mask   src/utils/gaussian_utils.py:101-147
    on_after_backward (hand)  src/modules/hand_dynamic.py:193-224
    on_after_backward (obj)   src/modules/object.py:66-81
    density_update            src/modules/base.py:87-98 -> src/utils/gaussian_utils.py:451-498
GPU tensors only; there is no CPU or PyTorch fallback.
"""
import torch

from ._lib import ManusHipError, check, f32c, lib, ptr, stream


def _byte_mask(mask, device):
    """Any mask layout the reference passes ((1,H,W,1) / (H,W,1) / (H,W), any dtype) -> (H,W) uint8 on `device`."""
    m = torch.as_tensor(mask).to(device)
    if m.dim() == 4:
        m = m[0]
    if m.dim() == 3:
        m = m[..., 0]
    return (m != 0).to(torch.uint8).contiguous()


def dilate_mask(mask, kernel_size=11):
    """(H,W) mask -> (H,W) bool, True where any pixel of the kernel_size x kernel_size window is set."""
    m = _byte_mask(mask, torch.as_tensor(mask).device)
    if not m.is_cuda:
        raise ManusHipError("manus_amd.density needs GPU tensors; there is no CPU fallback")
    H, W = m.shape
    tmp, out = torch.empty_like(m), torch.empty_like(m)
    check(lib().mgr_dilate_mask(H, W, int(kernel_size), ptr(m), ptr(tmp), ptr(out), stream()), "mgr_dilate_mask")
    return out.bool()


def get_points_outside_mask(camera, points, mask, keypoints=None, dilate=False):
    """Same signature and result as the reference: camera.K (3,3)/(1,3,3), camera.extr (3,4)/(4,4) (+ leading 1),
    points (N,3), mask (1,H,W,1)/(H,W,1) -> (N,1) bool, True = the Gaussian projects onto a pixel outside the
    (optionally dilated) mask; all False when any keypoint does."""
    pts = f32c(points)
    dev = pts.device
    K = torch.as_tensor(camera["K"] if isinstance(camera, dict) else camera.K, dtype=torch.float32).to(dev)
    E = torch.as_tensor(camera["extr"] if isinstance(camera, dict) else camera.extr, dtype=torch.float32).to(dev)
    if K.dim() == 3:
        K = K[0]
    if E.dim() == 3:
        E = E[0]
    K9, E12 = K.reshape(-1)[:9].contiguous(), E[:3, :4].reshape(-1).contiguous()
    m = _byte_mask(mask, dev)
    if dilate:
        m = dilate_mask(m).to(torch.uint8)
    H, W = m.shape
    N = pts.shape[0]
    kp = f32c(keypoints).to(dev).reshape(-1, 3) if keypoints is not None else None
    out = torch.empty((N,), dtype=torch.uint8, device=dev)
    check(lib().mgr_points_outside_mask(N, ptr(pts), ptr(K9), ptr(E12), H, W, ptr(m), 0 if kp is None else kp.shape[0],
                                        ptr(kp), ptr(out), stream()), "mgr_points_outside_mask")
    return out.bool().reshape(N, 1)


def keypoint_far_mask(points, keypoints, thresh=0.2):
    """torch.cdist(points, keypoints).mean(1) > thresh (hand_dynamic.py:217) -> (N,) bool."""
    pts, kp = f32c(points), f32c(keypoints).reshape(-1, 3)
    out = torch.empty((pts.shape[0],), dtype=torch.uint8, device=pts.device)
    check(lib().mgr_keypoint_far_mask(pts.shape[0], ptr(pts), kp.shape[0], ptr(kp.to(pts.device)), float(thresh), ptr(out),
                                      stream()), "mgr_keypoint_far_mask")
    return out.bool()


class DensityController:
    """`on_after_backward` of the hand / object training modules over a `GaussianOptimizer`.

    The reference trains one (frame, view) per step; with V views per iteration the pruning masks of the views are
    OR-ed (the reference's own `self.pts_mask += pts_mask` accumulation, hand_dynamic.py:208) and density_update
    runs once on the statistics summed over the views.  With V = 1 this is the reference step.

    views: list of dicts, one per view rendered this step:
        camera     object / dict with K, extr           (mask test)
        mask       (1,H,W,1) / (H,W,1) / (H,W)           (mask test)
        posed_xyz  (N,3) posed means of that view's pose (hand: LBS output; object: the canonical xyz)
        keypoints  (21,3) cat(heads[:1], tails)          (hand only)
    """

    def __init__(self, opt, extent, kind="hand", bg_white=True):
        if kind not in ("hand", "object"):
            raise ValueError("kind must be 'hand' or 'object'")
        self.opt, self.extent, self.kind, self.bg_white = opt, float(extent), kind, bool(bg_white)
        self.do_density_update = True
        self.on_train_epoch_start()

    def on_train_epoch_start(self):
        self.pts_mask = torch.zeros(self.opt.N, dtype=torch.bool, device=self.opt.device)

    def prune_mask(self, global_step, views):
        """The mask handed to density_update (None when no Gaussian is flagged)."""
        rse = self.opt.opts["remove_seg_end"]
        tested = False
        if global_step < rse:
            for v in views or ():
                if v.get("mask") is None:
                    continue
                if self.kind == "hand":
                    pm = get_points_outside_mask(v["camera"], v["posed_xyz"], v["mask"], v["keypoints"], dilate=True)
                else:
                    pm = get_points_outside_mask(v["camera"], v["posed_xyz"], v["mask"])
                self.pts_mask |= pm[..., 0]
                tested = True
        elif self.kind == "hand" and global_step % 100 == 0:
            for v in views or ():
                if v.get("keypoints") is None:
                    continue
                self.pts_mask |= keypoint_far_mask(v["posed_xyz"], v["keypoints"], 0.2)
                tested = True
        # `self.pts_mask.sum() > 0` (a host sync in the reference, every step): the mask can only be non-zero on a
        # step that ran a test, because every True is pruned in the same step
        if tested and bool(self.pts_mask.any()):
            return self.pts_mask.clone()
        return None

    def after_backward(self, global_step, stats, views=None, noise=None, noise_is_pool=False):
        """Returns True when leaves were replaced (N may have changed)."""
        if not self.do_density_update:
            return False
        mask = self.prune_mask(global_step, views)
        res = self.opt.density_update(stats, self.extent, global_step, self.bg_white, mask_to_prune=mask, noise=noise,
                                      noise_is_pool=noise_is_pool)
        if res:
            self.pts_mask = torch.zeros(self.opt.N, dtype=torch.bool, device=self.opt.device)
        return res
