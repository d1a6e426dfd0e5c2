"""Multi-view training step: V views batched per GPU, views sharded across GPUs,
one all-reduce of the per-Gaussian gradients (RCCL over xGMI via torch.distributed).

The reference trains one (frame, view) per step on one GPU (config/trainer/trainer.yaml:5,
main.py:84-87 DDP commented out).  "V views per iteration" is defined here as
    grad = (1/V) * sum_v grad(L_v)            (with V=1 it is the reference step)
and the densification statistics follow the reference's per-view rule
(src/models/gaussian.py:335-338, src/utils/gaussian_utils.py:469-471):
    xyz_gradient_accum += sum_v ||d L_v / d means2D[:, :2]||   (visible Gaussians only)
    denom              += sum_v visible_v
    max_radii2D         = max_v radii_v
"""
import torch
import torch.distributed as dist

# packed leaf-gradient layout: 59 floats per Gaussian
GRAD_LAYOUT = (("_xyz", 3), ("_features_dc", 3), ("_features_rest", 45), ("_opacity", 1),
               ("_scaling", 3), ("_rotation", 4))
GRAD_WIDTH = sum(w for _, w in GRAD_LAYOUT)


def shard_views(n_views, rank, world_size):
    """Round-robin assignment of view indices to ranks."""
    return list(range(rank, n_views, world_size))


def pack_grads(grads, N, device, out=None):
    """Flat SoA buffer: one contiguous segment per leaf, in GRAD_LAYOUT order."""
    buf = out if out is not None else torch.empty(N * GRAD_WIDTH, dtype=torch.float32, device=device)
    o = 0
    for name, w in GRAD_LAYOUT:
        g = grads[name]
        seg = buf[o:o + N * w]
        if not (g.data_ptr() == seg.data_ptr() and g.is_contiguous()):   # already written in place (grad arena)
            seg.copy_(g.reshape(-1))
        o += N * w
    return buf


def unpack_grads(buf, shapes, N):
    out, o = {}, 0
    for name, w in GRAD_LAYOUT:
        out[name] = buf[o:o + N * w].reshape(shapes[name])
        o += N * w
    return out


class ViewShardedStep:
    """Runs `compute_fn(view_ids)` on this rank's views and reduces across ranks.

    compute_fn returns a dict with
        grads      {leaf name: scale * sum over the given views of dL_v/dleaf}
        grad2d     (N,) sum over views of the visible 2D-gradient norms (unscaled)
        vis        (N,) number of views in which the Gaussian was visible
        radii      (N,) int32 max screen radius over views
        loss       scalar tensor, scale * sum of L_v
    where compute_fn is called as compute_fn(view_ids, scale).
    """

    def __init__(self, n_gaussians, shapes, compute_fn, n_views, rank=0, world_size=1, group=None):
        self.N, self.shapes, self.compute_fn = n_gaussians, shapes, compute_fn
        self.n_views, self.rank, self.world, self.group = n_views, rank, world_size, group
        self.local_views = shard_views(n_views, rank, world_size)
        self.always_pack = False   # tests: take the packing path without a process group
        self._flat = None

    def step(self):
        # compute_fn folds the 1/V of "grad = (1/V) sum_v grad L_v" into the loss scale
        N = self.N
        packed = self.world > 1 or self.always_pack
        if packed and self._flat is not None and hasattr(self.compute_fn, "grad_arena"):
            # the backward kernels write straight into the all-reduce buffer (no packing copies): one view per leaf
            flat, o, arena = self._flat, 0, {}
            for name, w in GRAD_LAYOUT:
                arena[name] = flat[o:o + N * w]
                o += N * w
            arena["grad2d"], arena["vis"] = flat[o:o + N], flat[o + N:o + 2 * N]
            self.compute_fn.grad_arena = arena
        out = self.compute_fn(self.local_views, 1.0 / float(self.n_views))
        dev = out["grad2d"].device
        if not packed:
            return dict(grads=out["grads"], grad2d=out["grad2d"], vis=out["vis"],
                        radii=out["radii"].to(torch.int32), loss=out["loss"])
        # one flat buffer -> one SUM all-reduce (+ one MAX all-reduce for the radii)
        if self._flat is None or self._flat.device != dev:
            self._flat = torch.empty(N * (GRAD_WIDTH + 2) + 1, dtype=torch.float32, device=dev)
        flat = self._flat
        pack_grads(out["grads"], N, dev, out=flat[: N * GRAD_WIDTH])
        for src, lo in ((out["grad2d"], N * GRAD_WIDTH), (out["vis"], N * (GRAD_WIDTH + 1))):
            seg = flat[lo: lo + N]
            if src.data_ptr() != seg.data_ptr():
                seg.copy_(src)
        flat[-1:].copy_(out["loss"].reshape(1))
        radii = out["radii"].to(torch.int32)
        if self.world > 1:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            dist.all_reduce(radii, op=dist.ReduceOp.MAX, group=self.group)
        grads = unpack_grads(flat[: N * GRAD_WIDTH], self.shapes, N)
        return dict(grads=grads, grad2d=flat[N * GRAD_WIDTH: N * (GRAD_WIDTH + 1)],
                    vis=flat[N * (GRAD_WIDTH + 1): N * (GRAD_WIDTH + 2)], radii=radii,
                    loss=flat[-1])


class HipViewCompute:
    """compute_fn over the HIP kernels: skin weights once, LBS per pose, SH colour and
    rasterisation per view, L1 image loss (rgb_loss of src/modules/base.py:329-331),
    backward to the six leaf tensors.  All local views go through every kernel launch
    together."""

    def __init__(self, scene, targets, cam_table, loss_weight=1.0, fused=True, loss="l1", w_rgb=0.8, w_ssim=0.2):
        from . import fused as fused_mod, ops, rasterizer
        self.ops, self.rz, self.fz, self.fused = ops, rasterizer, fused_mod, fused
        self.s = scene
        self.targets = targets          # (V_all,3,H,W) on the GPU
        self.cams = cam_table           # (V_all,40)
        self.loss_weight = loss_weight
        # "l1": mean|render - gt| (rgb_loss alone); "l1+ssim": w_rgb * rgb_loss + w_ssim * ssim_loss, the
        # image terms of config/HAND_GAUSSIAN.yaml:22-23 (src/modules/base.py:323-365), one fused kernel
        if loss not in ("l1", "l1+ssim"):
            raise ValueError("loss must be 'l1' or 'l1+ssim'")
        self.loss, self.w_rgb, self.w_ssim = loss, w_rgb, w_ssim
        self.is_hand = scene.get("grid") is not None and scene["kind"] == "hand"
        self.params = {k: v.detach().clone().requires_grad_(True) for k, v in scene["params"].items()}
        self.grad_arena = None   # set by ViewShardedStep: preallocated gradient outputs (fused path only)
        self._cache = {}
        self.grid = ops.SkinGrid(scene["grid"], scene["grid"].device) if self.is_hand else None

    def _select(self, view_ids):
        """Per-view constants for a set of views (cached: no per-step gather copies)."""
        key = tuple(view_ids)
        c = self._cache.get(key)
        if c is None:
            idx = list(view_ids)
            c = dict(cams=self.cams[idx].contiguous(), targets=self.targets[idx].contiguous(),
                     T=self.s["transforms"][idx].contiguous() if self.is_hand else None)
            self._cache = {key: c}
        return c

    def forward_views(self, view_ids):
        s, p, ops = self.s, self.params, self.ops
        sel = self._select(view_ids)
        cams = sel["cams"]
        V = len(view_ids)
        feats = torch.cat([p["_features_dc"], p["_features_rest"]], dim=1)
        opac = torch.sigmoid(p["_opacity"])
        if self.is_hand:
            w = ops.skin_weights(p["_xyz"], self.grid, s["grid_center"], s["grid_scale"])
            # one pose per view (the reference trains one (frame, view) per step)
            pxyz, pcov, tf = ops.lbs_cov(p["_xyz"], p["_scaling"], p["_rotation"], w, sel["T"])
            col = ops.sh_colors(feats, p["_xyz"], tf, cams)
        else:
            _, pcov1, _ = ops.lbs_cov(p["_xyz"], p["_scaling"], p["_rotation"], None, None)
            pxyz, pcov = p["_xyz"], pcov1[0]
            col = ops.sh_colors(feats, p["_xyz"], None, cams)
        N = p["_xyz"].shape[0]
        means2D = torch.zeros((V, N, 3), dtype=torch.float32, device=cams.device, requires_grad=True)
        img, radii = self.rz.rasterize_views(cams, pxyz, means2D, col, opac, pcov, s["bg"], s["width"], s["height"])
        return img, radii, means2D

    def forward_views_fused(self, view_ids, stats=None, grad2d_scale=1.0):
        s, p, ops = self.s, self.params, self.ops
        sel = self._select(view_ids)
        w = ops.skin_weights(p["_xyz"], self.grid, s["grid_center"], s["grid_scale"]) if self.is_hand else None
        return self.fz.render_views(p["_xyz"], p["_scaling"], p["_rotation"], p["_opacity"], p["_features_dc"],
                                    p["_features_rest"], w, sel["T"], sel["cams"], s["bg"], s["width"], s["height"],
                                    stats=stats, grad2d_scale=grad2d_scale, grad_arena=self.grad_arena)

    def _image_loss(self, img, tgt, scale):
        """(loss value, dL/dimg) of scale * sum over the views of the per-view image loss."""
        per_view = img[0].numel()
        k = self.loss_weight * scale / per_view
        if self.loss == "l1":
            loss_sum, g = self.ops.l1_loss_grad(img, tgt, scale=k)
            return loss_sum[0] * k, g
        const = self.w_ssim * self.loss_weight * scale * img.shape[0]   # the "1 -" of 1 - ssim, once per view
        sums, g = self.ops.image_loss_grad(img, tgt, self.w_rgb, self.w_ssim, k, const)
        return sums[2], g

    def _call_fused(self, view_ids, scale):
        for v in self.params.values():
            v.grad = None
        stats = self.fz.ViewStats()
        img, radii = self.forward_views_fused(view_ids, stats, 1.0 / scale)
        tgt = self._select(view_ids)["targets"]
        loss, g = self._image_loss(img, tgt, scale)
        img.backward(g)
        return dict(grads={n: v.grad for n, v in self.params.items()}, grad2d=stats.grad2d, vis=stats.vis,
                    radii=stats.radii, loss=loss)

    def __call__(self, view_ids, scale=1.0):
        if self.fused:
            return self._call_fused(view_ids, scale)
        for v in self.params.values():
            v.grad = None
        img, radii, means2D = self.forward_views(view_ids)
        tgt = self._select(view_ids)["targets"]
        loss, g = self._image_loss(img, tgt, scale)
        img.backward(g)
        vis = radii > 0
        g2 = means2D.grad[..., :2].norm(dim=-1) * (1.0 / scale)
        return dict(grads={n: v.grad for n, v in self.params.items()},
                    grad2d=(g2 * vis).sum(0), vis=vis.sum(0).float(),
                    radii=radii.max(dim=0).values, loss=loss)


class Trainer:
    """The per-step control flow of the reference's training_step (src/modules/hand_dynamic.py:230-282) on top of
    the kernels: multi-view step -> fused Adam step with the xyz schedule -> density_update (statistics, densify /
    prune every `densification_interval` steps, opacity reset).  After a densification the parameter tensors are
    new (different N): the compute object is re-pointed at them and the rasterizer workspaces of the old size are
    released.  Multi-GPU: every rank must draw the same split noise, so rank 0's is broadcast."""

    def __init__(self, compute, n_views, extent, opts=None, spatial_lr_scale=1.0, rank=0, world_size=1, group=None,
                 bg_white=True):
        from .optim import GaussianOptimizer
        self.compute, self.n_views, self.extent, self.bg_white = compute, n_views, float(extent), bg_white
        self.rank, self.world, self.group = rank, world_size, group
        self.opt = GaussianOptimizer(compute.params, opts=opts, spatial_lr_scale=spatial_lr_scale, adopt=True)
        self.global_step = 0
        self._rebuild_step()

    def _rebuild_step(self):
        if hasattr(self.compute, "grad_arena"):
            self.compute.grad_arena = None   # views of the previous step object's buffer (possibly another N)
        p = self.compute.params
        shapes = {k: v.shape for k, v in p.items()}
        self.stepper = ViewShardedStep(p["_xyz"].shape[0], shapes, self.compute, self.n_views, rank=self.rank,
                                       world_size=self.world, group=self.group)

    def _split_noise(self, n_sel, device):
        noise = torch.randn((2 * n_sel, 3), dtype=torch.float32, device=device)
        if self.world > 1:
            dist.broadcast(noise, src=0, group=self.group)
        return noise

    def train_step(self):
        """One optimisation step; returns the step's output dict (loss, statistics) plus "changed"."""
        from . import rasterizer
        out = self.stepper.step()
        self.global_step += 1
        self.opt.update_learning_rate(self.global_step)
        self.opt.step(out["grads"])
        o = self.opt.opts
        changed = False
        gs = self.global_step
        if gs < o["densify_until_step"]:
            self.opt.add_densification_stats(out["grad2d"], out["vis"], out["radii"])
            if gs > o["densify_from_step"] and gs % o["densification_interval"] == 0:
                size_threshold = o["size_threshold"] if gs > o["opacity_reset_interval"] else None
                # the plan's selection count is needed before the noise can be drawn: densify_and_prune draws it
                # itself on one GPU; with several ranks it is drawn here for the worst case and broadcast
                noise = None
                if self.world > 1:
                    noise_full = self._split_noise(self.opt.N, self.opt.device)
                    noise = noise_full
                info = self._densify(o, size_threshold, noise)
                changed = True
                out["densify"] = info
            if gs % o["opacity_reset_interval"] == 0 or (self.bg_white and gs == o["densify_from_step"]):
                self.opt.reset_opacity()
        if changed:
            self.compute.params = {k: v.detach().requires_grad_(True) for k, v in self.opt.parameters().items()}
            self.opt.p = {k: v.detach() for k, v in self.compute.params.items()}
            rasterizer._POOL.clear()
            rasterizer.set_sync_policy(True)   # the pair capacity of the new size has to be learnt again
            self._rebuild_step()
        out["changed"] = changed
        return out

    def _densify(self, o, size_threshold, noise_full):
        if noise_full is None:
            return self.opt.densify_and_prune(o["densify_grad_threshold"], o["min_opacity_threshold"], self.extent,
                                              size_threshold)
        return self.opt.densify_and_prune(o["densify_grad_threshold"], o["min_opacity_threshold"], self.extent,
                                          size_threshold, noise=noise_full, noise_is_pool=True)
