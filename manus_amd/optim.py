"""Optimizer step and densification of the Gaussian parameter model over the HIP kernels of
`csrc/optim.hip`.

Mirrors the training-side half of `GaussianModel` (brown-ivl/manus, src/models/gaussian.py):

    training_setup          :128-146   six Adam groups (xyz, f_dc, f_rest, opacity, scaling, rotation), eps=1e-15
    update_learning_rate    src/utils/gaussian_utils.py:501-508 + get_expon_lr_func :212-245 (xyz schedule)
    optimizer.step()        one fused kernel over all six groups (per-group step counts, like Adam's state["step"])
    add_densification_stats :335-338   (+ the max_radii2D update of density_update, gaussian_utils.py:470-473)
    densify_and_prune       :310-333   clone + split + prune + Adam-moment surgery: plan + apply kernels
    prune_points            :185-203   compaction of leaves, moments, skin weights and statistics by a mask
    reset_opacity           :148-165
    density_update          src/utils/gaussian_utils.py:451-498 (incl. the mask_to_prune branch)

The leaves keep the reference's names and shapes (`_xyz (N,3)`, `_features_dc (N,1,3)`,
`_features_rest (N,15,3)`, `_opacity (N,1)`, `_scaling (N,3)`, `_rotation (N,4)`).  GPU tensors only;
there is no CPU or PyTorch fallback.
"""
import ctypes
import math

import numpy as np
import torch

from ._lib import ManusHipError, check, f32c, lib, ptr, stream

# group order of training_setup (gaussian.py:133-140): (group name, attribute, row width)
GROUPS = (("xyz", "_xyz", 3), ("f_dc", "_features_dc", 3), ("f_rest", "_features_rest", 45), ("opacity", "_opacity", 1),
          ("scaling", "_scaling", 3), ("rotation", "_rotation", 4))

DEFAULT_OPTS = dict(  # config/model/gaussian/gaussian.yaml
    position_lr_init=0.0016, position_lr_final=0.0000016, position_lr_delay_mult=0.01, position_lr_max_steps=30000,
    feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001, percent_dense=0.000001,
    densification_interval=100, opacity_reset_interval=3000, densify_from_step=100, densify_until_step=50000,
    densify_grad_threshold=0.0002, min_opacity_threshold=0.005, size_threshold=20, remove_outliers_step=-1,
    remove_seg_end=1000)

ALL_GROUPS = frozenset(n for n, _, _ in GROUPS)


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """Log-linear learning-rate decay, gaussian_utils.py:212-245 (host arithmetic, float64)."""

    def helper(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        delay_rate = 1.0
        if lr_delay_steps > 0:
            delay_rate = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
        t = np.clip(step / max_steps, 0, 1)
        return float(delay_rate * np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t))

    return helper


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[ptr(t) if t is not None else None for t in tensors])


class GaussianOptimizer:
    """Leaves + Adam state + densification statistics of one Gaussian model on one GPU."""

    def __init__(self, params, opts=None, spatial_lr_scale=1.0, skin_weights=None, adopt=False):
        """params: {attribute name: tensor}.  adopt=True updates the given tensors in place (they stay the
        leaves other code holds, e.g. `engine.HipViewCompute.params`) until densification replaces them."""
        self.opts = dict(DEFAULT_OPTS, **(opts or {}))
        self.spatial_lr_scale = float(spatial_lr_scale)
        self.p = {}
        for _, attr, w in GROUPS:
            t = params[attr]
            if not t.is_cuda:
                raise ManusHipError("manus_amd.optim needs GPU tensors; there is no CPU fallback")
            self.p[attr] = t.detach() if (adopt and t.dtype == torch.float32 and t.is_contiguous()) else f32c(t.detach()).clone()
        self.skin_weights = f32c(skin_weights).clone() if skin_weights is not None else None
        self.training_setup()

    # -- training_setup, gaussian.py:128-146 ------------------------------------------------
    def training_setup(self):
        o, dev = self.opts, self.device
        self.m = {a: torch.zeros_like(self.p[a]) for _, a, _ in GROUPS}
        self.v = {a: torch.zeros_like(self.p[a]) for _, a, _ in GROUPS}
        # steps taken per group: torch.optim.Adam keeps state["step"] per parameter, and a group whose leaf was
        # replaced before optimizer.step() (reset_opacity at densify_from_step, :153-165) misses that step
        self.group_step = {n: 0 for n, _, _ in GROUPS}
        self.replaced = frozenset()                         # groups replaced since the last step() (no gradient yet)
        self.lr = {"xyz": o["position_lr_init"] * self.spatial_lr_scale, "f_dc": o["feature_lr"],
                   "f_rest": o["feature_lr"] / 20.0, "opacity": o["opacity_lr"], "scaling": o["scaling_lr"],
                   "rotation": o["rotation_lr"]}
        self.xyz_scheduler_args = get_expon_lr_func(o["position_lr_init"] * self.spatial_lr_scale,
                                                    o["position_lr_final"] * self.spatial_lr_scale,
                                                    lr_delay_mult=o["position_lr_delay_mult"],
                                                    max_steps=o["position_lr_max_steps"])
        self._reset_stats()

    def _reset_stats(self):
        n, dev = self.N, self.device
        self.xyz_gradient_accum = torch.zeros((n, 1), dtype=torch.float32, device=dev)
        self.denom = torch.zeros((n, 1), dtype=torch.float32, device=dev)
        self.max_radii2D = torch.zeros((n,), dtype=torch.float32, device=dev)

    @property
    def N(self):
        return self.p["_xyz"].shape[0]

    @property
    def device(self):
        return self.p["_xyz"].device

    @property
    def state_step(self):
        return max(self.group_step.values())

    @state_step.setter
    def state_step(self, t):
        self.group_step = {n: int(t) for n, _, _ in GROUPS}

    def parameters(self):
        """The six leaves by the reference's attribute names (live tensors, updated in place by step())."""
        return self.p

    # -- resume -------------------------------------------------------------------------------
    def state_dict(self):
        """Everything a resume needs beyond the leaves (CPU tensors): both Adam moments, the per-group step counts, the
        learning rates and the densification statistics.  (The reference resumes through Lightning, whose checkpoint
        carries torch.optim.Adam's exp_avg / exp_avg_sq / step per parameter: the same content.)"""
        c = lambda t: t.detach().cpu().clone()
        return {"exp_avg": {a: c(t) for a, t in self.m.items()}, "exp_avg_sq": {a: c(t) for a, t in self.v.items()},
                "group_step": dict(self.group_step), "lr": dict(self.lr),
                "xyz_gradient_accum": c(self.xyz_gradient_accum), "denom": c(self.denom), "max_radii2D": c(self.max_radii2D)}

    def load_state_dict(self, sd):
        """Inverse of `state_dict` on an optimizer built over leaves with the same number of rows."""
        for a in self.m:
            if tuple(sd["exp_avg"][a].shape) != tuple(self.m[a].shape):
                raise ValueError("optimizer state of %s has shape %s, the leaf %s" % (a, tuple(sd["exp_avg"][a].shape), tuple(self.m[a].shape)))
            self.m[a].copy_(sd["exp_avg"][a])
            self.v[a].copy_(sd["exp_avg_sq"][a])
        self.group_step = {n: int(t) for n, t in sd["group_step"].items()}
        self.lr.update(sd["lr"])
        self.xyz_gradient_accum.copy_(sd["xyz_gradient_accum"])
        self.denom.copy_(sd["denom"])
        self.max_radii2D.copy_(sd["max_radii2D"])

    # -- update_learning_rate, gaussian_utils.py:501-508 --------------------------------------
    def update_learning_rate(self, global_step):
        self.lr["xyz"] = self.xyz_scheduler_args(global_step)
        return self.lr["xyz"]

    # -- optimizer.step() ---------------------------------------------------------------------
    def step(self, grads, beta1=0.9, beta2=0.999, eps=1e-15, skip=None):
        """One Adam update from `grads` {attribute name: tensor of the leaf's shape}.  `skip`: group names left
        untouched (default: the groups replaced since the last step -- the reference's optimizer.step() runs after
        on_after_backward and skips a new nn.Parameter, which has no .grad; hand_dynamic.py:193-224,269-277)."""
        skip = self.replaced if skip is None else frozenset(skip)
        self.replaced = frozenset()
        active = [(n, a) for n, a, _ in GROUPS if n not in skip]
        if not active:
            return
        g = {a: f32c(grads[a]) for _, a in active}
        for _, a in active:
            if g[a].numel() != self.p[a].numel():
                raise ManusHipError("step: gradient of %s has the wrong size" % a)
        for n, _ in active:
            self.group_step[n] += 1
        on = {n for n, _ in active}
        counts = (ctypes.c_int64 * 6)(*[self.p[a].numel() for _, a, _ in GROUPS])
        lrs = (ctypes.c_double * 6)(*[self.lr[n] for n, _, _ in GROUPS])
        steps = (ctypes.c_int64 * 6)(*[self.group_step[n] if n in on else 0 for n, _, _ in GROUPS])
        check(lib().mgr_adam_step_groups(6, counts, _ptr_array([self.p[a] for _, a, _ in GROUPS]),
                                         _ptr_array([g.get(a) for _, a, _ in GROUPS]),
                                         _ptr_array([self.m[a] for _, a, _ in GROUPS]),
                                         _ptr_array([self.v[a] for _, a, _ in GROUPS]), lrs, steps, beta1, beta2, eps,
                                         stream()), "mgr_adam_step_groups")

    # -- sharded step: flat storage + Adam on an element range -------------------------------------------
    def flatten(self, padded):
        """Re-home the six leaves and both moments in three flat buffers of `padded` floats, group after group in
        training_setup order (= the layout of the engine's gradient buffer), so that an optimizer step can be taken
        on any element range and the parameters all-gathered as one tensor.  The leaves stay views with the
        reference's shapes."""
        dev = self.device
        flats = [torch.zeros(padded, dtype=torch.float32, device=dev) for _ in range(3)]
        for store, flat in zip((self.p, self.m, self.v), flats):
            o = 0
            for _, a, _ in GROUPS:
                t = store[a]
                flat[o:o + t.numel()].copy_(t.reshape(-1))
                store[a] = flat[o:o + t.numel()].view(t.shape)
                o += t.numel()
        self.pflat, self.mflat, self.vflat = flats
        return self.pflat

    def step_range(self, grad_flat, lo, hi, beta1=0.9, beta2=0.999, eps=1e-15, skip=None):
        """Adam on the elements [lo, hi) of the flat parameter buffer (`flatten` first) from the same elements of
        `grad_flat`; every rank of a sharded step calls this with its own range, so the per-group step counts advance
        for every group that is not skipped, whether or not the range touches it."""
        skip = self.replaced if skip is None else frozenset(skip)
        self.replaced = frozenset()
        counts, ps, gs, ms, vs, steps, lrs = [], [], [], [], [], [], []
        o = 0
        for n, a, _ in GROUPS:
            size = self.p[a].numel()
            on = n not in skip
            if on:
                self.group_step[n] += 1
            a0, a1 = max(lo, o), min(hi, o + size)
            take = on and a1 > a0
            counts.append(a1 - a0 if take else 0)
            ps.append(self.pflat[a0:a1] if take else None)
            gs.append(grad_flat[a0:a1] if take else None)
            ms.append(self.mflat[a0:a1] if take else None)
            vs.append(self.vflat[a0:a1] if take else None)
            steps.append(self.group_step[n] if take else 0)
            lrs.append(self.lr[n])
            o += size
        if not any(counts):
            return
        check(lib().mgr_adam_step_groups(6, (ctypes.c_int64 * 6)(*counts), _ptr_array(ps), _ptr_array(gs), _ptr_array(ms),
                                         _ptr_array(vs), (ctypes.c_double * 6)(*lrs), (ctypes.c_int64 * 6)(*steps), beta1,
                                         beta2, eps, stream()), "mgr_adam_step_groups")

    # -- add_densification_stats (+ max radii), gaussian.py:335-338, gaussian_utils.py:470-473 -
    def add_densification_stats(self, grad2d_sum, vis_count, radii_max):
        """Accumulate the statistics of one multi-view step (`fused.ViewStats` / `ViewShardedStep` outputs):
        sum over views of ||dL/dmeans2D[:, :2]||, number of views that saw the Gaussian, max screen radius."""
        N = self.N
        if N == 0:
            return
        g2, vis = f32c(grad2d_sum).reshape(-1), f32c(vis_count).reshape(-1)
        rad = radii_max.reshape(-1)
        rad = (rad if rad.dtype == torch.int32 else rad.to(torch.int32)).contiguous()
        if g2.numel() != N or vis.numel() != N or rad.numel() != N:
            raise ManusHipError("add_densification_stats: statistics must have %d entries" % N)
        for name in ("xyz_gradient_accum", "denom", "max_radii2D"):
            # the kernel writes N floats through a raw pointer into each accumulator: a tensor of another size (assigned by
            # a caller, or left behind by a resize) would be written out of bounds where the torch ops raised a shape error
            t = getattr(self, name)
            if not t.is_cuda or t.numel() != N:
                raise ManusHipError("add_densification_stats: %s has %d entries, the model %d" % (name, t.numel(), N))
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise ManusHipError("add_densification_stats: %s must be a contiguous float32 tensor (it is updated in place)" % name)
        check(lib().mgr_add_densification_stats(N, ptr(g2), ptr(vis), ptr(rad), ptr(self.xyz_gradient_accum), ptr(self.denom),
                                                ptr(self.max_radii2D), stream()), "mgr_add_densification_stats")

    # -- densify_and_prune, gaussian.py:310-333 ------------------------------------------------
    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size=None, remove_outliers=False, noise=None,
                          noise_is_pool=False):
        """Clone / split / prune exactly like the reference, including the order of the surviving rows and the
        zero Adam moments of new rows.  `max_screen_size` falsy (None before `opacity_reset_interval`,
        gaussian_utils.py:478-482): only low opacity (and NaN scales) prunes; set: Gaussians larger than
        0.1 * extent in world space are pruned as well (:316-320; the max_radii2D half of that test cannot fire,
        densification_postfix zeroes max_radii2D first, :249-251).
        `noise`: optional standard normals (2*n_selected, 3) for the split offsets (default: torch.randn).
        noise_is_pool=True: `noise` is a larger pool (>= 2*n_selected rows, e.g. one broadcast to all ranks before
        the selection count is known); its first 2*n_selected rows are used.
        remove_outliers (:323-326, scikit-learn LocalOutlierFactor-style filter on the host, off in every shipped
        config: remove_outliers_step -1) is not built."""
        if remove_outliers:
            raise ManusHipError("densify_and_prune: remove_outliers is not supported (SURVEY 8f: host-side "
                                "update_mask_based_on_outliers; remove_outliers_step is -1 in every shipped config)")
        N, dev = self.N, self.device
        nbytes = int(lib().mgr_densify_workspace_bytes(N))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        counts = (ctypes.c_int64 * 5)()
        check(lib().mgr_densify_plan(N, ptr(self.xyz_gradient_accum), ptr(self.denom), ptr(self.p["_scaling"]),
                                     ptr(self.p["_opacity"]), float(max_grad), float(min_opacity), float(extent),
                                     float(self.opts["percent_dense"]), float(max_screen_size or 0.0), ptr(ws), nbytes,
                                     counts, stream()), "mgr_densify_plan")
        n_sel, M = int(counts[2]), int(counts[4])
        if noise is None:
            noise = torch.randn((2 * n_sel, 3), dtype=torch.float32, device=dev)
        noise = f32c(noise)
        if noise_is_pool:
            if noise.shape[0] < 2 * n_sel:
                raise ManusHipError("densify_and_prune: the noise pool is smaller than 2*%d rows" % n_sel)
            noise = noise[: 2 * n_sel].contiguous()
        if noise.numel() != 2 * n_sel * 3:
            raise ManusHipError("densify_and_prune: noise must be (2*%d, 3)" % n_sel)
        shp = lambda a: (M,) + tuple(self.p[a].shape[1:])
        new_p = {a: torch.empty(shp(a), dtype=torch.float32, device=dev) for _, a, _ in GROUPS}
        new_m = {a: torch.empty(shp(a), dtype=torch.float32, device=dev) for _, a, _ in GROUPS}
        new_v = {a: torch.empty(shp(a), dtype=torch.float32, device=dev) for _, a, _ in GROUPS}
        B = self.skin_weights.shape[1] if self.skin_weights is not None else 0
        new_skin = torch.empty((M, B), dtype=torch.float32, device=dev) if B else None
        order = [a for _, a, _ in GROUPS]
        check(lib().mgr_densify_apply(N, M, n_sel, ptr(ws), _ptr_array([self.p[a] for a in order]),
                                      _ptr_array([self.m[a] for a in order]), _ptr_array([self.v[a] for a in order]),
                                      _ptr_array([new_p[a] for a in order]), _ptr_array([new_m[a] for a in order]),
                                      _ptr_array([new_v[a] for a in order]), ptr(self.skin_weights), ptr(new_skin), B,
                                      ptr(noise), stream()), "mgr_densify_apply")
        self.p, self.m, self.v, self.skin_weights = new_p, new_m, new_v, new_skin
        self._reset_stats()
        self.replaced = ALL_GROUPS
        return dict(kept=int(counts[0]), cloned=int(counts[1]), split_selected=n_sel, split_kept=int(counts[3]), total=M)

    # -- prune_points, gaussian.py:185-203 -------------------------------------------------------
    def prune_points(self, mask, extra=None):
        """Remove the Gaussians where `mask` (N,) is True: leaves, both Adam moments, skin weights and the three
        statistics keep the surviving rows in order.  `extra`: optional {name: (N, ...) 4-byte-element tensor}
        compacted the same way (returned as a dict).  Returns the number of survivors."""
        N, dev = self.N, self.device
        mask = torch.as_tensor(mask, device=dev).reshape(-1)
        if mask.numel() != N:
            raise ManusHipError("prune_points: mask must have %d entries" % N)
        mask = (mask != 0).to(torch.uint8).contiguous()
        nbytes = int(lib().mgr_densify_workspace_bytes(N))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        counts = (ctypes.c_int64 * 5)()
        check(lib().mgr_prune_plan(N, ptr(mask), ptr(ws), nbytes, counts, stream()), "mgr_prune_plan")
        M = int(counts[4])
        shp = lambda a: (M,) + tuple(self.p[a].shape[1:])
        order = [a for _, a, _ in GROUPS]
        new_p = {a: torch.empty(shp(a), dtype=torch.float32, device=dev) for a in order}
        new_m = {a: torch.empty(shp(a), dtype=torch.float32, device=dev) for a in order}
        new_v = {a: torch.empty(shp(a), dtype=torch.float32, device=dev) for a in order}
        B = self.skin_weights.shape[1] if self.skin_weights is not None else 0
        new_skin = torch.empty((M, B), dtype=torch.float32, device=dev) if B else None
        check(lib().mgr_densify_apply(N, M, 0, ptr(ws), _ptr_array([self.p[a] for a in order]),
                                      _ptr_array([self.m[a] for a in order]), _ptr_array([self.v[a] for a in order]),
                                      _ptr_array([new_p[a] for a in order]), _ptr_array([new_m[a] for a in order]),
                                      _ptr_array([new_v[a] for a in order]), ptr(self.skin_weights), ptr(new_skin), B,
                                      None, stream()), "mgr_densify_apply")

        def gather(t):
            t = t.contiguous()
            if t.element_size() != 4:
                raise ManusHipError("prune_points: side arrays must have 4-byte elements")
            out = torch.empty((M,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
            width = int(t.numel() // max(N, 1))
            check(lib().mgr_gather_rows(N, M, ptr(ws), ptr(t), ptr(out), width, stream()), "mgr_gather_rows")
            return out

        self.xyz_gradient_accum, self.denom = gather(self.xyz_gradient_accum), gather(self.denom)
        self.max_radii2D = gather(self.max_radii2D)
        extra_out = {k: gather(v) for k, v in (extra or {}).items()}
        self.p, self.m, self.v, self.skin_weights = new_p, new_m, new_v, new_skin
        self.replaced = ALL_GROUPS
        self.last_prune_extra = extra_out
        return M

    # -- row order (no reference counterpart) --------------------------------------------------------------------------
    def sort_rows(self, n_art=None):
        """Reorder the Gaussians along a Z-order curve of their canonical positions: leaves, both Adam moments, skin
        weights and the three statistics move together, so the model and its optimisation are the same up to the row
        permutation (returned, (N,) int64: new row r is old row perm[r]).  Meant for the moment densification has just
        rebuilt every tensor anyway: the rasterizer gathers 48-byte records of the Gaussians of a tile, and rows that
        are neighbours in space being neighbours in memory is worth ~3 % of the fused step on a shuffled 300 k model
        (LAB.md, "Row order"; `bench.py --gaussian-order morton`).  The reference keeps whatever order initialisation,
        cat() of clones / splits and boolean-mask pruning leave (gaussian.py:167-321).
        n_art: a composite model keeps its articulated rows in front of the static ones (the fused kernels skin the
        first n_art rows, and `skin_weights` has rows for those only): the two segments are sorted separately.  Default:
        the rows `skin_weights` covers when it covers a prefix only, else one segment."""
        N = self.N
        if N == 0:
            return torch.zeros(0, dtype=torch.long, device=self.device)
        if n_art is None and self.skin_weights is not None and self.skin_weights.shape[0] < N:
            n_art = int(self.skin_weights.shape[0])
        n_art = N if n_art is None else int(n_art)
        if not 0 <= n_art <= N:
            raise ManusHipError("sort_rows: n_art must lie in [0, N]")
        if self.skin_weights is not None and self.skin_weights.shape[0] not in (N, n_art):
            raise ManusHipError("sort_rows: skin_weights covers %d rows, neither N nor n_art" % self.skin_weights.shape[0])
        xyz = self.p["_xyz"]
        lo, hi = xyz.min(0).values, xyz.max(0).values
        q = ((xyz - lo) / (hi - lo).clamp_min(1e-12) * 1023.0).long().clamp_(0, 1023)
        code = torch.zeros(N, dtype=torch.long, device=xyz.device)
        for bit in range(10):
            for ax in range(3):
                code |= ((q[:, ax] >> bit) & 1) << (3 * bit + ax)
        if n_art < N:     # the static segment stays behind the articulated one
            code += (torch.arange(N, device=xyz.device) >= n_art).long() << 31
        perm = torch.argsort(code, stable=True)
        take = lambda t: t.index_select(0, perm).contiguous()
        self.p = {a: take(t) for a, t in self.p.items()}
        self.m = {a: take(t) for a, t in self.m.items()}
        self.v = {a: take(t) for a, t in self.v.items()}
        if self.skin_weights is not None:
            self.skin_weights = take(self.skin_weights) if self.skin_weights.shape[0] == N else \
                self.skin_weights.index_select(0, perm[:n_art]).contiguous()
        self.xyz_gradient_accum, self.denom, self.max_radii2D = take(self.xyz_gradient_accum), take(self.denom), take(self.max_radii2D)
        self.replaced = ALL_GROUPS
        extra = getattr(self, "last_prune_extra", None)       # side arrays of the last prune follow their rows
        if extra:
            self.last_prune_extra = {k: (take(t) if t.shape[0] == N else t) for k, t in extra.items()}
        return perm

    # -- reset_opacity, gaussian.py:148-165 -----------------------------------------------------
    def reset_opacity(self):
        check(lib().mgr_reset_opacity(self.N, ptr(self.p["_opacity"]), ptr(self.m["_opacity"]), ptr(self.v["_opacity"]),
                                      stream()), "mgr_reset_opacity")
        self.replaced = self.replaced | {"opacity"}

    # -- density_update, gaussian_utils.py:451-498 -------------------------------------------------
    def density_update(self, stats, extent, global_step, bg_white=True, mask_to_prune=None, noise=None,
                       noise_is_pool=False):
        """`density_update` on the statistics of a step (dict with grad2d, vis, radii as returned by
        `engine.ViewShardedStep.step`).  mask_to_prune given: prune those Gaussians and nothing else (:454-459).
        Returns True when leaves were replaced (`self.replaced` names the groups: the next step() skips them,
        like the reference's optimizer.step() skips a gradient-less new nn.Parameter)."""
        o, changed = self.opts, False
        if mask_to_prune is not None:
            self.prune_points(mask_to_prune)
            return True
        if global_step < o["densify_until_step"]:
            self.add_densification_stats(stats["grad2d"], stats["vis"], stats["radii"])
            if global_step > o["densify_from_step"] and global_step % o["densification_interval"] == 0:
                size_threshold = o["size_threshold"] if global_step > o["opacity_reset_interval"] else None
                self.last_densify = self.densify_and_prune(o["densify_grad_threshold"], o["min_opacity_threshold"], extent,
                                                           size_threshold, global_step == o["remove_outliers_step"],
                                                           noise=noise, noise_is_pool=noise_is_pool)
                changed = True
            if global_step % o["opacity_reset_interval"] == 0 or (bg_white and global_step == o["densify_from_step"]):
                if global_step != 0:
                    self.reset_opacity()
                    changed = True
        return changed
