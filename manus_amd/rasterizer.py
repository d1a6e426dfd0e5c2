"""Drop-in operator surface of `diff_gaussian_rasterization` on MI355X.

Mirrors the interface MANUS imports at src/utils/gaussian_utils.py:18-21 and calls
at :378-416 (reference tree brown-ivl/manus):

    GaussianRasterizationSettings(image_height, image_width, tanfovx, tanfovy, bg,
        scale_modifier, viewmatrix, projmatrix, sh_degree, campos, prefiltered, debug)
    GaussianRasterizer(raster_settings)(means3D, means2D, opacities, shs=None,
        colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None)
        -> (color (3,H,W), radii (N,) int32)

Everything is computed by hand-written HIP kernels through the C ABI of
libmanus_hip.so; there is no PyTorch fallback.  `rasterize_views` is the
multi-view batched form (V cameras in every launch) used by the training engine.
"""
from typing import NamedTuple

import os

import torch
import torch.nn as nn

from . import _lib
from ._lib import check, f32c, lib, ptr, stream


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


# ---------------------------------------------------------------------------
# per-device context: workspace pool, learnt pair capacities, overflow fences
# ---------------------------------------------------------------------------
class RasterWorkspace:
    """One opaque byte tensor that links a forward to its backward."""
    # sort items with a single depth bucket beyond the light launch's LDS that the light launch may take on itself: none --
    # such a bucket goes through its global-memory fallback, ~90 us, which with one view on the GPU is the whole kernel
    # (measured: 0.039 -> 0.093 ms)
    SORT_BIG_MAX = int(os.environ.get("MANUS_SORT_BIG_MAX", "0"))

    def __init__(self, device, V, N, W, H, cap):
        self.key = (V, N, W, H)
        self.cap = int(cap)
        self.nbytes = int(lib().mgr_raster_workspace_bytes(V, N, W, H, self.cap))
        # zero-filled once: pair tags start at 0 = "never written"
        self.buf = torch.zeros(self.nbytes, dtype=torch.uint8, device=device)
        self.busy = False
        # depth cut (mgr_views_forward, debug bit 8): whose views the per-tile hints in this workspace describe (set by
        # the caller that opts in), and "the last forward with the cut was flagged: run the next one without it"
        self.hint_key = self.prev_hint_key = None
        self.cut_block = False
        self.mirror = None
        # binning tiers (debug bits 16 / 32 of the forward): which LDS tiers beyond the smallest the most recent forward
        # whose header was read needed (None: unknown -- launch them all)
        self.tiers = None
        # set by a forward that runs without a host read but promises the caller a complete result (the operator route's
        # automatic fences): launches whose absence would have to be answered by a re-run are then never skipped
        self.no_flagging_skips = False

    def skip_bits(self):
        """debug bits that spare the forward the binning launches its views did not need last time (they are verified on
        the device: a view that needs a skipped launch flags the forward, which is then run again with all of them)."""
        if self.tiers is None:
            return 0
        # sort items of the previous forward beyond 13/16 of k_dbin_rank's small / large capacity, and beyond the small capacity
        near_s, near_l, large = (self.tiers >> 8) & 0xFF, (self.tiers >> 16) & 0xFF, (self.tiers >> 24) & 0x7F
        near = near_l if large else near_s          # (the count that belongs to the instantiation this forward asks for)
        bits = (0 if self.tiers & 1 else 16) | (0 if self.tiers & 2 else 32) | (128 if near <= RasterWorkspace.SORT_BIG_MAX else 0) | (256 if large else 0)
        bits |= 4096 if self.tiers & 4 else 0       # rectangles of more than 64 tiles met: k_bin_scatter's lane-spreading instantiation
        # (bit 128 = "the previous forward met no sort item near the LDS of k_dbin_rank: skip the launch behind it" -- verified on the
        # device like the tile-box tiers, MGR_OVF_TIER; bit 256 = "it met items beyond MGR_DB_RANK_MAX keys: k_dbin_rank's instantiation
        # for larger items" -- a dense depth slice then costs that kernel ~4 us instead of 33 us in the launch behind)
        return bits & ~(48 | 128) if self.no_flagging_skips else bits


def default_pair_capacity(V, N):
    return max(4096, 8 * V * max(N, 1))


class RasterContext:
    """All host-side rasterizer state of ONE device (SURVEY.md 8b "Threading": one host thread per device, re-entrant
    across devices): the workspace pool, the pair capacity learnt per (V,N,W,H), the sync policy and the overflow
    fences of forwards that ran without a host synchronisation."""

    MAX_FENCES = 64
    # Operator route (`GaussianRasterizer` under autograd, sync policy left at its default): after this many consecutive
    # forwards of one (V, N, W, H) whose pair count fitted the learnt capacity, the blocking read of the pair count per forward
    # (upstream's cudaMemcpy; 0.23 ms of a 1.47 ms step at 1280x720, profiles/r05_other_configs/dropin_c*) is replaced by a
    # fence that the NEXT forward of the context resolves -- by then it has long been written.  The capacity keeps its 25 %
    # headroom; should a forward outgrow it all the same, the next forward raises ManusHipError (the step before it used an
    # incomplete image) and the context returns to the blocking mode.  0 = never switch; a densification (new N) starts over.
    AUTO_FENCE_AFTER = int(os.environ.get("MANUS_AUTO_FENCE_AFTER", "8"))

    def __init__(self, device):
        self.device = torch.device(device)
        self.pool = {}
        self.cap_hint = {}
        self.sync_every_forward = True
        self.last_ws = None
        self._fences = []        # (workspace, pinned int32[2], event) per unsynchronised forward, oldest first
        self._free_pinned = []
        self._evicted_overflow = False   # an overflow seen while retiring old fences: raised by the next poll()
        self.cut_retries = 0             # forwards flagged MGR_OVF_CUT (each is answered by a forward without the depth cut)
        self.cut_repairs = 0             # (tile, quadrant) units repaired on the device by forwards with the depth cut (bit 2048)
        self._clean = {}                 # (V, N, W, H) -> consecutive synchronised forwards that fitted their capacity
        self.auto_fenced = 0             # forwards that ran on an automatic fence instead of the blocking read
        self.tier_retries = 0            # forwards flagged MGR_OVF_TIER (answered by a forward with every binning launch)

    # -- pool ---------------------------------------------------------------------------------
    def acquire(self, V, N, W, H, min_cap):
        lst = self.pool.setdefault((V, N, W, H), [])
        for ws in lst:
            if not ws.busy and ws.cap >= min_cap:
                ws.busy = True
                return ws
        lst[:] = [w for w in lst if w.busy]   # drop idle smaller workspaces before growing
        ws = RasterWorkspace(self.device, V, N, W, H, min_cap)
        ws.busy = True
        lst.append(ws)
        return ws

    def clear(self):
        """Forget pooled workspaces and learnt capacities (e.g. after densification changed N)."""
        if self._fences:                        # the forwards' last kernels write into the pinned words: wait before recycling
            torch.cuda.synchronize(self.device)
        for ws, pinned, ev in self._fences:
            if pinned is not None:
                self._free_pinned.append(pinned)
        for lst in self.pool.values():          # withdraw mirrors that no forward took
            for ws in lst:
                if getattr(ws, "mirror", None) is not None:
                    lib().mgr_raster_set_status_mirror(ptr(ws.buf), None)
                    self._free_pinned.append(ws.mirror)
                    ws.mirror = None
        self.pool.clear()
        self.cap_hint.clear()
        self._fences.clear()
        self._evicted_overflow = False
        self.last_ws = None
        self._clean.clear()

    def _learn(self, key, npairs):
        self.cap_hint[key] = max(self.cap_hint.get(key, 0), int(npairs * 1.25) + 4096)

    # -- forward driver -------------------------------------------------------------------------
    def forward(self, V, N, W, H, launch, sync_check=True, defer_fence=False, auto_fence=False):
        """Run `launch(ws)` (which enqueues one forward on the current stream) with a workspace large enough for
        the pairs it produces.  sync policy True: read the pair count back (one host sync, like upstream) and
        retry with a larger workspace on overflow; False: no host sync, an overflow fence is recorded instead
        (`poll()` / `check_overflow()`).  Returns (workspace, pair count or None).  defer_fence: the caller records the
        fence itself (`fence(ws)`) once everything that can raise a flag is queued -- a forward split at the blend raises
        the depth-cut flag in its second half."""
        key = (V, N, W, H)
        cap = max(self.cap_hint.get(key, 0), default_pair_capacity(V, N))
        # auto_fence (the operator route): see AUTO_FENCE_AFTER
        auto = bool(auto_fence) and sync_check and self.sync_every_forward and 0 < self.AUTO_FENCE_AFTER <= self._clean.get(key, 0)
        if auto:
            try:
                self.poll()              # the automatic fences of the forwards before (complete by now)
            except _lib.ManusHipError:
                self._clean[key] = 0
                raise _lib.ManusHipError("an unsynchronised forward of the previous step outgrew its pair capacity: its image and "
                                         "gradients were incomplete.  The capacity was enlarged and this context reads the pair count "
                                         "back again; set MANUS_AUTO_FENCE_AFTER=0 to keep every forward synchronous.")
        while True:
            ws = self.acquire(V, N, W, H, cap)
            # whoever launches may claim the depth-cut hints of the forward before (prev_hint_key) and name the views of
            # this one; a launch that does neither leaves hints nobody may use
            ws.prev_hint_key, ws.hint_key = ws.hint_key, None
            fenced = auto or not (sync_check and self.sync_every_forward)
            ws.no_flagging_skips = auto
            if fenced:
                self._arm_mirror(ws)
            launch(ws)
            self.last_ws = ws
            if fenced:
                if auto:
                    self.auto_fenced += 1
                if not defer_fence:
                    self._fence(ws)
                return ws, None
            import ctypes
            npairs, ovf, tiers = ctypes.c_int64(0), ctypes.c_int32(0), ctypes.c_int32(0)
            rc = lib().mgr_raster_status_tiers_sync(ptr(ws.buf), ctypes.byref(npairs), ctypes.byref(ovf), ctypes.byref(tiers), stream())
            ws.tiers = None if (ovf.value & 4) else int(tiers.value)
            if rc == 0:
                self._learn(key, npairs.value)
                self._clean[key] = self._clean.get(key, 0) + 1
                return ws, int(npairs.value)
            self._clean[key] = 0
            if rc == -7 and not (ovf.value & 3):   # MGR_ETIER: a skipped binning launch was needed; same workspace, all launches
                self.tier_retries += 1
                ws.busy = False
                continue
            if rc == -6 and not (ovf.value & 1):   # MGR_ECUT: the depth-cut hints no longer fit; same workspace, no cut
                ws.cut_block = True
                self.cut_retries += 1
                ws.busy = False
                continue
            if rc != -4:
                check(rc, "mgr_raster_status_sync")
            ws.busy = False  # overflow: retry with room for the observed count
            cap = int(npairs.value * 1.5) + 4096

    # -- overflow fences ------------------------------------------------------------------------
    def fence(self, ws):
        self._fence(ws)

    def fenced(self, sync_check=True):
        """True when forwards run without a host synchronisation (fences instead)."""
        return not (sync_check and self.sync_every_forward)

    def _arm_mirror(self, ws):
        """Before an unsynchronised forward: four pinned words the forward's last kernel writes its status to
        (mgr_raster_set_status_mirror) -- the fence is then an event only, no device-to-host copy on the stream."""
        while len(self._fences) >= self.MAX_FENCES:
            _, ovf = self._resolve(self._fences.pop(0))
            self._evicted_overflow = self._evicted_overflow or bool(ovf)
        pinned = self._free_pinned.pop() if self._free_pinned else torch.zeros(4, dtype=torch.int32).pin_memory()
        pinned[3] = 0
        import ctypes
        rc = lib().mgr_raster_set_status_mirror(ptr(ws.buf), ctypes.c_void_p(pinned.data_ptr()))
        ws.mirror = pinned if rc == 0 else None      # (not mappable: the fence falls back to a blocking read)

    def _fence(self, ws):
        """Remember the forward just queued: the host can later wait for THIS forward only, while the kernels queued after
        it keep the GPU busy.  With a mirror armed nothing at all goes onto the stream -- the forward's last kernel writes
        the status words and then their valid flag into pinned memory, and the host waits on that flag (an event behind
        the forward costs ~6 us of idle GPU: the next kernel does not start until it has signalled).  Without a mirror:
        an event."""
        mirror = getattr(ws, "mirror", None)
        ev = None
        if mirror is None:
            ev = torch.cuda.Event()
            ev.record()
        self._fences.append((ws, mirror, ev))
        ws.mirror = None

    @staticmethod
    def _wait_flag(pinned, seconds=5.0):
        """Spin (politely) until the device has written the valid flag of a status mirror."""
        import time
        t0, n = time.perf_counter(), 0
        while int(pinned[3]) != 1:
            n += 1
            if n > 200:
                time.sleep(0.00005)
            if time.perf_counter() - t0 > seconds:
                return False
        return True

    def _resolve(self, fence):
        ws, pinned, ev = fence
        if ev is not None:
            ev.synchronize()
        elif pinned is not None and not self._wait_flag(pinned):
            torch.cuda.synchronize(self.device)      # (a forward whose blend never ran: nothing will write the flag)
        if pinned is not None and int(pinned[3].item()) == 1:
            npairs, word, tiers_seen = int(pinned[0].item()) & 0xFFFFFFFF, int(pinned[1].item()) & 0xFFFFFFFF, int(pinned[2].item())
            ovf = word & 0xFFFF
            self.cut_repairs += word >> 16       # quadrants of depth-cut tiles the forward repaired on the device (no re-run)
        else:   # (a forward that did not run its blend, or no mirror: read the header -- valid if nothing ran on ws since)
            import ctypes
            n_, o_, t_ = ctypes.c_int64(0), ctypes.c_int32(0), ctypes.c_int32(0)
            lib().mgr_raster_status_tiers_sync(ptr(ws.buf), ctypes.byref(n_), ctypes.byref(o_), ctypes.byref(t_), stream())
            npairs, ovf, tiers_seen = int(n_.value), int(o_.value), int(t_.value)
        if pinned is not None:
            self._free_pinned.append(pinned)
        self._learn(ws.key, npairs)
        if ovf & 2:     # MGR_OVF_CUT: the caller re-runs the step; that forward must not use the hints
            ws.cut_block = True
            self.cut_retries += 1
        if ovf & 4:     # MGR_OVF_TIER: ... and with every binning launch
            ws.tiers = None
            self.tier_retries += 1
        else:
            ws.tiers = tiers_seen
        return npairs, ovf

    def poll(self):
        """Wait for the forwards recorded so far (not for what was queued after them) and raise ManusHipError if one
        of them overflowed its pair capacity -- its image and gradients are incomplete; the capacity hint has been
        enlarged, so re-running the step succeeds.  Returns the pair count of the most recent forward."""
        last, bad = 0, self._evicted_overflow
        self._evicted_overflow = False
        while self._fences:
            npairs, ovf = self._resolve(self._fences.pop(0))
            last, bad = npairs, bad or bool(ovf)
        if bad:
            raise _lib.ManusHipError("rasterizer forward incomplete: pair capacity exceeded or depth-cut hints outdated "
                                     "(retry: the capacity hint was enlarged / the next forward runs without the cut)")
        return last

    def check_overflow(self):
        """Blocking check of every forward since the last check (fences) and of the most recent workspace; returns
        the most recent pair count, raises ManusHipError on overflow (after enlarging the capacity hint)."""
        import ctypes
        polled = self.poll()
        ws = self.last_ws
        if ws is None:
            return polled
        npairs, ovf, tiers = ctypes.c_int64(0), ctypes.c_int32(0), ctypes.c_int32(0)
        rc = lib().mgr_raster_status_tiers_sync(ptr(ws.buf), ctypes.byref(npairs), ctypes.byref(ovf), ctypes.byref(tiers), stream())
        self._learn(ws.key, npairs.value)
        ws.tiers = None if (ovf.value & 4) else int(tiers.value)
        if ovf.value & 4:
            self.tier_retries += 1
        if rc == -6:
            ws.cut_block = True
            self.cut_retries += 1
        if rc != 0:
            check(rc, "rasterizer overflow check (retry: capacity hint was enlarged)")
        return int(npairs.value)


_CONTEXTS = {}


def context(device=None):
    """The RasterContext of `device` (default: the current device)."""
    idx = torch.cuda.current_device() if device is None else (torch.device(device).index
                                                              if torch.device(device).index is not None
                                                              else torch.cuda.current_device())
    ctx = _CONTEXTS.get(idx)
    if ctx is None:
        ctx = _CONTEXTS[idx] = RasterContext(torch.device("cuda", idx))
    return ctx


def set_sync_policy(sync_every_forward, device=None):
    """True (default, drop-in behaviour): every forward reads back the pair count
    (one host sync, like the upstream extension) and transparently retries with a
    larger workspace on overflow.  False (training engine): no host sync; the
    capacity learnt so far is used and `poll()` / `check_overflow()` report overflows."""
    context(device).sync_every_forward = bool(sync_every_forward)


def check_overflow(device=None):
    return context(device).check_overflow()


def poll(device=None):
    return context(device).poll()


class _Lease:
    """Releases the workspace when the autograd graph that needs it is freed."""

    def __init__(self, ws):
        self.ws = ws

    def __del__(self):
        self.ws.busy = False


def _run_forward(cams, V, N, W, H, bg, means3D, cov3D, colors, opacity, debug, sync_check=True):
    dev = means3D.device
    out = torch.empty((V, 3, H, W), dtype=torch.float32, device=dev)
    radii = torch.empty((V, N), dtype=torch.int32, device=dev)
    s_m = means3D.stride(0) if means3D.dim() == 3 else 0
    s_c = cov3D.stride(0) if cov3D.dim() == 3 else 0
    s_col = colors.stride(0) if colors.dim() == 3 else 0
    s_o = opacity.stride(0) if opacity.dim() == 2 else 0

    def launch(ws):
        check(lib().mgr_raster_forward(V, N, W, H, ptr(cams), ptr(bg), ptr(means3D), s_m, ptr(cov3D), s_c,
                                       ptr(colors), s_col, ptr(opacity), s_o, ptr(out), ptr(radii),
                                       ptr(ws.buf), ws.nbytes, ws.cap, int(bool(debug)) | ws.skip_bits(), stream()),
              "mgr_raster_forward")

    ws, npairs = context(dev).forward(V, N, W, H, launch, sync_check, auto_fence=True)
    return out, radii, ws, npairs


def _opacity_layout(op, V, N):
    """(N,1)/(N,) -> shared (N); (V,N,1)/(V,N) -> per view (V,N)."""
    if op.dim() == 3:
        return op.reshape(V, N)
    if op.dim() == 2 and op.shape[1] == 1:
        return op.reshape(N)
    if op.dim() == 2:
        return op.reshape(V, N)
    return op.reshape(N)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, colors, opacities, cov3D, cams, bg, W, H, debug):
        V = cams.shape[0]
        means3D, colors, cov3D = f32c(means3D), f32c(colors), f32c(cov3D)
        opac = f32c(opacities)
        N = means3D.shape[-2]
        opac = _opacity_layout(opac, V, N)
        bg = f32c(bg).reshape(-1)
        out, radii, ws, npairs = _run_forward(cams, V, N, W, H, bg, means3D, cov3D, colors, opac, debug)
        ctx.lease = _Lease(ws)
        ctx.meta = (V, N, W, H, bool(debug), means2D.shape, opacities.shape)
        ctx.num_rendered = npairs
        ctx.save_for_backward(means3D, colors, opac, cov3D, cams, bg, out)
        ctx.mark_non_differentiable(radii)
        return out, radii

    @staticmethod
    def backward(ctx, g_color, _g_radii):
        means3D, colors, opac, cov3D, cams, bg, out = ctx.saved_tensors
        V, N, W, H, debug, m2d_shape, op_shape = ctx.meta
        ws = ctx.lease.ws
        dev = means3D.device
        g_color = f32c(g_color)
        d_m3 = torch.empty((V, N, 3), dtype=torch.float32, device=dev)
        d_m2 = torch.empty((V, N, 3), dtype=torch.float32, device=dev)
        d_col = torch.empty((V, N, 3), dtype=torch.float32, device=dev)
        d_op = torch.empty((V, N), dtype=torch.float32, device=dev)
        d_cov = torch.empty((V, N, 6), dtype=torch.float32, device=dev)
        s_m = means3D.stride(0) if means3D.dim() == 3 else 0
        s_c = cov3D.stride(0) if cov3D.dim() == 3 else 0
        s_col = colors.stride(0) if colors.dim() == 3 else 0
        s_o = opac.stride(0) if opac.dim() == 2 else 0
        check(lib().mgr_raster_backward(V, N, W, H, ptr(cams), ptr(bg), ptr(means3D), s_m, ptr(cov3D), s_c,
                                        ptr(colors), s_col, ptr(opac), s_o, ptr(out), ptr(g_color), ptr(d_m3), ptr(d_m2),
                                        ptr(d_col), ptr(d_op), ptr(d_cov), ptr(ws.buf), ws.nbytes, ws.cap,
                                        int(debug), stream()), "mgr_raster_backward")

        def fold(g, shared, shape=None):
            # inputs shared by all views receive the sum over views
            if shared:
                g = g.sum(0) if V > 1 else g[0]
            return g.reshape(shape) if shape is not None else g

        g_m3 = fold(d_m3, means3D.dim() == 2)
        g_m2 = fold(d_m2, len(m2d_shape) == 2, m2d_shape)
        g_col = fold(d_col, colors.dim() == 2)
        g_op = fold(d_op, opac.dim() == 1, op_shape)
        g_cov = fold(d_cov, cov3D.dim() == 2)
        return g_m3, g_m2, g_col, g_op, g_cov, None, None, None, None, None


def rasterize_views(cams, means3D, means2D, colors, opacities, cov3D, bg, W, H, debug=False):
    """V views in one call.  cams (V,40) from `_lib.pack_cameras`; per-Gaussian
    inputs are (N,..) shared by all views or (V,N,..).  Returns color (V,3,H,W),
    radii (V,N)."""
    return _RasterizeGaussians.apply(means3D, means2D, colors, opacities, cov3D, cams, bg, int(W), int(H), debug)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """In-frustum mask (view-space z > 0.2), as the upstream helper."""
        with torch.no_grad():
            vm = self.raster_settings.viewmatrix.reshape(4, 4).to(positions)
            z = positions @ vm[:3, 2] + vm[3, 2]
            return z > 0.2

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None,
                rotations=None, cov3D_precomp=None):
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                (scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        dev = means3D.device
        if not means3D.is_cuda:
            raise _lib.ManusHipError("GaussianRasterizer needs GPU tensors; there is no CPU fallback")
        cams = _lib.pack_cameras(rs.tanfovx, rs.tanfovy, rs.viewmatrix, rs.projmatrix, rs.campos, dev)
        if cov3D_precomp is None:
            from .ops import lbs_cov
            _, cov, _ = lbs_cov(means3D, torch.log(scales * rs.scale_modifier), rotations, None, None)
            cov3D_precomp = cov[0]
        if colors_precomp is None:
            from .ops import sh_colors
            K = (rs.sh_degree + 1) ** 2
            sh = shs
            if sh.shape[1] < 16 or K < 16:
                full = torch.zeros((sh.shape[0], 16, 3), dtype=sh.dtype, device=dev)
                k = min(K, sh.shape[1])
                full[:, :k] = sh[:, :k]
                sh = full
            colors_precomp = sh_colors(sh, means3D, None, cams)[0]
        color, radii = _RasterizeGaussians.apply(means3D, means2D, colors_precomp, opacities, cov3D_precomp,
                                                 cams, rs.bg, int(rs.image_width), int(rs.image_height),
                                                 bool(rs.debug))
        return color[0], radii[0]
