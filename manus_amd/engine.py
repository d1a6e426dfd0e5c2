"""Multi-view training step: V views batched per GPU, views sharded across GPUs,
ONE all-reduce of the per-Gaussian gradients per step (RCCL over xGMI via torch.distributed).

The reference trains one (frame, view) per step on one GPU (config/trainer/trainer.yaml:5,
main.py:84-87 DDP commented out).  "V views per iteration" is defined here as
    grad = (1/V) * sum_v grad(L_v)            (with V=1 it is the reference step)
and the densification statistics follow the reference's per-view rule
(src/models/gaussian.py:335-338, src/utils/gaussian_utils.py:469-471):
    xyz_gradient_accum += sum_v ||d L_v / d means2D[:, :2]||   (visible Gaussians only)
    denom              += sum_v visible_v
    max_radii2D         = max_v radii_v

Collectives per step: one SUM all-reduce of the flat buffer
    [ 59 N leaf gradients | N grad2d | N visibility counts | loss | overflow flag ]
which the backward kernels write in place (no packing copies).  max_radii2D needs a MAX, not a SUM; max is
associative and idempotent, so every rank keeps the running maximum over ITS views and the ranks are only
combined (one MAX all-reduce) right before the statistic is consumed, i.e. at a densification step.
"""
import os

import torch
import torch.distributed as dist

# packed leaf-gradient layout: 59 floats per Gaussian
GRAD_LAYOUT = (("_xyz", 3), ("_features_dc", 3), ("_features_rest", 45), ("_opacity", 1),
               ("_scaling", 3), ("_rotation", 4))
GRAD_WIDTH = sum(w for _, w in GRAD_LAYOUT)
FLAT_TAIL = 2            # loss, overflow flag


def flat_size(N):
    return N * (GRAD_WIDTH + 2) + FLAT_TAIL


def shard_views(n_views, rank, world_size, weights=None):
    """View indices of `rank`.  Without weights: round-robin.  With weights (one cost per view, identical on every
    rank -- e.g. `view_costs` of the measured pair counts): greedy longest-processing-time assignment, the heaviest
    view first, each to the rank with the smallest load so far (ties: the lowest rank; views of a rank in ascending
    order), so that a rig whose cameras see the hand at very different sizes does not leave ranks idle (SURVEY.md 8e:
    "pad or balance by measured R per view")."""
    if weights is None:
        return list(range(rank, n_views, world_size))
    w = [float(x) for x in weights]
    assert len(w) == n_views
    load, mine = [0.0] * world_size, [[] for _ in range(world_size)]
    for v in sorted(range(n_views), key=lambda i: (-w[i], i)):
        r = min(range(world_size), key=lambda k: (load[k], k))
        load[r] += w[v]
        mine[r].append(v)
    return sorted(mine[rank])


def view_costs(pairs_per_view, n_gaussians):
    """Cost model of one view for `shard_views`: the blend / binning kernels scale with the view's (tile, Gaussian) pairs,
    the per-instance kernels with N.  Measured on the 300k / 1080p bench (profiles/): 0.16 ms per 2.6 M pairs against
    0.066 ms per 300k Gaussians, i.e. one Gaussian costs as much as 3.5 pairs."""
    return [float(p) + 3.5 * float(n_gaussians) for p in pairs_per_view]


def flat_views(flat, N):
    """Named views of the flat step buffer: one contiguous segment per leaf (GRAD_LAYOUT order), the two
    statistics, the loss and the overflow flag."""
    out, o = {}, 0
    for name, w in GRAD_LAYOUT:
        out[name] = flat[o:o + N * w]
        o += N * w
    out["grad2d"], out["vis"] = flat[o:o + N], flat[o + N:o + 2 * N]
    out["loss"], out["overflow"] = flat[o + 2 * N:o + 2 * N + 1], flat[o + 2 * N + 1:o + 2 * N + 2]
    return out


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
    """Runs `compute_fn(view_ids, scale)` on this rank's views and reduces across ranks.

    compute_fn returns a dict with
        grads      {leaf name: scale * sum over the given views of dL_v/dleaf}
        grad2d     (N,) sum over views of the visible 2D-gradient norms (unscaled)
        vis        (N,) number of views in which the Gaussian was visible
        radii      (N,) int32 max screen radius over views
        loss       scalar tensor, scale * sum of L_v
        overflow   optional scalar tensor, non-zero when the rasterizer ran out of pair capacity
    If compute_fn has a `grad_arena` attribute it is handed views of the step buffer and writes them in place.

    step() returns the reduced grads / grad2d / vis / loss / overflow (sums over all ranks) and the LOCAL radii
    (max over this rank's views); `reduce_max_radii` combines the ranks when the statistic is consumed.

    Reduction modes (world_size > 1):
        default        one SUM all-reduce of [59 N gradients | N grad2d | N vis | loss | overflow]
        compact=True   only the rows with a gradient on some rank travel (see `_compact_all_reduce`)
        scatter=True   sharded optimizer step: a reduce-scatter of the gradient part leaves every rank with the summed
                       slice it owns (`owned`, element range of the flat gradient buffer; the returned grads are then
                       only valid inside that slice) + one small all-reduce of the statistics; the caller updates the
                       parameters it owns and all-gathers them with `all_gather_params`.
    """

    def __init__(self, n_gaussians, shapes, compute_fn, n_views, rank=0, world_size=1, group=None, compact=False,
                 scatter=False, view_weights=None, force_collectives=False):
        self.N, self.shapes, self.compute_fn = n_gaussians, shapes, compute_fn
        self.n_views, self.rank, self.world, self.group = n_views, rank, world_size, group
        self.local_views = shard_views(n_views, rank, world_size, view_weights)
        self.always_pack = False   # tests: take the packing path without a process group
        # force_collectives: issue every collective even in a world of one rank (a single-process "nccl" group on a one-GPU
        # box runs RCCL's all-reduce / reduce-scatter / all-gather code paths with the dtypes and in-place forms used here)
        self.force = bool(force_collectives)
        self.compact = bool(compact)
        self.scatter = bool(scatter) and (world_size > 1 or self.force)
        self.last_rows = None
        self._store = None
        n_g = self.N * GRAD_WIDTH
        self.padded_g = (n_g + world_size - 1) // world_size * world_size      # reduce-scatter needs equal slices
        c = self.padded_g // world_size
        self.owned = (rank * c, min((rank + 1) * c, n_g))
        dev = getattr(compute_fn, "device", None)
        if dev is not None and (world_size > 1 or self.force):
            self._alloc(dev)

    def _alloc(self, dev):
        # [ 59 N gradients | padding to a multiple of the world size | N grad2d | N vis | loss | overflow ]
        self._store = torch.zeros(self.padded_g + 2 * self.N + FLAT_TAIL, dtype=torch.float32, device=dev)

    @property
    def _flat(self):   # (tests look at the buffer the kernels write into)
        return self._store

    def _views(self):
        N, st = self.N, self._store
        out, o = {}, 0
        for name, w in GRAD_LAYOUT:
            out[name] = st[o:o + N * w]
            o += N * w
        o = self.padded_g
        out["grad2d"], out["vis"] = st[o:o + N], st[o + N:o + 2 * N]
        out["loss"], out["overflow"] = st[o + 2 * N:o + 2 * N + 1], st[o + 2 * N + 1:o + 2 * N + 2]
        return out

    def exchanged_rows(self):
        """Rows in the union of a recent compact exchange (the count reaches the host through an asynchronous copy behind the
        step: this waits for the one in flight, if any).  None before the first compact step."""
        sc = getattr(self, "_xch", None)
        if sc is not None and sc.get("ev") is not None:
            sc["ev"].synchronize()
            self.last_rows, sc["ev"] = int(sc["host"][0]), None
            sc["cap_rows"] = min(self.N, int(self.last_rows * 1.25) + 1024)
        return self.last_rows

    def reduce_max_radii(self, radii):
        """MAX over ranks of a per-Gaussian radius statistic (in place; any integer or float dtype)."""
        if self.world > 1 or self.force:
            dist.all_reduce(radii, op=dist.ReduceOp.MAX, group=self.group)
        return radii

    def all_gather_params(self, store):
        """In place on a parameter buffer laid out like the gradient part (`padded_g` floats): every rank contributes
        the slice it owns."""
        c = self.padded_g // self.world
        mine = store[self.rank * c:(self.rank + 1) * c]
        if dist.get_backend(self.group) == "gloo":      # (CPU / single-GPU tests; RCCL takes the one-call form)
            parts = [torch.empty_like(mine) for _ in range(self.world)]
            dist.all_gather(parts, mine.clone(), group=self.group)
            for r, t in enumerate(parts):
                store[r * c:(r + 1) * c].copy_(t)
        else:
            dist.all_gather_into_tensor(store, mine, group=self.group)

    def step(self):
        # compute_fn folds the 1/V of "grad = (1/V) sum_v grad L_v" into the loss scale
        N = self.N
        packed = self.world > 1 or self.always_pack or self.force
        if packed and self._store is not None and hasattr(self.compute_fn, "grad_arena"):
            # the backward kernels write straight into the step buffer (no packing copies): one view per leaf
            self.compute_fn.grad_arena = self._views()
        out = self.compute_fn(self.local_views, 1.0 / float(self.n_views))
        dev = out["grad2d"].device
        radii = out["radii"].to(torch.int32)
        if not packed:
            return dict(grads=out["grads"], grad2d=out["grad2d"], vis=out["vis"], radii=radii, loss=out["loss"],
                        overflow=out.get("overflow"))
        if self._store is None or self._store.device != dev:
            self._alloc(dev)
        st, n_g = self._store, N * GRAD_WIDTH
        fv = self._views()
        pack_grads(out["grads"], N, dev, out=st[:n_g])
        for name in ("grad2d", "vis"):
            if out[name].data_ptr() != fv[name].data_ptr():
                fv[name].copy_(out[name])
        fv["loss"].copy_(out["loss"].reshape(1))
        ovf = out.get("overflow")
        if ovf is None:
            fv["overflow"].zero_()
        else:
            fv["overflow"].copy_(ovf.reshape(1).to(torch.float32))
        if self.scatter:
            c = self.padded_g // self.world
            mine = st[self.rank * c:(self.rank + 1) * c]
            if dist.get_backend(self.group) == "gloo":      # gloo has no reduce-scatter: same sums, more bytes
                dist.all_reduce(st[:self.padded_g], op=dist.ReduceOp.SUM, group=self.group)
            else:
                dist.reduce_scatter_tensor(mine, st[:self.padded_g], op=dist.ReduceOp.SUM, group=self.group)
            dist.all_reduce(st[self.padded_g:], op=dist.ReduceOp.SUM, group=self.group)    # 2 N + 2 floats
        elif self.world > 1 or self.force or (self.always_pack and self.compact):
            if self.compact:
                self._compact_all_reduce(st, fv, active=getattr(self.compute_fn, "last_active", None))
            else:
                dist.all_reduce(st, op=dist.ReduceOp.SUM, group=self.group)   # the step's ONE collective
        grads = unpack_grads(st[:n_g], self.shapes, N)
        return dict(grads=grads, grad2d=fv["grad2d"], vis=fv["vis"], radii=radii, loss=fv["loss"][0],
                    overflow=fv["overflow"][0])


def _row_mask(fv, N):
    """Rows with any non-zero gradient entry (uint8).  Exact by construction: a row outside the mask is all zeros."""
    m = fv["grad2d"] != 0
    for name, w in GRAD_LAYOUT:
        m = m | (fv[name].view(N, w).abs().amax(dim=1) > 0)
    return m.to(torch.uint8)


_XCH_SEGS = list(GRAD_LAYOUT) + [("grad2d", 1)]


def _compact_all_reduce(self, flat, fv, active=None):
    """Two collectives: (1) one byte-sized SUM all-reduce of [row mask | visibility count] (2 N bytes: the visibility
    count is needed for every Gaussian -- a hidden Gaussian still counts as visible, gaussian.py:335-338 -- but it is
    at most the number of views, so a byte carries it); (2) the SUM all-reduce of the 60 floats (59 gradients + the
    2D-gradient norm) of the rows that are active on some rank.

    On the GPU everything around the two collectives is five launches of the library (csrc/exchange.hip): the mask -- from
    the fused backward's own list of the Gaussians that received a gradient when the compute function hands it over
    (`active`), otherwise from the rows --, the ordered row list (two launches), one pack, one unpack.  The host reads
    NOTHING in the middle of the step (round 6): the second collective is sized by a row capacity chosen beforehand -- the
    union's size of the step before + 25 % (+ 1024; all N rows the first time) --, the pack / unpack kernels take the actual
    count from the device, rows between count and capacity travel as zeros, and a union that outgrows the capacity adds to
    the step's overflow word (summed over the ranks: `Trainer._run_step` runs the step again, by which time the count has
    arrived on the host through an asynchronous copy and the capacity has grown).  `self.last_rows`: the count of the most
    recent step whose copy has arrived.  Tensors on the CPU (the gloo tests of the rank logic, where a CPU stand-in
    computes the gradients) take the same steps in torch."""
    N = self.N
    if self.n_views > 255:
        raise ValueError("compact all-reduce carries the visibility count in one byte: at most 255 views per step (use the dense mode)")
    live = self.world > 1 or self.force
    if flat.is_cuda:
        import ctypes
        from ._lib import check, lib, ptr, stream
        L = lib()
        dev = flat.device
        nseg = len(_XCH_SEGS)
        base = flat.data_ptr()
        offs = (ctypes.c_int64 * nseg)(*[(fv[name].data_ptr() - base) // 4 for name, _ in _XCH_SEGS])
        widths = (ctypes.c_int * nseg)(*[w for _, w in _XCH_SEGS])
        vis_off, tail_off = (fv["vis"].data_ptr() - base) // 4, (fv["loss"].data_ptr() - base) // 4
        sc = getattr(self, "_xch", None)
        if sc is None or sc["small"].device != dev:
            sc = self._xch = dict(small=torch.empty(2 * N, dtype=torch.uint8, device=dev), idx=torch.empty(N, dtype=torch.int32, device=dev),
                                  count=torch.zeros(1, dtype=torch.int32, device=dev),
                                  ws=torch.empty(L.mgr_exchange_index_workspace_bytes(N), dtype=torch.uint8, device=dev),
                                  host=torch.zeros(1, dtype=torch.int32).pin_memory(), buf=None, ev=None, cap_rows=N)
        small, idx, count = sc["small"], sc["idx"], sc["count"]
        if sc["ev"] is not None and sc["ev"].query():          # the count of an earlier step has arrived: the capacity follows it
            n_seen = int(sc["host"][0])
            self.last_rows, sc["ev"] = n_seen, None
            sc["cap_rows"] = min(N, int(n_seen * 1.25) + 1024)
        cap_rows = int(getattr(self, "row_capacity", None) or sc["cap_rows"])      # (row_capacity: a test's fixed capacity)
        lst, cnt = active if active is not None else (None, None)
        check(L.mgr_exchange_mask(N, ptr(flat), nseg, offs, widths, vis_off, lst, cnt, ptr(small), stream()), "mgr_exchange_mask")
        if live:
            dist.all_reduce(small, op=dist.ReduceOp.SUM, group=self.group)
        check(L.mgr_exchange_index(N, ptr(small), ptr(idx), ptr(count), ptr(sc["ws"]), sc["ws"].numel(), stream()), "mgr_exchange_index")
        need = cap_rows * (GRAD_WIDTH + 1) + FLAT_TAIL
        if sc["buf"] is None or sc["buf"].numel() < need:
            sc["buf"] = torch.empty(need + 64, dtype=torch.float32, device=dev)
        buf = sc["buf"][:need]
        check(L.mgr_exchange_pack_rows(N, cap_rows, ptr(count), ptr(idx), ptr(flat), nseg, offs, widths, tail_off, ptr(buf), stream()),
              "mgr_exchange_pack_rows")
        if live:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
        check(L.mgr_exchange_unpack_rows(N, cap_rows, ptr(count), ptr(idx), ptr(flat), nseg, offs, widths, tail_off, ptr(buf),
                                         small[N:].data_ptr(), vis_off, stream()), "mgr_exchange_unpack_rows")
        if sc["ev"] is None:                                   # the count travels to the host behind the step (read by a later one)
            sc["host"].copy_(count, non_blocking=True)
            sc["ev"] = torch.cuda.Event()
            sc["ev"].record()
        self.last_cap_rows = cap_rows
        return
    small = torch.cat([_row_mask(fv, N), fv["vis"].to(torch.uint8)])
    if live:
        dist.all_reduce(small, op=dist.ReduceOp.SUM, group=self.group)
    fv["vis"].copy_(small[N:])
    idx = torch.nonzero(small[:N], as_tuple=False)[:, 0]                     # (host sync: the collective's size)
    n = idx.shape[0]
    self.last_rows = n
    width = GRAD_WIDTH + 1
    buf = torch.empty(n * width + FLAT_TAIL, dtype=torch.float32, device=flat.device)
    o, segs = 0, []
    for name, w in _XCH_SEGS:
        seg = buf[o:o + n * w].view(n, w)
        torch.index_select(fv[name].view(N, w), 0, idx, out=seg)
        segs.append((name, w, seg))
        o += n * w
    buf[o:o + FLAT_TAIL].copy_(flat[-FLAT_TAIL:])   # (loss, overflow: the last two floats of the step buffer)
    if live:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
    for name, w, seg in segs:                                                # rows outside the union are zero everywhere
        fv[name].view(N, w).index_copy_(0, idx, seg)
    flat[-FLAT_TAIL:].copy_(buf[o:o + FLAT_TAIL])


ViewShardedStep._compact_all_reduce = _compact_all_reduce


class HipViewCompute:
    """compute_fn over the HIP kernels: skin weights once, LBS per pose, SH colour and rasterisation per view,
    image loss (rgb_loss / ssim_loss of src/modules/base.py:323-365), backward to the six leaf tensors.  All local
    views go through every kernel launch together.

    scene["kind"]: "hand" (every Gaussian skinned, hand_dynamic.py:86-137), "object" (static, object.py:32-41) or
    "composite" (the first scene["n_hand"] Gaussians skinned, the rest static with the identity transform,
    composite.py:50-59).  fused=True runs the fused kernels through direct C-ABI calls (no autograd graph); fused=False
    the modular operators under autograd (the reference-shaped path).

    ONE set of defaults for the step, shared by `Trainer`, `bench.py` and a hand-built object: depth cut off, gradient /
    image buffers kept (`persistent_grads`).  With kept buffers the tensors in a step's output dict (`grads`, `grad2d`,
    `vis`) and `last_image` are the SAME storage every step, like `.grad` tensors: they hold the latest step's values --
    clone what must outlive the next call, or construct with persistent_grads=False for fresh tensors per call.  Writing
    into them is allowed: every step compares the tensors' torch version counters with those it recorded when it handed
    them out, and a buffer touched in between (any in-place torch op on it or on a view of it) is filled in full again
    instead of row- / tile-selectively (tests/test_gpu_fused.py::test_kept_buffers_survive_a_caller_writing_into_them).
    Writes torch cannot see (raw pointers, `.data`) are the caller's to avoid."""

    # bytes of parked depth-cut hint tables kept per compute object (4 bytes per tile and view each)
    MAX_CUT_HINT_BYTES = 64 << 20
    # per-view target maps kept (130 KB per 1080p view)
    MAX_TARGET_MAPS = 4096

    def __init__(self, scene, targets, cam_table, loss_weight=1.0, fused=True, loss="l1", w_rgb=0.8, w_ssim=0.2,
                 sh_storage="fp32", sparse_loss=True, overlap_loss=True, depth_cut=False, max_cut_hints=1024,
                 persistent_grads=True):
        from . import fused as fused_mod, ops, rasterizer
        # persistent_grads (fused step, no grad_arena): the leaf gradients, the skin-weight gradient, the statistics and the
        # image are written into buffers this object keeps (see the class docstring).  The backward then zeroes only the
        # rows that were written by the previous step and get nothing now, instead of every row of every gradient every
        # step (mgr_views_backward, debug bit 512: 97 MB of stores per bench step), and the forward writes the background
        # only into empty tiles that held something else (mgr_views_forward, bit 1024).  `_pg_ver`: the torch version
        # counter of every kept tensor as of the end of the last step -- a mismatch means somebody wrote into it.
        self.persistent_grads, self._pg, self._pg_ws, self._pimg_ws, self._pg_ver = bool(persistent_grads), None, None, None, {}
        # depth_cut (fused step only): every forward leaves, per tile whose pixels all saturated, the depth in front of
        # which they had stopped (+ a margin); the next forward of the SAME views leaves the instances behind it out of
        # that tile's list -- the binning kernels then handle a fraction of the pairs, the image and the gradients stay
        # bit for bit those of the full lists (a cut list that runs out under an unsaturated pixel is flagged like a
        # pair-capacity overflow and the step is run again without the cut).  The hints live in the workspace; when the
        # view set changes they are parked per view set (max_cut_hints sets of 4 bytes per tile and view) and brought
        # back when it returns -- a training run revisits its (frame, camera) pairs every epoch.  OFF by default: it pays only
        # while the model stands still between two forwards of its views (fwd+bwd loops without an optimizer, evaluation
        # sweeps); under a moving model it is a wash.  MANUS_DEPTH_CUT=0 in the environment forces it off.
        self.depth_cut = bool(depth_cut) and os.environ.get("MANUS_DEPTH_CUT", "1") != "0"
        self._cut_store, self._cut_max, self._cut_gen, self._cut_bit = {}, int(max_cut_hints), 0, 0
        # A flagged forward costs a whole step, and with the optimizer in the loop no margin prevents them all: a pixel whose
        # transmittance ends just under the threshold needs many more entries after the slightest change (measured on the
        # bench scene with Adam at the reference's learning rates: a flagged forward every ~30 steps at 4x the margins).  So
        # the cut backs off: a flagged forward doubles the margins (every clean one takes 2 % off again, 1x .. 32x the
        # library's defaults), restricts the hints to interior tiles, and suspends the cut for `backoff` forwards -- 4, then
        # 8, ... up to 512; 32 clean forwards in a row halve it again.  A model that stands still between two forwards of its
        # views (fwd+bwd benchmarks, evaluation sweeps, several losses on one state) keeps the cut on; one that moves every
        # step ends up trying it every few hundred steps, at a cost below the run-to-run noise.  Hints that have seen more
        # than `cut_max_age` parameter updates are not used at all (a dataset of thousands of views revisits each once per
        # epoch: the forward then simply runs uncut and leaves fresh hints).
        self._cut_scale, self._cut_seen, self.cut_max_age, self._cut_clock, self._cut_born = 1.0, 0, 16, 0, {}
        self._cut_pause, self._cut_backoff, self._cut_clean = 0, 4, 0
        # cut_repair (round 6): tiles whose cut list runs out are repaired on the device (mgr_views_forward, debug bit 2048) --
        # no flagged forward, no re-run, no back-off; cut_margin scales the library's default margins, cut_penalty = forwards
        # a tile that ran out goes without a hint.  MANUS_CUT_REPAIR=0 in the environment: the round-5 behaviour (A/B).
        self.cut_repair = os.environ.get("MANUS_CUT_REPAIR", "1") != "0"
        self.cut_margin, self.cut_penalty = float(os.environ.get("MANUS_CUT_MARGIN", "1.0")), int(os.environ.get("MANUS_CUT_PENALTY", "16"))
        # sparse_loss: the fused step hands the forward's tile-list offsets to the image loss, which then settles the
        # spans under empty tiles from the target alone (exact: the rasterizer writes the background colour there) and
        # leaves their gradient unwritten (the backward never reads it).  False: the loss reads both images everywhere.
        self.sparse_loss, self._ts_off = bool(sparse_loss), {}
        # overlap_loss: that span list is built on a second stream while the forward blend runs (MANUS_OVERLAP_LOSS=0
        # in the environment switches it off for A/B runs)
        self.overlap_loss, self._side = bool(overlap_loss) and os.environ.get("MANUS_OVERLAP_LOSS", "1") != "0", None
        # target_map: the span list is derived from column masks "target differs from the background here", which the list
        # kernel otherwise recomputes from the full target images every step (0.2 GB at 8 views of 1080p).  A target is a
        # constant of its view: the masks are computed once per view (1 bit per pixel column and row pair, 130 KB per
        # 1080p view) and the list is built from them, in-stream, in a few microseconds -- no second stream, no forward
        # split at the blend.  Same list, same loss and gradients.  MANUS_TARGET_MAP=0 switches it off for A/B runs.
        self.target_map, self._tmaps, self._lws = os.environ.get("MANUS_TARGET_MAP", "1") != "0", {}, {}
        # the mapped span list built by the forward's last kernel (mgr_views_forward_attach_loss_list) instead of a launch of its own (A/B: 0)
        self.attach_list = os.environ.get("MANUS_LOSS_LIST_ATTACH", "1") != "0"
        # sh_storage "fp16" (BASELINE config 5): the fused kernels read an fp16 copy of _features_rest (96 B instead of
        # 180 B per Gaussian and view group); arithmetic, gradients and the optimizer's master copy stay fp32.  The copy
        # is refreshed lazily after the leaves changed (`mark_params_changed`).  The reference has no fp16 mode:
        # parity is judged with "fp32".
        if sh_storage not in ("fp32", "fp16"):
            raise ValueError("sh_storage must be 'fp32' or 'fp16'")
        self.sh_half, self._sh_copy, self._sh_dirty = sh_storage == "fp16", None, True
        self.ops, self.rz, self.fz, self.fused = ops, rasterizer, fused_mod, fused
        self.s = scene
        self._cache, self._const_stamp = {}, None
        self.targets = targets          # (V_all,3,H,W) on the GPU (a property: replacing it drops what was derived from it)
        self.cams = cam_table           # (V_all,40)
        self.loss_weight = loss_weight
        # "l1": mean|render - gt| (rgb_loss alone); "l1+ssim": w_rgb * rgb_loss + w_ssim * ssim_loss, the
        # image terms of config/HAND_GAUSSIAN.yaml:22-23 (src/modules/base.py:323-365), one fused kernel
        if loss not in ("l1", "l1+ssim"):
            raise ValueError("loss must be 'l1' or 'l1+ssim'")
        self.loss, self.w_rgb, self.w_ssim = loss, w_rgb, w_ssim
        self.kind = scene["kind"]
        self.params = {k: v.detach().clone().requires_grad_(True) for k, v in scene["params"].items()}
        N = self.params["_xyz"].shape[0]
        has_grid = scene.get("grid") is not None
        self.n_art = N if (self.kind == "hand" and has_grid) else (int(scene["n_hand"]) if (self.kind == "composite" and has_grid) else 0)
        self.is_hand = self.n_art > 0
        self.device = self.params["_xyz"].device
        self.grad_arena = None   # set by ViewShardedStep: preallocated gradient outputs (fused path only)
        # sync_check True: every fused forward reads the pair count back and retries on overflow (like the drop-in
        # operator).  A Trainer sets it False on ITS compute object: no host sync, the forward leaves an overflow fence
        # that Trainer._run_step polls.  (The device-wide policy of rasterizer.set_sync_policy is left alone.)
        self.sync_check = True
        self._w_cache = None     # forward-only skin weights of the current model state (forward_views_fused under no_grad)
        self.grid = ops.SkinGrid(scene["grid"], scene["grid"].device) if self.is_hand else None

    @property
    def targets(self):
        return self._targets

    @targets.setter
    def targets(self, t):
        self._targets = t
        self._drop_view_constants()

    def _drop_view_constants(self):
        """Forget everything derived from the per-view constants (targets, background, cameras, poses)."""
        self._cache = {}
        if hasattr(self, "_tmaps"):
            self._tmaps.clear()
        self._const_stamp = None

    def _check_view_constants(self):
        """The target maps and the per-view-set gathers are functions of `targets`, `s["bg"]`, `cams` and the transforms:
        when one of those tensors was replaced or written in place (torch version counter) they are rebuilt."""
        s = self.s
        tfm = s.get("transforms") if self.is_hand else None
        stamp = tuple((id(t), t._version) if t is not None else None for t in (self._targets, s.get("bg"), self.cams, tfm))
        if stamp != self._const_stamp:
            if self._const_stamp is not None:
                self._drop_view_constants()
            self._const_stamp = stamp

    def set_params(self, params, n_art=None):
        """Re-point at new leaf tensors (after densification / pruning changed N)."""
        self.params = {k: v.detach().requires_grad_(True) for k, v in params.items()}
        if n_art is not None:
            self.n_art = int(n_art)
        elif self.kind == "hand" and self.is_hand:
            self.n_art = self.params["_xyz"].shape[0]
        self.grad_arena = None
        self._sh_dirty = True
        self._cut_gen += 1          # rows were added / removed: the hints of the old model are dropped
        self._cut_store.clear()
        self._cut_born.clear()

    def mark_params_changed(self):
        """The leaves were updated in place (optimizer step): derived storage copies are stale."""
        self._sh_dirty = True
        self._cut_clock += 1

    def _sh_storage(self, f_rest):
        """(pointer source tensor, sh_half flag) for the fused kernels."""
        if not self.sh_half:
            return f_rest, 0
        from ._lib import check, lib, ptr, stream
        N = f_rest.shape[0]
        if self._sh_copy is None or self._sh_copy.shape[0] != N:
            self._sh_copy, self._sh_dirty = torch.empty((N, 48), dtype=torch.float16, device=f_rest.device), True
        if self._sh_dirty:
            check(lib().mgr_sh_to_half(N, ptr(f_rest), ptr(self._sh_copy), stream()), "mgr_sh_to_half")
            self._sh_dirty = False
        return self._sh_copy, 1

    def _select(self, view_ids):
        """Per-view constants for a set of views (cached: no per-step gather copies)."""
        self._check_view_constants()
        key = tuple(view_ids)
        c = self._cache.get(key)
        if c is None:
            idx = list(view_ids)
            c = dict(cams=self.cams[idx].contiguous(), targets=self.targets[idx].contiguous(),
                     T=self.s["transforms"][idx].contiguous() if self.is_hand else None)
            self._cache = {key: c}
        return c

    # -- modular, reference-shaped path (autograd) -------------------------------------------------
    def _posed(self, T):
        """posed means / covariances / transforms of all Gaussians for the poses T (P,B,4,4): LBS for the first
        n_art rows, identity for the rest."""
        p, ops = self.params, self.ops
        N, na = p["_xyz"].shape[0], self.n_art
        if na == 0:
            _, pcov1, _ = ops.lbs_cov(p["_xyz"], p["_scaling"], p["_rotation"], None, None)
            return p["_xyz"], pcov1[0], None
        w = ops.skin_weights(p["_xyz"][:na], self.grid, self.s["grid_center"], self.s["grid_scale"])
        pxyz, pcov, tf = ops.lbs_cov(p["_xyz"][:na], p["_scaling"][:na], p["_rotation"][:na], w, T)
        if na < N:   # composite.py:50-59: concat, identity tf for the object
            P = T.shape[0]
            _, ocov, otf = ops.lbs_cov(p["_xyz"][na:], p["_scaling"][na:], p["_rotation"][na:], None, None)
            pxyz = torch.cat([pxyz, p["_xyz"][na:][None].expand(P, -1, -1)], dim=1)
            pcov = torch.cat([pcov, ocov.expand(P, -1, -1)], dim=1)
            tf = torch.cat([tf, otf.expand(P, -1, -1)], dim=1)
        return pxyz, pcov, tf

    def forward_views(self, view_ids):
        s, p, ops = self.s, self.params, self.ops
        sel = self._select(view_ids)
        cams = sel["cams"]
        V = len(view_ids)
        feats = torch.cat([p["_features_dc"], p["_features_rest"]], dim=1)
        opac = torch.sigmoid(p["_opacity"])
        pxyz, pcov, tf = self._posed(sel["T"])   # one pose per view (the reference trains one (frame, view) per step)
        col = ops.sh_colors(feats, p["_xyz"], tf, cams)
        N = p["_xyz"].shape[0]
        means2D = torch.zeros((V, N, 3), dtype=torch.float32, device=cams.device, requires_grad=True)
        img, radii = self.rz.rasterize_views(cams, pxyz, means2D, col, opac, pcov, s["bg"], s["width"], s["height"])
        return img, radii, means2D

    def forward_views_fused(self, view_ids, stats=None, grad2d_scale=1.0):
        """The fused kernels behind the autograd node `fused.render_views`."""
        s, p, ops = self.s, self.params, self.ops
        sel = self._select(view_ids)
        na = self.n_art
        w = None
        if na and not torch.is_grad_enabled():
            # forward only (evaluation sweeps, target rendering): the skin weights depend on `_xyz` alone, so they are kept per
            # model state -- (generation, parameter-update clock, the leaf's storage and version) -- instead of gathered from
            # the grid again for every batch of views (0.04 ms for 300 k Gaussians: 13 % of a one-view forward).  Training
            # steps always recompute them (their gradient flows back into `_xyz`).
            key = (self._cut_gen, self._cut_clock, p["_xyz"].data_ptr(), p["_xyz"]._version, na, id(self.grid))
            if self._w_cache is None or self._w_cache[0] != key:
                self._w_cache = (key, ops.skin_weights(p["_xyz"][:na], self.grid, s["grid_center"], s["grid_scale"]))
            w = self._w_cache[1]
        elif na:
            w = ops.skin_weights(p["_xyz"][:na], self.grid, s["grid_center"], s["grid_scale"])
        return self.fz.render_views(p["_xyz"], p["_scaling"], p["_rotation"], p["_opacity"], p["_features_dc"],
                                    p["_features_rest"], w, sel["T"], sel["cams"], s["bg"], s["width"], s["height"],
                                    stats=stats, grad2d_scale=grad2d_scale, grad_arena=self.grad_arena)

    def _image_loss(self, img, tgt, scale, tiles=None):
        """(loss value, dL/dimg) of scale * sum over the views of the per-view image loss.  tiles = (bg, device address
        of the tile-list offsets of the forward that rendered img): spans under empty tiles are not read (ops.image_loss_grad)."""
        per_view = img[0].numel()
        k = self.loss_weight * scale / per_view
        if self.loss == "l1":
            loss_sum, g = self.ops.l1_loss_grad(img, tgt, scale=k)
            return loss_sum[0] * k, g
        const = self.w_ssim * self.loss_weight * scale * img.shape[0]   # the "1 -" of 1 - ssim, once per view
        bg, ts = tiles if (tiles is not None and self.sparse_loss) else (None, None)
        sums, g = self.ops.image_loss_grad(img, tgt, self.w_rgb, self.w_ssim, k, const, bg=bg, tile_start_ptr=ts)
        return sums[2], g

    def _layout(self, ws, V, N, W, H):
        key = (V, N, W, H, ws.cap)
        off = self._ts_off.get(key)
        if off is None:
            import ctypes
            from ._lib import lib
            arr = (ctypes.c_size_t * 40)()
            lib().mgr_raster_layout(V, N, W, H, ws.cap, arr, 40)
            off = self._ts_off[key] = [int(x) for x in arr]
        return off

    def _target_map(self, view_ids, sel, bg):
        """(V, rows / 2, ceil(W / 32)) int32: the target-vs-background column masks of the views (computed once per view)."""
        m = sel.get("tmap")
        if m is None:
            from ._lib import check, lib, ptr, stream
            H, W = int(self.s["height"]), int(self.s["width"])
            per = int(lib().mgr_image_loss_target_map_words(1, H, W))
            rows = []
            for k, v in enumerate(view_ids):
                t = self._tmaps.get(v)
                if t is None:
                    t = torch.empty(per, dtype=torch.int32, device=self.device)
                    check(lib().mgr_image_loss_target_map(1, H, W, ptr(sel["targets"][k]), ptr(bg), ptr(t), stream()),
                          "mgr_image_loss_target_map")
                    while len(self._tmaps) >= self.MAX_TARGET_MAPS:
                        self._tmaps.pop(next(iter(self._tmaps)))
                    self._tmaps[v] = t
                rows.append(t)
            m = sel["tmap"] = torch.stack(rows).contiguous()
        return m

    def _tile_start_ptr(self, ws, V, N, W, H):
        return ws.buf.data_ptr() + self._layout(ws, V, N, W, H)[7]

    def _cut_flag(self, ws, view_ids, V, N, W, H):
        """debug bit 8 of mgr_views_forward for this forward on `ws`: set when the workspace holds the depth-cut hints of
        exactly these views (left by the previous forward, or parked earlier and brought back here) and the last forward
        with them was not flagged."""
        if not self.depth_cut or not self.rz.context(self.device).fenced(self.sync_check):
            return 0      # (with a host sync per forward the split forward would need a second one after the blend: not worth it)
        key = (id(self), self._cut_gen, tuple(view_ids))
        prev = ws.prev_hint_key
        ws.hint_key = key
        from ._lib import lib
        ctx = self.rz.context(self.device)
        if ctx.cut_retries != self._cut_seen:          # a forward of ours was flagged since the last launch
            self._cut_seen, self._cut_scale = ctx.cut_retries, min(32.0, self._cut_scale * 2.0)
            self._cut_pause, self._cut_backoff, self._cut_clean = self._cut_backoff, min(512, self._cut_backoff * 2), 0
        else:
            self._cut_scale = max(1.0, self._cut_scale * 0.98)
        k = self._cut_scale
        if self.cut_repair:
            # tiles that run out are completed on the device (bit 2048): a flagged forward is then a capacity matter, rare, and
            # the margins need neither widening nor the interior-only rule -- the library's per-tile countdown keeps the
            # repeat offenders out
            lib().mgr_raster_set_cut_margin(0.125 * self.cut_margin, int(64 * self.cut_margin), 0.0625 * self.cut_margin,
                                            2.0e-4 * self.cut_margin, 0)
            lib().mgr_raster_set_cut_penalty(int(self.cut_penalty))
        else:
            lib().mgr_raster_set_cut_margin(min(4.0, 0.125 * k), int(64 * k), min(4.0, 0.0625 * k), 2.0e-4 * k, 1 if k > 1.0 else 0)
        born, self._cut_born[key] = self._cut_born.get(key), self._cut_clock
        if len(self._cut_born) > 4 * self._cut_max:     # view sets not seen for cut_max_age updates have no usable hints
            self._cut_born = {k_: b_ for k_, b_ in self._cut_born.items() if self._cut_clock - b_ <= self.cut_max_age}
        too_old = born is None or self._cut_clock - born > self.cut_max_age
        if prev != key:
            T = ((W + 15) // 16) * ((H + 15) // 16)
            lay = self._layout(ws, V, N, W, H)
            # the hints (tile_zcut) and the depth windows of the repair that belong to them (tile_zwin): 2 x 4 bytes per tile and view
            regions = [ws.buf[o_: o_ + 4 * V * T] for o_ in (lay[26], lay[32])]
            nbytes = sum(r.numel() for r in regions)
            if prev is not None and prev[:2] == key[:2]:      # park the hints of the views rendered last
                max_sets = max(1, min(self._cut_max, self.MAX_CUT_HINT_BYTES // max(1, nbytes)))
                while len(self._cut_store) >= max_sets:
                    self._cut_store.pop(next(iter(self._cut_store)))
                self._cut_store[prev] = torch.cat(regions)
            saved = self._cut_store.pop(key, None)
            if saved is not None and saved.numel() == nbytes:
                o_ = 0
                for r in regions:
                    r.copy_(saved[o_: o_ + r.numel()])
                    o_ += r.numel()
            else:
                for r in regions:
                    r.zero_()                                               # no hints for these views yet
        if ws.cut_block:                  # the previous forward was flagged: this one rebuilds the hints from full lists
            ws.cut_block = False
            return 0
        if too_old:
            return 0
        if self.cut_repair:
            if self._cut_pause > 0:       # (a capacity of the repair was exceeded: a few forwards on full lists, no escalation)
                self._cut_pause = min(self._cut_pause, 4) - 1
                return 0
            return 8 | 2048
        if self._cut_pause > 0:           # backing off after a flagged forward
            self._cut_pause -= 1
            return 0
        self._cut_clean += 1
        if self._cut_clean >= 32:
            self._cut_clean, self._cut_backoff = 0, max(4, self._cut_backoff // 2)
        return 8

    # -- fused path, direct C-ABI calls ------------------------------------------------------------
    def _step_direct(self, view_ids, scale, g_img=None):
        """skin weights -> mgr_views_forward -> image loss -> mgr_views_backward -> skin-weight backward, every
        gradient written straight into the arena (the all-reduce buffer) when one is set.  g_img: optional dL/dimage
        (V,3,H,W) used instead of the image loss (parity tests)."""
        from ._lib import check, lib, ptr, stream
        s, p = self.s, {k: v.detach() for k, v in self.params.items()}
        sel = self._select(view_ids)
        cams, T = sel["cams"], sel["T"]
        dev = self.device
        N, na, V = p["_xyz"].shape[0], self.n_art, len(view_ids)
        W, H = int(s["width"]), int(s["height"])
        arena = self.grad_arena or {}
        own = None
        if self.persistent_grads and not arena and self.fused:
            key = (N, na, str(dev))
            if self._pg is None or self._pg[0] != key:
                self._pg, self._pg_ws, self._pimg_ws, self._pg_ver = (key, {}), None, None, {}
            own = self._pg[1]
            # a kept buffer somebody wrote into since it was handed out (torch bumps a tensor's version counter on every
            # in-place op, also through views) no longer holds what the row / tile bookkeeping says: fill everything again
            if any(own[n]._version != ver for n, ver in self._pg_ver.items() if n in own):
                self._pg_ws = self._pimg_ws = None

        def e(shape, name):
            if own is not None:
                t = own.get(name)
                if t is None:
                    t = own[name] = torch.zeros(shape, dtype=torch.float32, device=dev)
                return t
            t = arena.get(name)
            n = 1
            for d in shape:
                n *= d
            if t is not None and t.numel() == n and t.is_contiguous() and t.dtype == torch.float32 and t.device == dev:
                return t.view(shape)
            return torch.empty(shape, dtype=torch.float32, device=dev)

        sg, B, w = self.grid, 0, None
        if na:
            B = sg.B
            w = torch.empty((na, B), dtype=torch.float32, device=dev)
            check(lib().mgr_skin_weights_fwd(na, ptr(p["_xyz"]), ptr(sg.data), sg.D, sg.H, sg.W, sg.B, sg.stride,
                                             ptr(s["grid_center"]), ptr(s["grid_scale"]), ptr(w), stream()),
                  "mgr_skin_weights_fwd")
        if own is not None:   # persistent_grads: the image too is a buffer this object keeps -- a tile that held the background
            out = own.get(("image", V, H, W))          # after the previous forward on the same workspace and is empty again is not
            if out is None:                            # written again (mgr_views_forward, debug bit 1024)
                out = own[("image", V, H, W)] = torch.empty((V, 3, H, W), dtype=torch.float32, device=dev)
        else:
            out = torch.empty((V, 3, H, W), dtype=torch.float32, device=dev)
        radii = torch.empty((V, N), dtype=torch.int32, device=dev)
        op = p["_opacity"].reshape(-1)
        bg = s["bg"]
        f_rest, sh_half = self._sh_storage(p["_features_rest"])

        # The span list of the image loss needs the forward's tile offsets but not its image: with overlap_loss it is
        # built on a second stream while the forward blend runs (forward split at the blend, debug bits 1 / 2).
        mapped = g_img is None and self.loss == "l1+ssim" and self.sparse_loss and self.target_map
        overlap = g_img is None and self.loss == "l1+ssim" and self.sparse_loss and self.overlap_loss and not mapped

        def fwd(ws, phase):
            check(lib().mgr_views_forward(V, N, B, na, sh_half, W, H, ptr(cams), ptr(bg), ptr(p["_xyz"]), ptr(p["_scaling"]),
                                          ptr(p["_rotation"]), ptr(op), ptr(p["_features_dc"]), ptr(f_rest),
                                          ptr(w), ptr(T), ptr(out), ptr(radii), ptr(ws.buf), ws.nbytes, ws.cap,
                                          phase | self._cut_bit | ws.skip_bits() | (1024 if (own is not None and self._pimg_ws is ws) else 0),
                                          stream()), "mgr_views_forward")
            if own is not None:
                self._pimg_ws = ws

        # mapped: the span list of the loss is built by the forward itself (extra workgroups of its last kernel, attached per launch)
        lws = tmap = None
        lnbytes = 0
        if mapped:
            lnbytes = int(lib().mgr_image_loss_workspace_bytes(V, H, W))
            lws = self._lws.get((V, H, W))      # kept across steps, zero-filled once: list / finish pairs leave it clean
            if lws is None:
                lws = self._lws[(V, H, W)] = torch.zeros(lnbytes, dtype=torch.uint8, device=dev)
            tmap = self._target_map(view_ids, sel, bg)
        launches = [0]

        def launch(ws):
            self._cut_bit = self._cut_flag(ws, view_ids, V, N, W, H)
            if mapped and self.attach_list:
                if launches[0]:          # a forward of this step ran before (capacity / tier retry): its list was never finished
                    lws.zero_()
                launches[0] += 1
                check(lib().mgr_views_forward_attach_loss_list(ptr(ws.buf), V, H, W, ptr(tmap), ptr(lws), lnbytes),
                      "mgr_views_forward_attach_loss_list")
                try:
                    fwd(ws, 0)
                except Exception:        # (a forward that failed before its last kernel leaves the attachment pending: withdraw it)
                    lib().mgr_views_forward_attach_loss_list(ptr(ws.buf), V, H, W, None, None, 0)
                    raise
                return
            fwd(ws, 2 if overlap else 0)

        ctx = self.rz.context(dev)
        ws, _ = ctx.forward(V, N, W, H, launch, sync_check=self.sync_check, defer_fence=True)
        try:
            if not overlap and ctx.fenced(self.sync_check):
                ctx.fence(ws)
            if mapped:
                tgt = sel["targets"]
                nbytes = lnbytes
                g_img = torch.empty_like(out)
                sums = torch.empty(3, dtype=torch.float32, device=dev)
                if not self.attach_list:
                    import ctypes
                    check(lib().mgr_image_loss_tiles_list_mapped(V, H, W, ptr(tmap), ptr(bg), ctypes.c_void_p(self._tile_start_ptr(ws, V, N, W, H)),
                                                                 ptr(lws), nbytes, 1, stream()), "mgr_image_loss_tiles_list_mapped")
                per_view = out[0].numel()
                k = self.loss_weight * scale / per_view
                const = self.w_ssim * self.loss_weight * scale * V
                check(lib().mgr_image_loss_tiles_finish(V, H, W, ptr(out), ptr(tgt), self.w_rgb, self.w_ssim, k, const, ptr(g_img),
                                                        ptr(sums), ptr(lws), nbytes, stream()), "mgr_image_loss_tiles_finish")
                loss = sums[2]
            elif overlap:
                import ctypes
                tgt = sel["targets"]
                nbytes = int(lib().mgr_image_loss_workspace_bytes(V, H, W))
                lws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                g_img = torch.empty_like(out)
                sums = torch.empty(3, dtype=torch.float32, device=dev)
                cur = torch.cuda.current_stream(dev)
                if self._side is None:
                    self._side = torch.cuda.Stream(device=dev)
                self._side.wait_stream(cur)
                with torch.cuda.stream(self._side):
                    check(lib().mgr_image_loss_tiles_list(V, H, W, ptr(tgt), ptr(bg), ctypes.c_void_p(self._tile_start_ptr(ws, V, N, W, H)),
                                                          ptr(lws), nbytes, stream()), "mgr_image_loss_tiles_list")
                fwd(ws, 4)                                  # the blend, next to the list
                if ctx.fenced(self.sync_check):
                    ctx.fence(ws)                           # (after the blend: it is the blend that raises the depth-cut flag)
                cur.wait_stream(self._side)
                lws.record_stream(self._side)
                per_view = out[0].numel()
                k = self.loss_weight * scale / per_view
                const = self.w_ssim * self.loss_weight * scale * V
                check(lib().mgr_image_loss_tiles_finish(V, H, W, ptr(out), ptr(tgt), self.w_rgb, self.w_ssim, k, const, ptr(g_img),
                                                        ptr(sums), ptr(lws), nbytes, stream()), "mgr_image_loss_tiles_finish")
                loss = sums[2]
            elif g_img is None:
                loss, g_img = self._image_loss(out, sel["targets"], scale, tiles=(bg, self._tile_start_ptr(ws, V, N, W, H)))
            else:
                loss, g_img = (out * g_img).sum(), g_img.contiguous()
            d_xyz, d_ls, d_rot = e((N, 3), "_xyz"), e((N, 3), "_scaling"), e((N, 4), "_rotation")
            d_op, d_fdc, d_frest = e((N, 1), "_opacity"), e((N, 1, 3), "_features_dc"), e((N, 15, 3), "_features_rest")
            st_g, st_v = e((N,), "grad2d"), e((N,), "vis")
            st_r = torch.empty(N, dtype=torch.int32, device=dev)
            d_w = (e((na, B), "_skin_w") if own is not None else torch.empty((na, B), dtype=torch.float32, device=dev)) if na else None
            # the buffers are those of the previous backward on this very workspace, untouched since: the library may skip the
            # zero fill of the rows it knows to be zero (it checks that its row state is that call's; bit 512)
            kept = 512 if (own is not None and self._pg_ws is ws and V <= 8) else 0
            self._pg_ws = None
            check(lib().mgr_views_backward(V, N, B, na, sh_half, W, H, ptr(cams), ptr(bg), ptr(p["_xyz"]), ptr(p["_scaling"]),
                                           ptr(p["_rotation"]), ptr(op), ptr(p["_features_dc"]), ptr(f_rest),
                                           ptr(w), ptr(T), ptr(radii), ptr(out), ptr(g_img), 1.0 / scale, ptr(d_xyz),
                                           ptr(d_ls), ptr(d_rot), ptr(d_op), ptr(d_fdc), ptr(d_frest), ptr(d_w), ptr(st_g),
                                           ptr(st_v), ptr(st_r), ptr(ws.buf), ws.nbytes, ws.cap, kept, stream()),
                  "mgr_views_backward")
            if own is not None:
                self._pg_ws = ws
                self._pg_ver = {n: t._version for n, t in own.items()}
            active = None
            if V <= 8 and N > 0:
                # the backward's list of the Gaussians that received a gradient (device pointers: list, length)
                import ctypes
                lst, cnt = ctypes.c_void_p(), ctypes.c_void_p()
                check(lib().mgr_views_active_list(ptr(ws.buf), V, N, W, H, ws.cap, ctypes.byref(lst), ctypes.byref(cnt)),
                      "mgr_views_active_list")
                active = (lst, cnt)
            if na and V <= 8:
                # d xyz += d w . d(trilinear weights)/d xyz (the leaf is used twice: gaussian_utils.py:167-196), for the
                # Gaussians that received a gradient only (the others' d_w rows are zero)
                check(lib().mgr_skin_weights_bwd_indexed(na, ptr(p["_xyz"]), ptr(sg.data), sg.D, sg.H, sg.W, sg.B, sg.stride,
                                                         ptr(s["grid_center"]), ptr(s["grid_scale"]), ptr(d_w), ptr(d_xyz),
                                                         lst, cnt, N, stream()), "mgr_skin_weights_bwd_indexed")
            elif na:
                check(lib().mgr_skin_weights_bwd(na, ptr(p["_xyz"]), ptr(sg.data), sg.D, sg.H, sg.W, sg.B, sg.stride,
                                                 ptr(s["grid_center"]), ptr(s["grid_scale"]), ptr(d_w), ptr(d_xyz), 1,
                                                 stream()), "mgr_skin_weights_bwd")
            overflow = ws.buf[4:8].view(torch.int32)
        except BaseException:
            # a call that failed between the loss's list and finish passes leaves the kept loss workspace with a non-zero
            # span count (the finish pass is what resets it) and the kept buffers in an unknown state: start over
            self._lws.pop((V, H, W), None)
            self._pg_ws = self._pimg_ws = None
            raise
        finally:
            ws.busy = False
        self.last_image, self.last_radii = out, radii
        self.last_active = active      # (device pointers into the workspace of this step: valid until the next forward on it)
        return dict(grads={"_xyz": d_xyz, "_scaling": d_ls, "_rotation": d_rot, "_opacity": d_op, "_features_dc": d_fdc,
                           "_features_rest": d_frest}, grad2d=st_g, vis=st_v, radii=st_r, loss=loss, overflow=overflow)

    def __call__(self, view_ids, scale=1.0):
        if self.fused:
            return self._step_direct(view_ids, scale)
        self.last_active = None
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

    def pairs_per_view(self, view_ids=None, group=8):
        """Surviving (tile, Gaussian) pairs of every view (forward only, in groups of `group` views): the weights of the
        balanced view assignment.  Deterministic, so every rank computes the same list."""
        import ctypes
        from ._lib import lib
        ids = list(range(self.cams.shape[0])) if view_ids is None else list(view_ids)
        W, H = int(self.s["width"]), int(self.s["height"])
        T = ((W + 15) // 16) * ((H + 15) // 16)
        N = self.params["_xyz"].shape[0]
        out = []
        from ._lib import ManusHipError
        ctx = self.rz.context(self.device)
        with torch.no_grad():
            for k in range(0, len(ids), group):
                part = ids[k:k + group]
                for attempt in range(4):
                    self.forward_views_fused(part)
                    try:    # (the autograd node's forward follows the device's sync policy: whatever that is, the tile offsets
                        ctx.check_overflow()    # read below come from a forward that did not overflow its pair capacity)
                        break
                    except ManusHipError:
                        if attempt == 3:
                            raise
                ws = ctx.last_ws
                arr = (ctypes.c_size_t * 32)()
                lib().mgr_raster_layout(len(part), N, W, H, ws.cap, arr, 32)
                ts = ws.buf[int(arr[7]): int(arr[7]) + 4 * (len(part) * T + 1)].view(torch.int32)[::T].cpu().tolist()
                out += [int(b - a) for a, b in zip(ts[:-1], ts[1:])]
        return out

    # -- inputs of the pruning tests (on_after_backward) ---------------------------------------------
    def prune_views(self, view_ids):
        """One dict per view for `density.DensityController`: camera (K, extr), mask, posed means of the view's
        pose and its keypoints.  Views without a mask / keypoints in the scene are skipped by the controller."""
        s = self.s
        with torch.no_grad():
            if self.is_hand:
                pxyz, _, _ = self._posed(self._select(view_ids)["T"])
            else:
                pxyz = None
        out = []
        masks, keyp = s.get("masks"), s.get("keypoints")
        for k, v in enumerate(view_ids):
            out.append(dict(camera=s["cameras"][v], mask=masks[v] if masks is not None else None,
                            posed_xyz=(pxyz[k, : self.n_art] if pxyz is not None else self.params["_xyz"].detach()),
                            keypoints=keyp[v] if keyp is not None else None))
        return out


class Trainer:
    """One optimisation step in the reference's order (src/modules/hand_dynamic.py:230-282 + Lightning's hooks):

        training_step      render the views, loss, backward (+ the all-reduce)                at global_step g
        on_after_backward  pruning tests + density_update(g)   hand_dynamic.py:193-224 / object.py:66-81
        on_before_optimizer_step   xyz learning rate = schedule(g)   hand_dynamic.py:226-228
        optimizer.step()   Adam; leaves that density_update has just replaced carry no gradient and are skipped
        global_step += 1

    After a densification / pruning the parameter tensors are new (different N): the compute object is re-pointed
    at them and the rasterizer workspaces of the old size are released.  The rasterizer runs without host
    synchronisation; every step is fenced (`rasterizer.poll`: waits for the forward only) and re-run with a larger
    pair capacity if it overflowed, so no update is ever made from a truncated image.
    Multi-GPU: every rank must draw the same split noise, so rank 0's is broadcast; the pruning masks are OR-ed
    over the ranks; max_radii2D is MAX-reduced before it is consumed."""

    def __init__(self, compute, n_views, extent, opts=None, spatial_lr_scale=1.0, rank=0, world_size=1, group=None,
                 bg_white=True, kind=None, compact_allreduce=False, sharded_adam=False, view_weights=None, depth_cut=False,
                 sort_rows=False, persistent_grads=True):
        # sharded_adam (world_size > 1): reduce-scatter of the gradients -> every rank takes the Adam step on the 1/world
        # of the parameter elements it owns -> all-gather of the parameters.  The same bytes on the wire as the
        # all-reduce (which is a reduce-scatter followed by an all-gather), 1/world of the optimizer work per rank.
        # sort_rows: after a densification / pruning has rebuilt the tensors, put the rows in Z-order of their positions
        # (GaussianOptimizer.sort_rows: the same model up to the permutation; every rank computes the same one)
        self.sort_rows = bool(sort_rows)
        # persistent_grads (one rank): the compute object keeps the gradient buffers (HipViewCompute's own default): the
        # tensors in a step's `out["grads"]` are overwritten by the next step -- the optimizer has consumed them by then
        # (see train_step).  False: fresh tensors every step.
        # (only ever narrowed: a compute object built with persistent_grads=False keeps handing out fresh tensors)
        if hasattr(compute, "persistent_grads"):
            compute.persistent_grads = bool(compute.persistent_grads) and bool(persistent_grads) and world_size == 1
        self.compact_allreduce = compact_allreduce and not sharded_adam
        self.sharded_adam = bool(sharded_adam) and world_size > 1
        from . import rasterizer
        from .density import DensityController
        from .optim import GaussianOptimizer
        global ALL_GROUPS
        from .optim import ALL_GROUPS
        self.compute, self.n_views, self.extent, self.bg_white = compute, n_views, float(extent), bg_white
        self.view_weights = view_weights   # per-view costs for the balanced assignment (shard_views); None = round-robin
        self.rank, self.world, self.group = rank, world_size, group
        self.opt = GaussianOptimizer(compute.params, opts=opts, spatial_lr_scale=spatial_lr_scale, adopt=True)
        # A composite scene (hand + object in one launch) has no density control in the reference: composite.py has no
        # on_after_backward / density_update and its training_step is `pass` (composite.py:80-81); the two models are
        # densified by their own modules.  The Trainer therefore only renders, reduces and takes the Adam step there --
        # decided here, before any state exists, and independent of `opts` (densify_from_step etc. are ignored).
        self.density_enabled = getattr(compute, "kind", "hand") != "composite"
        kind = kind or ("object" if getattr(compute, "kind", "hand") == "object" else "hand")
        self.density = DensityController(self.opt, extent, kind=kind, bg_white=bg_white)
        self.global_step = 0
        self.retries = 0
        self._rz = rasterizer
        if getattr(compute, "fused", False) and hasattr(compute, "sync_check"):
            compute.sync_check = False      # this trainer's forwards are fenced and polled in _run_step (no global policy flip)
        # depth_cut: a Trainer moves the model every step, and under a moving model the depth cut of the fused forward is a
        # wash at best (flagged forwards are run twice; measured 567 against 578 iters/s with its back-off): off
        # -- HipViewCompute's own default -- unless asked for here.
        if hasattr(compute, "depth_cut"):
            compute.depth_cut = bool(depth_cut) and os.environ.get("MANUS_DEPTH_CUT", "1") != "0"
        self._rebuild_step()

    def _rebuild_step(self):
        if hasattr(self.compute, "grad_arena"):
            self.compute.grad_arena = None   # views of the previous step object's buffer (possibly another N)
        p = self.compute.params
        shapes = {k: v.shape for k, v in p.items()}
        self.stepper = ViewShardedStep(p["_xyz"].shape[0], shapes, self.compute, self.n_views, rank=self.rank,
                                       world_size=self.world, group=self.group, compact=self.compact_allreduce,
                                       scatter=self.sharded_adam, view_weights=self.view_weights)
        if self.sharded_adam:   # leaves and moments as views of flat buffers laid out like the gradient buffer
            self.opt.flatten(self.stepper.padded_g)
            self.compute.set_params(self.opt.parameters())
            self.opt.p = {k: v.detach() for k, v in self.compute.params.items()}

    def _split_noise(self, n_rows, device):
        noise = torch.randn((2 * n_rows, 3), dtype=torch.float32, device=device)
        if self.world > 1:
            dist.broadcast(noise, src=0, group=self.group)
        return noise

    def _run_step(self):
        """Forward + backward (+ all-reduce) with the overflow fence; re-runs the step after an overflow."""
        for _ in range(4):
            out = self.stepper.step()
            local_bad = False
            if torch.cuda.is_available() and getattr(self.compute, "fused", False):
                try:
                    self._rz.poll(getattr(self.compute, "device", None))
                except RuntimeError:
                    local_bad = True
            bad = local_bad
            if self.world > 1 and out.get("overflow") is not None:
                bad = float(out["overflow"]) > 0.0     # summed over the ranks by the all-reduce: the same on all of them
            if not bad:
                return out
            self.retries += 1
        raise RuntimeError("rasterizer pair capacity still exceeded after 4 attempts")

    def gather_moments(self):
        """Sharded optimizer step: each rank keeps only the Adam moments of the elements it owns up to date.  Before
        the rows move (prune / densify: ownership is by element range, so it moves with them), and before a checkpoint,
        the ranks exchange their slices."""
        if self.sharded_adam:
            self.stepper.all_gather_params(self.opt.mflat)
            self.stepper.all_gather_params(self.opt.vflat)

    def train_step(self, views=None):
        """One optimisation step; returns the step's output dict (loss, statistics) plus "changed".
        views: optional list of per-view dicts for the pruning tests (default: `compute.prune_views`).

        ALIASING: with kept buffers (the default at one rank) `out["grads"]`, `out["grad2d"]`, `out["vis"]` and
        `compute.last_image` are the same storage every step, like `.grad` tensors -- they hold THIS step's values until the
        next call; clone what is logged or compared across steps (or construct the Trainer with persistent_grads=False).
        Writing into them is safe (HipViewCompute notices by the tensors' version counters and refills them in full)."""
        o, gs = self.opt.opts, self.global_step
        out = self._run_step()
        if not self.density_enabled:     # composite: render + reduce + Adam only (see __init__)
            self.opt.update_learning_rate(gs)
            if self.sharded_adam:
                lo, hi = self.stepper.owned
                self.opt.step_range(self.stepper._store, lo, hi)
                self.stepper.all_gather_params(self.opt.pflat)
            else:
                self.opt.step(out["grads"])
            if hasattr(self.compute, "mark_params_changed"):
                self.compute.mark_params_changed()
            self.global_step += 1
            out["changed"] = False
            return out
        # ---- on_after_backward ----
        needs_tests = self.density.kind == "hand" and (gs < o["remove_seg_end"] or gs % 100 == 0) or \
            self.density.kind == "object" and gs < o["remove_seg_end"]
        if views is None and needs_tests and hasattr(self.compute, "prune_views"):
            views = self.compute.prune_views(self.stepper.local_views)
        mask = self.density.prune_mask(gs, views) if needs_tests else None
        if self.world > 1 and needs_tests:   # every rank tested its own views: OR them
            m8 = (mask if mask is not None else torch.zeros(self.opt.N, dtype=torch.bool, device=self.opt.device)).to(torch.uint8)
            dist.all_reduce(m8, op=dist.ReduceOp.MAX, group=self.group)
            mask = m8.bool() if bool(m8.any()) else None
        will_densify = mask is None and gs < o["densify_until_step"] and gs > o["densify_from_step"] and \
            gs % o["densification_interval"] == 0
        stats = dict(grad2d=out["grad2d"], vis=out["vis"], radii=out["radii"])
        noise = None
        if will_densify and self.world > 1:
            # the selection count is only known inside the plan: draw the worst case on rank 0 and broadcast
            noise = self._split_noise(self.opt.N, self.opt.device)
        n_before = self.opt.N
        if self.sharded_adam and (mask is not None or will_densify):
            self.gather_moments()     # rows are about to move: every rank needs the moments of every row
        if mask is not None:
            changed = self.opt.density_update(stats, self.extent, gs, self.bg_white, mask_to_prune=mask)
        else:
            if will_densify and self.world > 1:
                # max_radii2D is a running maximum over this rank's views; combine the ranks before it is consumed
                self.opt.add_densification_stats(out["grad2d"], out["vis"], out["radii"])
                self.stepper.reduce_max_radii(self.opt.max_radii2D)
                stats = dict(grad2d=torch.zeros_like(out["grad2d"]), vis=torch.zeros_like(out["vis"]),
                             radii=torch.zeros_like(out["radii"]))
            changed = self.opt.density_update(stats, self.extent, gs, self.bg_white, noise=noise,
                                              noise_is_pool=noise is not None)
        if changed:
            self.density.on_train_epoch_start()
        resized = changed and (self.opt.N != n_before or self.opt.replaced == ALL_GROUPS)   # new leaf tensors
        if resized and self.sort_rows and self.opt.replaced == ALL_GROUPS:
            na = getattr(self.compute, "n_art", 0)
            out["row_perm"] = self.opt.sort_rows(n_art=na if 0 < na < n_before and getattr(self.compute, "kind", "") == "composite" else None)
        # ---- on_before_optimizer_step + optimizer.step() ----
        self.opt.update_learning_rate(gs)
        if self.sharded_adam:
            if self.opt.replaced == ALL_GROUPS:
                self.opt.replaced = frozenset()          # every leaf is new: no gradient, no step (like opt.step)
            else:
                lo, hi = self.stepper.owned
                self.opt.step_range(self.stepper._store, lo, hi)
                self.stepper.all_gather_params(self.opt.pflat)
        else:
            self.opt.step(out["grads"])   # skips the groups replaced above (all of them after a densify / prune)
        if hasattr(self.compute, "mark_params_changed"):
            self.compute.mark_params_changed()
        self.global_step += 1
        if resized:
            self.compute.set_params(self.opt.parameters(), None)
            self.opt.p = {k: v.detach() for k, v in self.compute.params.items()}
            if torch.cuda.is_available():
                self._rz.context(getattr(self.compute, "device", None)).clear()
            self._rebuild_step()
        out["changed"] = changed
        if changed and getattr(self.opt, "last_densify", None) is not None and will_densify:
            out["densify"] = self.opt.last_densify
        return out
