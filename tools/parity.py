"""Parity helpers shared by bench.py's CPU leg and the -m gpu tests (tests/fused_oracle.py): reading the fused kernels'
per-(view, Gaussian) records back from the workspace, asking the device for its own alpha / end-of-walk decisions on the
pairs the oracle finds within rounding of a threshold, and forcing the oracle to them.  Test infrastructure: it talks to
the product only through the C ABI and is never imported by the product."""
import ctypes

import numpy as np
import torch

FLIP_EPS = 2e-4
DEV = "cuda:0"


def layout(V, N, W, H, cap):
    from manus_amd._lib import lib
    arr = (ctypes.c_size_t * 40)()
    n = lib().mgr_raster_layout(V, N, W, H, cap, arr, 40)
    names = ["header", "grec", "depth", "rect", "alive", "pair_off", "tile_count", "tile_start", "tile_cursor", "tile_done",
             "tile_queue", "chunk_start", "items", "ckpt", "keys", "sorted_gid", "final_T", "n_contrib", "pair_tag",
             "pair_grad", "total", "inst_grad", "inst_tag", "db_nvis", "db_bbox", "db_order", "tile_zcut", "tile_zused", "tile_qend",
             "tile_rep", "rep_unit", "rep_cnt", "tile_zwin"]
    assert n == len(names)
    return dict(zip(names, [int(x) for x in arr[:n]]))


def read_records(buf, offset, n):
    """(n, 12) float32: the content words of n per-(view, Gaussian) records starting at byte `offset` of a workspace tensor (the
    records sit in slots of mgr_raster_record_bytes() bytes)."""
    from manus_amd._lib import lib
    stride = int(lib().mgr_raster_record_bytes())
    return buf[offset: offset + n * stride].view(torch.float32).reshape(n, stride // 4)[:, :12].cpu().numpy().copy()


def fused_records(ws, V, N, W, H):
    """(grec (V,N,12), depth (V,N), gathered blend sums (N,G,12)) of the last forward / backward on workspace `ws`."""
    L = layout(V, N, W, H, ws.cap)
    P = W * H
    ncontrib = ws.buf[L["n_contrib"]: L["n_contrib"] + V * P * 4].view(torch.int32).reshape(V, H, W).cpu().numpy()
    grec = read_records(ws.buf, L["grec"], V * N).reshape(V, N, 12)
    depth = ws.buf[L["depth"]: L["depth"] + V * N * 4].view(torch.float32).reshape(V, N).cpu().numpy()
    G = 1 if V <= 1 else 2 if V <= 2 else 4 if V <= 4 else 8
    iacc = ws.buf[L["inst_grad"]: L["inst_grad"] + N * G * 48].view(torch.float32).reshape(N, G, 12).cpu().numpy().copy()
    # the sums of a (Gaussian, view) are those of the last backward where its tag is that call's epoch (the largest tag in
    # the workspace); elsewhere the instance had no record and the lane may hold an earlier call's sums (with run lists
    # the gather only writes the instances with records)
    tag = ws.buf[L["inst_tag"]: L["inst_tag"] + V * N * 4].view(torch.int32).reshape(V, N).cpu().numpy()
    if V <= 8 and tag.size:
        valid = np.zeros((N, G), bool)
        valid[:, :V] = (tag == tag.max()).T
        iacc[~valid] = 0.0
    return grec, depth, iacc, ncontrib


def device_alpha_decisions(rec6, px, py, device=DEV):
    """The blend kernels' own alpha / validity for (record, pixel) pairs (mgr_debug_pair_alpha)."""
    from manus_amd._lib import check, lib, ptr, stream
    n = int(len(px))
    if n == 0:
        return np.zeros(0, np.float32), np.zeros(0, np.int32)
    r = torch.as_tensor(np.ascontiguousarray(rec6, np.float32), device=device)
    x = torch.as_tensor(np.ascontiguousarray(px, np.int32), device=device)
    y = torch.as_tensor(np.ascontiguousarray(py, np.int32), device=device)
    al = torch.empty(n, dtype=torch.float32, device=device)
    va = torch.empty(n, dtype=torch.int32, device=device)
    check(lib().mgr_debug_pair_alpha(n, ptr(r), ptr(x), ptr(y), ptr(al), ptr(va), stream()), "mgr_debug_pair_alpha")
    return al.cpu().numpy(), va.cpu().numpy()


def kernel_last_gaussian(view, V, N, W, H, ncontrib_v):
    """(H,W) int32: the Gaussian each pixel's walk ended on in the kernels (-1: no contributor), from the per-pixel list
    position the forward saved (n_contrib, 1-based into the kernels' tile list -- the oracle's list minus the provably
    null pairs, so positions differ but Gaussians do not) and the tile lists (mgr_raster_debug_binning_sync)."""
    from manus_amd import rasterizer as rz
    from manus_amd._lib import lib, ptr, stream
    ws = rz.context().last_ws
    gx, gy = (W + 15) // 16, (H + 15) // 16
    ranges = np.zeros((gx * gy, 2), np.int32)
    npairs, ovf = ctypes.c_int64(0), ctypes.c_int32(0)
    lib().mgr_raster_status_sync(ptr(ws.buf), ctypes.byref(npairs), ctypes.byref(ovf), stream())
    pl = np.zeros((max(int(npairs.value), 1),), np.int32)
    rc = lib().mgr_raster_debug_binning_sync(ptr(ws.buf), V, N, W, H, ws.cap, view, ranges.ctypes.data_as(ctypes.c_void_p),
                                             pl.ctypes.data_as(ctypes.c_void_p), pl.shape[0], stream())
    assert rc == 0
    ys, xs = np.mgrid[0:H, 0:W]
    tile = (ys // 16) * gx + xs // 16
    start, size = ranges[tile, 0].astype(np.int64), (ranges[tile, 1] - ranges[tile, 0]).astype(np.int64)
    last = np.where(size > 0, ncontrib_v.astype(np.int64), 0)     # (nothing is written under empty tiles)
    assert (last <= size).all()
    return np.where(last > 0, pl[np.clip(start + last - 1, 0, pl.shape[0] - 1)], -1).astype(np.int32)


def align_threshold_decisions(bo, rec, colors, bg, W, kernel_last=None, device=DEV):
    """Force the oracle `bo` to the kernels' outcome of the alpha >= 1/255 test on the pairs where the two disagree, and
    -- with kernel_last (H,W), the Gaussian every pixel's walk ended on in the kernels -- to the kernels' end of the
    walk (the second threshold, T (1 - alpha) < 1e-4).  rec (N,12): the kernels' per-Gaussian records of this view.
    Returns (ambiguous pairs, alpha flips, stop flips, stop flips that were NOT within rounding of the threshold)."""
    pix, gid, al = bo.ambiguous_pairs(FLIP_EPS)
    flips = np.zeros(0, bool)
    dev_valid = np.zeros(0, np.int32)
    if len(pix):
        dev_alpha, dev_valid = device_alpha_decisions(rec[gid][:, :6], pix % W, pix // W, device)
        ora_keep = (al >= np.float32(1.0) / np.float32(255.0)).astype(np.int32)
        flips = dev_valid != ora_keep
        assert np.abs(dev_alpha - al).max() < 1e-6, "device and oracle alpha differ by more than rounding on identical inputs"
    need_stop = kernel_last is not None
    sf = sv = 0
    if flips.any() or need_stop:
        sf, sv = bo.reblend(pix[flips], gid[flips], dev_valid[flips], colors, bg, forced_last=kernel_last if need_stop else None)
    return int(len(pix)), int(flips.sum()), sf, sv
