"""Hand <-> object contact distances over the HIP kernel of `csrc/contact.hip`.

Mirrors src/utils/gaussian_utils.py of brown-ivl/manus:

    get_contact_map(pt1, pt2, chunk)    :514-518   chunked torch.cdist(...).min(1)[0]
    get_contact_dist(pt1, pt2)          :521-549   taichi brute-force nearest point, distance + index
    get_cmap(pt1, pt2, c_thresh)        :571-577   1 - clamp(dist, 0, c_thresh) / c_thresh (+ a matplotlib colour map,
                                                   which stays with the caller)

GPU tensors only; there is no CPU fallback.
"""
import torch

from ._lib import ManusHipError, check, f32c, lib, ptr, stream


def _nearest(pt1, pt2, want_idx):
    pt1, pt2 = f32c(pt1), f32c(pt2)
    if not pt1.is_cuda:
        raise ManusHipError("manus_amd.contact needs GPU tensors; there is no CPU fallback")
    if pt1.dim() != 2 or pt1.shape[1] != 3 or pt2.dim() != 2 or pt2.shape[1] != 3:
        raise ManusHipError("contact: points are (N,3)")
    n1, n2 = pt1.shape[0], pt2.shape[0]
    dist = torch.empty((n1,), dtype=torch.float32, device=pt1.device)
    idx = torch.empty((n1,), dtype=torch.int32, device=pt1.device) if want_idx else None
    nbytes = int(lib().mgr_contact_workspace_bytes(n1, n2))
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=pt1.device)
    check(lib().mgr_contact_dist(n1, ptr(pt1), n2, ptr(pt2), ptr(dist), ptr(idx), ptr(ws), nbytes, stream()),
          "mgr_contact_dist")
    return dist, idx


def get_contact_dist(pt1, pt2):
    """(distance to the nearest point of pt2, its index as float32 like the reference's taichi ndarray)."""
    dist, idx = _nearest(pt1, pt2, True)
    return dist, idx.to(torch.float32)


def get_contact_map(pt1, pt2, chunk=1024):
    """Distance of every point of pt1 to its nearest point of pt2 (`chunk` kept for signature parity)."""
    return _nearest(pt1, pt2, False)[0]


def get_cmap_values(pt1, pt2, c_thresh=0.004):
    """(1 - clamp(dist, 0, c_thresh)/c_thresh, indices): get_cmap without the colour-map lookup."""
    dist, idx = get_contact_dist(pt1, pt2)
    return 1 - torch.clamp(dist.clone(), 0, c_thresh) / c_thresh, idx
