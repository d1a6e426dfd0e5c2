"""CPU restatement of the skin-weight initialisation from the MANO rest mesh -- TEST INFRASTRUCTURE ONLY (see
oracle/__init__.py).  float64 numpy, small inputs.

* knn_mean_rows      torch.cdist(points, verts).topk(k, largest=False) + np.mean(rows[idx], axis=1)
                     (/root/reference/src/utils/train_utils.py:70-73), with exact float64 distances; pinned by
                     tests/golden/mano_init.npz (the reference's own init_mano_weights, filter_grid=False).
* mesh_sdf           what the reference asks of `pysdf.SDF(verts, faces)(points)` (train_utils.py:55-58): distance to
                     the surface, positive inside.  pysdf is an external package (github.com/sxyu/sdf, unpinned in the
                     reference's setup) that is not in this image: PARITY UNPINNED.  Restated from its published
                     contract -- exact point-triangle distance; containment decided here by the generalised winding
                     number (Jacobson et al. 2013) because the MANO mesh is open at the wrist.
* init_mano_weights  train_utils.py:48-89 around the two.
"""
import numpy as np

MANO_TO_OURS = [13, 14, 14, 15, 0, 1, 2, 3, 0, 4, 5, 6, 0, 10, 11, 12, 0, 7, 8, 9]


def knn_indices(points, refs, k):
    p, r = np.asarray(points, np.float64), np.asarray(refs, np.float64)
    d2 = ((p[:, None, :] - r[None, :, :]) ** 2).sum(-1)
    return np.argsort(d2, axis=1, kind="stable")[:, :k], np.sort(d2, axis=1)[:, :k + 1]


def knn_mean_rows(points, refs, rows, k):
    idx, _ = knn_indices(points, refs, k)
    return np.asarray(rows, np.float64)[idx].mean(1)


def _closest_on_triangle(p, a, b, c):
    """Closest points of triangles (a, b, c) (each (F,3)) to ONE point p (3,): Voronoi-region walk (Ericson 5.1.5),
    vectorised over the triangles."""
    ab, ac, ap = b - a, c - a, p - a
    d1, d2 = (ab * ap).sum(-1), (ac * ap).sum(-1)
    bp = p - b
    d3, d4 = (ab * bp).sum(-1), (ac * bp).sum(-1)
    cp = p - c
    d5, d6 = (ab * cp).sum(-1), (ac * cp).sum(-1)
    vc, vb, va = d1 * d4 - d3 * d2, d5 * d2 - d1 * d6, d3 * d6 - d5 * d4
    with np.errstate(divide="ignore", invalid="ignore"):
        den = 1.0 / (va + vb + vc)
        q = a + ab * (vb * den)[:, None] + ac * (vc * den)[:, None]                  # interior
        m = (va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0)                          # edge bc
        w = (d4 - d3) / ((d4 - d3) + (d5 - d6))
        q = np.where(m[:, None], b + (c - b) * w[:, None], q)
        m = (vb <= 0) & (d2 >= 0) & (d6 <= 0)                                        # edge ac
        q = np.where(m[:, None], a + ac * (d2 / (d2 - d6))[:, None], q)
        m = (d6 >= 0) & (d5 <= d6)                                                   # vertex c
        q = np.where(m[:, None], c, q)
        m = (vc <= 0) & (d1 >= 0) & (d3 <= 0)                                        # edge ab
        q = np.where(m[:, None], a + ab * (d1 / (d1 - d3))[:, None], q)
        m = (d3 >= 0) & (d4 <= d3)                                                   # vertex b
        q = np.where(m[:, None], b, q)
        m = (d1 <= 0) & (d2 <= 0)                                                    # vertex a
        q = np.where(m[:, None], a, q)
    return q


def mesh_sdf(points, verts, faces):
    """(signed distance, winding number) per point; the regions are tested in the reverse of the kernel's if-chain so
    that the FIRST matching region of the chain wins, like there."""
    v = np.asarray(verts, np.float64)
    f = np.asarray(faces, np.int64)
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    out, wind = np.zeros(len(points)), np.zeros(len(points))
    for i, p in enumerate(np.asarray(points, np.float64)):
        q = _closest_on_triangle(p, a, b, c)
        d = np.sqrt(((p - q) ** 2).sum(-1).min())
        A, B, C = a - p, b - p, c - p
        la, lb, lc = np.linalg.norm(A, axis=1), np.linalg.norm(B, axis=1), np.linalg.norm(C, axis=1)
        det = (A * np.cross(B, C)).sum(-1)
        den = la * lb * lc + (A * B).sum(-1) * lc + (B * C).sum(-1) * la + (C * A).sum(-1) * lb
        w = (2.0 * np.arctan2(det, den)).sum() / (4.0 * np.pi)
        wind[i] = w
        out[i] = d if abs(w) > 0.5 else -d
    return out, wind


def init_mano_weights(points, data, neighbors=20, filter_grid=True, threshold=-0.02):
    init_weights = np.asarray(data["weights"])[..., MANO_TO_OURS]
    weights = knn_mean_rows(points, data["verts"], init_weights, neighbors)
    mask = None
    if filter_grid:
        sdf, _ = mesh_sdf(points, data["verts"], data["face"])
        mask = sdf > threshold
        weights = np.concatenate([weights, np.zeros((weights.shape[0], 1))], axis=-1)
        weights[sdf < threshold] = 0
        weights[sdf < threshold, -1] = 1
    return weights / weights.sum(-1, keepdims=True), mask
