"""CPU: the host half of the MANO skin-weight initialisation and its float64 oracle against the reference's own outputs
(tests/golden/mano_init.npz; generator tests/golden/make_golden.py --mano)."""
import os

import numpy as np
import pytest
import torch


def test_skinning_grid_lattice_equals_reference(golden_dir):
    from manus_amd import mano_init as MI
    g = np.load(os.path.join(golden_dir, "mano_init.npz"))
    np.testing.assert_array_equal(MI.create_skinning_grid(3, 4, 5).numpy(), g["grid_3_4_5"])


def test_oracle_nearest_vertex_weights_equal_reference(golden_dir):
    from oracle import mesh_ref as R
    g = np.load(os.path.join(golden_dir, "mano_init.npz"))
    m = np.load(os.path.join(golden_dir, "mano_rest.npz"))
    data = {"verts": m["verts"], "weights": m["weights"], "face": m["faces"]}
    pts = g["points"][:800]
    for k in (4, 20):
        w, mask = R.init_mano_weights(pts, data, neighbors=k, filter_grid=False)
        assert mask is None
        _, d2 = R.knn_indices(pts, m["verts"], k)
        clear = (d2[:, k] - d2[:, k - 1]) > 1e-5 * d2[:, k]
        np.testing.assert_allclose(w[clear], g["weights_k%d" % k][:800][clear], atol=1e-6)
        idx, _ = R.knn_indices(pts, m["verts"], k)
        assert (np.sort(idx, 1) == np.sort(g["idx_k%d" % k][:800], 1))[clear].all()


def test_oracle_signed_distance_known_answers():
    from oracle import mesh_ref as R
    V = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]], float)
    F = np.array([[0, 2, 1], [0, 3, 2], [4, 5, 6], [4, 6, 7], [0, 1, 5], [0, 5, 4], [2, 3, 7], [2, 7, 6], [1, 2, 6], [1, 6, 5], [0, 4, 7], [0, 7, 3]])
    P = np.array([[0.5, 0.5, 0.5], [0.5, 0.5, 0.9], [1.5, 0.5, 0.5], [2, 2, 2], [0.2, 0.5, 0.5], [-1, -1, 0.5], [0.5, 2.0, 3.0]])
    sdf, w = R.mesh_sdf(P, V, F)
    np.testing.assert_allclose(sdf, [0.5, 0.1, -0.5, -np.sqrt(3.0), 0.2, -np.sqrt(2.0), -np.sqrt(5.0)], atol=1e-12)
    np.testing.assert_allclose(np.abs(w), [1, 1, 0, 0, 1, 0, 0], atol=1e-12)
    # an open box (lid removed): the winding number degrades gracefully, deep inside still counts as inside
    sdf_open, w_open = R.mesh_sdf(np.array([[0.5, 0.5, 0.2], [0.5, 0.5, 3.0]]), V, F[[0, 1, 4, 5, 6, 7, 8, 9, 10, 11]])
    assert abs(w_open[0]) > 0.5 and sdf_open[0] > 0 and abs(w_open[1]) < 0.5 and sdf_open[1] < 0


def test_init_fails_loudly_without_a_gpu(golden_dir):
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    from manus_amd import mano_init as MI
    from manus_amd._lib import ManusHipError
    m = np.load(os.path.join(golden_dir, "mano_rest.npz"))
    with pytest.raises(ManusHipError):
        MI.init_mano_weights(np.zeros((4, 3), np.float32), {"verts": m["verts"], "weights": m["weights"], "face": m["faces"]})
