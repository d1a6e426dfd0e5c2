"""CPU: the sequence reader (manus_amd/dataset.py) against the reference's own
`src/datasets/brics_dynamic.py::Dataset` run on the same two action files (tests/golden/seq/*.npz; outputs in
tests/golden/dataset.npz, generator tests/golden/make_golden.py --dataset)."""
import json
import os
import shutil

import numpy as np
import pytest
import torch

from manus_amd import dataset as D

CFGS = {"a": dict(num_time_steps=2, split_ratio=0.75, sequences="all", split_by_action=False),
        "b": dict(num_time_steps=-1, split_ratio=0.5, sequences=["grasp_10"], split_by_action=True)}
BASE = dict(resize_factor=1.0, bg_color="white", subject="s1", width=64, height=48, rand_views_per_timestep=-1, n_bones=20)


@pytest.fixture()
def seq_dir(golden_dir, tmp_path):
    for f in os.listdir(os.path.join(golden_dir, "seq")):
        shutil.copy(os.path.join(golden_dir, "seq", f), tmp_path / f)
    return str(tmp_path)


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("split", ["train", "val"])
def test_dataset_matches_reference(golden_dir, seq_dir, tag, split):
    g = np.load(os.path.join(golden_dir, "dataset.npz"))
    ds = D.SequenceDataset(seq_dir, dict(BASE, **CFGS[tag]), split)
    k = "%s_%s_" % (tag, split)
    assert ["|".join(map(str, t)) for t in ds.index_list] == list(g[k + "index"])
    assert [a.rsplit(".", 1)[0] for a in ds.actions] == [a.rsplit(".", 1)[0] for a in g[k + "actions"]]
    assert list(ds.cam_names) == list(g[k + "cam_names"])
    assert abs(float(ds.extent) - float(g[k + "extent"])) < 1e-12
    for f in ("K", "extr", "fovx", "fovy", "world_view_transform", "projection_matrix", "full_proj_transform", "camera_center"):
        np.testing.assert_allclose(np.asarray(getattr(ds.all_cameras, f)), g[k + "cams_" + f], rtol=1e-12, atol=1e-12)
    items = sorted({0, len(ds) // 2, len(ds) - 1})
    assert all(k + "item%d_rgb" % i in g for i in items)
    for idx in items:
        d, kk = ds[idx], k + "item%d_" % idx
        assert torch.equal(d["rgb"], torch.tensor(g[kk + "rgb"])) and torch.equal(d["mask"], torch.tensor(g[kk + "mask"]))
        assert d["rgb"].dtype == torch.float32 and d["rgb"].shape == (1, 48, 64, 3) and d["mask"].shape == (1, 48, 64, 1)
        assert torch.equal(d["bg_color"], torch.tensor(g[kk + "bg"]))
        assert [str(d["info"][0]), str(d["info"][1]), str(d["info"][2]), str(d["info"][3][0])] == list(g[kk + "info"])
        np.testing.assert_allclose(d["pose_latent"].numpy(), g[kk + "pose_latent"], atol=1e-6)
        for f in ("K", "extr", "world_view_transform", "full_proj_transform", "camera_center", "fovx"):
            np.testing.assert_allclose(np.asarray(getattr(d["camera"], f)), g[kk + "cam_" + f], rtol=1e-6, atol=1e-7)
        for f in ("heads", "tails", "transforms"):
            assert torch.equal(getattr(d["bones_rest"], f), torch.tensor(g[kk + "rest_" + f]))
        for f in ("heads", "tails", "transforms", "eulers", "eulers_c", "root_translation", "root_rotation"):
            assert torch.equal(getattr(d["bones_posed"], f), torch.tensor(g[kk + "posed_" + f])), f
        assert [int(d["bones_posed"].kintree[str(i)]) for i in range(20)] == list(g[kk + "kintree"])


def test_tree_store_protocol_and_helpers(seq_dir, tmp_path):
    with D.open_sequence(os.path.join(seq_dir, "grasp_2.npz")) as f:
        assert sorted(f.keys()) == ["K", "extr", "frames", "mano_rest"]
        assert f["frames"].keys() == ["11", "14", "8"]          # by name, like h5py; natsorted orders them for use
        assert D.natsorted(f["frames"].keys()) == ["8", "11", "14"]
        assert f.get("nope") is None and "K" in f and len(f["K"]) == 4
        assert f["frames"]["8"]["metadata"]["rest_matrixs"][:].shape == (20, 4, 4)
        assert dict(f["mano_rest"].items())["verts"].shape == (30, 3)
        with pytest.raises(KeyError):
            f["frames"]["9"]
    assert D.natsorted(["a10", "a2", "b1", "a1"]) == ["a1", "a2", "a10", "b1"]
    # an HDF5 file without h5py: a loud, explained failure (this image has no h5py)
    p = tmp_path / "x.hdf5"
    p.write_bytes(b"\x89HDF\r\n\x1a\n" + b"\0" * 64)
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(RuntimeError, match="h5py"):
            D.open_sequence(str(p))
    # area resize: 1/k block means with cv2's rounding; other factors need OpenCV
    img = np.arange(4 * 6 * 4, dtype=np.uint8).reshape(4, 6, 4)
    half = D._area_resize(img, 0.5)
    assert half.shape == (2, 3, 4) and half[0, 0, 0] == int(np.floor((0 + 4 + 24 + 28) / 4 + 0.5))
    assert D._area_resize(img, 1.0) is img
    # quaternions: unit norm, real part first, round trip through the rotation they encode
    e = torch.randn(50, 3, generator=torch.Generator().manual_seed(0))
    q = D.euler_angles_to_quats(e)
    assert torch.allclose(q.norm(dim=-1), torch.ones(50), atol=1e-6)
    from manus_amd import transforms as T
    R = T.euler_angles_to_matrix(e, "XYZ", intrinsic=True)
    w, x, y, z = q.unbind(-1)
    R2 = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                      2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                      2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)
    assert torch.allclose(R, R2, atol=1e-5)


def test_random_views_split_file_and_view_batch(seq_dir, tmp_path):
    ds = D.SequenceDataset(seq_dir, dict(BASE, num_time_steps=-1, rand_views_per_timestep=3), "train",
                           split_file_dir=str(tmp_path))
    assert len(ds) == 2 * 3 and all(t[2] is None for t in ds.index_list)      # one item per (action, frame)
    d = ds[1]
    assert d["rgb"].shape == (3, 48, 64, 3) and len(set(d["info"][3])) == 3 and d["camera"].K.shape == (3, 3, 3)
    assert json.load(open(tmp_path / "train_split.json")) == [list(t) for t in ds.index_list]
    ds1 = D.SequenceDataset(seq_dir, dict(BASE, sequences=["grasp_2"]), "train")
    assert len(ds1) == 3 * 4 and ds1.fetch_data_by_frame("grasp_2", "8", "cam01") is not None
    assert ds1.fetch_data_by_frame("grasp_2", "9", "cam01") is None
    b = ds1.view_batch([0, 1, 5])
    assert b["targets"].shape == (3, 3, 48, 64) and b["masks"].shape == (3, 48, 64) and b["posed"].shape == (3, 20, 4, 4)
    assert b["keypoints"].shape == (3, 21, 3) and len(b["cameras"]) == 3 and b["rest"].shape == (20, 4, 4)
    assert torch.equal(b["keypoints"][0, 0], ds1[0]["bones_posed"].heads[0])
    # the synthetic writer reproduces the committed fixture byte for byte
    arr = D.synthetic_sequence(11)
    with D.open_sequence(os.path.join(seq_dir, "grasp_2.npz")) as f:
        assert np.array_equal(f["frames"]["8"]["images"]["cam02"], arr["frames/8/images/cam02"])


# ---------------------------------------------------------------------------------------------------------------------
# evaluation trajectories: brics_dynamic.py::TestDataset (485-696); fixture tests/golden/testdataset.npz
# (generator: tests/golden/make_golden.py --testdataset), inputs tests/golden/eval_inputs/*.npz
# ---------------------------------------------------------------------------------------------------------------------
EVAL_CASES = {"a": dict(frame_sample_rate=2, test_on_canonical_pose=False, contact_render_type="default", color_bkgd_aug="white"),
              "b": dict(frame_sample_rate=1, test_on_canonical_pose=True, contact_render_type="default", color_bkgd_aug="black"),
              "c": dict(frame_sample_rate=1, test_on_canonical_pose=False, contact_render_type="gt_eval", color_bkgd_aug="white"),
              "d": dict(frame_sample_rate=3, test_on_canonical_pose=False, contact_render_type="acc_gt_eval", color_bkgd_aug="white")}


@pytest.mark.parametrize("tag", sorted(EVAL_CASES))
def test_eval_trajectory_dataset_matches_reference(golden_dir, tag):
    from manus_amd import dataset as D
    g = np.load(os.path.join(golden_dir, "testdataset.npz"))
    ind = os.path.join(golden_dir, "eval_inputs")
    ds = D.TestDataset(dict(subject="s1", cam_path=os.path.join(ind, "camera_path.npz"), cano_cam_path=os.path.join(ind, "cano_camera.npz"),
                            metadata_path=os.path.join(ind, "novel_pose.npz"), **EVAL_CASES[tag]))
    k = tag + "_"
    assert len(ds) == int(g[k + "len"])
    assert ["|".join(map(str, i)) for i in ds.infos] == list(g[k + "infos"])
    for f in ("K", "extr", "fovx", "fovy", "width", "height", "world_view_transform", "projection_matrix", "full_proj_transform",
              "camera_center"):
        np.testing.assert_array_equal(np.asarray(getattr(ds.all_cameras, f)), g[k + "cams_" + f], err_msg=f)
        np.testing.assert_array_equal(np.asarray(getattr(ds.cano_camera, f)), g[k + "cano_" + f], err_msg="cano " + f)
    assert [str(x) for x in ds.all_cameras.cam_name] == list(g[k + "cams_cam_name"])
    for f in ("heads", "tails", "transforms"):
        np.testing.assert_array_equal(getattr(ds.bones_rest, f).numpy(), g[k + "rest_" + f])
        np.testing.assert_array_equal(np.stack([getattr(b, f).numpy() for b in ds.bones_posed_list]), g[k + "posed_" + f])
    np.testing.assert_allclose(np.stack([p.numpy() for p in ds.pose_latent_list]), g[k + "pose_latent"], rtol=0, atol=1e-6)
    d = ds[len(ds) - 1]
    assert sorted(d.keys()) == list(g[k + "item_keys"])
    np.testing.assert_array_equal(d["bg_color"].numpy(), g[k + "item_bg"])
    assert d["idx"] == int(g[k + "item_idx"]) and d["scaling_modifier"] == 1.0
    np.testing.assert_array_equal(np.asarray(d["camera"].K), g[k + "item_cam_K"])
    assert d["bones_rest"] is ds.bones_rest and d["cano_camera"] is ds.cano_camera


def test_armature_to_world_is_a_point_and_frame_map(golden_dir):
    from manus_amd import dataset as D
    with np.load(os.path.join(golden_dir, "eval_inputs", "novel_pose.npz")) as z:
        md = {k: z[k] for k in z.files}
    w = D.convert_armature_space_to_world_space(md)
    h = np.concatenate([md["pose_heads"], np.ones(md["pose_heads"].shape[:-1] + (1,))], -1)
    np.testing.assert_allclose(w["pose_heads"], np.einsum("fjab,fjb->fja", md["pose_matrix_world"], h)[..., :3], atol=1e-12)
    np.testing.assert_allclose(w["rest_matrixs"], np.einsum("jab,jbc->jac", md["rest_matrix_world"], md["rest_matrixs"]), atol=1e-12)
    assert md["rest_matrixs"] is not w["rest_matrixs"] and "frame_nums" in w     # input left alone, other columns kept


def _calib_rows(n=3):
    """Rows of a calibration file in the column order of params.py:57-88 (names deliberately unsorted)."""
    g = np.random.default_rng(3)
    rows = []
    for k in range(n):
        q = g.normal(size=4)
        q /= np.linalg.norm(q)
        rows.append([k, 1280, 720, 900.0 + 10 * k, 905.0 + 7 * k, 633.0 + 3 * k, 371.0 - 2 * k, -0.08 + 0.01 * k, 0.03, 1e-3, -5e-4,
                     "brics-cam%02d" % (n - k), q[0], q[1], q[2], q[3], 0.1 * k, -0.2, 1.5 + 0.1 * k])
    return rows


def test_calibration_file_cameras(golden_dir, tmp_path):
    """brics_dynamic.py:513-531 (mode "acc_gt_eval" with a calibration .txt): read_params / get_intr / get_extr restate
    params.py; the optimal new camera matrix restates OpenCV's published algorithm -- PARITY UNPINNED (no OpenCV here, no
    vectors upstream), so what is checked are its closed forms."""
    from manus_amd import calib, dataset as D
    rows = _calib_rows()
    path = tmp_path / "calib.txt"
    path.write_text("\n".join(" ".join(str(v) for v in r) for r in rows) + "\n")
    params = calib.read_params(str(path))
    assert [str(p["cam_name"]) for p in params] == sorted(r[11] for r in rows)             # np.sort(order="cam_name")
    intr, dist = calib.get_intr(params[0])
    src = rows[-1]                                                                          # cam01 sorts first
    assert (intr[0, 0], intr[1, 1], intr[0, 2], intr[1, 2]) == tuple(src[3:7]) and tuple(dist) == tuple(src[7:11])
    e = calib.get_extr(params[0])
    assert e.shape == (3, 4)
    np.testing.assert_allclose(e[:, :3] @ e[:, :3].T, np.eye(3), atol=1e-12)                # a rotation
    np.testing.assert_allclose(e[:, 3], src[16:19])
    # no distortion: undistortion is the identity, both rectangles are the frame, and the centred matrix scales the
    # focal lengths by the largest of the four centre / border ratios
    w, h = 1280, 720
    K = np.array([[900.0, 0, 600.0], [0, 910.0, 380.0], [0, 0, 1]])
    pts = np.array([[0.0, 0.0], [w, h], [333.0, 77.0]])
    np.testing.assert_allclose(calib.undistort_points(pts, K, np.zeros(4), K), pts, atol=1e-9)
    new, roi = calib.optimal_new_camera_matrix(K, np.zeros(4), (w, h), alpha=0.0)
    cx, cy = (w - 1) / 2, (h - 1) / 2
    s = max(cx / 600.0, cy / 380.0, cx / (w - 1 - 600.0), cy / (h - 1 - 380.0))     # grid over the pixel centres 0 .. w - 1 (current OpenCV)
    np.testing.assert_allclose([new[0, 0], new[1, 1], new[0, 2], new[1, 2]], [900.0 * s, 910.0 * s, cx, cy], rtol=1e-6)
    old, _ = calib.optimal_new_camera_matrix(K, np.zeros(4), (w, h), alpha=0.0, legacy_grid=True)    # releases before 4.5.2: 0 .. w
    s_old = max(cx / 600.0, cy / 380.0, cx / (w - 600.0), cy / (h - 380.0))
    np.testing.assert_allclose([old[0, 0], old[1, 1]], [900.0 * s_old, 910.0 * s_old], rtol=1e-6)
    assert roi[0] >= 0 and roi[1] >= 0 and roi[0] + roi[2] <= w and roi[1] + roi[3] <= h
    # radial + tangential distortion: the fixed-point iteration inverts the forward model
    d = np.array([-0.12, 0.04, 1.5e-3, -8e-4])
    xy = np.random.default_rng(0).uniform(-0.45, 0.45, size=(50, 2))
    r2 = (xy ** 2).sum(-1)
    rad = 1 + d[0] * r2 + d[1] * r2 ** 2
    xd = xy[:, 0] * rad + 2 * d[2] * xy[:, 0] * xy[:, 1] + d[3] * (r2 + 2 * xy[:, 0] ** 2)
    yd = xy[:, 1] * rad + d[2] * (r2 + 2 * xy[:, 1] ** 2) + 2 * d[3] * xy[:, 0] * xy[:, 1]
    pix = np.stack([xd * K[0, 0] + K[0, 2], yd * K[1, 1] + K[1, 2]], -1)
    want = np.stack([xy[:, 0] * K[0, 0] + K[0, 2], xy[:, 1] * K[1, 1] + K[1, 2]], -1)
    np.testing.assert_allclose(calib.undistort_points(pix, K, d, K), want, atol=0.05)       # five iterations: ~1e-2 px
    np.testing.assert_allclose(calib.undistort_points(pix, K, d, K, iters=40), want, atol=1e-6)
    # barrel distortion (k1 < 0): the undistorted border bulges outwards, so the inscribed rectangle is the frame's own
    # corners' hull shrunk -- alpha = 0 must zoom IN relative to the undistorted bounding rectangle (alpha = 1)
    n0, _ = calib.optimal_new_camera_matrix(K, d, (w, h), alpha=0.0)
    n1, _ = calib.optimal_new_camera_matrix(K, d, (w, h), alpha=1.0)
    assert n0[0, 0] > n1[0, 0] and n0[1, 1] > n1[1, 1] and n0[0, 2] == n1[0, 2] == cx
    # and the dataset takes the file
    ind = os.path.join(golden_dir, "eval_inputs")
    ds = D.TestDataset(dict(cam_path=str(path), cano_cam_path=os.path.join(ind, "cano_camera.npz"), width=w, height=h,
                            metadata_path=os.path.join(ind, "novel_pose.npz"), contact_render_type="acc_gt_eval"))
    assert [i[3] for i in ds.infos] == sorted(r[11] for r in rows)       # the file's camera names label the items (:597)
    tab = calib.camera_table_from_calibration(str(path), w, h)
    np.testing.assert_allclose(np.asarray(ds.all_cameras.K)[:, 0, 0], [k[0] for k in tab["intrs"]], rtol=1e-6)
    assert len(ds) > 0 and "camera" in ds[0]
