"""CPU: the torch oracle (oracle/torch_ref.py) against the golden vectors produced by
importing the reference (tests/golden/make_golden.py).  This is what pins the oracle."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import torch_ref as tr

from util import max_rel_err


def _leaves(d):
    names = {"_xyz": "xyz", "_scaling": "scaling", "_rotation": "rotation", "_features_dc": "features_dc",
             "_features_rest": "features_rest", "_opacity": "opacity"}
    return {k: torch.tensor(d[v], requires_grad=True) for k, v in names.items()}


@pytest.mark.parametrize("fn", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "lbs_sh_hand_*.npz"))))
def test_hand_chain_matches_reference(fn):
    d = np.load(fn)
    p = _leaves(d)
    out = tr.hand_forward(p, torch.tensor(d["grid"]), torch.tensor(d["grid_center"]), torch.tensor(d["grid_scale"]),
                          torch.tensor(d["posed"]), torch.tensor(d["rest"]), torch.tensor(d["cam_center"]))
    for k in ("posed_xyz", "posed_cov", "tf", "skin_wts", "colors"):
        assert max_rel_err(out[k].detach().numpy(), d[k]) < 2e-6, k
    assert max_rel_err(out["opacity"].detach().numpy(), d["opacity_act"]) < 1e-6
    loss = ((out["posed_xyz"] * torch.tensor(d["r1"])).sum() + (out["posed_cov"] * torch.tensor(d["r2"])).sum()
            + (out["colors"] * torch.tensor(d["r3"])).sum())
    loss.backward()
    for k, v in p.items():
        ref = d["grad" + k]
        got = v.grad.numpy() if v.grad is not None else np.zeros_like(ref)
        assert max_rel_err(got, ref) < 2e-5, k


def test_object_chain_matches_reference(golden_dir):
    d = np.load(os.path.join(golden_dir, "lbs_sh_object_s3_n64.npz"))
    p = _leaves(d)
    out = tr.object_forward(p, torch.tensor(d["cam_center"]))
    for k in ("posed_xyz", "posed_cov", "colors"):
        assert max_rel_err(out[k].detach().numpy(), d[k]) < 2e-6, k
    loss = ((out["posed_xyz"] * torch.tensor(d["r1"])).sum() + (out["posed_cov"] * torch.tensor(d["r2"])).sum()
            + (out["colors"] * torch.tensor(d["r3"])).sum())
    loss.backward()
    for k, v in p.items():
        ref = d["grad" + k]
        got = v.grad.numpy() if v.grad is not None else np.zeros_like(ref)
        assert max_rel_err(got, ref) < 2e-5, k


def test_eval_sh(golden_dir):
    d = np.load(os.path.join(golden_dir, "sh_eval.npz"))
    for deg in range(4):
        got = tr.eval_sh(deg, torch.tensor(d["coeffs"]), torch.tensor(d["dirs"])).numpy()
        np.testing.assert_allclose(got, d[f"deg{deg}"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(tr.rgb2sh(torch.tensor(d["coeffs"][:, :, 0])).numpy(), d["rgb2sh"], rtol=1e-6)


def test_cameras(golden_dir):
    d = np.load(os.path.join(golden_dir, "cameras.npz"))
    for i in range(d["K"].shape[0]):
        o = tr.camera_attributes(d["K"][i].copy(), d["extr"][i].copy(), int(d["width"]), int(d["height"]))
        for k in ("fovx", "fovy", "world_view_transform", "projection_matrix", "full_proj_transform", "camera_center"):
            np.testing.assert_allclose(np.asarray(o[k]), d[k][i], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(tr.projection_matrix(0.01, 100.0, 0.7, 0.5), d["proj_0p01_100"], rtol=1e-13)


def test_fk_known_answer(golden_dir):
    """FK reproduces the reference's bundled pose_matrixs (data/meta_data/novel_pose.pkl)."""
    d = np.load(os.path.join(golden_dir, "fk_novel_pose.npz"))
    rest = torch.tensor(d["rest_matrixs"])
    pose = torch.tensor(d["pose_params"])
    F = pose.shape[0]
    fk = tr.fk_pose_wrt_root(rest, pose, torch.eye(3)[None].repeat(F, 1, 1), torch.zeros(F, 3), d["parents"])
    assert np.abs(fk.numpy() - d["pose_matrixs"]).max() < 2e-6          # the known-answer file itself
    assert np.abs(fk.numpy() - d["fk"]).max() < 1e-6                    # the reference function's output
    fkg = tr.fk_pose_wrt_root(rest, pose, torch.tensor(d["global_R"]), torch.tensor(d["global_t"]), d["parents"])
    assert np.abs(fkg.numpy() - d["fk_global"]).max() < 1e-6
    e = tr.euler_to_matrix(torch.tensor(d["root_rotation"]), "XYZ", intrinsic=True).numpy()
    assert np.abs(e - d["euler_intrinsic"]).max() < 1e-6
    assert np.abs(e - d["pose_matrix_world0"][:, :3, :3]).max() < 1e-6  # known answer from the pkl
    e2 = tr.euler_to_matrix(torch.tensor(d["root_rotation"]), "XYZ", intrinsic=False).numpy()
    assert np.abs(e2 - d["euler_extrinsic"]).max() < 1e-6
    eb = tr.euler_to_matrix(torch.tensor(d["eulers"]), "XYZ", intrinsic=True).numpy()
    assert np.abs(eb - d["euler_bones_intrinsic"]).max() < 1e-6


def test_project_points(golden_dir):
    d = np.load(os.path.join(golden_dir, "fk_novel_pose.npz"))
    got = tr.project_points(torch.tensor(d["pp_points"]), torch.tensor(d["pp_K"]), torch.tensor(d["pp_E"])).numpy()
    np.testing.assert_allclose(got, d["pp_out"], rtol=1e-5, atol=1e-3)


def test_image_losses_match_reference(golden_dir):
    """l1_loss / ssim as loss_func calls them (HWC images: the SSIM window runs over the (W,3)
    plane of each row), values and gradients, against the reference's own outputs."""
    d = np.load(os.path.join(golden_dir, "image_loss.npz"))
    for k in range(3):
        pred = torch.tensor(d[f"pred{k}"], requires_grad=True)
        gt = torch.tensor(d[f"gt{k}"])
        ss = tr.ssim_hwc(pred, gt)
        assert abs(ss.item() - float(d[f"ssim_{k}"])) < 2e-6
        (g,) = torch.autograd.grad(ss, pred)
        assert max_rel_err(g.numpy(), d[f"g_ssim_{k}"]) < 2e-5
        pred2 = torch.tensor(d[f"pred{k}"], requires_grad=True)
        full = tr.rgb_ssim_loss(pred2, gt, 0.8, 0.2)
        ref = 0.8 * float(d[f"l1_{k}"]) + 0.2 * (1.0 - float(d[f"ssim_{k}"]))
        assert abs(full.item() - ref) < 2e-6
        (g2,) = torch.autograd.grad(full, pred2)
        assert max_rel_err(g2.numpy(), 0.8 * d[f"g_l1_{k}"] - 0.2 * d[f"g_ssim_{k}"]) < 2e-5
