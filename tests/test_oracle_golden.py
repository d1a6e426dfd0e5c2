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


GAUSSIAN_OPTS = dict(position_lr_init=0.0016, position_lr_final=0.0000016, position_lr_max_steps=30000,
                     feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001,
                     densify_grad_threshold=0.0002, min_opacity_threshold=0.005)   # config/model/gaussian/gaussian.yaml


def _opt_state(d, tag):
    st = {}
    for n in tr.LEAVES:
        st[n] = torch.tensor(d[f"{tag}_{n}"])
        st[n + "_m"] = torch.tensor(d[f"{tag}_{n}_m"]) if f"{tag}_{n}_m" in d else torch.zeros_like(st[n])
        st[n + "_v"] = torch.tensor(d[f"{tag}_{n}_v"]) if f"{tag}_{n}_v" in d else torch.zeros_like(st[n])
    st["skin"] = torch.tensor(d[f"{tag}_skin"])
    return st


@pytest.mark.parametrize("name", ["optimizer_s0.npz", "optimizer_s1.npz", "optimizer_s2.npz", "optimizer_s3.npz"])
def test_optimizer_and_densify_match_reference(golden_dir, name):
    """Adam steps with the xyz schedule, densify_and_prune and reset_opacity restated in oracle/torch_ref.py
    against the reference's own GaussianModel run (tests/golden/make_golden.py::make_optimizer_golden)."""
    d = np.load(os.path.join(golden_dir, name))
    st = _opt_state(d, "init")
    K = int(d["K"])
    for k in range(K):
        lrs = tr.group_lrs(GAUSSIAN_OPTS, float(d["spatial_lr_scale"]), int(d[f"step{k}"]))
        np.testing.assert_allclose(lrs, d["lrs"][k], rtol=1e-6)  # spatial_lr_scale is stored as fp32
        for n, lr in zip(tr.LEAVES, lrs):
            st[n], st[n + "_m"], st[n + "_v"] = tr.adam_step(st[n], torch.tensor(d[f"grad{k}_{n}"]), st[n + "_m"],
                                                             st[n + "_v"], lr, k + 1)
    for n in tr.LEAVES:
        assert max_rel_err(st[n].numpy(), d[f"adam_{n}"]) < 2e-6, n
        assert max_rel_err(st[n + "_m"].numpy(), d[f"adam_{n}_m"]) < 2e-6, n
        assert max_rel_err(st[n + "_v"].numpy(), d[f"adam_{n}_v"]) < 2e-6, n
    st = _opt_state(d, "adam")
    std = d["split_std"]
    noise = torch.tensor(d["split_samples"] / std) if std.size else torch.zeros((0, 3))
    new = tr.densify_and_prune(st, torch.tensor(d["stat_accum"]), torch.tensor(d["stat_denom"]),
                               GAUSSIAN_OPTS["densify_grad_threshold"], GAUSSIAN_OPTS["min_opacity_threshold"],
                               float(d["extent"]), float(d["percent_dense"]), noise,
                               max_screen_size=float(d["size_threshold"]) or None)
    if name in ("optimizer_s2.npz", "optimizer_s3.npz"):   # these hold Gaussians larger than 0.1 * extent
        big = (new["scaling"].exp().max(1).values > 0.1 * float(d["extent"])).sum()
        assert (big > 0) == (name == "optimizer_s2.npz")    # kept while size_threshold is None, pruned once it is set
    assert new["xyz"].shape == d["dens_xyz"].shape and new["xyz"].shape[0] != d["adam_xyz"].shape[0]
    for n in tr.LEAVES:
        assert max_rel_err(new[n].numpy(), d[f"dens_{n}"]) < 2e-6, n
        np.testing.assert_array_equal(new[n + "_m"].numpy(), d[f"dens_{n}_m"])
        np.testing.assert_array_equal(new[n + "_v"].numpy(), d[f"dens_{n}_v"])
    np.testing.assert_array_equal(new["skin"].numpy(), d["dens_skin"])
    assert not d["dens_accum"].any() and not d["dens_denom"].any() and not d["dens_maxrad"].any()
    rs = tr.reset_opacity(new)
    assert max_rel_err(rs["opacity"].numpy(), d["reset_opacity"]) < 2e-6
    assert not d["reset_opacity_m"].any() and not d["reset_opacity_v"].any()
    np.testing.assert_array_equal(rs["xyz_m"].numpy(), d["reset_xyz_m"])


def test_prune_points_matches_reference(golden_dir):
    """prune_points restated against GaussianModel.prune_points (gaussian.py:167-203)."""
    d = np.load(os.path.join(golden_dir, "prune_points.npz"))
    st = _opt_state(d, "pre")
    st.update(accum=torch.tensor(d["pre_accum"]), denom=torch.tensor(d["pre_denom"]), maxrad=torch.tensor(d["pre_maxrad"]))
    new = tr.prune_points(st, torch.tensor(d["mask"]))
    assert new["xyz"].shape[0] == int((~d["mask"]).sum()) == d["post_xyz"].shape[0]
    for n in tr.LEAVES:
        for suf in ("", "_m", "_v"):
            np.testing.assert_array_equal(new[n + suf].numpy(), d[f"post_{n}{suf}"])
    for a, b in (("skin", "skin"), ("accum", "accum"), ("denom", "denom"), ("maxrad", "maxrad")):
        np.testing.assert_array_equal(new[a].numpy(), d["post_" + b])


def test_points_outside_mask_matches_reference(golden_dir):
    """dilate_mask / get_points_outside_mask restated against gaussian_utils.py:35-47,101-147."""
    d = np.load(os.path.join(golden_dir, "points_outside_mask.npz"))
    K, E, pts, mask = (torch.tensor(d[k]) for k in ("K", "extr", "points", "mask"))
    np.testing.assert_array_equal(tr.dilate_mask(mask[0, ..., 0]).numpy(), d["dilated"])
    f = lambda kp, dil: tr.points_outside_mask(pts, K[0], E[0], mask[0], None if kp is None else torch.tensor(d[kp]), dil).numpy()
    np.testing.assert_array_equal(f(None, False), d["obj"])
    np.testing.assert_array_equal(f("key_in", True), d["hand_in"])
    np.testing.assert_array_equal(f("key_in", False), d["hand_in_nodilate"])
    np.testing.assert_array_equal(f("key_out", True), d["hand_out"])
    assert d["obj"].any() and d["hand_in"].any() and not d["hand_out"].any()
    assert d["obj"].sum() > d["hand_in"].sum()           # the dilated mask keeps more points


def test_contact_distance_matches_reference(golden_dir):
    """The loop restatement of get_contact_dist against the reference's other implementation of the same
    quantity (get_contact_map = torch.cdist().min): distances agree to cdist's accuracy; the index attains it."""
    d = np.load(os.path.join(golden_dir, "contact.npz"))
    for k in range(4):
        pt1, pt2 = d[f"pt1_{k}"], d[f"pt2_{k}"]
        dist, idx = tr.contact_dist(pt1, pt2)
        np.testing.assert_allclose(dist, d[f"dist_{k}"], rtol=2e-4, atol=5e-5)   # torch.cdist is matmul-based: exact contacts come out as ~2e-5
        np.testing.assert_allclose(np.linalg.norm(pt1 - pt2[idx], axis=1), dist, rtol=1e-6, atol=1e-9)
    dist, idx = tr.contact_dist(d["pt1_2"], d["pt2_2"])
    assert (dist[50:60] == 0).all() and (idx[50:60] == np.arange(100, 110)).all()


def test_training_loss_matches_reference_loss_func(golden_dir):
    """0.8 rgb_loss + 0.2 ssim_loss + 0.1 isotropic_reg restated from the oracle pieces against the reference's own
    loss_func (base.py:323-365) and its gradients."""
    d = np.load(os.path.join(golden_dir, "loss_func.npz"))
    pred = torch.tensor(d["pred"], requires_grad=True)
    ls = torch.tensor(d["log_scale"], requires_grad=True)
    loss = tr.rgb_ssim_loss(pred, torch.tensor(d["gt"]), 0.8, 0.2) + 0.1 * tr.isotropic_reg(ls, 0.4)
    assert abs(loss.item() - float(d["loss"])) < 2e-6
    assert abs(tr.isotropic_reg(ls, 0.4).item() - float(d["iso"])) < 2e-6
    gp, gs = torch.autograd.grad(loss, [pred, ls])
    assert max_rel_err(gp.numpy(), d["g_pred"]) < 2e-5
    assert max_rel_err(gs.numpy(), d["g_log_scale"]) < 2e-5
