"""GPU: articulation kernels (skin weights, LBS of means/covariances, SH colour) against the
golden vectors produced by importing the reference, and against the torch oracle on larger
seeded inputs.  fp32 tolerance: max-rel-err (max|a-b|/max|b|) < 1e-4 for gradients, < 2e-5
for forward values."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import torch_ref as tr

from util import max_rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NAMES = {"_xyz": "xyz", "_scaling": "scaling", "_rotation": "rotation", "_features_dc": "features_dc",
         "_features_rest": "features_rest", "_opacity": "opacity"}


def _hip_hand(d):
    from manus_amd import ops
    from manus_amd.transforms import bone_transforms
    p = {k: torch.tensor(d[v], device=DEV, requires_grad=True) for k, v in NAMES.items()}
    grid = torch.tensor(d["grid"], device=DEV)
    w = ops.skin_weights(p["_xyz"], grid, torch.tensor(d["grid_center"], device=DEV), torch.tensor(d["grid_scale"], device=DEV))
    T = bone_transforms(torch.tensor(d["posed"], device=DEV), torch.tensor(d["rest"], device=DEV))
    pxyz, pcov, tf = ops.lbs_cov(p["_xyz"], p["_scaling"], p["_rotation"], w, T)
    cams = torch.zeros((1, 40), device=DEV)
    cams[0, 34:37] = torch.tensor(d["cam_center"], device=DEV).reshape(-1)
    feats = torch.cat([p["_features_dc"], p["_features_rest"]], 1)
    col = ops.sh_colors(feats, p["_xyz"], tf[0], cams)
    return p, dict(posed_xyz=pxyz[0], posed_cov=pcov[0], tf=tf[0], skin_wts=w, colors=col[0])


@pytest.mark.parametrize("fn", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "lbs_sh_hand_*.npz"))))
def test_hand_chain_vs_reference_golden(fn):
    d = np.load(fn)
    p, out = _hip_hand(d)
    assert max_rel_err(out["posed_xyz"].detach().cpu().numpy(), d["posed_xyz"]) < 2e-5
    assert max_rel_err(out["posed_cov"].detach().cpu().numpy(), d["posed_cov"]) < 2e-5
    assert max_rel_err(out["skin_wts"].detach().cpu().numpy(), d["skin_wts"]) < 2e-5
    assert max_rel_err(out["colors"].detach().cpu().numpy(), d["colors"]) < 2e-5
    assert max_rel_err(out["tf"].detach().cpu().numpy().reshape(-1, 3, 4), d["tf"][:, :3, :]) < 2e-5
    loss = ((out["posed_xyz"] * torch.tensor(d["r1"], device=DEV)).sum()
            + (out["posed_cov"] * torch.tensor(d["r2"], device=DEV)).sum()
            + (out["colors"] * torch.tensor(d["r3"], device=DEV)).sum())
    loss.backward()
    for k in ("_xyz", "_scaling", "_rotation", "_features_dc", "_features_rest"):
        e = max_rel_err(p[k].grad.cpu().numpy(), d["grad" + k])
        assert e < 1e-4, (k, e)


@pytest.mark.parametrize("fn", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "lbs_sh_hand_*.npz")))[:2])
def test_hand_chain_with_4x4_transform_rows_equals_the_12_float_rows(fn):
    """`lbs_cov(..., tf44=True)` returns the reference's (N,4,4) transforms written by the kernel (constant last row included)
    and `sh_colors` reads them in place: same outputs and the same leaf gradients, bit for bit, as the (N,12) rows -- and the 4x4
    equals the reference's golden transform, last row included."""
    from manus_amd import ops
    from manus_amd.transforms import bone_transforms
    d = np.load(fn)
    res = []
    for tf44 in (False, True):
        p = {k: torch.tensor(d[v], device=DEV, requires_grad=True) for k, v in NAMES.items()}
        w = ops.skin_weights(p["_xyz"], torch.tensor(d["grid"], device=DEV), torch.tensor(d["grid_center"], device=DEV),
                             torch.tensor(d["grid_scale"], device=DEV))
        T = bone_transforms(torch.tensor(d["posed"], device=DEV), torch.tensor(d["rest"], device=DEV))
        pxyz, pcov, tf = ops.lbs_cov(p["_xyz"], p["_scaling"], p["_rotation"], w, T, tf44=tf44)
        cams = torch.zeros((1, 40), device=DEV)
        cams[0, 34:37] = torch.tensor(d["cam_center"], device=DEV).reshape(-1)
        col = ops.sh_colors(torch.cat([p["_features_dc"], p["_features_rest"]], 1), p["_xyz"], tf[0], cams)
        # a loss that also reads the transforms directly (their gradient then has two sources: the SH operator and this term)
        r4 = torch.linspace(-1.0, 1.0, 12, device=DEV).reshape(3, 4)
        tf_rows = tf[0][:, :3, :] if tf44 else tf[0].reshape(-1, 3, 4)
        loss = ((pxyz[0] * torch.tensor(d["r1"], device=DEV)).sum() + (pcov[0] * torch.tensor(d["r2"], device=DEV)).sum()
                + (col[0] * torch.tensor(d["r3"], device=DEV)).sum() + (tf_rows * r4).sum())
        loss.backward()
        res.append((pxyz.detach(), pcov.detach(), tf.detach(), col.detach(), {k: v.grad.clone() for k, v in p.items() if v.grad is not None}))
    a, b = res
    assert b[2].shape[-2:] == (4, 4)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[3], b[3])
    assert torch.equal(a[2][0].reshape(-1, 3, 4), b[2][0][:, :3, :])
    assert torch.equal(b[2][0][:, 3, :], torch.tensor([0.0, 0.0, 0.0, 1.0], device=DEV).expand(b[2].shape[1], 4))
    assert max_rel_err(b[2][0].cpu().numpy(), d["tf"]) < 2e-5
    assert set(a[4]) == set(b[4]) and {"_xyz", "_scaling", "_rotation", "_features_dc", "_features_rest"} <= set(a[4])
    for k in a[4]:
        assert torch.equal(a[4][k], b[4][k]), k


def test_object_chain_vs_reference_golden(golden_dir):
    from manus_amd import ops
    d = np.load(os.path.join(golden_dir, "lbs_sh_object_s3_n64.npz"))
    p = {k: torch.tensor(d[v], device=DEV, requires_grad=True) for k, v in NAMES.items()}
    pxyz, pcov, _ = ops.lbs_cov(p["_xyz"], p["_scaling"], p["_rotation"], None, None)
    cams = torch.zeros((1, 40), device=DEV)
    cams[0, 34:37] = torch.tensor(d["cam_center"], device=DEV).reshape(-1)
    col = ops.sh_colors(torch.cat([p["_features_dc"], p["_features_rest"]], 1), p["_xyz"], None, cams)
    assert max_rel_err(pxyz[0].detach().cpu().numpy(), d["posed_xyz"]) < 1e-6
    assert max_rel_err(pcov[0].detach().cpu().numpy(), d["posed_cov"]) < 2e-5
    assert max_rel_err(col[0].detach().cpu().numpy(), d["colors"]) < 2e-5
    loss = ((pxyz[0] * torch.tensor(d["r1"], device=DEV)).sum() + (pcov[0] * torch.tensor(d["r2"], device=DEV)).sum()
            + (col[0] * torch.tensor(d["r3"], device=DEV)).sum())
    loss.backward()
    for k in ("_xyz", "_scaling", "_rotation", "_features_dc", "_features_rest"):
        e = max_rel_err(p[k].grad.cpu().numpy(), d["grad" + k])
        assert e < 1e-4, (k, e)


def test_multi_pose_multi_view_vs_oracle():
    """P = V = 3 poses/views in one launch against the torch oracle looped over views."""
    from manus_amd import ops
    from manus_amd.synthetic import camera_table, make_scene
    sc = make_scene(n_gaussians=5000, kind="hand", seed=3, grid_res=24, n_cameras=3, width=64, height=48, device="cpu")
    par_cpu = {k: v.clone().requires_grad_(True) for k, v in sc["params"].items()}
    par = {k: v.clone().to(DEV).requires_grad_(True) for k, v in sc["params"].items()}
    grid = sc["grid"].to(DEV)
    ct = camera_table(sc["cameras"], DEV)
    w = ops.skin_weights(par["_xyz"], grid, sc["grid_center"].to(DEV), sc["grid_scale"].to(DEV))
    pxyz, pcov, tf = ops.lbs_cov(par["_xyz"], par["_scaling"], par["_rotation"], w, sc["transforms"].to(DEV))
    col = ops.sh_colors(torch.cat([par["_features_dc"], par["_features_rest"]], 1), par["_xyz"], tf, ct)
    g = torch.Generator().manual_seed(0)
    r1, r2, r3 = torch.randn((3, 5000, 3), generator=g), torch.randn((3, 5000, 6), generator=g), torch.randn((3, 5000, 3), generator=g)
    ((pxyz * r1.to(DEV)).sum() + (pcov * r2.to(DEV) * 1e3).sum() + (col * r3.to(DEV)).sum()).backward()
    loss = 0
    for v in range(3):
        cc = torch.tensor(np.asarray(sc["cameras"][v]["camera_center"], np.float32))
        o = tr.hand_forward(par_cpu, sc["grid"], sc["grid_center"], sc["grid_scale"], sc["posed"][v], sc["rest"], cc)
        assert max_rel_err(pxyz[v].detach().cpu().numpy(), o["posed_xyz"].detach().numpy()) < 2e-5
        assert max_rel_err(pcov[v].detach().cpu().numpy(), o["posed_cov"].detach().numpy()) < 2e-5
        assert max_rel_err(col[v].detach().cpu().numpy(), o["colors"].detach().numpy()) < 5e-5
        loss = loss + (o["posed_xyz"] * r1[v]).sum() + (o["posed_cov"] * r2[v] * 1e3).sum() + (o["colors"] * r3[v]).sum()
    loss.backward()
    for k in ("_xyz", "_scaling", "_rotation", "_features_dc", "_features_rest"):
        e = max_rel_err(par[k].grad.cpu().numpy(), par_cpu[k].grad.numpy())
        assert e < 1e-4, (k, e)


def test_points_outside_grid_give_nan_like_reference():
    """0/0 -> NaN without epsilon (src/utils/gaussian_utils.py:183)."""
    from manus_amd import ops
    grid = torch.rand((4, 5, 6, 21), device=DEV)
    xyz = torch.tensor([[10.0, 0, 0], [0.0, 0, 0]], device=DEV)
    w = ops.skin_weights(xyz, grid, torch.zeros(3, device=DEV), torch.ones(3, device=DEV))
    assert torch.isnan(w[0]).all() and torch.isfinite(w[1]).all()
    assert abs(float(w[1].sum()) - 1) < 1e-5


def test_project_points_vs_golden(golden_dir):
    from manus_amd import ops
    d = np.load(os.path.join(golden_dir, "fk_novel_pose.npz"))
    got = ops.project_points(torch.tensor(d["pp_points"], device=DEV), torch.tensor(d["pp_K"], device=DEV),
                             torch.tensor(d["pp_E"], device=DEV)).cpu().numpy()
    np.testing.assert_allclose(got, d["pp_out"], rtol=1e-5, atol=1e-3)


_SKIN_EDGE = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
from manus_amd import ops
torch.manual_seed(0)
D, H, W, B = 7, 9, 11, 21
grid = torch.rand((D, H, W, B), device="cuda:0") ** 3
c = torch.tensor([0.01, -0.02, 0.03], device="cuda:0"); s = torch.tensor([[0.5, 0.4, 0.3]], device="cuda:0")
out = {}
for N in (1, 2, 7, 9, 63, 255, 257, 1000):
    xyz = (torch.rand((N, 3), device="cuda:0") * 2.4 - 1.2) * s + c      # some points outside the grid
    out[N] = ops.skin_weights(xyz, grid, c, s).cpu()
torch.save(out, sys.argv[2])
"""


def test_skin_forward_lane_split_equals_one_thread_kernel(tmp_path):
    """k_skin_fwd24x8 (eight lanes per Gaussian, the default) against k_skin_fwd24 (MGR_SKIN_FWD=thread; the switch is read
    once per process, hence two processes): ragged sizes, points outside the grid (same NaN rows), last-bit differences only."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for mode in ("x8", "thread"):
        env = dict(os.environ)
        env.pop("MGR_SKIN_FWD", None)
        if mode == "thread":
            env["MGR_SKIN_FWD"] = "thread"
        out = str(tmp_path / (mode + ".pt"))
        subprocess.run([sys.executable, "-c", _SKIN_EDGE, root, out], check=True, env=env, timeout=300)
        res[mode] = torch.load(out)
    for n, a in res["x8"].items():
        b = res["thread"][n]
        assert a.shape == b.shape == (n, 21)
        fin = torch.isfinite(b)
        assert torch.equal(torch.isfinite(a), fin)
        assert fin.any() and float((a[fin] - b[fin]).abs().max()) < 1e-6
        assert float((a[fin.all(1)].sum(1) - 1).abs().max()) < 1e-5
