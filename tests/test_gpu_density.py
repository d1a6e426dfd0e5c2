"""GPU: the pruning path and the per-step control flow around the optimizer -- prune_points, dilate_mask /
get_points_outside_mask, the keypoint-distance test, on_after_backward -> density_update -> optimizer.step() --
against golden runs of the reference itself (tests/golden/make_golden.py: make_prune_golden, make_mask_golden,
make_flow_golden) and the oracle restatement at full size."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import torch_ref as tr

from util import max_rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling",
        "rotation": "_rotation"}


def _load_state(go, d, tag):
    """Put the reference's recorded state `tag` (leaves, moments, skin, statistics) into the optimizer."""
    for n in tr.LEAVES:
        go.p[ATTR[n]] = torch.tensor(d[f"{tag}_{n}"], device=DEV)
        z = np.zeros_like(d[f"{tag}_{n}"])
        go.m[ATTR[n]] = torch.tensor(d[f"{tag}_{n}_m"] if f"{tag}_{n}_m" in d else z, device=DEV)
        go.v[ATTR[n]] = torch.tensor(d[f"{tag}_{n}_v"] if f"{tag}_{n}_v" in d else z, device=DEV)
    go.skin_weights = torch.tensor(d[f"{tag}_skin"], device=DEV) if f"{tag}_skin" in d else None
    go.xyz_gradient_accum = torch.tensor(d[f"{tag}_accum"], device=DEV)
    go.denom = torch.tensor(d[f"{tag}_denom"], device=DEV)
    go.max_radii2D = torch.tensor(d[f"{tag}_maxrad"], device=DEV)


def _assert_state(go, d, tag, tol=2e-6):
    for n in tr.LEAVES:
        got = go.p[ATTR[n]].cpu().numpy()
        assert got.shape == d[f"{tag}_{n}"].shape, (tag, n, got.shape, d[f"{tag}_{n}"].shape)
        assert max_rel_err(got, d[f"{tag}_{n}"]) < tol, (tag, n)
        if f"{tag}_{n}_m" in d:
            assert max_rel_err(go.m[ATTR[n]].cpu().numpy(), d[f"{tag}_{n}_m"]) < tol, (tag, n, "m")
            assert max_rel_err(go.v[ATTR[n]].cpu().numpy(), d[f"{tag}_{n}_v"]) < tol, (tag, n, "v")
    if f"{tag}_skin" in d:
        np.testing.assert_array_equal(go.skin_weights.cpu().numpy(), d[f"{tag}_skin"])
    assert max_rel_err(go.xyz_gradient_accum.cpu().numpy(), d[f"{tag}_accum"]) < tol
    np.testing.assert_array_equal(go.denom.cpu().numpy(), d[f"{tag}_denom"])
    np.testing.assert_array_equal(go.max_radii2D.cpu().numpy(), d[f"{tag}_maxrad"])


def test_prune_points_matches_reference(golden_dir):
    from manus_amd.optim import GaussianOptimizer
    d = np.load(os.path.join(golden_dir, "prune_points.npz"))
    go = GaussianOptimizer({ATTR[n]: torch.tensor(d[f"pre_{n}"], device=DEV) for n in tr.LEAVES})
    _load_state(go, d, "pre")
    side = torch.arange(d["mask"].shape[0], dtype=torch.int32, device=DEV)
    M = go.prune_points(torch.tensor(d["mask"], device=DEV), extra={"idx": side})
    assert M == int((~d["mask"]).sum()) == go.N
    for n in tr.LEAVES:          # pure row copies: bit-exact
        np.testing.assert_array_equal(go.p[ATTR[n]].cpu().numpy(), d[f"post_{n}"])
        np.testing.assert_array_equal(go.m[ATTR[n]].cpu().numpy(), d[f"post_{n}_m"])
        np.testing.assert_array_equal(go.v[ATTR[n]].cpu().numpy(), d[f"post_{n}_v"])
    np.testing.assert_array_equal(go.skin_weights.cpu().numpy(), d["post_skin"])
    np.testing.assert_array_equal(go.xyz_gradient_accum.cpu().numpy(), d["post_accum"])
    np.testing.assert_array_equal(go.denom.cpu().numpy(), d["post_denom"])
    np.testing.assert_array_equal(go.max_radii2D.cpu().numpy(), d["post_maxrad"])
    np.testing.assert_array_equal(go.last_prune_extra["idx"].cpu().numpy(), np.nonzero(~d["mask"])[0])
    assert go.replaced == frozenset(ATTR)      # the next optimizer.step() has no gradient for any leaf


@pytest.mark.parametrize("n", [1, 1024, 1025, 300000])
def test_prune_points_sizes(n):
    """Scan-block seams, everything / nothing pruned, and the bench size, against the oracle restatement."""
    from manus_amd.optim import GaussianOptimizer
    g = torch.Generator().manual_seed(n)
    st = {"xyz": torch.randn(n, 3, generator=g), "f_dc": torch.randn(n, 1, 3, generator=g),
          "f_rest": torch.randn(n, 15, 3, generator=g), "opacity": torch.randn(n, 1, generator=g),
          "scaling": torch.randn(n, 3, generator=g), "rotation": torch.randn(n, 4, generator=g)}
    for mode in ("random", "none", "all"):
        go = GaussianOptimizer({ATTR[k]: st[k].to(DEV) for k in tr.LEAVES})
        go.m = {a: torch.randn(v.shape, generator=g).to(DEV) for a, v in go.p.items()}
        mask = torch.rand(n, generator=g) < 0.4 if mode == "random" else torch.full((n,), mode == "all")
        want = tr.prune_points({k: st[k] for k in tr.LEAVES}, mask)
        m_before = {a: v.cpu() for a, v in go.m.items()}
        assert go.prune_points(mask.to(DEV)) == int((~mask).sum())
        for k in tr.LEAVES:
            np.testing.assert_array_equal(go.p[ATTR[k]].cpu().numpy(), want[k].numpy())
            np.testing.assert_array_equal(go.m[ATTR[k]].cpu().numpy(), m_before[ATTR[k]][~mask].numpy())


def test_points_outside_mask_matches_reference(golden_dir):
    from manus_amd.density import dilate_mask, get_points_outside_mask
    d = np.load(os.path.join(golden_dir, "points_outside_mask.npz"))
    cam = types.SimpleNamespace(K=torch.tensor(d["K"], device=DEV), extr=torch.tensor(d["extr"], device=DEV))
    pts, mask = torch.tensor(d["points"], device=DEV), torch.tensor(d["mask"], device=DEV)
    np.testing.assert_array_equal(dilate_mask(mask[0, ..., 0]).cpu().numpy(), d["dilated"])
    f = lambda kp, dil: get_points_outside_mask(cam, pts, mask, None if kp is None else torch.tensor(d[kp], device=DEV),
                                                dilate=dil).cpu().numpy()
    for got, want in ((f(None, False), d["obj"]), (f("key_in", True), d["hand_in"]),
                      (f("key_in", False), d["hand_in_nodilate"]), (f("key_out", True), d["hand_out"])):
        assert got.shape == want.shape and got.dtype == np.bool_
        np.testing.assert_array_equal(got, want)


def test_dilate_and_mask_test_full_size():
    """1920x1080 mask, 300k points: dilation and lookup against the oracle restatement (bit-exact but for points whose
    projection lies within fp32 rounding of a pixel boundary)."""
    from manus_amd.density import dilate_mask, get_points_outside_mask, keypoint_far_mask
    g = torch.Generator().manual_seed(3)
    H, W, n = 1080, 1920, 300000
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    mask = ((((xx - 900.0) / 500.0) ** 2 + ((yy - 500.0) / 300.0) ** 2) < 1.0) & ((xx + yy) % 37 != 0)
    np.testing.assert_array_equal(dilate_mask(mask.to(DEV)).cpu().numpy(), tr.dilate_mask(mask).numpy())
    K = torch.tensor([[2666.67, 0.0, 959.5], [0.0, 2666.67, 539.5], [0.0, 0.0, 1.0]])
    E = torch.tensor([[1.0, 0.0, 0.0, 0.01], [0.0, 1.0, 0.0, -0.02], [0.0, 0.0, 1.0, 1.2]])
    pts = torch.randn(n, 3, generator=g) * torch.tensor([0.25, 0.15, 0.05])
    key = torch.randn(21, 3, generator=g) * 0.03
    want = tr.points_outside_mask(pts, K, E, mask[..., None].to(torch.uint8), key, dilate=True)
    got = get_points_outside_mask(dict(K=K, extr=E), pts.to(DEV), mask[None, ..., None].to(DEV), key.to(DEV), dilate=True)
    assert want.any() and not want.all()
    bad = (got.cpu() != want).reshape(-1)
    if bad.any():        # only allowed at pixel-boundary ties of the projection
        uv = tr.project_points(pts[None].double(), K.double(), E.double())[0][bad]
        assert (torch.minimum((uv - uv.round()).abs().min(dim=1).values, torch.tensor(1.0)) < 1e-3).all()
    assert bad.float().mean() < 1e-4
    far = keypoint_far_mask(pts.to(DEV), key.to(DEV), 0.2).cpu()
    mean_d = torch.cdist(pts.double(), key.double()).mean(1)
    sure = (mean_d - 0.2).abs() > 1e-6
    assert torch.equal(far[sure], (mean_d > 0.2)[sure]) and far.any() and not far.all()


@pytest.mark.parametrize("kind", ["hand", "object"])
def test_step_flow_matches_reference(golden_dir, kind):
    """Every step of the reference's recorded run (mask prune -> keypoint-distance prune -> opacity reset -> densify with
    and without the size threshold, each followed by the optimizer step that skips replaced leaves), re-run here from the
    reference's own pre-step state with the recorded inputs; the post-step state must equal the reference's."""
    from manus_amd.density import DensityController
    from manus_amd.optim import GaussianOptimizer
    d = np.load(os.path.join(golden_dir, f"flow_{kind}.npz"))
    opts = dict(remove_seg_end=int(d["remove_seg_end"]), densify_from_step=100, densification_interval=100,
                opacity_reset_interval=int(d["opacity_reset_interval"]), percent_dense=float(d["percent_dense"]))
    go = GaussianOptimizer({ATTR[n]: torch.tensor(d[f"init_{n}"], device=DEV) for n in tr.LEAVES}, opts=opts,
                           skin_weights=torch.tensor(d["init_skin"], device=DEV) if "init_skin" in d else None)
    dc = DensityController(go, float(d["extent"]), kind=kind, bg_white=True)
    cam = dict(K=torch.tensor(d["K"], device=DEV), extr=torch.tensor(d["extr"], device=DEV))
    keyp = torch.tensor(np.concatenate([d["heads"][:1], d["tails"]], 0), device=DEV)
    mask = torch.tensor(d["mask"], device=DEV)
    prev, seen = "init", set()
    for k, gs in enumerate(int(x) for x in d["steps"]):
        _load_state(go, d, prev)
        dc.on_train_epoch_start() if k == 0 else None
        n0 = go.N
        radii = torch.tensor(d[f"s{k}_radii"], device=DEV)
        vg = torch.tensor(d[f"s{k}_vsp_grad"], device=DEV)
        vis = radii > 0
        stats = dict(grad2d=vg[:, :2].norm(dim=-1) * vis, vis=vis.float(), radii=radii)
        views = [dict(camera=cam, mask=mask, posed_xyz=go.p["_xyz"], keypoints=keyp if kind == "hand" else None)]
        noise = torch.tensor(d[f"s{k}_noise"], device=DEV) if d[f"s{k}_noise"].size else None
        changed = dc.after_backward(gs, stats, views, noise=noise)
        replaced = go.replaced
        go.update_learning_rate(gs)
        if go.N == n0:      # the recorded gradients belong to the pre-step tensors
            go.step({ATTR[n]: torch.tensor(d[f"s{k}_grad_{n}"], device=DEV) for n in tr.LEAVES})
        else:
            assert replaced == frozenset(ATTR)
            go.step({})      # every group is skipped
        assert go.N == int(d[f"s{k}_n_after"]), (k, gs)
        _assert_state(go, d, f"s{k}")
        seen.add("prune" if (changed and go.N < n0) else "grow" if go.N > n0 else "reset" if changed else "plain")
        assert dc.pts_mask.shape[0] == go.N and not dc.pts_mask.any()
        prev = f"s{k}"
    assert {"prune", "grow", "plain"} <= seen
