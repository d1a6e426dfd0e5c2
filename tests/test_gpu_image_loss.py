"""GPU: fused L1 + SSIM image loss (mgr_image_loss) against the reference's golden vectors, the
oracle restatement, and size-independent properties at the bench resolution."""
import os

import numpy as np
import pytest
import torch

from oracle import torch_ref as tr

from util import max_rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_reference_shaped_losses_match_golden(golden_dir):
    """manus_amd.losses.{l1_loss, ssim} called like base.py:329-347 (pred HWC, gt (1,H,W,3))."""
    from manus_amd import losses
    d = np.load(os.path.join(golden_dir, "image_loss.npz"))
    for k in range(3):
        gt = torch.tensor(d[f"gt{k}"], device=DEV)
        pred = torch.tensor(d[f"pred{k}"], device=DEV, requires_grad=True)
        l1 = losses.l1_loss(pred, gt)
        (g,) = torch.autograd.grad(l1, pred)
        assert abs(l1.item() - float(d[f"l1_{k}"])) < 1e-6
        np.testing.assert_allclose(g.cpu().numpy(), d[f"g_l1_{k}"], rtol=0, atol=1e-9)
        # the form loss_func itself uses: l1_loss(pred, gt, mean=False) followed by torch.mean (base.py:329-331)
        pred = torch.tensor(d[f"pred{k}"], device=DEV, requires_grad=True)
        for reduce in (torch.mean, lambda m: m.mean()):
            l1m = reduce(losses.l1_loss(pred, gt, mean=False))
            (g,) = torch.autograd.grad(l1m, pred)
            assert abs(l1m.item() - float(d[f"l1_{k}"])) < 1e-6
            np.testing.assert_allclose(g.cpu().numpy(), d[f"g_l1_{k}"], rtol=0, atol=1e-9)
        assert abs(float((losses.l1_loss(pred, gt, mean=False) * 2.0).mean().detach()) - 2 * float(d[f"l1_{k}"])) < 1e-6  # any other use
        pred = torch.tensor(d[f"pred{k}"], device=DEV, requires_grad=True)
        ss = losses.ssim(pred, gt)
        (g,) = torch.autograd.grad(ss, pred)
        assert abs(ss.item() - float(d[f"ssim_{k}"])) < 5e-6          # fp32, tolerance 5e-6 absolute on a value in [-1,1]
        assert max_rel_err(g.cpu().numpy(), d[f"g_ssim_{k}"]) < 1e-4  # gradient max-rel-err bar of the path
        pred = torch.tensor(d[f"pred{k}"], device=DEV, requires_grad=True)
        full = losses.rgb_ssim_loss(pred, gt, 0.8, 0.2)
        (g,) = torch.autograd.grad(full, pred)
        ref = 0.8 * float(d[f"l1_{k}"]) + 0.2 * (1.0 - float(d[f"ssim_{k}"]))
        assert abs(full.item() - ref) < 5e-6
        assert max_rel_err(g.cpu().numpy(), 0.8 * d[f"g_l1_{k}"] - 0.2 * d[f"g_ssim_{k}"]) < 1e-4


@pytest.mark.parametrize("V,H,W", [(1, 1, 1), (2, 3, 5), (1, 9, 256), (2, 4, 257), (1, 6, 700)])
def test_kernel_matches_oracle(V, H, W):
    """Ragged widths (workgroup seams at 256), tiny images, several views."""
    from manus_amd import ops
    g = torch.Generator().manual_seed(V * 1000 + W)
    gt = torch.rand((V, 3, H, W), generator=g)
    pred = (gt + 0.1 * torch.randn((V, 3, H, W), generator=g)).clamp(0, 1.1)
    sums, grad = ops.image_loss_grad(pred.to(DEV), gt.to(DEV), 0.8, 0.2, 0.37, 0.125)
    l1 = s_sum = 0.0
    gref = torch.zeros_like(pred)
    for v in range(V):
        p = pred[v].permute(1, 2, 0).clone().requires_grad_(True)
        t = gt[v].permute(1, 2, 0)
        a = (p - t).abs().sum()
        s = tr.ssim_hwc(p, t) * (3 * H * W)
        (gg,) = torch.autograd.grad(0.37 * (0.8 * a - 0.2 * s), p)
        gref[v] = gg.permute(2, 0, 1)
        l1 += a.item(); s_sum += s.item()
    got = sums.cpu().numpy()
    assert abs(got[0] - l1) <= 2e-6 * max(1.0, abs(l1)) * 3
    assert abs(got[1] - s_sum) <= 1e-5 * max(1.0, abs(s_sum))
    want = 0.37 * (0.8 * l1 - 0.2 * s_sum) + 0.125                  # the loss value the gradient belongs to
    assert abs(got[2] - want) <= 1e-5 * max(1.0, abs(want))
    assert max_rel_err(grad.cpu().numpy(), gref.numpy()) < 1e-4


def test_full_size_properties():
    """1920x1080, 8 views: (i) identical images: L1 = 0, SSIM map = 1 everywhere, gradient 0;
    (ii) the result does not depend on how views are batched; (iii) run-to-run bitwise equal
    (no float atomics: the loss value is reproducible too)."""
    from manus_amd import ops
    V, H, W = 8, 1080, 1920
    g = torch.Generator(device=DEV).manual_seed(5)
    gt = torch.rand((V, 3, H, W), device=DEV, generator=g)
    sums, grad = ops.image_loss_grad(gt, gt, 0.8, 0.2, 1.0)
    s = sums.cpu().numpy()
    assert s[0] == 0.0 and abs(s[1] / (V * 3 * H * W) - 1.0) < 1e-6
    assert float(grad.abs().max()) < 1e-5
    pred = (gt + 0.05 * torch.randn(gt.shape, device=DEV, generator=g)).clamp(0, 1)
    s_all, g_all = ops.image_loss_grad(pred, gt, 0.8, 0.2, 1.0)
    s_b, g_b = ops.image_loss_grad(pred, gt, 0.8, 0.2, 1.0)
    assert torch.equal(g_all, g_b) and torch.equal(s_all, s_b)
    s_one, g_one = ops.image_loss_grad(pred[3:4], gt[3:4], 0.8, 0.2, 1.0)
    assert torch.equal(g_all[3:4], g_one)
    acc = np.zeros(2)
    for v in range(V):
        acc += ops.image_loss_grad(pred[v:v + 1], gt[v:v + 1], 0.8, 0.2, 1.0)[0][:2].double().cpu().numpy()
    np.testing.assert_allclose(s_all[:2].double().cpu().numpy(), acc, rtol=1e-6)


def test_training_loss_matches_reference_loss_func(golden_dir):
    """The three image/shape terms of the reference's loss_func (base.py:323-365; OBJ_GAUSSIAN.yaml:22-23) from the
    fused kernels, value and gradients w.r.t. the render and the log-scales."""
    from manus_amd import losses
    d = np.load(os.path.join(golden_dir, "loss_func.npz"))
    pred = torch.tensor(d["pred"], device=DEV, requires_grad=True)
    ls = torch.tensor(d["log_scale"], device=DEV, requires_grad=True)
    gt = torch.tensor(d["gt"], device=DEV)
    loss = losses.rgb_ssim_loss(pred, gt, 0.8, 0.2) + 0.1 * losses.isotropic_reg(ls, 0.4)
    assert abs(loss.item() - float(d["loss"])) < 5e-6
    assert abs(losses.isotropic_reg(ls, 0.4).item() - float(d["iso"])) < 2e-6
    gp, gs = torch.autograd.grad(loss, [pred, ls])
    assert max_rel_err(gp.cpu().numpy(), d["g_pred"]) < 1e-4
    assert max_rel_err(gs.cpu().numpy(), d["g_log_scale"]) < 1e-4
    # accumulate form used by the engine: gradient added to an existing tensor, value from the fold kernel
    from manus_amd import ops
    base = torch.randn(ls.shape, device=DEV)
    lo1, g1 = ops.isotropic_reg_grad(ls.detach(), 0.4, 0.1)
    lo2, g2 = ops.isotropic_reg_grad(ls.detach(), 0.4, 0.1, grad_out=base.clone())
    assert abs(lo1.item() - 0.1 * float(d["iso"])) < 2e-6 and torch.equal(lo1, lo2)
    assert torch.allclose(g2, base + g1, rtol=0, atol=1e-7)


def test_identical_background_fast_path_matches_oracle():
    """Capture-like frame: rendered and target images share an exact background and differ in a patch.  Workgroups
    whose whole span is identical skip the filters; the result must equal the oracle everywhere, including next
    to the patch (spans that see a difference within +-10 px take the full path)."""
    from manus_amd import ops
    g = torch.Generator().manual_seed(9)
    V, H, W = 2, 7, 1300
    gt = torch.ones((V, 3, H, W))
    pred = torch.ones((V, 3, H, W))
    gt[:, :, 2:5, 600:700] = torch.rand((V, 3, 3, 100), generator=g)
    pred[:, :, 2:6, 590:705] = torch.rand((V, 3, 4, 115), generator=g)
    pred[1, :, 0, 0:3] = 0.5                                   # a difference at the image border
    sums, grad = ops.image_loss_grad(pred.to(DEV), gt.to(DEV), 0.8, 0.2, 1.0)
    l1 = s_sum = 0.0
    gref = torch.zeros_like(pred)
    for v in range(V):
        p = pred[v].permute(1, 2, 0).clone().requires_grad_(True)
        t = gt[v].permute(1, 2, 0)
        a = (p - t).abs().sum()
        s = tr.ssim_hwc(p, t) * (3 * H * W)
        (gg,) = torch.autograd.grad(0.8 * a - 0.2 * s, p)
        gref[v] = gg.permute(2, 0, 1)
        l1 += a.item(); s_sum += s.item()
    got = sums.cpu().numpy()
    assert abs(got[0] - l1) <= 1e-5 * max(1.0, abs(l1))
    assert abs(got[1] - s_sum) <= 1e-5 * abs(s_sum)
    assert max_rel_err(grad.cpu().numpy(), gref.numpy()) < 1e-4
    assert float(grad[0, :, 0, :500].abs().max()) == 0.0       # far from any difference: exactly zero


def test_tile_aware_loss_equals_the_full_comparison():
    """mgr_image_loss_tiles (spans under empty tiles settled from the target alone, their gradient left unwritten)
    against mgr_image_loss on a rendered scene: same sums; the same gradient wherever a tile holds a Gaussian or the
    target differs from the background; and the training step's leaf gradients do not change (bitwise)."""
    from manus_amd import ops, rasterizer as rz
    from manus_amd.engine import HipViewCompute
    from manus_amd.synthetic import camera_table, make_scene
    V, W, H, n = 3, 500, 330, 4000      # W not a multiple of 16 or 246, H not a multiple of 16
    sc = make_scene(n_gaussians=n, kind="hand", seed=9, grid_res=24, n_cameras=V, width=W, height=H, cam_radius=0.9,
                    sigma_range=(2e-3, 8e-3), device=DEV)
    ct = camera_table(sc["cameras"], DEV)
    g = torch.Generator(device="cpu").manual_seed(3)
    tgt_scene = dict(sc)
    tgt_scene["params"] = {k: (v + 0.02 * v.abs().mean() * torch.randn(v.shape, generator=g).to(DEV)) for k, v in sc["params"].items()}
    with torch.no_grad():
        targets = HipViewCompute(tgt_scene, torch.zeros((V, 3, H, W), device=DEV), ct).forward_views_fused(list(range(V)))[0].contiguous()
    targets[1, :, 5:9, 440:470] = 0.25          # target foreground where nothing is rendered (empty tiles)
    rz.set_sync_policy(True)
    res = {}
    for sparse in (False, True):
        hc = HipViewCompute(sc, targets, ct, loss="l1+ssim", sparse_loss=sparse)
        out = hc._step_direct(list(range(V)), 1.0 / V)
        res[sparse] = ({k: v.clone() for k, v in out["grads"].items()}, out["loss"].clone(), out["grad2d"].clone())
    for k in res[False][0]:
        assert torch.equal(res[False][0][k], res[True][0][k]), k
    assert torch.equal(res[False][2], res[True][2])
    assert abs(float(res[False][1]) - float(res[True][1])) <= 2e-7 * abs(float(res[False][1]))
    # the operator itself, on the last forward's image and tile lists
    ws = rz.context().last_ws
    img = hc.last_image
    ts_ptr = hc._tile_start_ptr(ws, V, n, W, H)
    s0, g0 = ops.image_loss_grad(img, targets, 0.8, 0.2, 0.37, 0.11)
    g1 = torch.full_like(img, float("nan"))
    # (ops allocates its own gradient tensor: call the ABI with a NaN-filled one to see what is written)
    import ctypes
    from manus_amd._lib import check, lib, ptr, stream
    sums = torch.empty(3, device=DEV)
    nbytes = int(lib().mgr_image_loss_workspace_bytes(V, H, W))
    wsb = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    check(lib().mgr_image_loss_tiles(V, H, W, ptr(img), ptr(targets), ptr(sc["bg"]), ctypes.c_void_p(ts_ptr), 0.8, 0.2, 0.37,
                                     0.11, ptr(g1), ptr(sums), ptr(wsb), nbytes, stream()), "mgr_image_loss_tiles")
    assert torch.allclose(sums, s0, rtol=2e-7, atol=0)
    written = ~torch.isnan(g1)
    # (a span that is identical under NON-empty tiles is zero-filled by the full comparison and computed here: fp32
    # noise of the order 1e-9 instead of exact zeros, like the reference's own result there)
    assert float((g1[written] - g0[written]).abs().max()) < 1e-6
    both = written & (g0 != 0)
    assert torch.equal(g1[both], g0[both])
    assert not g0[~written].any()                       # what was left unwritten is zero in the full comparison
    frac = float(written.float().mean())
    assert 0.02 < frac < 0.9, frac                      # most of the frame is background and was skipped
    assert written[1, :, 5:9, 440:470].all()            # the target-only foreground was computed (it counts in the loss)


def test_span_list_from_target_maps_equals_the_list_from_the_targets():
    """mgr_image_loss_target_map + mgr_image_loss_tiles_list_mapped (the target's column masks computed once per view) against
    mgr_image_loss_tiles_list (the masks recomputed from the full targets every call): the same spans listed, the same
    sums for the unlisted ones, and a training step that is bit for bit the same -- loss included."""
    import ctypes
    from manus_amd import rasterizer as rz
    from manus_amd._lib import check, lib, ptr, stream
    from manus_amd.engine import HipViewCompute
    from manus_amd.synthetic import camera_table, make_scene
    V, W, H, n = 3, 500, 331, 4000      # W not a multiple of 16, 32 or 246; H odd
    sc = make_scene(n_gaussians=n, kind="hand", seed=9, grid_res=24, n_cameras=V, width=W, height=H, cam_radius=0.9,
                    sigma_range=(2e-3, 8e-3), device=DEV)
    ct = camera_table(sc["cameras"], DEV)
    g = torch.Generator(device="cpu").manual_seed(3)
    tgt_scene = dict(sc)
    tgt_scene["params"] = {k: (v + 0.02 * v.abs().mean() * torch.randn(v.shape, generator=g).to(DEV)) for k, v in sc["params"].items()}
    with torch.no_grad():
        targets = HipViewCompute(tgt_scene, torch.zeros((V, 3, H, W), device=DEV), ct).forward_views_fused(list(range(V)))[0].contiguous()
    targets[1, :, 5:9, 440:470] = 0.25          # target foreground where nothing is rendered (empty tiles)
    targets[2, 1, 330, 499] = 0.5               # ... and in the last column of the last (unpaired) row
    rz.set_sync_policy(True)
    res = {}
    for mapped in (False, True):
        hc = HipViewCompute(sc, targets, ct, loss="l1+ssim")
        hc.target_map = mapped
        out = hc._step_direct(list(range(V)), 1.0 / V)
        res[mapped] = ({k: v.clone() for k, v in out["grads"].items()}, out["loss"].clone(), hc)
    for k in res[False][0]:
        assert torch.equal(res[False][0][k], res[True][0][k]), k
    assert torch.equal(res[False][1], res[True][1])
    # the two list builders on the last forward's tile offsets
    hc = res[True][2]
    ws = rz.context().last_ws
    ts_ptr = ctypes.c_void_p(hc._tile_start_ptr(ws, V, n, W, H))
    nbytes = int(lib().mgr_image_loss_workspace_bytes(V, H, W))
    nb = V * ((H + 1) // 2) * ((W + 245) // 246)
    lists = []
    words = int(lib().mgr_image_loss_target_map_words(V, H, W))
    assert words == V * ((H + 1) // 2) * ((W + 31) // 32)
    tmap = torch.empty(words, dtype=torch.int32, device=DEV)
    check(lib().mgr_image_loss_target_map(V, H, W, ptr(targets), ptr(sc["bg"]), ptr(tmap), stream()), "mgr_image_loss_target_map")
    assert torch.equal(tmap.view(V, -1), hc._target_map(list(range(V)), hc._select(list(range(V))), sc["bg"]).view(V, -1))
    for use_map in (False, True):
        wsb = torch.full((nbytes,), 0xAB, dtype=torch.uint8, device=DEV)
        if use_map:
            check(lib().mgr_image_loss_tiles_list_mapped(V, H, W, ptr(tmap), ptr(sc["bg"]), ts_ptr, ptr(wsb), nbytes, 0, stream()), "list_mapped")
        else:
            check(lib().mgr_image_loss_tiles_list(V, H, W, ptr(targets), ptr(sc["bg"]), ts_ptr, ptr(wsb), nbytes, stream()), "list")
        count = int(wsb[nb * 12 + 64: nb * 12 + 68].view(torch.int32).item())
        work = wsb[nb * 8: nb * 8 + 4 * count].view(torch.int32).sort().values.cpu()
        partial = wsb[: nb * 8].view(torch.float32).view(nb, 2).cpu()
        lists.append((count, work, partial))
    assert lists[0][0] == lists[1][0] and 0 < lists[0][0] < nb
    assert torch.equal(lists[0][1], lists[1][1])
    listed = torch.zeros(nb, dtype=torch.bool)
    listed[lists[0][1].long()] = True
    assert torch.equal(lists[0][2][~listed], lists[1][2][~listed])     # (the listed spans' sums are written by the finish pass)
    # the map marks exactly the columns where the target differs from the background in some channel of the row pair
    diff = (targets != sc["bg"].view(1, 3, 1, 1)).any(1)                                   # (V, H, W)
    diff = torch.nn.functional.pad(diff, (0, (-W) % 32, 0, H % 2))
    diff = (diff[:, 0::2] | diff[:, 1::2]).view(V, (H + 1) // 2, -1, 32)
    want = (diff.long() << torch.arange(32, device=DEV)).sum(-1)
    assert torch.equal(want, tmap.view(V, (H + 1) // 2, -1).long() & 0xFFFFFFFF)
