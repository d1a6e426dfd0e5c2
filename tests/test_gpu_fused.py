"""GPU: the fused articulated path (mgr_views_forward/backward) against the modular operators
(ops.lbs_cov + ops.sh_colors + rasterizer.rasterize_views) that are themselves checked against
the oracles.  Same math (csrc/instance_math.h) in different kernels: values agree to fp32
roundoff; the only larger deviations are isolated alpha-threshold flips (see test_gpu_misc)."""
import numpy as np
import pytest
import torch

from util import max_rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _scene(kind, n=4000, views=3):
    from manus_amd.synthetic import camera_table, make_scene
    sc = make_scene(n_gaussians=n, kind=kind, seed=6, grid_res=24, n_cameras=views, width=96, height=64,
                    cam_radius=0.5, sigma_range=(2e-3, 8e-3), device=DEV)
    return sc, camera_table(sc["cameras"], DEV)


@pytest.mark.parametrize("kind", ["hand", "object"])
def test_fused_equals_modular(kind):
    from manus_amd.engine import HipViewCompute
    sc, ct = _scene(kind)
    tg = torch.rand((3, 3, 64, 96), device=DEV)
    mod = HipViewCompute(sc, tg, ct, fused=False)
    fus = HipViewCompute(sc, tg, ct, fused=True)
    with torch.no_grad():
        im_m, rad_m, _ = mod.forward_views([0, 1, 2])
        im_f, rad_f = fus.forward_views_fused([0, 1, 2])
    assert torch.equal(rad_m, rad_f)
    d = (im_m - im_f).abs()
    assert float(d.max()) < 5e-3 and float(d.mean()) < 2e-6
    om, of = mod([0, 1, 2], 1.0 / 3), fus([0, 1, 2], 1.0 / 3)
    assert abs(float(om["loss"]) - float(of["loss"])) < 1e-6
    for k in om["grads"]:
        a, b = of["grads"][k].cpu().numpy().astype(np.float64), om["grads"][k].cpu().numpy().astype(np.float64)
        assert a.shape == b.shape, k
        assert max_rel_err(a, b) < 5e-3, (k, max_rel_err(a, b))
        rows = np.abs(a - b).reshape(a.shape[0], -1).max(1) > 2e-5 * np.abs(b).max()
        assert rows.mean() < 0.03, (k, rows.sum())
    assert torch.equal(of["vis"], om["vis"])
    assert torch.equal(of["radii"].to(torch.int32), om["radii"].to(torch.int32))
    assert max_rel_err(of["grad2d"].cpu().numpy(), om["grad2d"].cpu().numpy()) < 5e-3


def test_fused_run_to_run_determinism():
    from manus_amd.engine import HipViewCompute
    sc, ct = _scene("hand", n=6000, views=2)
    tg = torch.rand((2, 3, 64, 96), device=DEV)
    hc = HipViewCompute(sc, tg, ct, fused=True)
    a = hc([0, 1], 0.5)
    a = {k: ({n: g.clone() for n, g in v.items()} if isinstance(v, dict) else v.clone()) for k, v in a.items()}
    b = hc([0, 1], 0.5)
    for k in a["grads"]:
        assert torch.equal(a["grads"][k], b["grads"][k]), k
    assert torch.equal(a["grad2d"], b["grad2d"])
    assert abs(float(a["loss"]) - float(b["loss"])) < 1e-6  # the loss scalar is summed with float atomics


@pytest.mark.parametrize("views", [1, 5, 8, 11, 16])
def test_fused_equals_modular_view_counts(views):
    """Lane-group sizes 1, 8 (5 and 8 views) and more than 8 views (two view groups: the second accumulates; 16 = two full groups, the XCD-aware block orders with more than one view per XCD)."""
    from manus_amd.engine import HipViewCompute
    sc, ct = _scene("hand", n=3000, views=views)
    tg = torch.rand((views, 3, 64, 96), device=DEV)
    ids = list(range(views))
    om = HipViewCompute(sc, tg, ct, fused=False)(ids, 1.0 / views)
    of = HipViewCompute(sc, tg, ct, fused=True)(ids, 1.0 / views)
    assert abs(float(om["loss"]) - float(of["loss"])) < 1e-6
    for k in om["grads"]:
        a, b = of["grads"][k].cpu().numpy().astype(np.float64), om["grads"][k].cpu().numpy().astype(np.float64)
        assert max_rel_err(a, b) < 5e-3, (k, max_rel_err(a, b))
        rows = np.abs(a - b).reshape(a.shape[0], -1).max(1) > 2e-5 * np.abs(b).max()
        assert rows.mean() < 0.03, (k, rows.sum())
    assert torch.equal(of["vis"], om["vis"])
    assert torch.equal(of["radii"].to(torch.int32), om["radii"].to(torch.int32))
    assert max_rel_err(of["grad2d"].cpu().numpy(), om["grad2d"].cpu().numpy()) < 5e-3


@pytest.mark.parametrize("kind,views,n", [("hand", 8, 6000), ("hand", 5, 6000), ("composite", 7, 6000), ("hand", 11, 6000),
                                            ("hand", 8, 37), ("hand", 6, 1), ("composite", 8, 263),
                                            ("hand", 4, 6000), ("hand", 3, 6000), ("hand", 2, 6000), ("hand", 1, 6000), ("composite", 4, 263),
                                            ("object", 1, 6000), ("hand", 2, 1)])
def test_run_lists_equal_one_lane_per_view(kind, views, n):
    """mgr_views_backward_run_lists: an active Gaussian on 8 / 4 / 2 lanes by the number of its views with records (and the
    gather over a compacted list of the instances with records; one view: the gather alone) against one lane per view.  Same per-view values, summed over a tree of fewer terms: leaf gradients agree to rounding (1e-5 of
    the largest entry: the views' terms of a Gaussian can be larger than their sum), rows no view contributes to stay exactly zero, statistics and the active list are the same set;
    each setting is bit-reproducible.  (11 views: the second view group accumulates.)"""
    from manus_amd._lib import lib
    from manus_amd.engine import HipViewCompute
    sc, ct = _scene(kind, n=n, views=views)      # (37 / 1 / 263 Gaussians: partial lane groups, partial workgroups, a composite boundary inside one)
    tg = torch.rand((views, 3, 64, 96), device=DEV, generator=torch.Generator(device=DEV).manual_seed(views))
    ids = list(range(views))
    hc = HipViewCompute(sc, tg, ct, fused=True)
    clone = lambda o: {k: ({q: t.clone() for q, t in v.items()} if isinstance(v, dict) else v.clone()) for k, v in o.items()}
    prev = lib().mgr_views_backward_run_lists(1)
    try:
        on1 = clone(hc(ids, 1.0 / views))
        on2 = clone(hc(ids, 1.0 / views))
        assert lib().mgr_views_backward_run_lists(0) == 1
        off = clone(hc(ids, 1.0 / views))
    finally:
        lib().mgr_views_backward_run_lists(prev)
    for k in on1["grads"]:
        assert torch.equal(on1["grads"][k], on2["grads"][k]), k
        a, b = on1["grads"][k].double(), off["grads"][k].double()
        assert float((a - b).abs().max()) <= 1e-5 * max(float(b.abs().max()), 1e-30), (k, float((a - b).abs().max()), float(b.abs().max()))
        assert torch.equal(a.reshape(a.shape[0], -1).abs().sum(1) == 0, b.reshape(b.shape[0], -1).abs().sum(1) == 0), k
    assert torch.equal(on1["vis"], off["vis"]) and torch.equal(on1["radii"], off["radii"])
    assert float((on1["grad2d"].double() - off["grad2d"].double()).abs().max()) <= 1e-5 * float(off["grad2d"].abs().max())
    assert abs(float(on1["loss"]) - float(off["loss"])) < 1e-6   # (the loss scalar is summed with float atomics)


@pytest.mark.parametrize("kind,views", [("hand", 8), ("hand", 4), ("composite", 7), ("hand", 2)])
def test_persistent_gradient_buffers_equal_fresh_ones(kind, views):
    """HipViewCompute(persistent_grads=True): gradients land in buffers the object keeps, and the backward zeroes only the
    rows the previous step wrote and this one does not (mgr_views_backward, debug bit 512), and the image into a buffer whose
    empty tiles are written once (mgr_views_forward, debug bit 1024).  Over steps between which the
    model moves (Gaussians come into view and leave it) every step's gradients, statistics and loss are bit for bit those
    of a compute object that gets fresh, fully zeroed buffers; the same after the rows were disturbed by another
    compute object's backward on the same workspace, and with the run lists switched off in between."""
    from manus_amd._lib import lib
    from manus_amd.engine import HipViewCompute
    sc, ct = _scene(kind, n=5000, views=views)
    tg = torch.rand((views, 3, 64, 96), device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    ids = list(range(views))
    a = HipViewCompute(sc, tg, ct, fused=True, depth_cut=False, persistent_grads=False)
    b = HipViewCompute(sc, tg, ct, fused=True, depth_cut=False, persistent_grads=True)
    c = HipViewCompute(sc, tg, ct, fused=True, depth_cut=False, persistent_grads=True)   # shares the pooled workspace with b
    gen = torch.Generator(device=DEV).manual_seed(7)
    n_rows = []
    for it in range(7):
        with torch.no_grad():   # move the model: positions jitter, a tenth of the Gaussians turn transparent / opaque
            noise = 0.004 * torch.randn(a.params["_xyz"].shape, device=DEV, generator=gen)
            flip = torch.rand(a.params["_opacity"].shape, device=DEV, generator=gen) < 0.1
            for comp in (a, b, c):
                comp.params["_xyz"].add_(noise)
                comp.params["_opacity"][flip] = -comp.params["_opacity"][flip]
                comp.mark_params_changed()
        if it == 3:
            c(ids, 1.0 / views)                                  # another object's backward on the same workspace
        if it == 5:
            prev = lib().mgr_views_backward_run_lists(0)         # a backward through the other lane layout in between
            try:
                b(ids, 1.0 / views)
            finally:
                lib().mgr_views_backward_run_lists(prev)
        oa, ob = a(ids, 1.0 / views), b(ids, 1.0 / views)
        for k in oa["grads"]:
            assert torch.equal(oa["grads"][k], ob["grads"][k]), (it, k)
        assert torch.equal(oa["grad2d"], ob["grad2d"]) and torch.equal(oa["vis"], ob["vis"]) and torch.equal(oa["radii"], ob["radii"])
        assert torch.equal(a.last_image, b.last_image), it     # (the image is kept too: tiles that stay empty are not written again)
        n_rows.append(int((oa["grads"]["_xyz"].abs().sum(1) != 0).sum()))
    assert len(set(n_rows)) > 1          # the set of rows with a gradient did change from step to step


@pytest.mark.parametrize("kind,views", [("hand", 8), ("composite", 3)])
def test_kept_buffers_survive_a_caller_writing_into_them(kind, views):
    """The kept gradient / image buffers are handed out every step; the row- and tile-selective fills rest on their content
    being what the previous step left.  A caller that writes into them (in place, through torch: also via views) is noticed
    by the tensors' version counters and answered with a full fill: the next step is bit for bit the step of an object
    with fresh buffers -- rows that get no gradient are zero again, empty tiles hold the background again."""
    from manus_amd import rasterizer
    from manus_amd.engine import HipViewCompute
    from util import keep
    sc, ct = _scene(kind, n=5000, views=views)
    tg = torch.rand((views, 3, 64, 96), device=DEV, generator=torch.Generator(device=DEV).manual_seed(2))
    ids = list(range(views))
    fresh = HipViewCompute(sc, tg, ct, fused=True, loss="l1+ssim", persistent_grads=False)
    want = keep(fresh(ids, 1.0 / views))
    want_img = fresh.last_image.clone()
    kept = HipViewCompute(sc, tg, ct, fused=True, loss="l1+ssim")        # the default: kept buffers
    assert kept.persistent_grads and not kept.depth_cut
    for mode in ("sync", "fenced"):
        rasterizer.set_sync_policy(mode == "sync")
        o = kept(ids, 1.0 / views)
        o = kept(ids, 1.0 / views)                 # second step: the selective fills are in use
        rasterizer.check_overflow()
        assert kept._pg_ws is not None and kept._pimg_ws is not None
        for k in want["grads"]:
            assert torch.equal(o["grads"][k], want["grads"][k]), (mode, k)
        assert torch.equal(kept.last_image, want_img)
        # the caller scribbles over what it was handed: whole tensors, a view of one, the image
        o["grads"]["_xyz"].add_(1.0)
        o["grads"]["_features_rest"][:, 3].fill_(7.0)
        o["grad2d"].fill_(5.0)
        o["vis"].zero_()
        kept.last_image.mul_(0.25)
        o2 = kept(ids, 1.0 / views)
        rasterizer.check_overflow()
        for k in want["grads"]:
            assert torch.equal(o2["grads"][k], want["grads"][k]), (mode, k)
        assert torch.equal(o2["grad2d"], want["grad2d"]) and torch.equal(o2["vis"], want["vis"]) and torch.equal(o2["radii"], want["radii"])
        assert torch.equal(kept.last_image, want_img), mode
        o3 = kept(ids, 1.0 / views)                # and the selective fills resume on the refilled buffers
        rasterizer.check_overflow()
        assert kept._pg_ws is not None
        for k in want["grads"]:
            assert torch.equal(o3["grads"][k], want["grads"][k]), (mode, k)
        assert torch.equal(kept.last_image, want_img), mode
    rasterizer.set_sync_policy(True)


# ---------------------------------------------------------------------------------------------------------------
# The fused path against the ORACLE on identical blend inputs (north_star bars: PSNR delta < 0.01 dB, grad
# max-rel-err < 1e-4), alpha-threshold flips accounted for instead of avoided by the choice of seed: see
# tests/fused_oracle.py.  The same comparison runs at the BASELINE sizes in tests/test_gpu_fullsize.py.
# ---------------------------------------------------------------------------------------------------------------
from fused_oracle import assert_north_star, layout as _layout, run_fused_vs_oracle  # noqa: E402  (_layout: used by test_gpu_raster)


@pytest.mark.parametrize("seed", [1, 2, 3, 5, 8, 12, 13, 15, 21, 34])
@pytest.mark.parametrize("kind,views", [("hand", 3), ("object", 2), ("composite", 3)])
def test_fused_matches_oracle_on_identical_blend_inputs(kind, views, seed):
    res = run_fused_vs_oracle(kind, views, 4000, 96, 64, seed)
    assert_north_star(res, (kind, views, seed))
    print(kind, views, seed, "ambiguous pairs", res["ambiguous"], "flips", res["flips"], {k: "%.1e" % e for k, e in res["grads"].items()})


def test_fused_matches_oracle_eight_views():
    res = run_fused_vs_oracle("hand", 8, 4000, 96, 64, 12)
    assert_north_star(res, "hand-8")


@pytest.mark.parametrize("kind", ["hand", "composite"])
def test_fp16_sh_storage_is_exact_on_representable_coefficients(kind):
    """BASELINE config 5's "fp16 SH coeffs" is a STORAGE option (the reference has none: everything is fp32): with
    coefficients that are exactly representable in fp16 the fp16-storage step equals the fp32-storage step to fp32
    roundoff (same fp32 arithmetic on the same values), and with arbitrary coefficients it differs from the fp32 render by
    no more than the storage rounding (2^-11 relative per coefficient)."""
    from manus_amd.engine import HipViewCompute
    sc, ct = _scene(kind, n=5000, views=3)
    tg = torch.rand((3, 3, 64, 96), device=DEV)
    exact = dict(sc)
    exact["params"] = dict(sc["params"], _features_rest=sc["params"]["_features_rest"].half().float())
    a = HipViewCompute(exact, tg, ct, loss="l1+ssim", sh_storage="fp32")([0, 1, 2], 1.0 / 3)
    h16 = HipViewCompute(exact, tg, ct, loss="l1+ssim", sh_storage="fp16")
    b = h16([0, 1, 2], 1.0 / 3)
    assert h16._sh_copy.dtype == torch.float16 and h16._sh_copy.shape == (5000, 48)
    for k in a["grads"]:   # (two instantiations of the same source: fma contraction may differ in the last bit)
        assert max_rel_err(a["grads"][k].cpu().numpy(), b["grads"][k].cpu().numpy()) < 2e-5, k
    assert max_rel_err(a["grad2d"].cpu().numpy(), b["grad2d"].cpu().numpy()) < 2e-5
    assert abs(float(a["loss"]) - float(b["loss"])) < 1e-7
    # arbitrary coefficients: the only difference is the rounding of the stored coefficients
    f32 = HipViewCompute(sc, tg, ct, loss="l1+ssim", sh_storage="fp32")
    f16 = HipViewCompute(sc, tg, ct, loss="l1+ssim", sh_storage="fp16")
    f32([0, 1, 2], 1.0 / 3), f16([0, 1, 2], 1.0 / 3)
    d = (f32.last_image - f16.last_image).abs()
    assert 0 < float(d.max()) < 2e-3 and float(d.mean()) < 2e-5
    # the copy follows the leaves: after an in-place update it is refreshed lazily
    with torch.no_grad():
        f16.params["_features_rest"].mul_(0.5)
    f16.mark_params_changed()
    f16([0, 1, 2], 1.0 / 3)
    assert torch.equal(f16._sh_copy[:, :45].float(), f16.params["_features_rest"].detach().reshape(5000, 45).half().float())


def test_forward_only_skin_weights_follow_the_model():
    """forward_views_fused under no_grad keeps the skin weights per model state: the image must follow every way the
    positions can change -- an in-place torch update (version counter), an optimizer-style update announced by
    mark_params_changed, new leaves through set_params -- and equal a fresh object's render each time."""
    from manus_amd.engine import HipViewCompute
    sc, ct = _scene("hand", n=4000, views=2)
    tg = torch.zeros((2, 3, 64, 96), device=DEV)
    hc = HipViewCompute(sc, tg, ct)

    def fresh_image(params):
        s2 = dict(sc)
        s2["params"] = {k: v.detach().clone() for k, v in params.items()}
        with torch.no_grad():
            return HipViewCompute(s2, tg, ct).forward_views_fused([0, 1])[0].clone()

    with torch.no_grad():
        a = hc.forward_views_fused([0, 1])[0].clone()
        b = hc.forward_views_fused([0, 1])[0].clone()          # second call: cached weights
        assert hc._w_cache is not None and torch.equal(a, b) and torch.equal(a, fresh_image(hc.params))
        hc.params["_xyz"].add_(0.004 * torch.randn(hc.params["_xyz"].shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1)))
        c = hc.forward_views_fused([0, 1])[0].clone()
        assert not torch.equal(c, a) and torch.equal(c, fresh_image(hc.params))
        # an update torch does not see (what the fused Adam kernel does through raw pointers), announced like the Trainer does
        hc.params["_xyz"].data.mul_(1.01)
        hc.mark_params_changed()
        d = hc.forward_views_fused([0, 1])[0].clone()
        assert not torch.equal(d, c) and torch.equal(d, fresh_image(hc.params))
    hc.set_params({k: (v.detach() * (0.99 if k == "_xyz" else 1.0)).clone() for k, v in hc.params.items()})
    with torch.no_grad():
        e = hc.forward_views_fused([0, 1])[0].clone()
    assert torch.equal(e, fresh_image(hc.params))
    out = hc([0, 1], 0.5)                                   # a training step never uses the cache: d xyz includes the grid path
    assert float(out["grads"]["_xyz"].abs().sum()) > 0
