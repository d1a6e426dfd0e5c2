"""GPU: the HIP rasterizer (through the C ABI and the Python drop-in surface) against the
scalar oracle on identical inputs.

Bars (BASELINE.json north_star): image PSNR delta < 0.01 dB, gradient max-rel-err < 1e-4
(max|a-b| / max|b| per gradient tensor, against the fp32 oracle); integer / index work
(radii, pair count, tile ranges, blend order) bit-exact."""
import ctypes
import math

import numpy as np
import pytest
import torch

from manus_amd.synthetic import look_at_extrinsics

from oracle import RasterOracle

from util import cam_args, cam_table_np, make_camera, max_rel_err, psnr, random_gaussians

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
BG = np.array([1.0, 1.0, 1.0], np.float32)


def _oracle(cam, m, c, col, op, bg=BG, dtype=np.float32):
    a = cam_args(cam)
    return RasterOracle(a["W"], a["H"], a["tanfovx"], a["tanfovy"], a["view"], a["proj"], m, c, col, op, bg, dtype=dtype)


def _hip(cams, m, c, col, op, bg=BG, grad_img=None):
    """cams: list of camera dicts (same size).  m/c/col/op numpy (N,..) shared by the views."""
    from manus_amd.rasterizer import rasterize_views
    W, H = cams[0]["width"], cams[0]["height"]
    ct = torch.from_numpy(cam_table_np(cams)).to(DEV)
    tm = torch.tensor(m, device=DEV, requires_grad=True)
    tc = torch.tensor(c, device=DEV, requires_grad=True)
    tcol = torch.tensor(col, device=DEV, requires_grad=True)
    top = torch.tensor(op, device=DEV, requires_grad=True)
    m2d = torch.zeros((len(cams), m.shape[0], 3), device=DEV, requires_grad=True)
    img, radii = rasterize_views(ct, tm, m2d, tcol, top, tc, torch.tensor(bg, device=DEV), W, H)
    out = dict(img=img.detach().cpu().numpy(), radii=radii.cpu().numpy())
    if grad_img is not None:
        img.backward(torch.tensor(grad_img, device=DEV))
        out.update(means3D=tm.grad.cpu().numpy(), cov3D=tc.grad.cpu().numpy(), colors=tcol.grad.cpu().numpy(),
                   opacity=top.grad.cpu().numpy(), means2D=m2d.grad.cpu().numpy())
    return out


def _binning(view, V, N, W, H):
    from manus_amd import rasterizer as rz
    from manus_amd._lib import lib, ptr, stream
    ws = rz.context().last_ws
    T = ((W + 15) // 16) * ((H + 15) // 16)
    ranges = np.zeros((T, 2), np.int32)
    npairs = ctypes.c_int64(0)
    ovf = ctypes.c_int32(0)
    lib().mgr_raster_status_sync(ptr(ws.buf), ctypes.byref(npairs), ctypes.byref(ovf), stream())
    pl = np.zeros((max(int(npairs.value), 1),), np.int32)
    rc = lib().mgr_raster_debug_binning_sync(ptr(ws.buf), V, N, W, H, ws.cap, view,
                                             ranges.ctypes.data_as(ctypes.c_void_p),
                                             pl.ctypes.data_as(ctypes.c_void_p), pl.shape[0], stream())
    assert rc == 0
    return int(npairs.value), ranges, pl  # header pair count (all views, before null-pair culling)


def _check_lists_vs_oracle(o, ranges, pl, W, H):
    """The HIP tile lists are the oracle's lists (same order, ties included) minus pairs that
    provably contribute to no pixel of the tile (exact null-pair culling)."""
    opl, org = o.binning()
    g = o.geom()
    gx = (W + 15) // 16
    n_omitted = 0
    for t in range(org.shape[0]):
        mine = pl[ranges[t, 0]:ranges[t, 1]].tolist()
        ref = opl[org[t, 0]:org[t, 1]].tolist()
        it = iter(ref)
        assert all(x in it for x in mine), t          # subsequence, same relative order
        dropped = set(ref) - set(mine)
        if dropped:
            ty, tx = divmod(t, gx)
            ys, xs = np.mgrid[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16]
            for gid in dropped:
                dx, dy = g["xy"][gid, 0] - xs, g["xy"][gid, 1] - ys
                A, B, C, op = g["conic_opacity"][gid]
                power = -0.5 * (A * dx * dx + C * dy * dy) - B * dx * dy
                alpha = op * np.exp(np.minimum(power, 0.0))
                assert alpha.max() < 1.0 / 255.0, (t, gid, alpha.max())
            n_omitted += len(dropped)
    return n_omitted


def test_extension_is_loaded_and_versioned():
    from manus_amd._lib import lib
    assert lib().mgr_version() == 100
    assert torch.cuda.is_available()


@pytest.mark.parametrize("seed,n,W,H", [(0, 1500, 128, 96), (1, 4000, 200, 120), (2, 300, 33, 47)])
def test_integer_state_bit_exact(seed, n, W, H):
    cam = make_camera(W, H)
    m, c, col, op = random_gaussians(n, seed=seed)
    o = _oracle(cam, m, c, col, op)
    h = _hip([cam], m, c, col, op)
    assert (h["radii"][0] == o.radii).all()
    npairs, ranges, pl = _binning(0, 1, n, W, H)
    assert npairs == o.num_rendered                     # sum of tile rectangles (upstream's num_rendered)
    _check_lists_vs_oracle(o, ranges, pl, W, H)         # blend order incl. equal-depth ties


@pytest.mark.parametrize("seed,n,W,H", [(0, 1500, 128, 96), (3, 6000, 256, 160)])
def test_image_and_gradient_parity(seed, n, W, H):
    cam = make_camera(W, H)
    m, c, col, op = random_gaussians(n, seed=seed)
    rng = np.random.default_rng(seed + 50)
    gimg = rng.normal(size=(1, 3, H, W)).astype(np.float32)
    o = _oracle(cam, m, c, col, op)
    ob = o.backward(gimg[0])
    h = _hip([cam], m, c, col, op, grad_img=gimg)
    # image: PSNR against a common target, oracle vs HIP
    m2 = m + rng.normal(size=m.shape).astype(np.float32) * 0.01
    target = _oracle(cam, m2, c, col, op).color
    d_psnr = abs(psnr(h["img"][0], target) - psnr(o.color, target))
    assert d_psnr < 0.01, d_psnr
    assert np.abs(h["img"][0] - o.color).max() < 5e-3          # isolated threshold flips only
    assert np.mean(np.abs(h["img"][0] - o.color)) < 2e-6
    for k in ("means3D", "cov3D", "colors", "opacity"):
        e = max_rel_err(h[k], ob[k])
        assert e < 1e-4, (k, e)
    e = max_rel_err(h["means2D"][0], ob["means2D"])
    assert e < 1e-4, ("means2D", e)
    assert (h["means2D"][0][:, 2] == 0).all()
    # and against the fp64 oracle (the "true" gradient), looser: fp32 roundoff of both sides
    o64 = _oracle(cam, m, c, col, op, dtype=np.float64)
    b64 = o64.backward(gimg[0])
    for k in ("means3D", "cov3D", "colors", "opacity"):
        assert max_rel_err(h[k], b64[k]) < 2e-3, k


def test_edge_cases_empty_and_culled():
    from manus_amd.rasterizer import rasterize_views
    W, H = 64, 48
    cam = make_camera(W, H, pos=(0, 0, -2.0), target=(0, 0, 0), focal=80.0)
    ct = torch.from_numpy(cam_table_np([cam])).to(DEV)
    bg = torch.tensor([0.2, 0.4, 0.6], device=DEV)
    # N = 0: background image, empty radii, nothing launched on the Gaussians
    z = lambda *s: torch.zeros(s, device=DEV)
    img, radii = rasterize_views(ct, z(0, 3), z(1, 0, 3), z(0, 3), z(0), z(0, 6), bg, W, H)
    assert radii.shape == (1, 0)
    assert torch.allclose(img[0, 0], torch.full((H, W), 0.2, device=DEV))
    assert torch.allclose(img[0, 2], torch.full((H, W), 0.6, device=DEV))
    # everything culled (behind the camera / near plane / off screen)
    m = np.array([[0, 0, -1.9], [0, 0, -3.0], [50.0, 0, 0]], np.float32)
    c = np.repeat(np.array([[4e-4, 0, 0, 4e-4, 0, 4e-4]], np.float32), 3, 0)
    h = _hip([cam], m, c, np.ones((3, 3), np.float32), np.full(3, 0.5, np.float32), bg=np.array([0.2, 0.4, 0.6], np.float32),
             grad_img=np.ones((1, 3, H, W), np.float32))
    assert (h["radii"] == 0).all()
    assert np.allclose(h["img"][0, 1], 0.4)
    for k in ("means3D", "cov3D", "colors", "opacity", "means2D"):
        assert (h[k] == 0).all(), k


def test_single_gaussian_closed_form():
    W = H = 64
    cam = make_camera(W, H, pos=(0, 0, -2.0), target=(0, 0, 0), focal=80.0)
    sig, opac = 0.05, 0.8
    col = np.array([[0.9, 0.1, 0.3]], np.float32)
    c = np.array([[sig * sig, 0, 0, sig * sig, 0, sig * sig]], np.float32)
    h = _hip([cam], np.zeros((1, 3), np.float32), c, col, np.array([opac], np.float32))
    var2d = (80.0 * sig / 2.0) ** 2 + 0.3
    assert h["radii"][0, 0] == math.ceil(3 * math.sqrt(var2d))
    ys, xs = np.mgrid[0:H, 0:W]
    r2 = (xs - (W - 1) / 2) ** 2 + (ys - (H - 1) / 2) ** 2
    alpha = np.minimum(0.99, opac * np.exp(-0.5 * r2 / var2d))
    alpha = np.where(alpha < 1 / 255, 0, alpha)
    rad = h["radii"][0, 0]
    cx = (W - 1) / 2
    x0, x1 = int((cx - rad) / 16), int((cx + rad + 15) / 16)
    mask = (xs // 16 >= x0) & (xs // 16 < x1) & (ys // 16 >= x0) & (ys // 16 < x1)
    alpha = np.where(mask, alpha, 0)
    for ch in range(3):
        assert np.abs(h["img"][0, ch] - (col[0, ch] * alpha + (1 - alpha))).max() < 3e-5


def test_capacity_overflow_is_detected_and_retried():
    from manus_amd import rasterizer as rz
    W, H = 128, 96
    cam = make_camera(W, H)
    m, c, col, op = random_gaussians(3000, seed=7, sigma=(0.03, 0.1))
    o = _oracle(cam, m, c, col, op)
    assert o.num_rendered > 8 * 3000  # exceeds the default capacity 8*N -> first try overflows
    rz.context().clear()
    h = _hip([cam], m, c, col, op)
    assert np.abs(h["img"][0] - o.color).max() < 5e-3
    assert rz.check_overflow() == o.num_rendered


def test_skipped_binning_tiers_are_verified_and_retried():
    """debug bits 16 / 32 of the forward (include/manus_hip.h): a forward whose views' tile boxes fit the smallest LDS tier
    lets the next one skip the launches of the larger tiers; when the next one needs them after all -- the same Gaussians
    spread over the whole 1080p frame: a box of more than 2048 tiles -- it is flagged (MGR_ETIER) and run again with every
    launch.  Both forwards must equal a forward on a fresh context, bit for bit, and the large box then keeps its launches."""
    from manus_amd import rasterizer as rz
    W, H = 1920, 1080
    cam = make_camera(W, H)
    m, c, col, op = random_gaussians(20000, seed=11, spread=0.03, sigma=(0.002, 0.006))
    g = np.random.default_rng(1).normal(size=(1, 3, H, W)).astype(np.float32)
    ctx = rz.context()
    ctx.clear()
    want_small = _hip([cam], m, c, col, op, grad_img=g)
    ctx.clear()
    want_big = _hip([cam], 6.0 * m, c, col, op, grad_img=g)
    ctx.clear()
    ctx.tier_retries = 0
    first = _hip([cam], m, c, col, op, grad_img=g)
    ws = ctx.last_ws
    # (the instance sort's buckets span the depth range of THIS forward's visible instances: small items from the first forward on)
    assert ws.tiers == 0 and ws.skip_bits() == 48 + 128
    got_small = _hip([cam], m, c, col, op, grad_img=g)
    assert ws is ctx.last_ws and ws.tiers == 0 and ws.skip_bits() == 48 + 128      # box of at most 1536 tiles, no sort item beyond k_dbin_rank
    got_big = _hip([cam], 6.0 * m, c, col, op, grad_img=g)  # skips them, is flagged, runs again
    assert ctx.tier_retries == 1 and (ctx.last_ws.tiers & 0xFF) == 1 and (ctx.last_ws.skip_bits() & 48) == 32
    again = _hip([cam], 6.0 * m, c, col, op, grad_img=g)
    assert ctx.tier_retries == 1 and ctx.last_ws.skip_bits() == 32 + 128
    for k in want_small:
        assert np.array_equal(first[k], want_small[k]), k
    for k in want_small:
        assert np.array_equal(got_small[k], want_small[k]), k
        assert np.array_equal(got_big[k], want_big[k]), k
        assert np.array_equal(again[k], want_big[k]), k
    # the instance sort (bit 128): every Gaussian in one depth plane of the camera -> one depth bucket of 20 000 keys whatever the
    # range, beyond k_dbin_rank's items: the launch behind it sorts it, every time (bit 128 never set for this scene)
    E = look_at_extrinsics((0.3, -0.2, -1.5), (0, 0, 0), up=(0, 1, 0))
    R, tvec = np.asarray(E)[:3, :3], np.asarray(E)[:3, 3]
    pc = (m.astype(np.float64) @ R.T + tvec)
    pc[:, 2] = 1.5
    m_plane = ((pc - tvec) @ R).astype(np.float32)            # same image positions, all at view depth 1.5
    ctx.clear()
    want_plane = _hip([cam], m_plane, c, col, op, grad_img=g)        # (no record of the tiers yet: both launches)
    assert (ctx.last_ws.tiers >> 8) >= 1 and (ctx.last_ws.skip_bits() & 256) and not (ctx.last_ws.skip_bits() & 128)
    got_plane = _hip([cam], m_plane, c, col, op, grad_img=g)
    assert (ctx.last_ws.tiers >> 8) >= 1 and not (ctx.last_ws.skip_bits() & 128)
    for k in want_plane:
        assert np.array_equal(got_plane[k], want_plane[k]), k
    # ... and the same plane met while the launch behind is being skipped: flagged, run again, exact
    ctx.clear()
    ctx.tier_retries = 0
    _hip([cam], m, c, col, op, grad_img=g)
    assert ctx.last_ws.skip_bits() & 128
    got_plane2 = _hip([cam], m_plane, c, col, op, grad_img=g)
    assert ctx.tier_retries == 1 and not (ctx.last_ws.skip_bits() & 128)
    for k in want_plane:
        assert np.array_equal(got_plane2[k], want_plane[k]), k
    # a cloud pulled apart in depth between two forwards: the buckets follow the forward's own range, nothing to re-run
    far = m.copy()
    far[:, 2] += 0.6 * np.sign(far[:, 2])
    ctx.clear()
    want_far = _hip([cam], far, c, col, op, grad_img=g)
    ctx.clear()
    ctx.tier_retries = 0
    _hip([cam], m, c, col, op, grad_img=g)
    got_far = _hip([cam], far, c, col, op, grad_img=g)
    assert (ctx.last_ws.tiers >> 8) == 0          # (no sort item near k_dbin_rank's limit; the nearer half's larger tile box may ask for its tier)
    for k in want_far:
        assert np.array_equal(got_far[k], want_far[k]), k


def test_multi_view_batch_equals_single_views_bitwise():
    W, H = 160, 96
    cams = [make_camera(W, H, pos=p) for p in [(0.3, -0.2, -1.5), (-0.8, 0.1, -1.2), (0.1, 0.9, -1.3)]]
    m, c, col, op = random_gaussians(2500, seed=11)
    g = np.random.default_rng(0).normal(size=(3, 3, H, W)).astype(np.float32)
    hb = _hip(cams, m, c, col, op, grad_img=g)
    acc = {k: 0 for k in ("means3D", "cov3D", "colors", "opacity")}
    for v, cam in enumerate(cams):
        hs = _hip([cam], m, c, col, op, grad_img=g[v:v + 1])
        assert (hs["img"][0] == hb["img"][v]).all()
        assert (hs["radii"][0] == hb["radii"][v]).all()
        assert (hs["means2D"][0] == hb["means2D"][v]).all()
        for k in acc:
            acc[k] = acc[k] + hs[k].astype(np.float64)
    for k in acc:  # shared inputs receive the sum over views
        assert max_rel_err(hb[k], acc[k]) < 1e-6, k


def test_run_to_run_determinism():
    W, H = 128, 96
    cam = make_camera(W, H)
    m, c, col, op = random_gaussians(5000, seed=13, sigma=(0.01, 0.08))
    g = np.random.default_rng(1).normal(size=(1, 3, H, W)).astype(np.float32)
    a = _hip([cam], m, c, col, op, grad_img=g)
    b = _hip([cam], m, c, col, op, grad_img=g)
    for k in a:
        assert (a[k] == b[k]).all(), k  # no float atomics anywhere: bitwise reproducible


def test_deep_tile_uses_global_sort_path():
    """> 16384 pairs in one tile: the in-place global-memory sort network is used."""
    W = H = 16
    cam = make_camera(W, H, pos=(0, 0, -2.0), target=(0, 0, 0), focal=30.0)
    n = 20000
    g = np.random.default_rng(3)
    m = (g.normal(size=(n, 3)) * np.array([0.05, 0.05, 0.3])).astype(np.float32)
    m[5] = m[6]  # an exact depth tie
    c = np.repeat(np.array([[1e-4, 0, 0, 1e-4, 0, 1e-4]], np.float32), n, 0)
    col = g.uniform(0, 1, size=(n, 3)).astype(np.float32)
    op = np.full(n, 0.02, np.float32)
    o = _oracle(cam, m, c, col, op)
    h = _hip([cam], m, c, col, op)
    npairs, ranges, pl = _binning(0, 1, n, W, H)
    assert npairs == o.num_rendered and int(ranges[0, 1] - ranges[0, 0]) > 16384
    _check_lists_vs_oracle(o, ranges, pl, W, H)
    assert np.abs(h["img"][0] - o.color).max() < 1e-3


def test_full_size_properties_1080p():
    """BASELINE-size checks through size-independent properties (the oracle is too slow here)."""
    from manus_amd.synthetic import make_scene
    from manus_amd.engine import HipViewCompute
    from manus_amd.synthetic import camera_table
    from manus_amd import rasterizer as rz
    sc = make_scene(n_gaussians=100000, kind="object", seed=1, n_cameras=2, device=DEV)
    ct = camera_table(sc["cameras"], DEV)
    hc = HipViewCompute(sc, torch.zeros((2, 3, 1080, 1920), device=DEV), ct)
    with torch.no_grad():
        img, radii, _ = hc.forward_views([0, 1])
    assert torch.isfinite(img).all()
    # pixels of tiles no Gaussian touches are exactly the background
    npairs, ranges, pl = _binning(0, 2, sc["N"], 1920, 1080)
    empty = np.argwhere(ranges[:, 0] == ranges[:, 1])[:, 0]
    assert len(empty) > 100
    t = int(empty[len(empty) // 2])
    ty, tx = divmod(t, 120)
    blk = img[0, :, ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16]
    assert (blk == 1.0).all()
    # every tile list is sorted by view-space depth
    view = ct[0, 2:18].reshape(4, 4)
    z = (sc["params"]["_xyz"] @ view[:3, 2] + view[3, 2]).cpu().numpy()
    busy = np.argsort(ranges[:, 1] - ranges[:, 0])[-20:]
    for t in busy:
        zz = z[pl[ranges[t, 0]:ranges[t, 1]]]
        assert (np.diff(zz) >= 0).all()
    # the surviving pairs never exceed the sum of tile rectangles (the header count, both views)
    assert 0 < int(ranges[-1, 1]) <= npairs
    assert rz.check_overflow() == npairs


def test_closeup_deep_tiles_parity():
    """Close-up rig (hand fills the frame, deep tile lists, several checkpointed chunks per tile):
    image and gradients against the oracle on a reduced image."""
    from manus_amd.synthetic import make_scene
    sc = make_scene(n_gaussians=20000, kind="hand", seed=9, grid_res=16, n_cameras=1, width=160, height=96,
                    cam_radius=0.30, sigma_range=(4e-3, 1.5e-2), device="cpu")
    sc["params"]["_opacity"] = sc["params"]["_opacity"] - 3.0   # faint Gaussians: pixels saturate late
    from oracle import torch_ref as tr
    c = sc["cameras"][0]
    o = tr.hand_forward(sc["params"], sc["grid"], sc["grid_center"], sc["grid_scale"], sc["posed"][0], sc["rest"],
                        torch.tensor(c["camera_center"], dtype=torch.float32))
    m, cov = o["posed_xyz"].numpy(), o["posed_cov"].numpy()
    col, op = o["colors"].numpy(), o["opacity"].numpy()[:, 0]
    ro = _oracle(c, m, cov, col, op)
    assert ro.num_rendered > 4 * 20000            # deep lists
    g = np.random.default_rng(4).normal(size=(1, 3, 96, 160)).astype(np.float32)
    ob = ro.backward(g[0])
    h = _hip([c], m, cov, col, op, grad_img=g)
    ft, nc = ro.image_state()
    assert nc.max() > 256                          # more than two chunks consumed somewhere
    assert np.abs(h["img"][0] - ro.color).max() < 5e-3 and np.mean(np.abs(h["img"][0] - ro.color)) < 2e-6
    for k in ("means3D", "cov3D", "colors", "opacity"):
        e = max_rel_err(h[k], ob[k])
        assert e < 1e-4, (k, e)


@pytest.mark.parametrize("kind,n,views", [("object", 100000, 1), ("hand", 300000, 8), ("composite", 500000, 7),
                                          ("composite", 500000, 6)])
def test_baseline_configs_full_size(kind, n, views):
    """BASELINE.json configs 2 (static object, 100k, 1 view), 3 (articulated hand, 300k, 8 views) and 4 (hand+object
    composite, 500k; 53 cameras over 8 ranks = 7 or 6 views per rank) at 1920x1080 through size-independent
    properties (the oracle comparison at these sizes is tests/test_gpu_fullsize.py): finite, bit-reproducible, linear in dL/dimage, tile
    lists sorted by depth and made of valid, visible Gaussians, background where nothing lands, statistics consistent
    with the radii, exactly-zero gradient rows for Gaussians no view sees, and the fused image equal to the modular
    operators' (which are checked against the oracle at sizes it can run)."""
    from manus_amd import rasterizer as rz
    from manus_amd.engine import HipViewCompute
    from manus_amd.synthetic import camera_table, make_scene
    W, H = 1920, 1080
    sc = make_scene(n_gaussians=n, kind=kind, seed=2, n_cameras=views, device=DEV)
    ct = camera_table(sc["cameras"], DEV)
    hc = HipViewCompute(sc, torch.zeros((views, 3, H, W), device=DEV), ct)
    assert hc.n_art == {"object": 0, "hand": n, "composite": sc["n_hand"]}[kind]
    ids = list(range(views))
    rz.set_sync_policy(True)
    gen = torch.Generator(device=DEV).manual_seed(1)
    g1 = torch.randn((views, 3, H, W), device=DEV, generator=gen)
    g2 = torch.randn((views, 3, H, W), device=DEV, generator=gen)
    clone = lambda o: {k: ({q: t.clone() for q, t in v.items()} if isinstance(v, dict) else v.clone()) for k, v in o.items()}
    a = clone(hc._step_direct(ids, 1.0, g_img=g1))
    img, radii = hc.last_image.clone(), hc.last_radii.clone()
    b = clone(hc._step_direct(ids, 1.0, g_img=g1))
    assert torch.isfinite(img).all() and float(img.min()) >= 0.0
    for k in a["grads"]:
        assert torch.isfinite(a["grads"][k]).all(), k
        assert torch.equal(a["grads"][k], b["grads"][k]), k           # bit-reproducible
    assert torch.equal(img, hc.last_image)
    # ---- binning state of view 0 and of the last view: sorted by the recorded depth, valid visible Gaussians
    ws = rz.context().last_ws
    from test_gpu_fused import _layout
    L = _layout(views, n, W, H, ws.cap)
    for v in (0, views - 1):
        npairs, ranges, pl = _binning(v, views, n, W, H)
        depth = ws.buf[L["depth"] + v * n * 4: L["depth"] + (v + 1) * n * 4].view(torch.float32).cpu().numpy()
        rad_v = radii[v].cpu().numpy()
        pl = pl[: int(ranges[-1, 1])]                                 # this view's surviving pairs
        assert 0 < pl.shape[0] <= npairs and pl.min() >= 0 and pl.max() < n and (rad_v[pl] > 0).all()
        sizes = ranges[:, 1] - ranges[:, 0]
        for t in list(np.argsort(sizes)[-12:]) + list(np.nonzero(sizes > 0)[0][::997]):
            zz = depth[pl[ranges[t, 0]:ranges[t, 1]]]
            assert (np.diff(zz) >= 0).all(), (v, t)
        empty = np.nonzero(sizes == 0)[0]
        assert len(empty) > 100
        ty, tx = divmod(int(empty[len(empty) // 2]), 120)
        assert (img[v, :, ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16] == 1.0).all()      # white background
    assert float((img[0] == 1.0).float().mean()) > 0.2
    # ---- statistics
    assert torch.equal(a["radii"], radii.max(dim=0).values)
    assert torch.equal(a["vis"], (radii > 0).sum(0).float())
    unseen = (radii > 0).sum(0) == 0
    for k in a["grads"]:
        assert not a["grads"][k].reshape(n, -1)[unseen].any(), k
    assert not a["grad2d"][unseen].any() and (a["grad2d"] >= 0).all()
    # ---- the backward is linear in dL/dimage
    c = clone(hc._step_direct(ids, 1.0, g_img=g2))
    s12 = hc._step_direct(ids, 1.0, g_img=g1 + g2)
    for k in a["grads"]:
        want = a["grads"][k].double() + c["grads"][k].double()
        err = float((s12["grads"][k].double() - want).abs().max()) / max(float(want.abs().max()), 1e-30)
        assert err < 2e-5, (k, err)
    # ---- fused image == modular operators (per view; the modular path is the one checked against the oracle)
    del a, b, c, s12
    with torch.no_grad():
        for v in (0, views - 1):
            im_m, rad_m, _ = hc.forward_views([v])
            assert torch.equal(rad_m[0], radii[v])
            d = (im_m[0] - img[v]).abs()
            assert float(d.max()) < 5e-3 and float(d.mean()) < 2e-6, (v, float(d.max()), float(d.mean()))


@pytest.mark.parametrize("seed", list(range(8)))
def test_randomised_sizes_against_oracle(seed):
    """Random image sizes (not multiples of 16, down to a single tile), Gaussian counts from 1 up, random
    backgrounds: integer state exact, image and gradients within the path's bars."""
    rng = np.random.default_rng(1000 + seed)
    W, H = int(rng.integers(5, 230)), int(rng.integers(5, 150))
    n = int(rng.choice([1, 2, 17, 300, 2500]))
    bg = rng.uniform(0, 1, size=3).astype(np.float32)
    cam = make_camera(W, H)
    m, c, col, op = random_gaussians(n, seed=seed + 77)
    gimg = rng.normal(size=(1, 3, H, W)).astype(np.float32)
    o = _oracle(cam, m, c, col, op, bg=bg)
    ob = o.backward(gimg[0])
    h = _hip([cam], m, c, col, op, bg=bg, grad_img=gimg)
    assert (h["radii"][0] == o.radii).all()
    npairs, ranges, pl = _binning(0, 1, n, W, H)
    assert npairs == o.num_rendered
    _check_lists_vs_oracle(o, ranges, pl, W, H)
    assert np.abs(h["img"][0] - o.color).max() < 5e-3 and np.mean(np.abs(h["img"][0] - o.color)) < 2e-6
    for k in ("means3D", "cov3D", "colors", "opacity"):
        if np.abs(ob[k]).max() > 0:
            assert max_rel_err(h[k], ob[k]) < 1e-4, (k, W, H, n)
    if np.abs(ob["means2D"]).max() > 0:
        assert max_rel_err(h["means2D"][0], ob["means2D"]) < 1e-4


@pytest.mark.parametrize("kind,n,views,size,sigma", [("hand", 300000, 3, (1920, 1080), None), ("object", 60000, 2, (1000, 700), None),
                                                     ("hand", 5000, 2, (96, 64), None),
                                                     # rectangles of more than 16 and more than 64 tiles, boxes of more than 2048 tiles
                                                     ("object", 3000, 2, (1000, 900), (5e-3, 6e-2))])
def test_ordered_binning_equals_sorted_route(kind, n, views, size, sigma, monkeypatch):
    """The depth-ordered binning (instances sorted once, pairs generated in order) and the per-tile sorts must produce
    the same tile lists entry for entry -- the (depth, index) keys are unique, so there is exactly one correct order --
    and therefore bit-identical images."""
    from manus_amd import rasterizer as rz
    from manus_amd.engine import HipViewCompute
    from manus_amd.synthetic import camera_table, make_scene
    W, H = size
    kw = {} if W > 1000 else dict(grid_res=24, cam_radius=0.5)
    if sigma is not None:
        kw["sigma_range"] = sigma
    sc = make_scene(n_gaussians=n, kind=kind, seed=5, n_cameras=views, width=W, height=H, device=DEV, **kw)
    ct = camera_table(sc["cameras"], DEV)
    hc = HipViewCompute(sc, torch.zeros((views, 3, H, W), device=DEV), ct, loss="l1")
    rz.set_sync_policy(True)
    from test_gpu_fused import _layout
    got = {}
    # ("ordered" twice: the second forward runs with what the first one learnt -- skipped tiers, and for the scene with large
    #  rectangles k_bin_scatter's lane-spreading instantiation instead of the instance-by-instance route)
    for route in ("sorted", "ordered", "ordered2"):
        monkeypatch.setenv("MGR_BINNING", "ordered" if route == "ordered2" else route)
        with torch.no_grad():
            img = hc.forward_views_fused(list(range(views)))[0].clone()
        ws = rz.context().last_ws
        if route == "ordered2" and sigma is not None:
            assert ws.tiers & 4 and ws.skip_bits() & 4096
        L = _layout(views, n, W, H, ws.cap)
        T = ((W + 15) // 16) * ((H + 15) // 16)
        ts = ws.buf[L["tile_start"]: L["tile_start"] + 4 * (views * T + 1)].view(torch.int32).clone()
        total = int(ts[-1])
        assert 0 < total <= ws.cap
        gid = ws.buf[L["sorted_gid"]: L["sorted_gid"] + 4 * total].view(torch.int32).clone()
        got[route] = (img, ts, gid)
        if sigma is not None and route == "ordered":      # the scene does exercise the general routes
            assert ws.tiers & 4                           # (k_bin_scatter met rectangles of more than 64 tiles and said so)
            rect = ws.buf[L["rect"]: L["rect"] + 8 * views * n].view(torch.int16).view(-1, 4).int()
            tiles = (rect[:, 2] - rect[:, 0]) * (rect[:, 3] - rect[:, 1])
            assert int((tiles > 64).sum()) > 50 and int(((tiles > 16) & (tiles <= 64)).sum()) > 50
            bb = ws.buf[L["db_bbox"]: L["db_bbox"] + 8 * views].view(torch.int16).view(-1, 4).int()
            assert int((bb[:, 2] * bb[:, 3]).max()) > 2048
    for route in ("ordered", "ordered2"):
        assert torch.equal(got["sorted"][1], got[route][1])
        neq = (got["sorted"][2] != got[route][2]).nonzero()
        assert neq.numel() == 0, (route, int(neq.numel()), neq[:5].flatten().tolist())
        assert torch.equal(got["sorted"][0], got[route][0])


def test_row_of_64_tiles_on_an_odd_box_row(monkeypatch):
    """k_bin_scatter splits a batch's pairs into the tiles on even and on odd rows of the view's tile box (one consumer wave
    each).  A rectangle exactly 64 tiles wide and one tile high -- a 1000-px Gaussian clipped to the image's last tile row --
    on an ODD box row is the case where the even-row comb must not be shifted by 64 (= not at all): its tiles belong to
    the odd list, next to those of the small Gaussians that share the row in the same batch of 64 depth-consecutive
    instances (round-5 advisor finding: they went to the even list, two waves then raced on the same cursors)."""
    W, H = 1920, 1080
    cam = make_camera(W, H, pos=(0, 0, -1.5))
    f = 1.2 * W
    rng = np.random.default_rng(3)
    n = 600
    px, py, Z = rng.uniform(560, 1360, n), rng.uniform(1067.5, 1078, n), rng.uniform(-0.05, 0.05, n)
    d = 1.5 + Z
    m = np.stack([(959.5 - px) * d / f, (539.5 - py) * d / f, Z], 1).astype(np.float32)
    s, sw = 0.0019, 0.103
    c = np.tile(np.array([s * s, 0, 0, s * s, 0, s * s], np.float32), (n, 1))
    op = rng.uniform(0.3, 0.95, n).astype(np.float32)
    # the wide one: isotropic, 497 px of radius, centred 490 px below the image, in the middle of the depth range
    m = np.concatenate([m, np.array([[-0.0005, -0.671, 0.0]], np.float32)])
    c = np.concatenate([c, np.array([[sw * sw, 0, 0, sw * sw, 0, sw * sw]], np.float32)])
    op = np.concatenate([op, np.array([0.99], np.float32)])
    col = rng.uniform(0, 1, size=(n + 1, 3)).astype(np.float32)
    o = _oracle(cam, m, c, col, op)
    r = o.geom()["rect"]
    assert r[n, 2] - r[n, 0] == 64 and r[n, 3] - r[n, 1] == 1 and r[n, 1] == 67        # 64 x 1 tiles on the last tile row
    assert r[:n, 1].min() == 66                                                          # ... which is row 1 of the box
    opl, org = o.binning()
    shared = [t for t in range(67 * 120, 68 * 120) if n in opl[org[t, 0]:org[t, 1]] and org[t, 1] - org[t, 0] > 8]
    assert len(shared) > 30                                                              # tiles it shares with many others
    from manus_amd import rasterizer as rz
    from test_gpu_fused import _layout
    lists = {}
    for route in ("ordered", "ordered", "ordered", "sorted"):      # (the race of the two waves was not deterministic: three runs)
        monkeypatch.setenv("MGR_BINNING", route)
        h = _hip([cam], m, c, col, op)
        assert (h["radii"][0] == o.radii).all()
        npairs, ranges, pl = _binning(0, 1, n + 1, W, H)
        assert npairs == o.num_rendered
        _check_lists_vs_oracle(o, ranges, pl, W, H)
        assert np.abs(h["img"][0] - o.color).max() < 5e-3
        if route == "ordered":
            ws = rz.context().last_ws
            L = _layout(1, n + 1, W, H, ws.cap)
            bb = ws.buf[L["db_bbox"]: L["db_bbox"] + 8].view(torch.int16).int().tolist()
            assert (67 - bb[1]) % 2 == 1 and bb[2] * bb[3] <= 1536, bb                  # odd box row, the lane-mask route
        prev = lists.setdefault(route, (ranges.copy(), pl.copy(), h["img"].copy()))
        assert (prev[0] == ranges).all() and (prev[1] == pl).all() and (prev[2] == h["img"]).all()
    assert (lists["ordered"][0] == lists["sorted"][0]).all() and (lists["ordered"][1] == lists["sorted"][1]).all()
    assert (lists["ordered"][2] == lists["sorted"][2]).all()


@pytest.mark.parametrize("size", [(3840, 2160), (6016, 4000)])
def test_large_tile_grids_against_oracle(size):
    """Tile grids beyond the usual: 4K (32 400 tiles: LDS histograms and the ordered binning's whole-grid cursors, one
    wave per CU) and 24 Mpx (94 000 tiles: no LDS histogram at all -- global tile counters, k_emit and the per-tile sorts).
    Few Gaussians, some of them hundreds of tiles large, against the scalar oracle: integer state exact, image and
    gradients within the path's bars."""
    W, H = size
    n = 160
    cam = make_camera(W, H, focal=0.9 * W)
    m, c, col, op = random_gaussians(n, seed=11, spread=0.45, sigma=(0.004, 0.12), opacity=(0.1, 0.9))
    rng = np.random.default_rng(5)
    gimg = rng.normal(size=(1, 3, H, W)).astype(np.float32)
    o = _oracle(cam, m, c, col, op)
    ob = o.backward(gimg[0])
    h = _hip([cam], m, c, col, op, grad_img=gimg)
    assert (h["radii"][0] == o.radii).all() and (o.radii > 0).sum() > n // 2
    npairs, ranges, pl = _binning(0, 1, n, W, H)
    assert npairs == o.num_rendered
    _check_lists_vs_oracle(o, ranges, pl, W, H)
    assert np.abs(h["img"][0] - o.color).max() < 5e-3 and np.mean(np.abs(h["img"][0] - o.color)) < 2e-6
    # at 24 Mpx a single Gaussian sums up to 10^6 pixel terms in fp32, in a different order here (tree) and in the
    # scalar oracle (running sum): the bar is widened for that case only
    bar = 1e-4 if W * H < 10_000_000 else 5e-4
    for k in ("means3D", "cov3D", "colors", "opacity"):
        assert max_rel_err(h[k], ob[k]) < bar, k
    assert max_rel_err(h["means2D"][0], ob["means2D"]) < bar
