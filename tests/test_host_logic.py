"""CPU: host-side logic of the product package — camera / skeleton mirrors against the
reference golden vectors, the C-ABI library (loads, exports every declared symbol, argument
validation without a GPU), loud failure without a GPU, workspace sizing, and the view-sharded
engine under a world_size-2 gloo group (with the torch oracle standing in for the kernels)."""
import ctypes
import os
import re
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------------------
# mirrors of the reference's host code
# ---------------------------------------------------------------------------
def test_cam_utils_match_reference(golden_dir):
    from manus_amd.cam_utils import get_opengl_camera_attributes, getProjectionMatrix
    d = np.load(os.path.join(golden_dir, "cameras.npz"))
    for i in range(d["K"].shape[0]):
        o = get_opengl_camera_attributes(d["K"][i].copy(), d["extr"][i].copy(), int(d["width"]), int(d["height"]))
        for k in ("fovx", "fovy", "world_view_transform", "projection_matrix", "full_proj_transform", "camera_center"):
            np.testing.assert_allclose(np.asarray(o[k]), d[k][i], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(getProjectionMatrix(0.01, 100.0, 0.7, 0.5), d["proj_0p01_100"], rtol=1e-13)
    # layout contract of the operator: element (row r, col c) of P*E sits at flat index 4*c + r
    o = get_opengl_camera_attributes(d["K"][0].copy(), d["extr"][0].copy(), 1920, 1080)
    E = np.concatenate([d["extr"][0], [[0, 0, 0, 1]]], 0)
    PE = getProjectionMatrix(0.01, 100.0, o["fovx"], o["fovy"]) @ E
    flat = np.asarray(o["full_proj_transform"]).reshape(-1)
    for r in range(4):
        for c in range(4):
            assert abs(flat[4 * c + r] - PE[r, c]) < 1e-9


def test_transforms_match_reference(golden_dir):
    from manus_amd import transforms as T
    d = np.load(os.path.join(golden_dir, "fk_novel_pose.npz"))
    kintree = {str(i): int(p) for i, p in enumerate(d["parents"])}
    rest, pose = torch.tensor(d["rest_matrixs"]), torch.tensor(d["pose_params"])
    F = pose.shape[0]
    fk = T.get_pose_wrt_root(rest, pose, torch.eye(3)[None].repeat(F, 1, 1), torch.zeros(F, 3), kintree)
    assert np.abs(fk.numpy() - d["pose_matrixs"]).max() < 2e-6   # known-answer data of the reference
    fkg = T.get_pose_wrt_root(rest, pose, torch.tensor(d["global_R"]), torch.tensor(d["global_t"]), kintree)
    assert np.abs(fkg.numpy() - d["fk_global"]).max() < 1e-6
    e = T.euler_angles_to_matrix(torch.tensor(d["root_rotation"]), "XYZ", intrinsic=True)
    assert np.abs(e.numpy() - d["euler_intrinsic"]).max() < 1e-6
    e = T.euler_angles_to_matrix(torch.tensor(d["eulers"]), "XYZ", intrinsic=True)
    assert np.abs(e.numpy() - d["euler_bones_intrinsic"]).max() < 1e-6
    with pytest.raises(ValueError):
        T.euler_angles_to_matrix(torch.zeros(2, 3), "XXY")
    assert T.build_kintree(["a", "b", "c"], ["None", "a", "b"]) == {"0": -1, "1": 0, "2": 1}
    # full armature pipeline = euler -> matrix -> FK
    eul = torch.cat([torch.zeros(F, 1, 3), torch.tensor(d["eulers"])], 1)
    arm = T.euler_angles_to_armature_space(eul, kintree, rest, torch.zeros(F, 3))
    assert arm.shape == (F, 20, 4, 4)
    Tb = T.bone_transforms(torch.tensor(d["pose_matrixs"][0]), rest)
    assert Tb.shape == (21, 4, 4) and torch.equal(Tb[20], torch.eye(4))


def test_structures_index_like_reference():
    from manus_amd.structures import Bones, Cameras
    b = Bones(np.array(["a", "b"]), np.zeros((2, 3)), np.ones((2, 3)), np.zeros((2, 4, 4)))
    assert b[1].tails.shape == (3,) and b[1].eulers is None
    c = Cameras(*[np.zeros((2,) + s) for s in [(), (3, 3), (4, 4), (), (), (), (), (4, 4), (4, 4), (4, 4), (3,)]])
    assert c[0].K.shape == (3, 3)


def test_synthetic_scene_shapes():
    from manus_amd.synthetic import camera_table, make_scene
    sc = make_scene(n_gaussians=900, kind="hand", seed=0, grid_res=16, n_cameras=3, width=64, height=48)
    assert sc["params"]["_xyz"].shape == (900, 3) and sc["params"]["_features_rest"].shape == (900, 15, 3)
    assert sc["grid"].shape[-1] == 21 and sc["transforms"].shape == (3, 21, 4, 4)
    assert torch.allclose(sc["grid"].sum(-1), torch.ones(sc["grid"].shape[:3]), atol=1e-5)
    ct = camera_table(sc["cameras"], "cpu")
    assert ct.shape == (3, 40)
    # every Gaussian projects inside a generous frustum of every camera (capture-like rig)
    for c in sc["cameras"]:
        E = torch.tensor(c["extr"][:3], dtype=torch.float32)
        z = sc["params"]["_xyz"] @ E[2, :3] + E[2, 3]
        assert (z > 0.2).all()
    sc2 = make_scene(n_gaussians=500, kind="composite", seed=0, grid_res=8, n_cameras=1, width=32, height=32)
    assert sc2["N"] == 500 and 0 < sc2["n_hand"] < 500


# ---------------------------------------------------------------------------
# C ABI
# ---------------------------------------------------------------------------
def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "manus_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(mgr_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from manus_amd import _lib
    from manus_amd.build import build
    build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), n
        assert n in _lib.SIGNATURES, "python binding missing for " + n
    assert set(_lib.SIGNATURES) == set(names)
    assert _lib.lib().mgr_version() == 100
    # the product build carries no knock-out / instrumentation switch (tools/instr variants report their bits here)
    assert _lib.lib().mgr_build_variant() == 0


def test_argument_validation_without_gpu():
    from manus_amd._lib import lib
    L = lib()
    assert L.mgr_raster_workspace_bytes(1, 1000, 64, 64, 8000) > 8000 * 60
    a = L.mgr_raster_workspace_bytes(8, 300000, 1920, 1080, 24000000)
    b = L.mgr_raster_workspace_bytes(8, 300000, 1920, 1080, 48000000)
    assert b - a >= 24000000 * 64                       # keys 8 + gid 4 + tag 4 + record 48 per pair
    assert L.mgr_knn3_workspace_bytes(300000) > 300000 * 16
    # bad sizes / null pointers are reported, never abort
    rc = L.mgr_raster_forward(0, 10, 64, 64, None, None, None, 0, None, 0, None, 0, None, 0, None, None, None, 0, 0, 0, None)
    assert rc == -1 and b"bad sizes" in L.mgr_last_error()
    rc = L.mgr_raster_forward(1, 10, 64, 64, None, None, None, 0, None, 0, None, 0, None, 0, None, None, None, 0, 100, 0, None)
    assert rc == -1 and b"null" in L.mgr_last_error()
    assert L.mgr_skin_weights_fwd(5, None, None, 4, 4, 4, 64, 64, None, None, None, None) == -1   # > MGR_MAX_BONES
    assert L.mgr_lbs_cov_fwd(0, 5, 21, None, None, None, None, None, None, None, None, None) == -1
    assert L.mgr_knn3_mean_dist2(-1, None, None, None, 0, None) == -1
    assert L.mgr_sh_color_fwd(1, 0, None, None, 0, None, 0, None, None, None) == 0          # N = 0 is a no-op


def test_ops_fail_loudly_without_gpu():
    import manus_amd
    from manus_amd._lib import ManusHipError
    from manus_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    x = torch.zeros(4, 3)
    with pytest.raises(ManusHipError):
        manus_amd.distCUDA2(x)
    with pytest.raises(ManusHipError):
        manus_amd.lbs_cov(x, x, torch.ones(4, 4), None, None)
    with pytest.raises(ManusHipError):
        manus_amd.skin_weights(x, torch.ones(2, 2, 2, 21), torch.zeros(3), torch.ones(3))
    st = GaussianRasterizationSettings(8, 8, 1.0, 1.0, torch.ones(3), 1, torch.eye(4), torch.eye(4), 3, torch.zeros(3), False, False)
    r = GaussianRasterizer(st)
    with pytest.raises(Exception, match="exc?a?c?tly one"):
        r(x, x, torch.ones(4, 1))
    with pytest.raises(ManusHipError):
        r(means3D=x, means2D=x, opacities=torch.ones(4, 1), colors_precomp=x, cov3D_precomp=torch.ones(4, 6))
    # the drop-in package names resolve to this implementation
    import diff_gaussian_rasterization as dgr
    import simple_knn._C as knn
    assert dgr.GaussianRasterizer is GaussianRasterizer and knn.distCUDA2 is manus_amd.distCUDA2
    assert dgr.GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "sh_degree", "campos", "prefiltered", "debug")


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "manus_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                code = re.sub(r"#.*", "", src)
                assert "oracle" not in code and "tools.parity" not in code and "import tools" not in code, f


# ---------------------------------------------------------------------------
# view sharding + gradient all-reduce, world_size 2, gloo, CPU
# ---------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_compute(scene):
    """compute_fn with the engine's contract, on the CPU torch oracle (LBS + SH only: the
    'image' is a fixed linear functional of the per-view outputs — enough to exercise sharding,
    packing and the reductions)."""
    from oracle import torch_ref as tr
    g = torch.Generator().manual_seed(7)
    N = scene["N"]
    R = [(torch.randn((N, 3), generator=g), torch.randn((N, 6), generator=g), torch.randn((N, 3), generator=g))
         for _ in range(len(scene["cameras"]))]

    def fn(view_ids, scale):
        P = {k: v.clone().requires_grad_(True) for k, v in scene["params"].items()}
        loss = 0.0
        g2 = torch.zeros(N); vis = torch.zeros(N); rad = torch.zeros(N, dtype=torch.int32)
        for v in view_ids:
            cc = torch.tensor(np.asarray(scene["cameras"][v]["camera_center"], np.float32))
            o = tr.hand_forward(P, scene["grid"], scene["grid_center"], scene["grid_scale"], scene["posed"][v], scene["rest"], cc)
            loss = loss + ((o["posed_xyz"] * R[v][0]).sum() + (o["posed_cov"] * R[v][1]).sum() * 1e3 + (o["colors"] * R[v][2]).sum()
                           + (o["opacity"][:, 0] * R[v][0][:, 0]).sum())
            g2 += o["posed_xyz"].detach().norm(dim=1)
            vis += 1
            rad = torch.maximum(rad, (o["posed_xyz"].detach()[:, 0].abs() * 1000).to(torch.int32))
        (loss * scale).backward()
        return dict(grads={k: p.grad for k, p in P.items()}, grad2d=g2, vis=vis, radii=rad, loss=(loss * scale).detach())
    return fn


def _sparse(fn, n):
    """Wrap a compute_fn so that two of every seven Gaussians receive no gradient at all (rows exactly zero, like
    Gaussians hidden behind saturated pixels), keeping the visibility count dense."""
    def g(view_ids, scale):
        o = fn(view_ids, scale)
        keep = (torch.arange(n) % 7) < 5
        for k in o["grads"]:
            o["grads"][k] = o["grads"][k] * keep.reshape((-1,) + (1,) * (o["grads"][k].dim() - 1))
        o["grad2d"] = o["grad2d"] * keep
        return o
    return g


NG = 301      # 59 * 301 is odd: the scatter layout needs its padding with two ranks


def _worker(rank, world, port, q, mode="dense"):
    compact = mode == "compact"
    sys.path.insert(0, ROOT)
    from manus_amd.engine import ViewShardedStep
    from manus_amd.synthetic import make_scene
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    sc = make_scene(n_gaussians=NG, kind="hand", seed=4, grid_res=12, n_cameras=5, width=32, height=32)
    shapes = {k: v.shape for k, v in sc["params"].items()}
    fn = _oracle_compute(sc)
    st = ViewShardedStep(NG, shapes, _sparse(fn, NG) if compact else fn, 5, rank=rank, world_size=world, compact=compact,
                         scatter=mode == "scatter")
    out = st.step()
    if compact:
        assert 0 < st.last_rows < NG             # fewer rows travelled than there are Gaussians
    if mode == "scatter":
        c = (59 * NG + 1) // 2
        assert st.padded_g == 2 * c and st.owned == (rank * c, min((rank + 1) * c, 59 * NG))   # the pad element has no owner
        # the parameter all-gather: every rank contributes its slice of a buffer laid out like the gradients
        buf = torch.full((st.padded_g,), float("nan"))
        buf[rank * c:(rank + 1) * c] = torch.arange(rank * c, (rank + 1) * c, dtype=torch.float32)
        st.all_gather_params(buf)
        assert torch.equal(buf, torch.arange(st.padded_g, dtype=torch.float32))
    assert float(out["overflow"]) == 0.0
    st.reduce_max_radii(out["radii"])        # the per-step collective carries sums only; the maximum is combined on demand
    if rank == 0:
        # numpy: by value (a torch tensor travels as a file descriptor the consumer must fetch while this process lives)
        q.put({k: ({n: g.detach().numpy().copy() for n, g in v.items()} if isinstance(v, dict) else v.detach().numpy().copy())
               for k, v in out.items()
               if v is not None})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["dense", "compact", "scatter"])
def test_view_sharded_step_gloo_world2(mode):
    compact = mode == "compact"
    from manus_amd.engine import GRAD_WIDTH, ViewShardedStep, shard_views
    from manus_amd.synthetic import make_scene
    assert GRAD_WIDTH == 59
    assert shard_views(5, 0, 2) == [0, 2, 4] and shard_views(5, 1, 2) == [1, 3]
    assert sorted(sum((shard_views(53, r, 8) for r in range(8)), [])) == list(range(53))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    got = {k: ({n: torch.from_numpy(np.asarray(g)) for n, g in v.items()} if isinstance(v, dict) else torch.from_numpy(np.asarray(v)))
           for k, v in got.items()}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # single-process reference: all 5 views on one rank
    sc = make_scene(n_gaussians=NG, kind="hand", seed=4, grid_res=12, n_cameras=5, width=32, height=32)
    shapes = {k: v.shape for k, v in sc["params"].items()}
    fn = _oracle_compute(sc)
    ref = ViewShardedStep(NG, shapes, _sparse(fn, NG) if compact else fn, 5).step()
    for k in ref["grads"]:
        assert got["grads"][k].shape == ref["grads"][k].shape
        assert torch.allclose(got["grads"][k], ref["grads"][k], rtol=1e-4, atol=1e-6 * float(ref["grads"][k].abs().max())), k
    assert torch.allclose(got["grad2d"], ref["grad2d"], rtol=1e-5)
    assert torch.equal(got["vis"], ref["vis"]) and torch.equal(got["radii"], ref["radii"])
    assert abs(float(got["loss"]) - float(ref["loss"])) < 1e-4 * max(1.0, abs(float(ref["loss"])))


def test_checkpoint_format_matches_reference_loader(golden_dir, tmp_path):
    """manus_amd.checkpoint against the reference's own load_checkpoint / find_best_checkpoint outputs
    (tests/golden/checkpoint.npz): NaN rows dropped from every leaf, prefix stripped, the same file chosen
    (including the reference's string comparison of steps / epochs)."""
    import numpy as np
    import torch
    from manus_amd import checkpoint as ck
    d = np.load(os.path.join(golden_dir, "checkpoint.npz"))
    leaves = ["_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity"]
    sd = {"model." + k: torch.tensor(d["in_" + k]) for k in leaves}
    names = [str(x) for x in d["names"]]
    torch.save({"epoch": 3, "global_step": 1200, "state_dict": sd, "extra_params": {"num_gaussians": 40}},
               str(tmp_path / names[0]))
    for nm in names[1:]:
        (tmp_path / nm).write_bytes(b"")
    w, e = ck.load_checkpoint(str(tmp_path / names[0]))
    assert e["num_gaussians"] == int(d["out_num_gaussians"]) == w["_xyz"].shape[0]
    for k in leaves:
        np.testing.assert_array_equal(w[k].numpy(), d["out_" + k])
    assert os.path.basename(ck.find_best_checkpoint(str(tmp_path), "epoch")) == str(d["best_epoch"])
    assert os.path.basename(ck.find_best_checkpoint(str(tmp_path), "loss")) == str(d["best_loss"])
    # round trip through the writer: the file name format and the keys the reference reads back
    params = {k: w[k] for k in leaves}
    grid = {"grid_scale": torch.ones(3), "grid_center": torch.zeros(3), "grid_points": torch.zeros(2, 3),
            "grid_weights": torch.rand(2, 2, 2, 21)}
    path = ck.save_checkpoint(str(tmp_path / "out"), params, epoch=7, step=2800, loss=0.0123456, grid=grid)
    assert os.path.basename(path) == "epoch=007-step=2800-loss=0.012346.ckpt"
    raw = torch.load(path, weights_only=False)
    assert sorted(raw["state_dict"]) == sorted("model." + k for k in leaves)
    assert raw["extra_params"]["num_gaussians"] == 37 and "grid_weights" in raw["extra_params"]
    w2, e2 = ck.load_checkpoint(path)
    for k in leaves:
        np.testing.assert_array_equal(w2[k].numpy(), d["out_" + k])
    assert ck.get_num_gaussians_from_checkpoint(path) == 37


def test_balanced_view_assignment_is_a_partition_and_balances_the_load():
    """shard_views with weights (balance by measured pairs per view): every view goes to exactly one rank,
    every rank computes the same assignment, and the heaviest rank carries at most the mean load plus one view."""
    from manus_amd.engine import shard_views, view_costs
    rng = np.random.default_rng(3)
    for n_views, world in ((8, 8), (53, 8), (7, 2), (5, 3), (3, 4)):
        pairs = rng.integers(200000, 5000000, size=n_views)
        w = view_costs(pairs, 300000)
        parts = [shard_views(n_views, r, world, w) for r in range(world)]
        assert sorted(v for p in parts for v in p) == list(range(n_views))
        loads = [sum(w[v] for v in p) for p in parts]
        assert max(loads) <= sum(w) / world + max(w) + 1e-6
        if n_views >= world:
            assert all(len(p) >= 1 for p in parts)
        # round-robin (no weights) is unchanged
        assert shard_views(n_views, 1 % world, world) == list(range(1 % world, n_views, world))
    # equal weights: as many views per rank as round-robin gives
    assert sorted(len(shard_views(53, r, 8, [1.0] * 53)) for r in range(8)) == [6, 6, 6, 7, 7, 7, 7, 7]


def test_checkpoint_loader_refuses_pickled_objects_unless_asked(tmp_path):
    """A file the safe unpickler rejects is the one that can run code while loading: full unpickling is opt-in."""
    import argparse
    from manus_amd import checkpoint as ck
    sd = {"model." + k: torch.zeros(s) for k, s in (("_xyz", (4, 3)), ("_features_dc", (4, 1, 3)), ("_features_rest", (4, 15, 3)),
                                                    ("_scaling", (4, 3)), ("_rotation", (4, 4)), ("_opacity", (4, 1)))}
    path = str(tmp_path / "epoch=000-step=1-loss=0.5.ckpt")
    torch.save({"epoch": 0, "global_step": 1, "state_dict": sd, "extra_params": {"num_gaussians": 4},
                "hyper_parameters": argparse.Namespace(lr=1.0)}, path)
    with pytest.raises(RuntimeError, match="allow_pickle"):
        ck.load_checkpoint(path)
    w, e = ck.load_checkpoint(path, allow_pickle=True)
    assert e["num_gaussians"] == 4 and w["_xyz"].shape == (4, 3)
    assert ck.get_num_gaussians_from_checkpoint(path, allow_pickle=True) == 4
    # a missing file is a missing file, not advice to unpickle
    with pytest.raises(FileNotFoundError):
        ck.load_checkpoint(str(tmp_path / "absent.ckpt"))


def test_nan_rows_leave_the_optimizer_state_of_a_checkpoint_too(tmp_path):
    from manus_amd import checkpoint as ck
    n = 5
    sd = {"model." + k: torch.rand((n,) + s) for k, s in (("_xyz", (3,)), ("_features_dc", (1, 3)), ("_features_rest", (15, 3)),
                                                          ("_scaling", (3,)), ("_rotation", (4,)), ("_opacity", (1,)))}
    sd["model._scaling"][2, 1] = float("nan")
    opt = {"exp_avg": {"_xyz": torch.arange(n * 3, dtype=torch.float32).reshape(n, 3)}, "steps": [3, 3], "denom": torch.arange(n, dtype=torch.float32)}
    path = str(tmp_path / "epoch=000-step=1-loss=0.5.ckpt")
    torch.save({"epoch": 0, "global_step": 1, "state_dict": sd, "extra_params": {"num_gaussians": n}, "manus_amd_optimizer": opt}, path)
    w, e, full = ck.load_checkpoint(path, return_checkpoint=True)
    assert e["num_gaussians"] == 4 and w["_xyz"].shape[0] == 4
    o = full["manus_amd_optimizer"]
    assert o["exp_avg"]["_xyz"].shape == (4, 3) and o["denom"].tolist() == [0.0, 1.0, 3.0, 4.0] and o["steps"] == [3, 3]


def test_sequence_container_is_read_lazily_and_opened_once(golden_dir, tmp_path):
    import shutil
    from manus_amd import dataset as D
    p = str(tmp_path / "grasp_2.npz")
    shutil.copy(os.path.join(golden_dir, "seq", "grasp_2.npz"), p)
    a, b = D.open_sequence(p), D.open_sequence(p)
    assert a is b                                             # one store per action file (and process)
    assert len(a._a._cache) == 0                              # nothing decompressed yet
    _ = a["frames"]["8"]["metadata"]["rest_matrixs"][:]
    assert list(a._a._cache) == ["frames/8/metadata/rest_matrixs"]
    # a forked child (a DataLoader worker) opens its own descriptor instead of sharing the parent's
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:
        try:
            c = D.open_sequence(p)
            ok = (c is not a) and c._z.fid is not a._z.fid and len(D._STORES) == 1
            os.write(w, b"1" if ok else b"0")
        finally:
            os._exit(0)
    os.waitpid(pid, 0)
    assert os.read(r, 1) == b"1"
    # a file rewritten under the same name replaces its stale store
    os.utime(p, (1, 1))
    assert D.open_sequence(p) is not a
    D.close_sequences()
    assert not D._STORES


def test_predicted_ms_rides_in_the_multi_gpu_line():
    """bench.predicted_ms (DESIGN 7's model) at the headline size: compute of views / rank + the dense exchange bounds."""
    import bench
    for n in (2, 4, 8):
        p = bench.predicted_ms(n, 300000, 8, "hand", 1920, 1080)
        assert (p["step_ms_low"] < p["step_ms_high"] or n == 2) and p["compute_ms"] == bench.COMPUTE_MS_BY_VIEWS[8 // n]
    assert bench.predicted_ms(1, 300000, 8, "hand", 1920, 1080) is None
    assert bench.predicted_ms(8, 30000, 8, "hand", 480, 270) is None
    # BASELINE config 4: composite, 500 k Gaussians, 53 cameras over 8 ranks (seven views on the fullest rank, 122 MB dense)
    p = bench.predicted_ms(8, 500000, 53, "composite", 1920, 1080)
    assert p["views_on_the_fullest_rank"] == 7 and p["compute_ms"] == bench.COMPOSITE_MS_BY_VIEWS[7] and p["exchange_bytes_dense"] == (61 * 500000 + 2) * 4
    assert p["step_ms_low"] < p["step_ms_high"]


def test_packed_camera_tables_are_cached_per_unmodified_tensor_object():
    """_lib.cached_pack: one build per set of source tensor OBJECTS at unchanged versions; an in-place write, another tensor of
    equal content, or a non-tensor source builds again."""
    from manus_amd import _lib
    calls = []

    def build():
        calls.append(1)
        return len(calls)

    a, b = torch.zeros(4), torch.ones(3)
    assert _lib.cached_pack([a, b], [0.5, "cpu"], build) == 1
    assert _lib.cached_pack([a, b], [0.5, "cpu"], build) == 1 and len(calls) == 1
    assert _lib.cached_pack([a, b], [0.25, "cpu"], build) == 2            # another scalar
    a.add_(1.0)                                                            # in-place write: the version moves
    assert _lib.cached_pack([a, b], [0.5, "cpu"], build) == 3
    assert _lib.cached_pack([a, b], [0.5, "cpu"], build) == 3
    assert _lib.cached_pack([a.clone(), b], [0.5, "cpu"], build) == 4     # equal content, another object
    assert _lib.cached_pack([a.numpy(), b], [0.5, "cpu"], build) == 5     # not a tensor: never cached
    assert _lib.cached_pack([a.numpy(), b], [0.5, "cpu"], build) == 6
