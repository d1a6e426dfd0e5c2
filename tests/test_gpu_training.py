"""GPU: the whole training step end to end -- multi-view render + fused loss + backward, fused Adam with the
schedule, densification statistics, densify/prune with optimizer-state surgery, opacity reset -- on a small
synthetic hand: the loss must go down, the Gaussian count must change at the densification steps, and the loop
must keep running on the re-allocated tensors."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_training_loop_learns_and_densifies():
    from manus_amd import rasterizer
    from manus_amd.engine import HipViewCompute, Trainer
    from manus_amd.synthetic import camera_table, make_scene
    torch.manual_seed(0)
    V, W, H = 3, 128, 96
    sc = make_scene(n_gaussians=6000, kind="hand", seed=4, grid_res=32, n_cameras=V, width=W, height=H, cam_radius=0.5,
                    sigma_range=(2e-3, 6e-3), device=DEV)
    ct = camera_table(sc["cameras"], DEV)
    # targets: the same hand with different colours / opacities -> something to learn
    g = torch.Generator(device="cpu").manual_seed(5)
    tgt_scene = dict(sc)
    tgt_scene["params"] = {k: (v + (1.5 * torch.randn(v.shape, generator=g).to(DEV) if k == "_features_dc" else 0))
                           for k, v in sc["params"].items()}
    with torch.no_grad():
        hp = HipViewCompute(tgt_scene, torch.zeros((V, 3, H, W), device=DEV), ct)
        targets = torch.cat([hp.forward_views([v])[0] for v in range(V)]).contiguous()
    compute = HipViewCompute(sc, targets, ct, loss="l1+ssim")
    # (no opacity reset inside the window: like in the reference it makes the image vanish and the loss jump)
    opts = dict(densify_from_step=10, densification_interval=10, densify_until_step=1000, opacity_reset_interval=100000,
                percent_dense=0.01, densify_grad_threshold=5e-5)
    tr = Trainer(compute, V, extent=0.3, opts=opts, spatial_lr_scale=0.05, bg_white=False)
    rasterizer.set_sync_policy(True)
    losses, counts = [], []
    for it in range(60):
        out = tr.train_step()
        losses.append(float(out["loss"]))
        counts.append(tr.opt.N)
        assert np.isfinite(losses[-1])
    assert len(set(counts)) > 1, "densify_and_prune never changed the number of Gaussians"
    print("loss every 5 steps:", [round(x, 4) for x in losses[::5]], "N:", counts[::10])
    # (every densification -- each 10 steps here -- replaces Gaussians by freshly split ones and the loss jumps, as in
    # the reference; between densifications it must fall, and end below where it started)
    assert min(losses) < 0.7 * losses[0] and losses[-1] < 0.9 * losses[0], (losses[0], losses[-5:], counts[::5])
    assert all(losses[k + 9] < losses[k] for k in (0, 11, 21, 31, 41))
    tr.opt.reset_opacity()                                     # and the reset itself leaves a usable state
    out = tr.train_step()
    assert np.isfinite(float(out["loss"]))
    for k, v in tr.compute.params.items():
        assert v.shape[0] == tr.opt.N and torch.isfinite(v).all(), k
    # Adam moments travelled with their rows: same count, finite
    assert tr.opt.m["_xyz"].shape[0] == tr.opt.N and torch.isfinite(tr.opt.v["_features_rest"]).all()


def test_trainer_sort_rows_is_the_same_model_up_to_the_row_order():
    """Trainer(sort_rows=True): after a densification the rows follow a Z-order curve of the positions; leaves, Adam moments
    and statistics are those of a Trainer without it, row for row under the returned permutation (bit-equal at the step of
    the densification); the steps after it see the same model (losses agree to rounding: sums over Gaussians in another order)."""
    from manus_amd import rasterizer
    from manus_amd.engine import HipViewCompute, Trainer
    from manus_amd.synthetic import camera_table, make_scene
    V, W, H = 3, 128, 96
    opts = dict(densify_from_step=5, densification_interval=5, densify_until_step=1000, opacity_reset_interval=100000,
                percent_dense=0.01, densify_grad_threshold=5e-5)
    rasterizer.set_sync_policy(True)

    def make(sort):
        torch.manual_seed(0)
        sc = make_scene(n_gaussians=5000, kind="hand", seed=4, grid_res=32, n_cameras=V, width=W, height=H, cam_radius=0.5,
                        sigma_range=(2e-3, 6e-3), device=DEV)
        ct = camera_table(sc["cameras"], DEV)
        tg = torch.rand((V, 3, H, W), device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
        return Trainer(HipViewCompute(sc, tg, ct, loss="l1+ssim"), V, extent=0.3, opts=opts, spatial_lr_scale=0.05,
                       bg_white=False, sort_rows=sort)

    a, b = make(False), make(True)
    perm = None
    for it in range(14):
        torch.manual_seed(100 + it)          # the split noise of both trainers
        oa = a.train_step()
        torch.manual_seed(100 + it)
        ob = b.train_step()
        if perm is None and "row_perm" in ob:
            perm = ob["row_perm"]
            assert a.opt.N == b.opt.N == perm.numel() and a.opt.N != 5000
            for k in a.opt.p:
                assert torch.equal(a.opt.p[k][perm], b.opt.p[k]), k
                assert torch.equal(a.opt.m[k][perm], b.opt.m[k]) and torch.equal(a.opt.v[k][perm], b.opt.v[k]), k
            assert torch.equal(a.opt.max_radii2D[perm], b.opt.max_radii2D) and torch.equal(a.opt.denom[perm], b.opt.denom)
            x = b.opt.p["_xyz"]
            # Z-order: consecutive rows are close (median step well under the model's extent; shuffled rows are not)
            step_sorted = (x[1:] - x[:-1]).norm(dim=1).median()
            step_given = (a.opt.p["_xyz"][1:] - a.opt.p["_xyz"][:-1]).norm(dim=1).median()
            assert float(step_sorted) < 0.25 * float(step_given)
        assert abs(float(oa["loss"]) - float(ob["loss"])) <= 2e-5 * abs(float(oa["loss"])), (it, float(oa["loss"]), float(ob["loss"]))
    assert perm is not None
    assert "row_perm" not in oa


def test_trainer_follows_reference_step_order_and_prunes_by_mask():
    """on_after_backward before optimizer.step(): at step 0 (< remove_seg_end) Gaussians that project outside the
    segmentation mask are pruned and that step's Adam update is skipped for every (replaced) leaf; the following steps
    update all leaves; the keypoint-distance test runs at step 100.  Overflow fencing: the first step starts from the
    default pair capacity, which a close-up scene exceeds, and must be re-run transparently."""
    from manus_amd.engine import HipViewCompute, Trainer
    from manus_amd.synthetic import camera_table, make_masks, make_scene
    torch.manual_seed(0)
    V, W, H, n = 2, 128, 96, 5000
    sc = make_scene(n_gaussians=n, kind="hand", seed=8, grid_res=32, n_cameras=V, width=W, height=H, cam_radius=0.5,
                    sigma_range=(4e-3, 1.2e-2), device=DEV)
    ct = camera_table(sc["cameras"], DEV)
    hp = HipViewCompute(sc, torch.zeros((V, 3, H, W), device=DEV), ct)
    with torch.no_grad():
        pxyz, _, _ = hp._posed(sc["transforms"][:V])
        targets = torch.cat([hp.forward_views([v])[0] for v in range(V)]).contiguous()
    # masks: discs of 6 px around the projected keypoints (dilated by another 5 px in the test itself): the Gaussians
    # between the bones project outside them; the keypoints themselves are inside, so the test is not disabled
    # (a keypoint outside the mask turns it off, gaussian_utils.py:125-131)
    sc["masks"] = make_masks(sc, sc["keypoints"][:V], margin=6).to(DEV)
    from oracle import torch_ref as tr
    want = torch.zeros(n, dtype=torch.bool)
    for v in range(V):
        c = sc["cameras"][v]
        want |= tr.points_outside_mask(pxyz[v].cpu(), torch.tensor(c["K"], dtype=torch.float32),
                                       torch.tensor(c["extr"], dtype=torch.float32), sc["masks"][v].cpu(),
                                       sc["keypoints"][v].cpu(), dilate=True)[:, 0]
    assert 10 < int(want.sum()) < n // 4, int(want.sum())
    compute = HipViewCompute(sc, targets, ct, loss="l1+ssim")
    tr_ = Trainer(compute, V, extent=0.3, opts=dict(remove_seg_end=1, densify_from_step=1000), spatial_lr_scale=0.05)
    p0 = {k: v.detach().clone() for k, v in compute.params.items()}
    out = tr_.train_step()
    assert out["changed"] and tr_.global_step == 1
    n1 = tr_.opt.N
    assert abs(n1 - (n - int(want.sum()))) <= 2, (n1, int(want.sum()))   # exactly the Gaussians outside the masks (+- boundary ties)
    # the step that pruned made no Adam update: survivors are bit-identical to initial rows, no moment was touched
    assert tr_.opt.group_step["xyz"] == 0 and not tr_.opt.m["_xyz"].any()
    surv = compute.params["_xyz"].detach()
    assert (surv[::50, None, :] == p0["_xyz"][None, :, :]).all(-1).any(1).all()
    out = tr_.train_step()
    assert not out["changed"] and tr_.opt.group_step["xyz"] == 1 and tr_.opt.N == n1
    assert tr_.opt.m["_xyz"].abs().sum() > 0
    assert float(tr_.opt.denom.sum()) > 0                 # statistics accumulate on ordinary steps
    for _ in range(3):
        tr_.train_step()
    assert tr_.opt.group_step == {k: 4 for k in tr_.opt.group_step}
    # keypoint-distance test (every 100 steps past remove_seg_end): nothing is 20 cm from the hand -> plain step
    tr_.global_step = 100
    out = tr_.train_step()
    assert tr_.opt.N == n1 and np.isfinite(float(out["loss"]))
    # white background: step densify_from_step resets the opacity (and only the opacity group misses its Adam step)
    tr_.opt.opts["densify_from_step"] = 101
    gs0 = dict(tr_.opt.group_step)
    out = tr_.train_step()
    assert out["changed"] and tr_.opt.N == n1
    assert tr_.opt.group_step["opacity"] == gs0["opacity"] and tr_.opt.group_step["xyz"] == gs0["xyz"] + 1
    assert float(torch.sigmoid(tr_.opt.p["_opacity"]).max()) <= 0.0100001


def test_trainer_reruns_a_step_that_overflowed_the_pair_capacity(monkeypatch):
    """No-sync mode: the step is fenced; an overflow (here forced by a tiny default capacity) discards the step and
    re-runs it with the enlarged capacity -- the result equals a run that never overflowed."""
    from manus_amd import rasterizer as rz
    from manus_amd.engine import HipViewCompute, Trainer
    from manus_amd.synthetic import camera_table, make_scene
    V, W, H, n = 2, 96, 64, 3000
    sc = make_scene(n_gaussians=n, kind="hand", seed=3, grid_res=24, n_cameras=V, width=W, height=H, cam_radius=0.5,
                    sigma_range=(2e-3, 8e-3), device=DEV)
    ct = camera_table(sc["cameras"], DEV)
    tg = torch.rand((V, 3, H, W), device=DEV)
    res = []
    for tiny in (False, True):
        rz.context().clear()
        if tiny:
            monkeypatch.setattr(rz, "default_pair_capacity", lambda V, N: 4096)
        tr_ = Trainer(HipViewCompute(sc, tg, ct, loss="l1"), V, extent=0.3, opts=dict(remove_seg_end=0))
        out = tr_.train_step()
        res.append((tr_.retries, float(out["loss"]), {k: v.detach().clone() for k, v in tr_.compute.params.items()}))
    assert res[0][0] == 0 and res[1][0] >= 1
    assert abs(res[0][1] - res[1][1]) < 1e-6          # (the L1 loss value is summed with float atomics; gradients are not)
    for k in res[0][2]:
        assert torch.equal(res[0][2][k], res[1][2][k]), k


def test_training_loop_config5_shape():
    """BASELINE config 5 in miniature (the full 30-scene sweep on 8 GPUs is not runnable here): the whole training loop
    with densify / prune active, fp16 SH storage, several views per step at a quarter of 1080p, 40 k Gaussians: the loss
    falls, the model grows at the densification steps (with an opacity reset in between), the fp16 copy tracks the fp32
    leaves, everything stays finite."""
    from manus_amd.engine import HipViewCompute, Trainer
    from manus_amd.synthetic import camera_table, make_scene
    torch.manual_seed(0)
    V, W, H, n = 4, 960, 540, 40000
    sc = make_scene(n_gaussians=n, kind="hand", seed=11, grid_res=64, n_cameras=V, width=W, height=H, cam_radius=0.8,
                    sigma_range=(1e-3, 4e-3), device=DEV)
    ct = camera_table(sc["cameras"], DEV)
    g = torch.Generator(device="cpu").manual_seed(5)
    tgt_scene = dict(sc)
    tgt_scene["params"] = {k: (v + (1.0 * torch.randn(v.shape, generator=g).to(DEV) if k == "_features_dc" else 0))
                           for k, v in sc["params"].items()}
    with torch.no_grad():
        targets = HipViewCompute(tgt_scene, torch.zeros((V, 3, H, W), device=DEV), ct).forward_views_fused(list(range(V)))[0]
    compute = HipViewCompute(sc, targets.contiguous(), ct, loss="l1+ssim", sh_storage="fp16")
    opts = dict(remove_seg_end=0, densify_from_step=10, densification_interval=10, densify_until_step=1000,
                opacity_reset_interval=35, percent_dense=0.01, densify_grad_threshold=2e-5)
    tr_ = Trainer(compute, V, extent=0.3, opts=opts, spatial_lr_scale=0.05, bg_white=False)
    losses, counts = [], []
    for _ in range(48):
        out = tr_.train_step()
        losses.append(float(out["loss"]))
        counts.append(tr_.opt.N)
    assert all(np.isfinite(losses)) and min(losses) < 0.8 * losses[0]
    assert max(counts) > n and len(set(counts)) >= 3, counts[::6]   # several densifications changed the model
    assert compute._sh_copy.shape[0] == tr_.opt.N
    compute(list(range(V)), 1.0 / V)          # refreshes the copy from the leaves the optimizer just updated
    assert torch.equal(compute._sh_copy[:, :45].float(),
                       tr_.opt.p["_features_rest"].reshape(tr_.opt.N, 45).half().float())
    for k, v in tr_.opt.p.items():
        assert torch.isfinite(v).all(), k


def test_training_loop_config5_full_size():
    """BASELINE config 5 at the size of one scene of the sweep: 300 000 Gaussians, 8 views of 1920x1080 per step, fp16
    SH storage, the reference's step order with a mask prune at step 0, densifications at steps 3 and 6 (clone + split
    + prune with optimizer-state surgery) and an opacity reset: every step finite, the model shrinks at the prune and
    grows at the densifications, the workspace follows the new sizes, the loss after the first densification is below
    the initial one."""
    from manus_amd import rasterizer
    from manus_amd.engine import HipViewCompute, Trainer
    from manus_amd.synthetic import camera_table, make_masks, make_scene
    torch.manual_seed(0)
    V, W, H, n = 8, 1920, 1080, 300000
    sc = make_scene(n_gaussians=n, kind="hand", seed=0, n_cameras=V, width=W, height=H, device=DEV)
    ct = camera_table(sc["cameras"], DEV)
    g = torch.Generator(device="cpu").manual_seed(5)
    tgt_scene = dict(sc)
    tgt_scene["params"] = {k: (v + (0.5 * torch.randn(v.shape, generator=g).to(DEV) if k == "_features_dc" else 0))
                           for k, v in sc["params"].items()}
    with torch.no_grad():
        hp = HipViewCompute(tgt_scene, torch.zeros((V, 3, H, W), device=DEV), ct)
        targets = hp.forward_views_fused(list(range(V)))[0].contiguous()
        del hp
    rasterizer.context().clear()
    sc["masks"] = make_masks(sc, sc["keypoints"][:V], margin=40).to(DEV)     # discs around the keypoints: some Gaussians fall outside
    compute = HipViewCompute(sc, targets, ct, loss="l1+ssim", sh_storage="fp16")
    opts = dict(remove_seg_end=1, densify_from_step=2, densification_interval=3, densify_until_step=1000,
                opacity_reset_interval=5, percent_dense=0.01, densify_grad_threshold=2e-6)
    tr_ = Trainer(compute, V, extent=0.3, opts=opts, spatial_lr_scale=0.05, bg_white=True)
    losses, counts, changed = [], [], []
    for _ in range(9):
        out = tr_.train_step()
        losses.append(float(out["loss"]))
        counts.append(tr_.opt.N)
        changed.append(bool(out["changed"]))
    assert all(np.isfinite(losses)), losses
    assert counts[0] < n, counts                     # step 0: the mask test pruned
    assert counts[3] > counts[2] and counts[6] > counts[5], counts     # densifications at global steps 3 and 6
    assert changed[5], changed                       # the opacity reset at step 5
    # (the prune of step 0 removes part of the hand: the loss jumps there, like in the reference; then it must fall)
    assert losses[2] < losses[1] and np.isfinite(losses[-1]), losses
    for k, v in tr_.opt.p.items():
        assert v.shape[0] == tr_.opt.N and torch.isfinite(v).all(), k
    assert compute._sh_copy is None or compute._sh_copy.shape[0] in (tr_.opt.N, counts[-2], counts[-1])
    out = compute(list(range(V)), 1.0 / V)           # one more plain step on the final model
    assert torch.isfinite(out["loss"]) and compute._sh_copy.shape[0] == tr_.opt.N
    print("N:", counts, "loss:", [round(x, 5) for x in losses])


def test_sequence_dataset_drives_the_trainer(golden_dir, tmp_path):
    """SURVEY 8 f4 end to end: `dataset.SequenceDataset` over the capture schema (tests/golden/seq) -> `view_batch` ->
    `engine.HipViewCompute` -> `engine.Trainer.train_step`, with the mask prune of step 0 running on the DATASET's own
    masks, cameras and keypoints (hand_dynamic.py:193-224)."""
    import os
    import shutil
    from manus_amd import dataset as D
    from manus_amd.engine import HipViewCompute, Trainer
    from manus_amd.synthetic import camera_table
    for f in os.listdir(os.path.join(golden_dir, "seq")):
        shutil.copy(os.path.join(golden_dir, "seq", f), tmp_path / f)
    cfg = dict(resize_factor=1.0, bg_color="white", subject="s1", width=64, height=48, rand_views_per_timestep=-1, n_bones=20,
               num_time_steps=-1, split_ratio=1.0, sequences=["grasp_2"], split_by_action=False)
    ds = D.SequenceDataset(str(tmp_path), cfg, "train")
    ids = [0, 1, 2, 3]                                   # the four cameras of the first frame
    batch = ds.view_batch(ids)
    scene, targets = D.hand_scene_from_batch(batch, ds[0]["bones_rest"], 3000, grid_res=24, seed=1, device=DEV)
    assert targets.shape == (4, 3, 48, 64) and scene["masks"].shape == (4, 48, 64)
    ct = camera_table(scene["cameras"], DEV)
    compute = HipViewCompute(scene, targets, ct, loss="l1+ssim")
    opts = dict(remove_seg_end=1, densify_from_step=2, densification_interval=2, densify_until_step=100, opacity_reset_interval=100000,
                percent_dense=0.01, densify_grad_threshold=2e-5)
    tr = Trainer(compute, 4, extent=float(ds.extent), opts=opts, spatial_lr_scale=0.05, bg_white=True)
    n0 = tr.opt.N
    out = tr.train_step()                                # step 0: mask test on the dataset's masks
    assert np.isfinite(float(out["loss"]))
    assert tr.opt.N <= n0
    losses = [float(tr.train_step()["loss"]) for _ in range(6)]
    assert all(np.isfinite(x) for x in losses)
    for k, v in tr.compute.params.items():
        assert v.shape[0] == tr.opt.N and torch.isfinite(v).all(), k


def test_composite_trainer_with_default_opts_never_touches_density_control():
    """composite.py has no on_after_backward / density_update (its training_step is `pass`, composite.py:80-81): a Trainer
    over a composite scene -- default opts, masks present -- only renders, reduces and takes the Adam step, past the
    default densify_from_step / densification_interval (advisor finding of round 2: it used to densify at step 200 and
    then raise with the optimizer, the compute object and n_art out of step)."""
    from manus_amd.engine import HipViewCompute, Trainer
    from manus_amd.synthetic import camera_table, make_masks, make_scene
    V, W, H = 2, 96, 64
    sc = make_scene(n_gaussians=3000, kind="composite", seed=3, grid_res=24, n_cameras=V, width=W, height=H, cam_radius=0.5,
                    sigma_range=(3e-3, 9e-3), device=DEV)
    sc["masks"] = make_masks(sc, sc["params"]["_xyz"].detach(), margin=2).to(DEV)     # tight masks: a hand/object trainer would prune
    ct = camera_table(sc["cameras"], DEV)
    compute = HipViewCompute(sc, torch.rand((V, 3, H, W), device=DEV), ct, loss="l1+ssim")
    tr = Trainer(compute, V, extent=0.3, spatial_lr_scale=0.05, bg_white=True)
    assert tr.density_enabled is False
    n0, na0 = tr.opt.N, compute.n_art
    l0 = None
    for it in range(205):
        out = tr.train_step()
        assert out["changed"] is False and tr.opt.N == n0 and compute.n_art == na0
        l0 = float(out["loss"]) if l0 is None else l0
    assert tr.global_step == 205 and np.isfinite(float(out["loss"])) and float(out["loss"]) < l0


def test_checkpoint_carries_the_optimizer_state(tmp_path):
    """save_checkpoint(optimizer=...) -> load_checkpoint(return_checkpoint=True) -> load_state_dict: the resumed trainer
    takes bit-identical steps (moments, per-group step counts, statistics restored)."""
    from manus_amd import checkpoint as ck
    from manus_amd.engine import HipViewCompute, Trainer
    from manus_amd.synthetic import camera_table, make_scene
    V, W, H = 2, 96, 64
    opts = dict(densify_from_step=100000, densify_until_step=0, opacity_reset_interval=100000, remove_seg_end=0)

    def build(params=None):
        sc = make_scene(n_gaussians=2500, kind="object", seed=5, grid_res=24, n_cameras=V, width=W, height=H, cam_radius=0.5,
                        sigma_range=(3e-3, 9e-3), device=DEV)
        if params is not None:
            sc["params"] = {k: v.to(DEV) for k, v in params.items()}
        ct = camera_table(sc["cameras"], DEV)
        tg = torch.rand((V, 3, H, W), device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
        return Trainer(HipViewCompute(sc, tg, ct, loss="l1+ssim"), V, extent=0.3, opts=opts, spatial_lr_scale=0.05, kind="object")

    a = build()
    for _ in range(3):
        a.train_step()
    path = ck.save_checkpoint(str(tmp_path), a.opt.p, epoch=0, step=a.global_step, loss=0.5, optimizer=a.opt)
    w, extra, full = ck.load_checkpoint(path, return_checkpoint=True)
    b = build({k: w[k] for k in a.opt.p})
    b.opt.load_state_dict(full["manus_amd_optimizer"])
    b.global_step = int(full["global_step"])
    for _ in range(2):
        a.train_step()
        b.train_step()
    for k in a.opt.p:
        assert torch.equal(a.opt.p[k], b.opt.p[k]), k
        assert torch.equal(a.opt.m[k], b.opt.m[k]), k


def test_eval_trajectory_dataset_renders_through_the_kernels(golden_dir):
    """SURVEY 8 f4, evaluation half: `dataset.TestDataset` (brics_dynamic.py:485-696) over the reference's own camera
    path and novel-pose skeleton (thinned: tests/golden/eval_inputs) -> `view_batch` -> `engine.HipViewCompute` at the
    reference's 1080 x 1080; the hand is in frame from every camera and the fused and per-operator routes agree."""
    import os
    from manus_amd import dataset as D
    from manus_amd.engine import HipViewCompute
    from manus_amd.synthetic import camera_table
    ind = os.path.join(golden_dir, "eval_inputs")
    ds = D.TestDataset(dict(cam_path=os.path.join(ind, "camera_path.npz"), cano_cam_path=os.path.join(ind, "cano_camera.npz"),
                            metadata_path=os.path.join(ind, "novel_pose.npz"), frame_sample_rate=2))
    ids = [0, 3, len(ds) - 1]
    batch = ds.view_batch(ids)
    scene, targets = D.hand_scene_from_batch(batch, ds.bones_rest, 20000, grid_res=24, seed=2, device=DEV)
    assert targets is None and (scene["width"], scene["height"]) == (1080, 1080)
    ct = camera_table(scene["cameras"], DEV)
    blank = torch.zeros((len(ids), 3, 1080, 1080), device=DEV)
    with torch.no_grad():
        im_m, rad_m, _ = HipViewCompute(scene, blank, ct, fused=False).forward_views([0, 1, 2])
        im_f, rad_f = HipViewCompute(scene, blank, ct, fused=True).forward_views_fused([0, 1, 2])
    assert torch.equal(rad_m, rad_f)
    d = (im_m - im_f).abs()
    assert float(d.max()) < 5e-3 and float(d.mean()) < 2e-6
    for v in range(len(ids)):
        covered = (im_f[v] < 0.999).any(0).float().mean()
        assert 0.002 < float(covered) < 0.9, (v, float(covered))          # the hand, on a white background
        assert int((rad_f[v] > 0).sum()) > 1000
