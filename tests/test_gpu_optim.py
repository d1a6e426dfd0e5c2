"""GPU: fused Adam step, densify_and_prune and reset_opacity (manus_amd.optim over csrc/optim.hip) against
the reference's own GaussianModel run (golden vectors) and the oracle restatement at full size."""
import os

import numpy as np
import pytest
import torch

from oracle import torch_ref as tr

from util import max_rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling",
        "rotation": "_rotation"}


def _model(d, tag, percent_dense):
    from manus_amd.optim import GaussianOptimizer
    params = {ATTR[n]: torch.tensor(d[f"{tag}_{n}"], device=DEV) for n in tr.LEAVES}
    return GaussianOptimizer(params, opts={"percent_dense": percent_dense}, spatial_lr_scale=float(d["spatial_lr_scale"]),
                             skin_weights=torch.tensor(d[f"{tag}_skin"], device=DEV))


@pytest.mark.parametrize("name", ["optimizer_s0.npz", "optimizer_s1.npz", "optimizer_s2.npz", "optimizer_s3.npz"])
def test_matches_reference_run(golden_dir, name):
    """Same calls, in the same order, as the reference run recorded by make_optimizer_golden."""
    d = np.load(os.path.join(golden_dir, name))
    go = _model(d, "init", float(d["percent_dense"]))
    for k in range(int(d["K"])):
        lr = go.update_learning_rate(int(d[f"step{k}"]))
        assert abs(lr - d["lrs"][k][0]) <= 1e-6 * d["lrs"][k][0]
        go.step({ATTR[n]: torch.tensor(d[f"grad{k}_{n}"], device=DEV) for n in tr.LEAVES})
    for n in tr.LEAVES:   # fp32 Adam: parameters to 2e-6 relative of the tensor's scale, moments likewise
        assert max_rel_err(go.p[ATTR[n]].cpu().numpy(), d[f"adam_{n}"]) < 2e-6, n
        assert max_rel_err(go.m[ATTR[n]].cpu().numpy(), d[f"adam_{n}_m"]) < 2e-6, n
        assert max_rel_err(go.v[ATTR[n]].cpu().numpy(), d[f"adam_{n}_v"]) < 2e-6, n
    # continue from the reference's exact post-Adam state so that the integer outcome (who is cloned / split /
    # pruned) is compared on identical inputs
    go = _model(d, "adam", float(d["percent_dense"]))
    for n in tr.LEAVES:
        go.m[ATTR[n]] = torch.tensor(d[f"adam_{n}_m"], device=DEV)
        go.v[ATTR[n]] = torch.tensor(d[f"adam_{n}_v"], device=DEV)
    go.state_step = int(d["K"])
    go.xyz_gradient_accum = torch.tensor(d["stat_accum"], device=DEV)
    go.denom = torch.tensor(d["stat_denom"], device=DEV)
    go.max_radii2D = torch.tensor(d["stat_maxrad"], device=DEV)
    std = d["split_std"]
    noise = torch.tensor(d["split_samples"] / std, device=DEV) if std.size else None
    size_thr = float(d["size_threshold"]) or None
    info = go.densify_and_prune(0.0002, 0.005, float(d["extent"]), size_thr, noise=noise)
    assert info["total"] == d["dens_xyz"].shape[0] and info["split_selected"] * 2 == d["split_samples"].shape[0]
    for n in tr.LEAVES:
        got = go.p[ATTR[n]].cpu().numpy()
        assert got.shape == d[f"dens_{n}"].shape, n
        assert max_rel_err(got, d[f"dens_{n}"]) < 2e-6, n
        np.testing.assert_array_equal(go.m[ATTR[n]].cpu().numpy(), d[f"dens_{n}_m"])   # copied or zero: exact
        np.testing.assert_array_equal(go.v[ATTR[n]].cpu().numpy(), d[f"dens_{n}_v"])
    np.testing.assert_array_equal(go.skin_weights.cpu().numpy(), d["dens_skin"])
    assert not go.xyz_gradient_accum.any() and not go.denom.any() and not go.max_radii2D.any()
    go.reset_opacity()
    assert max_rel_err(go.p["_opacity"].cpu().numpy(), d["reset_opacity"]) < 2e-6
    assert not go.m["_opacity"].any() and not go.v["_opacity"].any()
    np.testing.assert_array_equal(go.m["_xyz"].cpu().numpy(), d["reset_xyz_m"])


def _random_state(n, g, nb=21):
    rn = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    st = {"xyz": rn(n, 3, sc=0.05), "f_dc": rn(n, 1, 3), "f_rest": rn(n, 15, 3, sc=0.1), "opacity": rn(n, 1, sc=2.5),
          "scaling": torch.rand(n, 3, generator=g) * 4.0 - 8.5, "rotation": rn(n, 4)}
    w = torch.rand(n, nb, generator=g)
    st["skin"] = w / w.sum(1, keepdim=True)
    return st


@pytest.mark.parametrize("n", [1, 5, 1023, 1025, 300000])
def test_adam_and_densify_match_oracle(n):
    """Ragged sizes (scan-block seams at 1024, odd counts that break float4 alignment) and the bench size."""
    from manus_amd.optim import GaussianOptimizer
    g = torch.Generator().manual_seed(n)
    st = _random_state(n, g)
    go = GaussianOptimizer({ATTR[k]: st[k].to(DEV) for k in tr.LEAVES}, opts={"percent_dense": 0.01},
                           spatial_lr_scale=1.0, skin_weights=st["skin"].to(DEV))
    ref = {k: st[k].clone() for k in tr.LEAVES}
    for k in tr.LEAVES:
        ref[k + "_m"], ref[k + "_v"] = torch.zeros_like(st[k]), torch.zeros_like(st[k])
    ref["skin"] = st["skin"]
    opts = dict(position_lr_init=0.0016, position_lr_final=0.0000016, position_lr_max_steps=30000, feature_lr=0.0025,
                opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001)
    for t in (1, 2):
        grads = {k: torch.randn(st[k].shape, generator=g) * 1e-3 for k in tr.LEAVES}
        go.update_learning_rate(100 * t)
        go.step({ATTR[k]: grads[k].to(DEV) for k in tr.LEAVES})
        for k, lr in zip(tr.LEAVES, tr.group_lrs(opts, 1.0, 100 * t)):
            ref[k], ref[k + "_m"], ref[k + "_v"] = tr.adam_step(ref[k], grads[k], ref[k + "_m"], ref[k + "_v"], lr, t)
    for k in tr.LEAVES:
        assert max_rel_err(go.p[ATTR[k]].cpu().numpy(), ref[k].numpy()) < 2e-6, k
        assert max_rel_err(go.v[ATTR[k]].cpu().numpy(), ref[k + "_v"].numpy()) < 2e-6, k
    # densify from the oracle's state (identical inputs on both sides)
    for k in tr.LEAVES:
        go.p[ATTR[k]], go.m[ATTR[k]], go.v[ATTR[k]] = ref[k].to(DEV), ref[k + "_m"].to(DEV), ref[k + "_v"].to(DEV)
    ref["scaling"][: max(1, n // 50), 0] = -2.5           # exp(-2.5) = 0.082 > 0.1 * extent: pruned only when size_thr is set
    go.p["_scaling"] = ref["scaling"].to(DEV)
    accum = torch.rand(n, 1, generator=g) * 8e-4
    denom = torch.randint(0, 4, (n, 1), generator=g).float()
    go.xyz_gradient_accum, go.denom = accum.to(DEV), denom.to(DEV)
    extent = 0.5
    grads_n = torch.nan_to_num(accum / denom, nan=0.0, posinf=float("inf")).reshape(-1)
    n_sel = int(((grads_n >= 0.0002) & (ref["scaling"].exp().max(1).values > 0.01 * extent)).sum())
    noise = torch.randn(2 * n_sel, 3, generator=g)
    size_thr = 20 if n % 2 else None                     # both branches of `if max_screen_size:` (gaussian.py:316)
    want = tr.densify_and_prune(ref, accum, denom, 0.0002, 0.005, extent, 0.01, noise, max_screen_size=size_thr)
    info = go.densify_and_prune(0.0002, 0.005, extent, size_thr, noise=noise.to(DEV))
    assert info["split_selected"] == n_sel and info["total"] == want["xyz"].shape[0]
    assert info["kept"] + info["cloned"] + 2 * info["split_kept"] == info["total"]
    for k in tr.LEAVES:
        if info["total"] == 0:
            continue
        assert max_rel_err(go.p[ATTR[k]].cpu().numpy(), want[k].numpy()) < 2e-6, k
        np.testing.assert_array_equal(go.m[ATTR[k]].cpu().numpy(), want[k + "_m"].numpy())
    if info["total"]:
        np.testing.assert_array_equal(go.skin_weights.cpu().numpy(), want["skin"].numpy())


def test_sort_rows_keeps_the_articulated_rows_in_front():
    """GaussianOptimizer.sort_rows on a composite model (the first n_art rows are skinned, skin_weights has rows for those
    only): each segment is sorted along the Z-order curve on its own, every array moves with its rows, and an empty
    model returns an empty permutation."""
    from manus_amd.optim import GaussianOptimizer
    g = torch.Generator().manual_seed(4)
    N, na = 3000, 1800
    p = {"_xyz": torch.rand(N, 3, generator=g), "_features_dc": torch.rand(N, 1, 3, generator=g), "_features_rest": torch.rand(N, 15, 3, generator=g),
         "_opacity": torch.rand(N, 1, generator=g), "_scaling": torch.rand(N, 3, generator=g), "_rotation": torch.rand(N, 4, generator=g)}
    skin = torch.rand(na, 21, generator=g)
    go = GaussianOptimizer({k: v.to(DEV) for k, v in p.items()}, skin_weights=skin.to(DEV))
    go.m["_xyz"].copy_(p["_xyz"].to(DEV) * 2)
    go.max_radii2D.copy_(torch.arange(N, dtype=torch.float32))
    perm = go.sort_rows().cpu()                       # n_art from the rows skin_weights covers
    assert sorted(perm.tolist()) == list(range(N))
    assert bool((perm[:na] < na).all()) and bool((perm[na:] >= na).all())
    for k in p:
        assert torch.equal(go.p[k].cpu(), p[k][perm]), k
    assert torch.equal(go.m["_xyz"].cpu(), p["_xyz"][perm] * 2)
    assert torch.equal(go.skin_weights.cpu(), skin[perm[:na]])
    assert torch.equal(go.max_radii2D.cpu(), perm.float())
    assert go.replaced == frozenset(ATTR)
    # along the curve: the codes of each segment ascend
    def code(x):
        lo, hi = p["_xyz"].min(0).values, p["_xyz"].max(0).values
        q = ((x - lo) / (hi - lo).clamp_min(1e-12) * 1023.0).long().clamp_(0, 1023)
        c = torch.zeros(x.shape[0], dtype=torch.long)
        for bit in range(10):
            for ax in range(3):
                c |= ((q[:, ax] >> bit) & 1) << (3 * bit + ax)
        return c
    c = code(go.p["_xyz"].cpu())
    assert bool((c[1:na] >= c[:na - 1]).all()) and bool((c[na + 1:] >= c[na:-1]).all())
    with pytest.raises(Exception):
        go.sort_rows(n_art=N + 1)
    empty = GaussianOptimizer({k: v[:0].to(DEV) for k, v in p.items()})
    assert empty.sort_rows().numel() == 0


def test_add_densification_stats_kernel():
    """mgr_add_densification_stats = the reference's three updates (gaussian.py:335-338, gaussian_utils.py:470-473) in one
    launch: exactly the torch expressions, for the dtypes the step hands over and for others."""
    from manus_amd.optim import GaussianOptimizer
    g = torch.Generator().manual_seed(9)
    N = 5003
    p = {"_xyz": torch.rand(N, 3, generator=g), "_features_dc": torch.rand(N, 1, 3, generator=g), "_features_rest": torch.rand(N, 15, 3, generator=g),
         "_opacity": torch.rand(N, 1, generator=g), "_scaling": torch.rand(N, 3, generator=g), "_rotation": torch.rand(N, 4, generator=g)}
    go = GaussianOptimizer({k: v.to(DEV) for k, v in p.items()})
    acc, den, mx = torch.zeros(N, 1), torch.zeros(N, 1), torch.zeros(N)
    for it in range(3):
        g2 = torch.rand(N, generator=g) * (torch.rand(N, generator=g) < 0.4)
        vis = torch.randint(0, 9, (N,), generator=g).float()
        rad = torch.randint(0, 60, (N,), generator=g).to(torch.int32)
        acc += g2.reshape(-1, 1); den += vis.reshape(-1, 1); mx = torch.maximum(mx, rad.float())
        if it == 1:      # other dtypes / shapes are converted, not misread
            go.add_densification_stats(g2.double().to(DEV).reshape(-1, 1), vis.to(torch.int64).to(DEV), rad.float().to(DEV))
        else:
            go.add_densification_stats(g2.to(DEV), vis.to(DEV), rad.to(DEV))
    assert torch.equal(go.xyz_gradient_accum.cpu(), acc) and torch.equal(go.denom.cpu(), den) and torch.equal(go.max_radii2D.cpu(), mx)
    assert go.xyz_gradient_accum.shape == (N, 1) and go.max_radii2D.shape == (N,)
    # an accumulator of another size (a caller's tensor, a missed resize) is refused, not written out of bounds
    from manus_amd._lib import ManusHipError
    go.denom = go.denom[: N - 7].clone()
    with pytest.raises(ManusHipError):
        go.add_densification_stats(g2.to(DEV), vis.to(DEV), rad.to(DEV))
