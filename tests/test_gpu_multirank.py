"""GPU: the multi-rank training path with two processes on one device (gloo stands in for RCCL here; bench.py and the
driver use backend "nccl" = RCCL over xGMI with one GPU per rank).  Views are sharded across the ranks; after every
step -- including a mask prune and a densification -- both ranks must hold bit-identical parameters, and the result
must equal a single-rank run over all views up to the summation order of the all-reduce."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"
V, W, H, N0 = 5, 96, 64, 4000
OPTS = dict(remove_seg_end=1, densify_from_step=2, densification_interval=3, densify_until_step=1000,
            opacity_reset_interval=100000, percent_dense=0.01, densify_grad_threshold=2e-5)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(rank, world, compact, sharded=False):
    from manus_amd.engine import HipViewCompute, Trainer
    from manus_amd.synthetic import camera_table, make_masks, make_scene
    sc = make_scene(n_gaussians=N0, kind="hand", seed=21, grid_res=24, n_cameras=V, width=W, height=H, cam_radius=0.5,
                    sigma_range=(3e-3, 9e-3), device=DEV)
    ct = camera_table(sc["cameras"], DEV)
    g = torch.Generator(device="cpu").manual_seed(5)
    tgt_scene = dict(sc)
    tgt_scene["params"] = {k: (v + (1.0 * torch.randn(v.shape, generator=g).to(DEV) if k == "_features_dc" else 0))
                           for k, v in sc["params"].items()}
    with torch.no_grad():
        hp = HipViewCompute(tgt_scene, torch.zeros((V, 3, H, W), device=DEV), ct)
        targets = hp.forward_views_fused(list(range(V)))[0].contiguous()
    sc["masks"] = make_masks(sc, sc["keypoints"][:V], margin=6).to(DEV)
    compute = HipViewCompute(sc, targets, ct, loss="l1+ssim")
    return Trainer(compute, V, extent=0.3, opts=OPTS, spatial_lr_scale=0.05, bg_white=False, rank=rank, world_size=world,
                   compact_allreduce=compact, sharded_adam=sharded)


def _run(tr, steps=5):
    torch.manual_seed(1234)          # the split noise: rank 0's is broadcast, so only rank 0's seed matters
    hist = []
    for _ in range(steps):
        out = tr.train_step()
        hist.append((tr.opt.N, float(out["loss"]), bool(out["changed"])))
    return hist


def _worker(rank, world, port, q, compact, sharded=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    tr = _build(rank, world, compact, sharded)
    assert tr.stepper.local_views == list(range(rank, V, world))
    hist = _run(tr)
    # identical state on every rank: compare through the process group itself
    for k, v in tr.opt.p.items():
        lo, hi = v.detach().clone(), v.detach().clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert torch.equal(lo, hi), k
    tr.gather_moments()              # (sharded step: a rank only keeps the moments of its own slice current)
    for t in (tr.opt.xyz_gradient_accum, tr.opt.denom, tr.opt.m["_features_rest"]):
        lo, hi = t.clone(), t.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert torch.equal(lo, hi)
    if rank == 0:
        q.put((hist, {k: v.detach().cpu().numpy() for k, v in tr.opt.p.items()}, tr.stepper.exchanged_rows()))   # by value
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("compact,sharded", [(False, False), (True, False), (False, True)])
def test_two_ranks_train_identically_through_prune_and_densify(compact, sharded):
    """dense all-reduce / row-compacted all-reduce / sharded optimizer step (reduce-scatter -> Adam on the owned slice
    -> all-gather of the parameters)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, compact, sharded)) for r in range(2)]
    for p in procs:
        p.start()
    hist, params, rows = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # single rank over all five views
    ref = _build(0, 1, False)
    ref_hist = _run(ref)
    ns = [h[0] for h in hist]
    assert ns[0] < N0, ns                                   # step 0: the mask test pruned
    assert any(b > a for a, b in zip(ns, ns[1:])), ns       # a densification grew the model
    assert [h[2] for h in hist] == [h[2] for h in ref_hist]
    if compact:
        assert rows is not None and 0 < rows < ns[-1]       # only part of the rows travelled in the last step
    # same trajectory as the single-rank run (the all-reduce changes the summation order: fp32 roundoff; a Gaussian
    # sitting exactly on a densification threshold could in principle change sides, hence the small allowance on N)
    for a, b in zip(hist, ref_hist):
        assert abs(a[0] - b[0]) <= max(2, b[0] // 500), (hist, ref_hist)
        assert abs(a[1] - b[1]) < 1e-4 * max(1.0, abs(b[1]))
    if ns == [h[0] for h in ref_hist]:
        for k, v in ref.opt.p.items():
            d = (torch.from_numpy(params[k]) - v.detach().cpu()).abs().max() / (v.abs().max().cpu() + 1e-12)
            assert float(d) < 1e-3, (k, float(d))


def _composite_trainer(rank, world):
    from manus_amd.engine import HipViewCompute, Trainer
    from manus_amd.synthetic import camera_table, make_scene
    Vc = 7                                   # 7 cameras over 2 ranks: 4 + 3 views (uneven shards, like 53 cameras over 8 GPUs)
    sc = make_scene(n_gaussians=5000, kind="composite", seed=23, grid_res=24, n_cameras=Vc, width=W, height=H, cam_radius=0.5,
                    sigma_range=(3e-3, 9e-3), device=DEV)
    ct = camera_table(sc["cameras"], DEV)
    g = torch.Generator(device="cpu").manual_seed(7)
    tgt = dict(sc)
    tgt["params"] = {k: (v + (1.0 * torch.randn(v.shape, generator=g).to(DEV) if k == "_features_dc" else 0)) for k, v in sc["params"].items()}
    with torch.no_grad():
        targets = HipViewCompute(tgt, torch.zeros((Vc, 3, H, W), device=DEV), ct).forward_views_fused(list(range(Vc)))[0].contiguous()
    compute = HipViewCompute(sc, targets, ct, loss="l1+ssim")
    opts = dict(remove_seg_end=0, densify_from_step=100000, densify_until_step=0, opacity_reset_interval=100000)
    return Trainer(compute, Vc, extent=0.3, opts=opts, spatial_lr_scale=0.05, bg_white=False, rank=rank, world_size=world,
                   kind="object", compact_allreduce=True), Vc     # (density tests of the static kind: none are due here)


def _composite_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    tr, Vc = _composite_trainer(rank, world)
    assert tr.stepper.local_views == list(range(rank, Vc, world))
    losses = [float(tr.train_step()["loss"]) for _ in range(4)]
    for k, v in tr.opt.p.items():
        lo, hi = v.detach().clone(), v.detach().clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert torch.equal(lo, hi), k
    if rank == 0:
        q.put((losses, {k: v.detach().cpu().numpy() for k, v in tr.opt.p.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_composite_scene_uneven_view_shards():
    """BASELINE config 4 in miniature: hand + object composite (the first rows skinned, the rest static), 7 cameras over
    2 ranks (4 + 3 views), row-compacted all-reduce: both ranks hold identical parameters after four Adam steps and the
    trajectory equals the single-rank run over all seven views."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_composite_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    losses, params = q.get(timeout=90)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref, _ = _composite_trainer(0, 1)
    ref_losses = [float(ref.train_step()["loss"]) for _ in range(4)]
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) < 1e-4 * max(1.0, abs(b)), (losses, ref_losses)
    assert losses[-1] < losses[0]
    for k, v in ref.opt.p.items():
        d = (torch.from_numpy(params[k]) - v.detach().cpu()).abs().max() / (v.abs().max().cpu() + 1e-12)
        assert float(d) < 1e-3, (k, float(d))


def test_bench_runs_under_torch_distributed_run_with_two_ranks():
    """The driver's multi-GPU command line (python -m torch.distributed.run ... bench.py --gpus N) on a tiny
    configuration, two ranks sharing this GPU through gloo: it must not die on a Python error and must print one JSON
    line with the fields the driver reads (the timing itself means nothing here)."""
    import json
    import subprocess
    env = dict(os.environ, MANUS_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--gaussians", "6000", "--views", "5", "--width", "160", "--height", "96", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0 and d["headline"] is False
    assert d["config"]["view_assignment"].startswith("balanced") and d["config"]["allreduce"]["mode"] in ("dense", "compact")
    assert set(d["config"]["allreduce"]["ms_per_step_by_mode"]) == {"dense", "compact"}


def _run_bench(nproc, extra, timeout=1500):
    import json
    import subprocess
    env = dict(os.environ, MANUS_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    if nproc > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())]
    else:
        cmd = [sys.executable]
    cmd += [os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "3", "--warmup", "2", "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("name,extra,views", [
    ("cfg3", ["--gaussians", "30000", "--views", "8", "--width", "480", "--height", "270"], 8),
    ("cfg4", ["--kind", "composite", "--gaussians", "30000", "--views", "53", "--width", "480", "--height", "270"], 53)])
def test_bench_eight_rank_dry_run(name, extra, views):
    """The 8-rank shape of bench.py -- what the driver's SCALE run launches -- at a reduced size, eight gloo ranks sharing this
    GPU: BASELINE config 3 (8 views -> one per rank, LPT over 8 ranks, the `auto` dense-vs-compact timing) and config 4
    (53 cameras -> 7 / 6 per rank, composite).  One JSON line, n_gpus 8, the view assignment a partition of the views,
    and the all-reduced gradients equal to the one-rank step's (same scene, same views: the sums differ by their order only)."""
    d8 = _run_bench(8, extra)
    assert d8["n_gpus"] == 8 and d8["steps"] == 3 and d8["value"] > 0 and d8["headline"] is False
    c = d8["config"]
    assert c["view_assignment"].startswith("balanced")
    by_rank = c["views_by_rank"]
    assert len(by_rank) == 8 and sorted(v for r in by_rank for v in r) == list(range(views))
    sizes = sorted(len(r) for r in by_rank)
    assert sizes[0] >= 1 and (views != 8 or sizes == [1] * 8), sizes      # (LPT balances cost, not counts: 53 views -> ~5..8 per rank)
    assert c["allreduce"]["mode"] in ("dense", "compact") and set(c["allreduce"]["ms_per_step_by_mode"]) == {"dense", "compact"}
    assert "predicted_ms" in c      # (None away from the 300 k / 1080p configuration)
    d1 = _run_bench(1, extra + ["--no-hints-variant"])
    assert d1["n_gpus"] == 1
    g8, g1 = c["grad_digest"], d1["config"]["grad_digest"]
    assert abs(g8["loss"] - g1["loss"]) < 1e-5 * max(1.0, abs(g1["loss"])), (g8["loss"], g1["loss"])
    for k in g1:
        if k == "loss":
            continue
        s8, a8 = g8[k]
        s1, a1 = g1[k]
        assert abs(a8 - a1) <= 2e-5 * a1 + 1e-12, (name, k, a8, a1)
        assert abs(s8 - s1) <= 2e-5 * a1 + 1e-12, (name, k, s8, s1)


_RCCL_ONE_RANK = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", sys.argv[2]
import torch, torch.distributed as dist
from manus_amd.engine import HipViewCompute, ViewShardedStep
from manus_amd.optim import GaussianOptimizer
from manus_amd.synthetic import camera_table, make_scene
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)          # RCCL, one rank
DEV = "cuda:0"
V, W, H, N = 3, 96, 64, 3000
sc = make_scene(n_gaussians=N, kind="hand", seed=4, grid_res=24, n_cameras=V, width=W, height=H, cam_radius=0.5, sigma_range=(3e-3, 9e-3), device=DEV)
ct = camera_table(sc["cameras"], DEV)
tg = torch.rand((V, 3, H, W), device=DEV, generator=torch.Generator(device=DEV).manual_seed(2))
def run(**kw):
    hc = HipViewCompute(sc, tg, ct, loss="l1+ssim")
    shapes = {k: v.shape for k, v in hc.params.items()}
    st = ViewShardedStep(N, shapes, hc, V, rank=0, world_size=1, **kw)
    out = st.step()
    torch.cuda.synchronize()
    return st, hc, {k: v.clone() for k, v in out["grads"].items()}, out["grad2d"].clone(), out["vis"].clone(), float(out["loss"]), out["radii"].clone()
_, _, g0, s0, v0, l0, r0 = run()
for kw in (dict(force_collectives=True), dict(force_collectives=True, compact=True)):
    st, hc, g, s, v, l, r = run(**kw)
    for k in g0:
        assert torch.equal(g[k].reshape(g0[k].shape), g0[k]), (kw, k)      # a sum over one rank is the identity
    assert torch.equal(s, s0) and torch.equal(v, v0) and abs(l - l0) < 1e-7 and torch.equal(r, r0), kw
    rr = st.reduce_max_radii(r.clone())
    assert torch.equal(rr, r0)
# sharded optimizer step: reduce_scatter_tensor (in place on the owned slice) + all_gather_into_tensor
st, hc, g, s, v, l, r = run(force_collectives=True, scatter=True)
assert st.owned == (0, N * 59)
for k in g0:
    assert torch.equal(g[k].reshape(g0[k].shape), g0[k]), ("scatter", k)
opt = GaussianOptimizer(hc.params, adopt=True)
opt.flatten(st.padded_g)
before = opt.pflat.clone()
opt.step_range(st._store, *st.owned)
st.all_gather_params(opt.pflat)
torch.cuda.synchronize()
assert not torch.equal(opt.pflat, before) and torch.isfinite(opt.pflat).all()
dist.destroy_process_group()
print("RCCL-ONE-RANK-OK")
"""


def test_every_collective_runs_on_rccl_in_a_world_of_one():
    """The gloo tests take the alternate branches (no reduce-scatter, list all-gather).  A single-process "nccl" group is
    RCCL with one rank: the float and uint8 SUM all-reduces, the MAX all-reduce, reduce_scatter_tensor in place on the
    owned slice and all_gather_into_tensor run through RCCL's own code with the dtypes and aliasing used by the step
    (sums over one rank must be the identity, bit for bit).  Multi-rank RCCL over xGMI remains the driver's to measure."""
    import subprocess
    r = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK, ROOT, str(_free_port())], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0 and "RCCL-ONE-RANK-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
