"""Row-compacted gradient exchange (csrc/exchange.hip) against the torch statement of the same steps
(engine._compact_all_reduce on CPU tensors): mask from the rows and from an active list, ordered row list, pack, unpack."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _flat(N, density, seed, world=1):
    from manus_amd.engine import FLAT_TAIL, GRAD_LAYOUT, GRAD_WIDTH
    g = torch.Generator().manual_seed(seed)
    padded = (N * GRAD_WIDTH + world - 1) // world * world
    flat = torch.zeros(padded + 2 * N + FLAT_TAIL)
    rows = torch.nonzero(torch.rand(N, generator=g) < density)[:, 0]
    o = 0
    for _, w in GRAD_LAYOUT:
        seg = flat[o:o + N * w].view(N, w)
        seg[rows] = torch.randn((len(rows), w), generator=g)
        o += N * w
    flat[padded:padded + N][rows] = torch.rand(len(rows), generator=g)
    flat[padded + N:padded + 2 * N] = torch.randint(0, 9, (N,), generator=g).float()
    flat[-2], flat[-1] = 0.375, 0.0
    return flat, rows, padded


class _Fn:
    pass


@pytest.mark.parametrize("N,density", [(1, 1.0), (1000, 0.0), (5000, 0.4), (70001, 0.07), (3000, 1.0)])
@pytest.mark.parametrize("use_list", [False, True])
def test_compact_exchange_kernels_equal_the_torch_steps(N, density, use_list):
    from manus_amd.engine import GRAD_LAYOUT, ViewShardedStep
    flat_cpu, rows, padded = _flat(N, density, seed=N)
    shapes = {name: (N, w) for name, w in GRAD_LAYOUT}
    outs = []
    for dev in ("cpu", DEV):
        st = ViewShardedStep(N, shapes, _Fn(), 8)
        st._store = flat_cpu.clone().to(dev)
        fv = st._views()
        active = None
        if use_list and dev != "cpu":
            # a superset of the non-zero rows, in arbitrary order, with entries beyond the count that must be ignored
            extra = torch.randperm(N)[: min(N, len(rows) + 7)]
            lst = torch.unique(torch.cat([rows, extra[:3]]))[torch.randperm(len(torch.unique(torch.cat([rows, extra[:3]]))))]
            pad = torch.full((5,), 2 ** 31 - 1, dtype=torch.int64)
            lst_d = torch.cat([lst, pad]).to(torch.int32).to(DEV)
            cnt_d = torch.tensor([len(lst)], dtype=torch.int32, device=DEV)
            active = (ctypes.c_void_p(lst_d.data_ptr()), ctypes.c_void_p(cnt_d.data_ptr()))
            n_expect = len(lst)
        st._compact_all_reduce(st._store, fv, active=active)
        if dev != "cpu":
            torch.cuda.synchronize()
            # (the union's size reaches the host through an asynchronous copy behind the step; the first step travels with all N rows)
            assert int(st._xch["host"][0]) == (n_expect if use_list else len(rows)) and st.last_cap_rows == N
            st._compact_all_reduce(st._store, fv, active=active)      # ... and sizes the next step's collective
            torch.cuda.synchronize()
            assert st.last_rows == (n_expect if use_list else len(rows))
            assert st.last_cap_rows == min(N, int(st.last_rows * 1.25) + 1024)
        outs.append(st._store.cpu())
    # one rank: the exchange is the identity on the rows and rewrites the visibility counts from their byte form
    assert torch.equal(outs[0], flat_cpu) and torch.equal(outs[1], flat_cpu)


def test_exchange_index_is_ordered_and_pack_layout_matches_the_torch_buffer():
    from manus_amd._lib import check, lib, ptr, stream
    from manus_amd.engine import GRAD_LAYOUT, GRAD_WIDTH
    L = lib()
    N = 4097
    flat_cpu, rows, padded = _flat(N, 0.3, seed=5)
    flat = flat_cpu.to(DEV)
    segs = list(GRAD_LAYOUT) + [("grad2d", 1)]
    offs, o = [], 0
    for _, w in GRAD_LAYOUT:
        offs.append(o)
        o += N * w
    offs.append(padded)
    offs_c = (ctypes.c_int64 * 7)(*offs)
    widths_c = (ctypes.c_int * 7)(*[w for _, w in segs])
    small = torch.empty(2 * N, dtype=torch.uint8, device=DEV)
    check(L.mgr_exchange_mask(N, ptr(flat), 7, offs_c, widths_c, padded + N, None, None, ptr(small), stream()), "mask")
    assert torch.equal(torch.nonzero(small[:N].cpu())[:, 0], rows)
    assert torch.equal(small[N:].cpu().float(), flat_cpu[padded + N:padded + 2 * N])
    idx = torch.empty(N, dtype=torch.int32, device=DEV)
    cnt = torch.zeros(1, dtype=torch.int32, device=DEV)
    ws = torch.empty(L.mgr_exchange_index_workspace_bytes(N), dtype=torch.uint8, device=DEV)
    check(L.mgr_exchange_index(N, ptr(small), ptr(idx), ptr(cnt), ptr(ws), ws.numel(), stream()), "index")
    n = int(cnt.item())
    assert n == len(rows) and torch.equal(idx[:n].cpu().long(), rows)
    buf = torch.full((n * (GRAD_WIDTH + 1) + 2,), float("nan"), device=DEV)
    check(L.mgr_exchange_pack(N, n, ptr(idx), ptr(flat), 7, offs_c, widths_c, flat.numel() - 2, ptr(buf), stream()), "pack")
    ref, o = [], 0
    for (_, w), off in zip(segs, offs):
        ref.append(flat_cpu[off:off + N * w].view(N, w)[rows].reshape(-1))
    ref.append(flat_cpu[-2:])
    assert torch.equal(buf.cpu(), torch.cat(ref))
    # unpack of doubled rows (what a two-rank sum of identical buffers gives)
    check(L.mgr_exchange_unpack(N, n, ptr(idx), ptr(flat), 7, offs_c, widths_c, flat.numel() - 2, ptr(buf * 2), small[N:].data_ptr(), padded + N,
                                stream()), "unpack")
    exp = flat_cpu.clone()
    for (_, w), off in zip(segs, offs):
        exp[off:off + N * w] *= 2
    exp[-2:] *= 2
    assert torch.equal(flat.cpu(), exp)
    with pytest.raises(Exception):
        check(L.mgr_exchange_pack(N, N + 1, ptr(idx), ptr(flat), 7, offs_c, widths_c, 0, ptr(buf), stream()), "pack")


def test_compact_exchange_row_capacity_is_enforced_on_the_device():
    """mgr_exchange_pack_rows / _unpack_rows: the collective is sized by a row capacity the host chose beforehand, the count
    stays on the device.  A capacity that holds the union: the exchange is the identity on one rank, padding rows travel as
    zeros.  A capacity below the union: the overflow word of the step buffer is raised (Trainer._run_step runs the step again)."""
    from manus_amd.engine import GRAD_LAYOUT, GRAD_WIDTH, ViewShardedStep
    N = 6000
    flat_cpu, rows, padded = _flat(N, 0.3, seed=21)
    shapes = {name: (N, w) for name, w in GRAD_LAYOUT}
    n = len(rows)
    for cap, ok in ((n, True), (n + 57, True), (n - 1, False), (N, True)):
        st = ViewShardedStep(N, shapes, _Fn(), 8)
        st._store = flat_cpu.clone().to(DEV)
        st.row_capacity = cap
        st._compact_all_reduce(st._store, st._views())
        torch.cuda.synchronize()
        out = st._store.cpu()
        buf = st._xch["buf"][: cap * (GRAD_WIDTH + 1) + 2].cpu()
        assert st.last_cap_rows == cap and int(st._xch["host"][0]) == n
        if ok:
            assert torch.equal(out, flat_cpu)
            # rows [n, cap) of every segment are zeros
            o = 0
            for _, w in list(GRAD_LAYOUT) + [("grad2d", 1)]:
                assert not buf[o + n * w: o + cap * w].any()
                o += cap * w
        else:
            assert float(out[-1]) == 1.0 and float(out[-2]) == float(flat_cpu[-2])     # the overflow word; the loss still travels
