"""GPU: contact nearest-point distance (mgr_contact_dist) against the loop oracle, the reference's cdist
golden vectors, and properties at the composite-scene size."""
import os

import numpy as np
import pytest
import torch

from oracle import torch_ref as tr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_matches_golden_and_oracle(golden_dir):
    from manus_amd.contact import get_contact_dist, get_contact_map
    d = np.load(os.path.join(golden_dir, "contact.npz"))
    for k in range(4):
        pt1, pt2 = torch.tensor(d[f"pt1_{k}"], device=DEV), torch.tensor(d[f"pt2_{k}"], device=DEV)
        dist, idx = get_contact_dist(pt1, pt2)
        assert idx.dtype == torch.float32                          # the reference returns float indices
        rd, ri = tr.contact_dist(d[f"pt1_{k}"], d[f"pt2_{k}"])
        np.testing.assert_array_equal(dist.cpu().numpy(), rd)      # same fp32 arithmetic: bit-exact
        np.testing.assert_array_equal(idx.cpu().numpy().astype(np.int64), ri)
        np.testing.assert_allclose(get_contact_map(pt1, pt2).cpu().numpy(), d[f"dist_{k}"], rtol=2e-4, atol=5e-5)


@pytest.mark.parametrize("n1,n2", [(1, 0), (3, 1), (513, 1024), (1000, 1025), (5000, 70000)])
def test_ragged_sizes_and_ties(n1, n2):
    """Empty target set, LDS-tile and segment seams, duplicated target points (lowest index must win)."""
    from manus_amd.contact import get_contact_dist
    g = np.random.default_rng(n1 + n2)
    pt1 = (g.normal(size=(n1, 3)) * 0.05).astype(np.float32)
    pt2 = (g.normal(size=(n2, 3)) * 0.05).astype(np.float32)
    if n2 > 40:
        pt2[n2 // 2:n2 // 2 + 20] = pt2[:20]                       # every one of these has an earlier duplicate
        pt1[: min(n1, 20)] = pt2[: min(n1, 20)]
    dist, idx = get_contact_dist(torch.tensor(pt1, device=DEV), torch.tensor(pt2.reshape(-1, 3), device=DEV))
    rd, ri = tr.contact_dist(pt1, pt2)
    np.testing.assert_array_equal(dist.cpu().numpy(), rd)
    np.testing.assert_array_equal(idx.cpu().numpy().astype(np.int64), ri)


def test_composite_scene_size_properties():
    """300k hand x 200k object points (BASELINE cfg4): exact for a sampled subset, zero for planted contacts,
    distance attained by the returned index everywhere, run-to-run identical."""
    from manus_amd.contact import get_contact_dist
    g = torch.Generator(device=DEV).manual_seed(3)
    pt1 = torch.randn((300000, 3), device=DEV, generator=g) * 0.05
    pt2 = torch.randn((200000, 3), device=DEV, generator=g) * 0.05 + torch.tensor([0.04, 0.0, 0.0], device=DEV)
    pt2[1000:1100] = pt1[5000:5100]
    dist, idx = get_contact_dist(pt1, pt2)
    d2, i2 = get_contact_dist(pt1, pt2)
    assert torch.equal(dist, d2) and torch.equal(idx, i2)
    assert (dist[5000:5100] == 0).all() and torch.equal(idx[5000:5100].long(), torch.arange(1000, 1100, device=DEV))
    att = (pt1 - pt2[idx.long()]).norm(dim=1)
    assert float((att - dist).abs().max()) < 1e-7
    sub = torch.arange(0, 300000, 997, device=DEV)
    rd, ri = tr.contact_dist(pt1[sub].cpu().numpy(), pt2.cpu().numpy())
    np.testing.assert_array_equal(dist[sub].cpu().numpy(), rd)
    np.testing.assert_array_equal(idx[sub].cpu().numpy().astype(np.int64), ri)
