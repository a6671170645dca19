"""GPU: eprecon_sparsify_async (threshold + guard counts + compaction of the kept rows in one call) against the reference's
sequence of torch calls (models/neucon_network.py:454-507 without the random sub-sampling)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,c_all,c_feat,bs,with_target", [(1, 8, 4, 1, False), (255, 48, 24, 1, True), (256, 48, 24, 2, True),
                                                           (100003, 176, 96, 2, True), (371596, 48, 24, 1, True),
                                                           (5000, 33, 9, 3, False)])
def test_sparsify_matches_the_torch_sequence(n, c_all, c_feat, bs, with_target):
    from eprecon_amd import grid_ops as GO
    g = torch.Generator(device="cuda").manual_seed(n + c_all)
    dev = torch.device("cuda")
    occ = torch.randn((n, 1), device=dev, generator=g)
    tsdf = torch.randn((n, 1), device=dev, generator=g)
    wide = torch.randn((n, c_all + 3), device=dev, generator=g)
    feat_all = wide[:, :c_all]                                   # a column slice: row pitch != channels
    coords = torch.randint(0, 96, (n, 4), device=dev, generator=g, dtype=torch.int32)
    coords[:, 0] = torch.sort(torch.randint(0, bs, (n,), device=dev, generator=g, dtype=torch.int32))[0]
    target = (torch.rand((n, 1), device=dev, generator=g) > 0.3) if with_target else None
    thr = 0.1
    host, pc, pt, po, ka, pf = GO.sparsify(occ, thr, target, coords, tsdf, feat_all, c_feat, bs)
    occupancy = occ.squeeze(1) > thr
    keep = torch.nonzero(occupancy).squeeze(1)
    assert host[0] == keep.numel()
    tgt = target.reshape(-1) if target is not None else torch.ones_like(occupancy)
    for b in range(bs):
        rows = coords[:, 0] == b
        assert host[1 + b] == int((occupancy & rows).sum())
        assert host[1 + bs + b] == int((occupancy & tgt & rows).sum())
    assert torch.equal(pc, coords.index_select(0, keep))
    assert torch.equal(pt, tsdf.index_select(0, keep)) and torch.equal(po, occ.index_select(0, keep))
    kept = feat_all.index_select(0, keep)
    assert torch.equal(ka, kept)
    assert torch.equal(pf, torch.cat([kept[:, :c_feat], pt, po], dim=1))


def test_sparsify_with_nothing_kept_and_everything_kept():
    from eprecon_amd import grid_ops as GO
    dev = torch.device("cuda")
    n = 1000
    occ = torch.full((n, 1), -1.0, device=dev)
    coords = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    tsdf = torch.zeros((n, 1), device=dev)
    feat = torch.arange(n * 8, dtype=torch.float32, device=dev).view(n, 8)
    host, pc, pt, po, ka, pf = GO.sparsify(occ, 0.0, None, coords, tsdf, feat, 4, 1)
    assert host == [0, 0, 0] and pc.shape == (0, 4) and ka.shape == (0, 8) and pf.shape == (0, 6)
    host, pc, pt, po, ka, pf = GO.sparsify(-occ, 0.0, None, coords, tsdf, feat, 4, 1)
    assert host == [n, n, n] and torch.equal(ka, feat) and torch.equal(pf[:, :4], feat[:, :4])
