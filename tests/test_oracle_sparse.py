"""CPU: self-consistency of the sparse oracle (the restated torchsparse / spconv semantics).
These layers are "parity unpinned" (no reference tests, libraries not installable); what can be
checked on CPU is that the restatement agrees with independent dense formulations."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import sparse as OS


def random_coords(rng, n, extent=20, batch=2, stride=1):
    c = rng.integers(-extent, extent, size=(n * 2, 3)) * stride
    b = rng.integers(0, batch, size=(n * 2, 1))
    rows = np.unique(np.concatenate([b, c], 1), axis=0)
    rng.shuffle(rows)
    return rows[:n].astype(np.int32)


def test_unique_first_occurrence_order():
    rng = np.random.default_rng(0)
    c = random_coords(rng, 500)
    dup = np.concatenate([c, c[rng.integers(0, 500, 300)]])
    u, inv = OS.unique_first(dup, 1)
    assert len(u) == 500 and np.array_equal(u, c)          # first occurrences, in order
    assert np.array_equal(u[inv], dup)
    u2, inv2 = OS.unique_first(c, 4)
    assert np.array_equal(u2[inv2][:, 1:], np.floor_divide(c[:, 1:], 4) * 4)
    first_seen = {}
    for i, k in enumerate(map(tuple, u2[inv2])):
        first_seen.setdefault(k, len(first_seen))
        assert inv2[i] == first_seen[k]


def test_submanifold_conv_matches_dense_conv3d():
    """k=3 stride-1 conv on a sparse set == dense conv3d evaluated at the active sites
    (inactive sites hold zeros), with the x-fastest offset enumeration."""
    rng = np.random.default_rng(1)
    D, cin, cout = 10, 5, 7
    occ = rng.random((D, D, D)) < 0.3
    xyz = np.argwhere(occ)
    coords = np.concatenate([np.zeros((len(xyz), 1), int), xyz], 1).astype(np.int32)
    x = rng.standard_normal((len(xyz), cin)).astype(np.float32)
    w = rng.standard_normal((27, cin, cout)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    nbr = OS.kernel_map(coords, coords, 3, 1)
    got = OS.sparse_conv(x, nbr, w, b)
    dense = np.zeros((1, cin, D, D, D), np.float32)
    dense[0, :, xyz[:, 0], xyz[:, 1], xyz[:, 2]] = x
    # dense weight [cout, cin, kx, ky, kz] for input laid out (x, y, z); k = (dz*3 + dy)*3 + dx
    wd = w.reshape(3, 3, 3, cin, cout).transpose(4, 3, 2, 1, 0)  # (kz,ky,kx,ci,co) -> (co,ci,kx,ky,kz)
    ref = F.conv3d(torch.from_numpy(dense), torch.from_numpy(np.ascontiguousarray(wd)),
                   torch.from_numpy(b), padding=1)[0].numpy()
    ref = ref[:, xyz[:, 0], xyz[:, 1], xyz[:, 2]].T
    np.testing.assert_allclose(got, ref, atol=1e-4)


def test_strided_and_transposed_maps():
    rng = np.random.default_rng(2)
    fine = random_coords(rng, 400, extent=8)
    coarse, parent = OS.unique_first(fine, 2)
    down = OS.kernel_map(fine, coarse, 2, 1)
    up = OS.transpose_map(fine, parent, 1)
    # every fine voxel appears exactly once in the down map, at (parent, child slot), and the
    # transposed map is its exact inverse
    for k in range(8):
        live = np.nonzero(down[k] >= 0)[0]
        assert np.array_equal(up[k][down[k][live]], live)
    assert (down >= 0).sum() == len(fine) == (up >= 0).sum()
    off = OS.offsets(2)
    for k in range(8):
        live = down[k] >= 0
        assert np.array_equal(fine[down[k][live]][:, 1:], coarse[live][:, 1:] + off[k])


def test_norms_match_torch():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((300, 24)).astype(np.float32) * 3 + 1
    r = rng.standard_normal((300, 24)).astype(np.float32)
    g = rng.standard_normal(24).astype(np.float32)
    b = rng.standard_normal(24).astype(np.float32)
    bn = torch.nn.BatchNorm1d(24)
    bn.weight.data, bn.bias.data = torch.from_numpy(g), torch.from_numpy(b)
    bn.train()
    ref = torch.relu(bn(torch.from_numpy(x)) + torch.from_numpy(r)).detach().numpy()
    np.testing.assert_allclose(OS.batchnorm_train(x, g, b, 1e-5, r, True), ref, atol=2e-5)
    t = torch.relu(torch.from_numpy(x)) + torch.from_numpy(r)
    ref = F.layer_norm(t, (24,), torch.from_numpy(g), torch.from_numpy(b), 1e-5).numpy()
    np.testing.assert_allclose(OS.layernorm_rows(x, g, b, 1e-5, r, True, False), ref, atol=2e-5)


def test_spconv_checkpoint_layout_is_converted():
    """A reference checkpoint stores SubMConv3d weights in spconv's [C_out, kx, ky, kz, C_in] layout
    (indices are (b, x, y, z) rows, models/modules.py:267) under `.sparsesubmconv3d.weight` or a 5-D
    `.conv.weight`.  Loading it into this package's layers must reproduce the dense conv3d the
    spconv layer is equivalent to on an (x, y, z) volume."""
    from eprecon_amd.modules import SparseSubMConv3d, SubMconv3dBlock
    rng = np.random.default_rng(4)
    D, cin, cout = 7, 3, 4
    w_sp = rng.standard_normal((cout, 3, 3, 3, cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    layer = SparseSubMConv3d(cin, cout, 3)
    missing = layer.load_state_dict({"sparsesubmconv3d.weight": torch.from_numpy(w_sp),
                                     "sparsesubmconv3d.bias": torch.from_numpy(b)})
    assert not missing.missing_keys and not missing.unexpected_keys
    blk = SubMconv3dBlock(cin, cout, 3, 1, 1)
    sd = blk.state_dict()
    sd["conv.weight"], sd["conv.bias"] = torch.from_numpy(w_sp), torch.from_numpy(b)
    blk.load_state_dict(sd)
    assert torch.equal(blk.conv.weight, layer.weight)
    k1 = SparseSubMConv3d(cin, cout, 1)
    w1 = rng.standard_normal((cout, 1, 1, 1, cin)).astype(np.float32)
    k1.load_state_dict({"sparsesubmconv3d.weight": torch.from_numpy(w1), "sparsesubmconv3d.bias": torch.from_numpy(b)})
    assert np.array_equal(k1.weight.detach().numpy()[0], w1[:, 0, 0, 0, :].T)
    # asymmetric occupancy so that an x/z swap cannot go unnoticed
    occ = rng.random((D, D + 1, D + 2)) < 0.4
    xyz = np.argwhere(occ)
    coords = np.concatenate([np.zeros((len(xyz), 1), int), xyz], 1).astype(np.int32)
    x = rng.standard_normal((len(xyz), cin)).astype(np.float32)
    got = OS.sparse_conv(x, OS.kernel_map(coords, coords, 3, 1), layer.weight.detach().numpy(), b)
    dense = np.zeros((1, cin) + occ.shape, np.float32)
    dense[0, :, xyz[:, 0], xyz[:, 1], xyz[:, 2]] = x
    wd = torch.from_numpy(w_sp).permute(0, 4, 1, 2, 3).contiguous()      # [co, ci, kx, ky, kz]
    ref = F.conv3d(torch.from_numpy(dense), wd, torch.from_numpy(b), padding=1)[0].numpy()
    np.testing.assert_allclose(got, ref[:, xyz[:, 0], xyz[:, 1], xyz[:, 2]].T, atol=1e-4)
