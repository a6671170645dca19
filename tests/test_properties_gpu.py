"""GPU property tests (hypothesis) for the integer / index paths, as SURVEY.md section 4 asks for: stable compaction
order of the back-projection, hash-grid query == brute-force membership, upsample child order, union == set union in
raster order — on small random inputs, plus the documented limits of the ABI (error codes instead of silent drops)."""
import ctypes

import numpy as np
import pytest

torch = pytest.importorskip("torch")
hyp = pytest.importorskip("hypothesis")
from hypothesis import given, settings, strategies as st  # noqa: E402

pytestmark = pytest.mark.gpu

from eprecon_amd import synthetic as S  # noqa: E402

SETTINGS = dict(max_examples=25, deadline=None)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@settings(**SETTINGS)
@given(seed=st.integers(0, 10_000), n=st.integers(1, 3000), min_view=st.integers(0, 4), batch=st.integers(1, 2))
def test_back_projection_compaction_is_stable_and_complete(seed, n, min_view, batch):
    """valid voxels = exactly those seen by >= min_view views, in INPUT order (models/occupancy_initialization.py:236)"""
    from eprecon_amd import back_project as BP
    rng = np.random.default_rng(seed)
    window = S.make_window(seed=seed % 7, width=320, height=240, n_vox=(32, 32, 32))
    xyz = rng.integers(-8, 40, size=(n, 3))
    b = np.sort(rng.integers(0, batch, size=(n, 1)), axis=0)
    coords = np.concatenate([b, xyz], 1).astype(np.int32)
    c, h, w = S.pyramid_shapes(240, 320)[1]
    feats = S.make_features(seed, 9, (c, h, w), batch=batch)
    kr = np.ascontiguousarray(np.repeat(window["proj_matrices"][:, 1][:, None], batch, 1))
    origin = np.repeat(window["vol_origin_partial"][None], batch, 0).copy()
    res = BP.run(dev(coords), dev(origin), 0.04, dev(feats), dev(kr), min_view)
    if res is None:
        return
    count = res["count"].cpu().numpy()
    keep = count >= min_view
    assert res["n_valid"] == int(keep.sum())
    assert np.array_equal(res["coords"].cpu().numpy(), coords[keep])           # stable, complete
    assert res["n_valid_per_batch"] == [int((keep & (coords[:, 0] == k)).sum()) for k in range(batch)]
    hidden = res["feats"].cpu().numpy()[count[keep] == 0]
    assert not hidden.any()                                                    # unseen voxels carry zeros


@settings(**SETTINGS)
@given(seed=st.integers(0, 10_000), n=st.integers(1, 4000), m=st.integers(1, 2000), q=st.sampled_from([1, 2, 4]))
def test_hash_grid_query_is_set_membership(seed, n, m, q):
    from eprecon_amd.sparse import HashGrid
    rng = np.random.default_rng(seed)
    pts = np.concatenate([rng.integers(0, 3, (n, 1)), rng.integers(-30, 30, (n, 3))], 1).astype(np.int32)
    qry = np.concatenate([rng.integers(0, 3, (m, 1)), rng.integers(-32, 32, (m, 3))], 1).astype(np.int32)
    grid = HashGrid(n, torch.device("cuda")).build(dev(pts), quantum=q)
    got = grid.query(dev(qry), quantum=q).cpu().numpy()
    key = lambda a: [tuple([r[0]] + [int(np.floor(v / q)) * q for v in r[1:]]) for r in a.tolist()]
    first = {}
    for i, k in enumerate(key(pts)):
        first.setdefault(k, i)                                                  # value = smallest row with that key
    assert got.tolist() == [first.get(k, -1) for k in key(qry)]


@settings(**SETTINGS)
@given(seed=st.integers(0, 10_000), n=st.integers(1, 500), c=st.integers(0, 9), interval=st.sampled_from([1, 2]))
def test_upsample_children_are_parent_major_in_the_reference_order(seed, n, c, interval):
    from eprecon_amd import grid_ops as GO
    rng = np.random.default_rng(seed)
    coords = np.concatenate([rng.integers(0, 2, (n, 1)), rng.integers(0, 24, (n, 3)) * 2 * interval], 1).astype(np.int32)
    feat = rng.standard_normal((n, max(c, 1))).astype(np.float32)
    uf, uc = GO.upsample(dev(feat), dev(coords), interval)
    order = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (1, 0, 1), (0, 1, 1), (1, 1, 1)]   # neucon_network.py:204-209
    exp = np.repeat(coords, 8, 0)
    exp[:, 1:] += np.tile(np.array(order, np.int32) * interval, (n, 1))
    assert np.array_equal(uc.cpu().numpy(), exp) and np.array_equal(uf.cpu().numpy(), np.repeat(feat, 8, 0))


@settings(**SETTINGS)
@given(seed=st.integers(0, 10_000), n_cur=st.integers(0, 300), n_map=st.integers(0, 300), dim=st.sampled_from([4, 8, 12]))
def test_map_union_is_the_raster_ordered_set_union(seed, n_cur, n_map, dim):
    from eprecon_amd.global_map import GlobalMap
    rng = np.random.default_rng(seed)
    c = 3
    cur = np.unique(rng.integers(0, dim, (n_cur, 3)), axis=0)
    glob = np.unique(rng.integers(-dim, 2 * dim, (n_map, 3)), axis=0)
    cur_f = rng.standard_normal((len(cur), c)).astype(np.float32)
    cur_f[rng.random(len(cur)) < 0.2] = 0                                       # all-zero rows do not activate a voxel
    glob_f = rng.standard_normal((len(glob), c)).astype(np.float32)
    rel = rng.integers(-3, 4, 3)
    gm = GlobalMap(c, torch.device("cuda"))
    gm.set(dev(glob.astype(np.int32)), dev(glob_f))
    cur4 = np.concatenate([np.zeros((len(cur), 1), int), cur], 1).astype(np.int32)
    upd, sc, sg, inside = gm.crop_union(dev(cur4), dev(cur_f), dim, 1, rel.tolist())
    loc = glob - rel
    ins = ((loc >= 0) & (loc < dim)).all(1)
    active = {tuple(x) for x, f in zip(cur.tolist(), cur_f) if f.any()} | {tuple(x) for x in loc[ins].tolist()}
    assert inside == int(ins.sum())
    assert upd.cpu().numpy().tolist() == [list(x) for x in sorted(active)]       # lexicographic == raster order
    vals = rng.standard_normal((upd.shape[0], c)).astype(np.float32)
    gm.update(upd, dev(vals))
    want_c = np.concatenate([glob[~ins], upd.cpu().numpy() + rel]) if len(glob) else upd.cpu().numpy() + rel
    assert np.array_equal(gm.C.cpu().numpy(), want_c.reshape(-1, 3))
    assert np.array_equal(gm.F.cpu().numpy(), np.concatenate([glob_f[~ins], vals]).reshape(-1, c))


def test_documented_limits_return_error_codes():
    """batch > 14 in a hash key, n_views > 32: status codes, never silent truncation"""
    from eprecon_amd import _lib
    from eprecon_amd import back_project as BP
    from eprecon_amd.sparse import HashGrid
    lib = _lib.load()
    g = HashGrid(4, torch.device("cuda")).build(dev(np.array([[15, 0, 0, 0], [0, 1, 1, 1]], np.int32)))
    with pytest.raises(_lib.EpreconError):
        g.status_ok()
    window = S.make_window(seed=0, width=320, height=240, n_vox=(32, 32, 32))
    coords = dev(S.dense_coords((32, 32, 32), 4))
    feats = torch.zeros((33, 1, 8, 15, 20), device="cuda")                      # 33 views
    kr = torch.zeros((33, 1, 4, 4), device="cuda")
    with pytest.raises(_lib.EpreconError):
        BP.run(coords, dev(window["vol_origin_partial"][None]), 0.04, feats, kr, 0)
