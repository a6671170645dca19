"""Scene mesh extraction.  CPU: the generated case table (C++ generator == independent numpy
generator) and the properties that pin the construction without skimage — closed surfaces are watertight, every
vertex lies on a grid edge at the linear zero crossing.  GPU: the HIP kernels against the numpy oracle, exactly."""
import ctypes
import os
from collections import Counter

import numpy as np
import pytest

from oracle import marching_cubes as OM


def blob_volume(n=16, seed=0):
    rng = np.random.default_rng(seed)
    g = np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing="ij"), -1).astype(np.float32)
    vol = np.full((n, n, n), 1e9, np.float32)
    for _ in range(4):
        c, r = rng.uniform(5, n - 6, 3), rng.uniform(2.0, 3.4)     # blobs stay inside the volume: closed surfaces
        vol = np.minimum(vol, np.linalg.norm(g - c, axis=-1) - r)
    return np.clip(vol / 3.0, -1, 1).astype(np.float32)


def edge_counts(faces):
    e = Counter()
    for a, b, c in faces:
        for p, q in ((a, b), (b, c), (c, a)):
            e[(min(p, q), max(p, q))] += 1
    return e


def test_case_table_generators_agree_and_cover_all_cases():
    from eprecon_amd import _lib
    table, most = OM.build_table()
    assert most == 5 and (table[0] == -1).all() and (table[255] == -1).all() and (table[1:255, 0] >= 0).all()
    got = np.zeros((256, 16), np.int8)
    assert _lib.load().eprecon_marching_cubes_table(got.ctypes.data_as(ctypes.c_void_p)) == 0
    assert np.array_equal(got, table)
    # complementary sign patterns cut the same edges
    for cs in range(256):
        assert set(table[cs][table[cs] >= 0]) == set(table[255 - cs][table[255 - cs] >= 0])


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_oracle_mesh_is_watertight_and_on_the_level_set(seed):
    vol = blob_volume(16, seed)
    verts, faces = OM.marching_cubes(vol, 0.0)
    assert len(verts) > 100 and set(edge_counts(faces).values()) == {2}       # closed 2-manifold: no cracks, no flaps
    # every vertex sits on one grid edge, where the trilinear field is zero
    frac = verts - np.floor(verts)
    assert ((frac > 0).sum(1) <= 1).all()
    lo = np.floor(verts).astype(int)
    hi = np.minimum(lo + (frac > 0), np.array(vol.shape) - 1)
    t = frac.max(1)
    val = vol[lo[:, 0], lo[:, 1], lo[:, 2]] * (1 - t) + vol[hi[:, 0], hi[:, 1], hi[:, 2]] * t
    assert np.abs(val).max() < 1e-5
    # faces are wound outwards (the field grows away from the blobs): positive enclosed volume
    a, b, c = (verts[faces[:, k]].astype(np.float64) for k in range(3))
    assert np.einsum("ij,ij->i", a, np.cross(b, c)).sum() / 6.0 > 0


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(12, 14, 10), (16, 16, 16)])
def test_hip_mesh_equals_oracle(shape):
    torch = pytest.importorskip("torch")
    from eprecon_amd import save_scene as SS
    rng = np.random.default_rng(shape[0])
    vol = blob_volume(16, 3)[: shape[0], : shape[1], : shape[2]].copy()
    sem = rng.integers(0, 21, shape).astype(np.int32)
    ins = rng.integers(0, 90, shape).astype(np.int32)
    verts, faces, normals, vs, vi = SS.marching_cubes(torch.from_numpy(vol).cuda(), 0.0,
                                                      labels=(torch.from_numpy(sem).cuda(), torch.from_numpy(ins).cuda()))
    ov, of = OM.marching_cubes(vol, 0.0)
    assert np.array_equal(verts.cpu().numpy(), ov) and np.array_equal(faces.cpu().numpy(), of)
    n = normals.cpu().numpy()
    assert np.abs(np.linalg.norm(n, axis=1) - 1).max() < 1e-5
    r = np.clip(np.rint(ov).astype(int), 0, np.array(shape) - 1)               # utils.py:236-239
    assert np.array_equal(vs.cpu().numpy(), sem[r[:, 0], r[:, 1], r[:, 2]])
    assert np.array_equal(vi.cpu().numpy(), ins[r[:, 0], r[:, 1], r[:, 2]])


@pytest.mark.gpu
def test_save_scene_writes_meshes_of_a_fused_scene(tmp_path, monkeypatch):
    """SaveScene(cfg)(outputs, inputs, epoch) on scene volumes shaped like GRUFusion.save_mesh's
    (models/gru_fusion.py:217-257): three .ply files + the .npz of utils.py:345-372"""
    torch = pytest.importorskip("torch")
    from types import SimpleNamespace
    from eprecon_amd import save_scene as SS
    from eprecon_amd import synthetic as S
    monkeypatch.chdir(tmp_path)
    window = S.make_window(seed=0)
    tsdf = S.analytic_tsdf(window, 0)                                          # 96^3 analytic room
    sem = (np.arange(96)[:, None, None] // 20 + np.zeros((96, 96, 96), np.int32)).astype(np.int32)
    outputs = {"scene_name": ["scene0000/00"], "origin": [torch.tensor([-1.92, 0.2, -0.4]).cuda()],
               "scene_tsdf": [torch.from_numpy(tsdf).cuda()], "scene_semantic": [torch.from_numpy(sem).cuda()],
               "scene_instance": [torch.from_numpy(sem * 3).cuda()]}
    cfg = SimpleNamespace(LOGDIR="logs/run", DATASET="scannet", SAVE_SCENE_MESH=True, MODEL=SimpleNamespace(VOXEL_SIZE=0.04))
    saver = SS.SaveScene(cfg)
    saver(outputs, {}, 7)
    d = os.path.join("results", "scene_scannet_run_fusion_eval_7")
    files = sorted(os.listdir(d))
    assert files == ["mesh_instance_scene0000-00.ply", "mesh_semantic_scene0000-00.ply", "scene0000-00.npz", "scene0000-00.ply"]
    z = np.load(os.path.join(d, "scene0000-00.npz"))
    assert np.array_equal(z["tsdf"], tsdf) and float(z["voxel_size"]) == 0.04
    head = open(os.path.join(d, "mesh_semantic_scene0000-00.ply"), "rb").read(400).decode("ascii", "ignore")
    assert head.startswith("ply\nformat binary_little_endian 1.0\nelement vertex ") and "property uchar red" in head
    mesh = SS.tsdf2mesh(0.04, outputs["origin"][0], outputs["scene_tsdf"][0])
    assert len(mesh["vertices"]) > 5000 and mesh["faces"].max() < len(mesh["vertices"])
    # the floor of the analytic room is at z = 0: there are mesh vertices within half a voxel of it
    assert (np.abs(mesh["vertices"][:, 2]) < 0.02).sum() > 500


def test_load_scene_npz_rebuilds_the_reference_dense_arrays(tmp_path):
    """sparse scene file (voxel rows) -> the dense arrays of utils.py:360-366: TSDF default 1, ids default 0"""
    from eprecon_amd.save_scene import load_scene_npz
    rng = np.random.default_rng(0)
    dims = (9, 7, 11)
    cells = rng.choice(np.prod(dims), 200, replace=False)
    c = np.stack(np.unravel_index(cells, dims), 1).astype(np.int32)
    tsdf = rng.uniform(-1, 1, 200).astype(np.float32)
    sem, ins = rng.integers(0, 21, 200).astype(np.int32), rng.integers(0, 50, 200).astype(np.int32)
    p = tmp_path / "s.npz"
    np.savez_compressed(p, origin=np.zeros(3, np.float32), voxel_size=0.04, dims=np.array(dims), sparse_coords=c,
                        sparse_tsdf=tsdf, sparse_semantic=sem, sparse_instance=ins)
    z = load_scene_npz(str(p))
    ref = np.ones(dims, np.float32)
    ref[c[:, 0], c[:, 1], c[:, 2]] = tsdf
    assert np.array_equal(z["tsdf"], ref) and z["semantic"].sum() == sem.sum() and z["instance"].dtype == np.int32
    assert z["semantic"][c[0, 0], c[0, 1], c[0, 2]] == sem[0] and (z["semantic"] != 0).sum() <= 200
    dense = tmp_path / "d.npz"
    np.savez_compressed(dense, origin=np.zeros(3), voxel_size=0.04, tsdf=ref, semantic=z["semantic"], instance=z["instance"])
    assert np.array_equal(load_scene_npz(str(dense))["tsdf"], ref)
    # a file of the EARLIER sparse layout (rows under the dense key names + dims) is rebuilt too, not handed back as 1-D rows
    legacy = tmp_path / "l.npz"
    np.savez_compressed(legacy, origin=np.zeros(3, np.float32), voxel_size=0.04, dims=np.array(dims), coords=c, tsdf=tsdf,
                        semantic=sem, instance=ins)
    zl = load_scene_npz(str(legacy))
    assert np.array_equal(zl["tsdf"], ref) and np.array_equal(zl["semantic"], z["semantic"]) and np.array_equal(zl["instance"], z["instance"])


@pytest.mark.gpu
def test_scene_fusion_sparse_export_and_incremental_saving(tmp_path, monkeypatch):
    """GRUFusion(direct_substitute).save_mesh also hands out the scene as voxel rows (`scene_sparse`); SaveScene writes the
    .npz from them (no dense device -> host copy), load_scene_npz gives back exactly the dense volumes the reference stores
    (models/gru_fusion.py:217-257, utils.py:345-372); SAVE_INCREMENTAL writes the per-keyframe meshes and images
    (utils.py:318-360)"""
    torch = pytest.importorskip("torch")
    from types import SimpleNamespace
    from eprecon_amd import save_scene as SS
    from eprecon_amd.config import ModelCfg
    from eprecon_amd.scene_fusion import SceneFusion
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(1)
    fus = SceneFusion(ModelCfg())
    fus.reset(torch.device("cuda"))
    m = 5000
    cells = rng.choice(40 * 30 * 20, m, replace=False)
    c = np.stack(np.unravel_index(cells, (40, 30, 20)), 1).astype(np.int32) + np.array([5, -3, 2], np.int32)
    fus.C = torch.from_numpy(c).cuda()
    # a thin shell around the plane x = 20 (so that there is a zero crossing to mesh)
    fus.F = torch.from_numpy(np.clip((c[:, :1] - 25 + rng.normal(0, 0.1, (m, 1))) / 3, -1, 1).astype(np.float32)).cuda()
    fus.instance = torch.from_numpy(rng.integers(0, 40, m).astype(np.int32)).cuda()
    fus.semantic = torch.from_numpy(rng.integers(0, 21, m).astype(np.int32)).cuda()
    outputs = fus.save_mesh({}, "scene0007/01")
    sp = outputs["scene_sparse"][0]
    assert sp["dims"] == tuple(outputs["scene_tsdf"][0].shape) and sp["coords"].shape == (m, 3)
    cfg = SimpleNamespace(LOGDIR="logs/run", DATASET="scannet", SAVE_SCENE_MESH=True, SAVE_INCREMENTAL=True,
                          SAVE_SCENE_NPZ="sparse", MODEL=SimpleNamespace(VOXEL_SIZE=0.04))
    saver = SS.SaveScene(cfg)
    saver.keyframe_id = 3                                                      # main.py:388
    imgs = torch.from_numpy(rng.integers(0, 255, (1, 9, 3, 24, 32)).astype(np.float32)).cuda()
    saver(outputs, {"imgs": imgs}, 2)
    z = SS.load_scene_npz(os.path.join("results", "scene_scannet_run_fusion_eval_2", "scene0007-01.npz"))
    assert np.array_equal(z["tsdf"], outputs["scene_tsdf"][0].cpu().numpy())
    assert np.array_equal(z["semantic"], outputs["scene_semantic"][0].cpu().numpy())
    assert np.array_equal(z["instance"], outputs["scene_instance"][0].cpu().numpy())
    raw = np.load(os.path.join("results", "scene_scannet_run_fusion_eval_2", "scene0007-01.npz"))
    # rows, not volumes, were written — under keys a dense reader (tools/generate_semantic_instance.py) fails loudly on
    assert "sparse_coords" in raw.files and raw["sparse_tsdf"].shape == (m,) and "tsdf" not in raw.files
    inc = os.path.join("incremental_results", "scene_scannet_run_2", "scene0007-01")
    inc = "incremental_" + os.path.join("results", "scene_scannet_run") + "_2" + os.sep + "scene0007-01"
    assert sorted(os.listdir(inc)) == ["mesh", "mesh_image", "mesh_instance", "mesh_semantic"]
    assert os.listdir(os.path.join(inc, "mesh")) == ["mesh_3.ply"]
    assert len(os.listdir(os.path.join(inc, "mesh_image"))) == 9
    from PIL import Image
    px = np.asarray(Image.open(os.path.join(inc, "mesh_image", "image_3_0.png")))
    assert np.array_equal(px, imgs[0, 0].permute(1, 2, 0).cpu().numpy().astype(np.uint8))
    # the DEFAULT is the reference's dense format (utils.py:360-366): [X,Y,Z] volumes under tsdf / semantic / instance
    del cfg.SAVE_SCENE_NPZ
    cfg.SAVE_INCREMENTAL = False
    SS.SaveScene(cfg)(outputs, {}, 5)
    raw = np.load(os.path.join("results", "scene_scannet_run_fusion_eval_5", "scene0007-01.npz"))
    assert "sparse_coords" not in raw.files and raw["tsdf"].shape == tuple(sp["dims"]) == raw["semantic"].shape
