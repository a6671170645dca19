"""GPU: GRUFusion(direct_substitute=True) (= NeuralRecon.fuse_to_global) against golden vectors
captured from the reference's own forward / panoptic_fusion / save_mesh: scene map coordinates, TSDF,
instance and semantic ids bit-exact over three overlapping fragments."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from eprecon_amd.config import ModelCfg  # noqa: E402


def scene_fusion_inputs(seed=21, n_vox=24, n_frag=3):
    rng = np.random.default_rng(seed)
    shifts = [(0, 0, 0), (8, 0, 0), (8, 8, 0)][:n_frag]
    frags = []
    for k, sh in enumerate(shifts):
        occ = rng.random((n_vox,) * 3) < 0.2
        gx, gy, gz = np.meshgrid(*[np.arange(n_vox)] * 3, indexing="ij")
        sx, sy, sz = gx + sh[0], gy + sh[1], gz + sh[2]
        blob = ((sx - 14) ** 2 + (sy - 10) ** 2 + (sz - 12) ** 2) < 30
        occ |= blob
        xyz = np.argwhere(occ)
        tsdf = np.clip(rng.standard_normal(len(xyz)) * 0.7, -1.2, 1.2).astype(np.float32)[:, None]
        seg = np.zeros(len(xyz), np.int32)
        inblob = blob[xyz[:, 0], xyz[:, 1], xyz[:, 2]]
        seg[inblob] = 1
        seg[(~inblob) & (xyz[:, 2] < 3)] = 2
        seg[(~inblob) & (xyz[:, 2] >= 3) & (rng.random(len(xyz)) < 0.2)] = 3
        info = [{"id": 1, "isthing": True, "category_id": 5}, {"id": 2, "isthing": False, "category_id": 2},
                {"id": 3, "isthing": True, "category_id": 7}]
        frags.append({"coords": np.concatenate([np.zeros((len(xyz), 1), np.int64), xyz], 1).astype(np.int32),
                      "tsdf": tsdf, "seg": seg, "info": info,
                      "origin_partial": (np.array([-0.96, 0.2, -0.4]) + np.array(sh) * 0.04).astype(np.float32)})
    return frags, np.array([-0.96, 0.2, -0.4], np.float32)


def test_scene_fusion_matches_reference(golden_dir):
    from eprecon_amd.gru_fusion import GRUFusion
    gold = np.load(os.path.join(golden_dir, "scene_fusion.npz"))
    fus = GRUFusion(ModelCfg(N_VOX=[24, 24, 24]), direct_substitute=True, trianing=False)
    frags, origin = scene_fusion_inputs()
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    outputs = {}
    for k, fr in enumerate(frags):
        inputs = {"fragment": ["f"], "scene": ["sceneA"], "vol_origin": dev(origin[None]),
                  "vol_origin_partial": dev(fr["origin_partial"][None])}
        infos = [{"panoptic_seg": [dev(fr["seg"]), [dict(d) for d in fr["info"]]]}]
        outputs = fus(dev(fr["coords"]), dev(fr["tsdf"]), inputs, 2, outputs, save_mesh=(k == len(frags) - 1),
                      panoptic_infos=infos)
        sc = fus._scene
        key = f"f{k}_"
        assert np.array_equal(sc.C.cpu().numpy(), gold[key + "map_C"])
        assert np.array_equal(sc.F.cpu().numpy(), gold[key + "map_F"])
        assert np.array_equal(sc.instance.cpu().numpy(), gold[key + "instance"])
        assert np.array_equal(sc.semantic.cpu().numpy(), gold[key + "semantic"])
    assert np.array_equal(outputs["scene_tsdf"][0].cpu().numpy(), gold["scene_tsdf"])
    assert np.array_equal(outputs["scene_instance"][0].cpu().numpy(), gold["scene_instance"])
    assert np.array_equal(outputs["scene_semantic"][0].cpu().numpy(), gold["scene_semantic"])
    assert np.allclose(outputs["origin"][0].cpu().numpy(), gold["scene_origin"])
    assert outputs["scene_name"] == ["sceneA"]
