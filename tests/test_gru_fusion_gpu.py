"""GPU: GRUFusion (HIP union / gather + ConvGRU) against the reference golden vectors (bookkeeping,
identity fusion) and the numpy oracle (with the ConvGRUs)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from eprecon_amd.config import ModelCfg  # noqa: E402
from oracle import gru_fusion as OGF  # noqa: E402
from oracle import pointvoxel as PV  # noqa: E402
from oracle import spvcnn as ON  # noqa: E402
from test_oracle_gru_fusion import sequence  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def make_inputs(fr, origin, lists):
    w2ac = np.eye(4, dtype=np.float32)
    w2ac[:3, :3] = np.array([[0.8, 0.6, 0], [-0.6, 0.8, 0], [0, 0, 1]], np.float32)
    w2ac[:3, 3] = [0.1, -0.2, 0.3]
    return {"fragment": ["f"], "scene": ["scene0"], "vol_origin": dev(origin[None]),
            "vol_origin_partial": dev(fr["origin_partial"][None]), "world_to_aligned_camera": dev(w2ac[None]),
            "occ_list": lists[0], "tsdf_list": lists[1]}, w2ac


def gt_lists(scale, fr):
    occ, tsdf = [None, None, None], [None, None, None]
    occ[2 - scale] = dev(fr["occ"][None])
    tsdf[2 - scale] = dev(fr["tsdf"][None])
    return occ, tsdf


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "gru_fusion.npz"))


@pytest.mark.parametrize("scale", [1, 2])
def test_bookkeeping_matches_reference_golden(gold, scale):
    from eprecon_amd.gru_fusion import GRUFusion
    cfg = ModelCfg(N_VOX=[24, 24, 24])
    fus = GRUFusion(cfg, ch_in=[6, 4, 3], ch_voxel=[4, 3, 2]).cuda()
    fus._identity_fusion = True
    frags, origin, interval, d, c = sequence(scale)
    for k, fr in enumerate(frags):
        inputs, _ = make_inputs(fr, origin, gt_lists(scale, fr))
        coords, values, tsdf_t, occ_t = fus(dev(fr["coords"]), dev(fr["values"]), inputs, scale)
        key = f"s{scale}_f{k}_"
        assert np.array_equal(coords.cpu().numpy()[:, 1:], gold[key + "updated"] * interval)
        assert np.array_equal(values.cpu().numpy(), gold[key + "values"])
        assert np.array_equal(fus.global_volume[scale].C.cpu().numpy(), gold[key + "map_C"])
        assert np.array_equal(fus.global_volume[scale].F.cpu().numpy(), gold[key + "map_F"])
        assert np.array_equal(fus.target_tsdf_volume[scale].C.cpu().numpy(), gold[key + "tgt_C"])
        # ground-truth twin: equal wherever the reference's duplicate scatter is well defined
        same = tsdf_t.cpu().numpy()[:, 0] == gold[key + "tsdf_target"][:, 0]
        assert same.mean() > 0.9
        fus.target_tsdf_volume[scale].F = dev(gold[key + "tgt_F"])  # follow the reference's pick


@pytest.mark.parametrize("scale", [1, 2])
def test_fusion_with_convgru_matches_oracle(scale):
    from eprecon_amd.gru_fusion import GRUFusion
    cfg = ModelCfg(N_VOX=[24, 24, 24])
    torch.manual_seed(scale)
    ch_in, ch_v = [6, 4, 3], [4, 3, 2]
    fus = GRUFusion(cfg, ch_in=ch_in, ch_voxel=ch_v).cuda()
    sd = {k: v.detach().cpu().numpy() for k, v in fus.state_dict().items()}
    frags, origin, interval, d, c = sequence(scale)
    st = OGF.ScaleState(c, origin)
    vres = 0.04 * interval
    for k, fr in enumerate(frags):
        inputs, w2ac = make_inputs(fr, origin, gt_lists(scale, fr))
        with torch.no_grad():
            coords, values, tsdf_t, occ_t = fus(dev(fr["coords"]), dev(fr["values"]), inputs, scale)

        def fuse(gvals, vals, updated, rel):
            c4 = np.concatenate([np.zeros((len(updated), 1), np.int32), (updated * interval).astype(np.int32)], 1)
            pts = PV.aligned_coords(c4, fr["origin_partial"][None], 0.04, w2ac[None])
            cv = ch_v[scale]
            fv = ON.convgru(sd, f"fusion_nets_voxel.{scale}", gvals[:, :cv], vals[:, :cv], pts, 1, vres)
            fi = ON.convgru(sd, f"fusion_nets_img.{scale}", gvals[:, cv:], vals[:, cv:], pts, 1, vres)
            return np.concatenate([fv, fi], 1)

        r = OGF.fuse_fragment(st, fr["coords"], fr["values"], fr["origin_partial"], fr["tsdf"], fr["occ"],
                              interval, d, fuse=fuse)
        assert np.array_equal(coords.cpu().numpy()[:, 1:], r["updated"] * interval)
        assert np.abs(values.cpu().numpy() - r["fused"]).max() < 1e-3
        assert np.array_equal(tsdf_t.cpu().numpy(), r["tsdf_target"])
        assert np.array_equal(occ_t.cpu().numpy(), r["occ_target"])
        assert np.array_equal(fus.global_volume[scale].C.cpu().numpy(), st.C)


@pytest.mark.parametrize("scale", [1, 2])
def test_stage_call_equals_the_separate_calls(scale):
    """the level's bookkeeping queued as ONE call with device-side counts (eprecon_gru_stage_begin_async + one host read)
    against the separate blocking calls (crop_union, gathers, target_fuse, aligned coordinates, two voxelisations): every
    output and both maps bit for bit over the fragment sequence, with one host read per level instead of four"""
    from eprecon_amd import _lib
    from eprecon_amd.gru_fusion import GRUFusion
    cfg = ModelCfg(N_VOX=[24, 24, 24])
    frags, origin, interval, d, c = sequence(scale)
    outs, reads = [], []
    for staged in (True, False):
        torch.manual_seed(scale + 10)
        fus = GRUFusion(cfg, ch_in=[6, 4, 3], ch_voxel=[4, 3, 2]).cuda()
        fus.stage_call = staged
        seq = []
        before = _lib.HOST_READS
        for fr in frags:
            inputs, _ = make_inputs(fr, origin, gt_lists(scale, fr))
            with torch.no_grad():
                res = fus(dev(fr["coords"]), dev(fr["values"]), inputs, scale)
            seq.append([t.cpu().numpy() for t in res] + [fus.global_volume[scale].C.cpu().numpy(), fus.global_volume[scale].F.cpu().numpy(),
                                                        fus.target_tsdf_volume[scale].C.cpu().numpy(),
                                                        fus.target_tsdf_volume[scale].F.cpu().numpy()])
        reads.append(_lib.HOST_READS - before)
        outs.append(seq)
    for a, b in zip(*outs):
        for x, y in zip(a, b):
            assert x.shape == y.shape and np.array_equal(x, y)
    # counted blocking reads per fragment and level: the origins (these inputs carry no host copy) + ONE for the stage call,
    # against origins + crop_union + target_fuse + the two unique-voxel counts
    assert reads == [2 * len(frags), 5 * len(frags)]
