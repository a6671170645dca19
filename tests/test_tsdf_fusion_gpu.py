"""GPU: HIP TSDF integration (csrc/tsdf_fusion.hip through the C ABI) against the reference's own
TSDFVolumeTorch.integrate (tests/golden/tsdf_fusion.npz) and the numpy oracle: integration counts and occupancy
bit exact, TSDF values bit exact (fp32, same operation order); frame-at-a-time == all views in one launch."""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import cases  # noqa: E402
from oracle import tsdf_fusion as OT  # noqa: E402
from test_oracle_tsdf_fusion import check_level  # noqa: E402


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "tsdf_fusion.npz"))


@pytest.fixture(scope="module")
def case():
    window, depths, intr, poses = cases.tsdf_case()
    return window, depths, intr, poses, torch.from_numpy(depths).cuda()


@pytest.mark.parametrize("lvl", cases.TSDF_LEVELS)
def test_matches_reference_golden(gold, case, lvl):
    from eprecon_amd.tsdf_fusion import TSDFVolumeHIP
    window, depths, intr, poses, d_dev = case
    dims = [n // 2 ** lvl for n in window["n_vox"]]
    vol = TSDFVolumeHIP(torch.tensor(dims), torch.from_numpy(window["vol_origin_partial"]), 0.04 * 2 ** lvl, margin=3)
    # the world->camera matrices the reference computed (torch.inverse(pose) in the build container): the last bits
    # of a float 4x4 inverse differ between hosts (LAPACK build), and a pixel decision can hinge on them
    w2c = gold["world2cam"]
    vol.integrate_views(d_dev, torch.from_numpy(intr), torch.from_numpy(poses), world2cam=w2c)
    here = np.stack([torch.inverse(torch.from_numpy(p).float()).numpy() for p in poses])
    assert np.abs(here - w2c).max() < 1e-6
    tsdf, weight = (t.cpu().numpy() for t in vol.get_volume())
    check_level(gold, lvl, tsdf, weight, vol.occupancy().cpu().numpy(), 0.0 if lvl < 2 else 2e-6)
    # and the numpy oracle: everything bit exact
    o_t, o_w, o_occ = OT.fuse_views(dims, window["vol_origin_partial"], 0.04 * 2 ** lvl, depths, intr, w2c)
    assert np.array_equal(tsdf, o_t) and np.array_equal(weight, o_w) and np.array_equal(vol.occupancy().cpu().numpy(), o_occ)


def test_frame_by_frame_equals_fused_launch_and_cuda_variant(case):
    from eprecon_amd.tsdf_fusion import TSDFVolumeHIP
    window, depths, intr, poses, d_dev = case
    dims, vs = [48, 48, 48], 0.08
    a = TSDFVolumeHIP(torch.tensor(dims), torch.from_numpy(window["vol_origin_partial"]), vs)
    b = TSDFVolumeHIP(torch.tensor(dims), torch.from_numpy(window["vol_origin_partial"]), vs)
    a.integrate_views(d_dev, torch.from_numpy(intr), torch.from_numpy(poses))
    for v in range(len(depths)):     # the reference's call pattern (datasets/transforms.py:288-293)
        b.integrate(d_dev[v], torch.from_numpy(intr[v]), torch.from_numpy(poses[v]), obs_weight=1.)
    assert torch.equal(a.get_volume()[0], b.get_volume()[0]) and torch.equal(a.get_volume()[1], b.get_volume()[1])
    assert torch.equal(a.occupancy(), b.occupancy())
    c = TSDFVolumeHIP(torch.tensor(dims), torch.from_numpy(window["vol_origin_partial"]), vs, variant="cuda")
    c.integrate_views(d_dev, torch.from_numpy(intr), torch.from_numpy(poses))
    ref = OT.fuse_views(dims, window["vol_origin_partial"], vs, depths, intr, poses, variant="cuda")
    assert np.array_equal(c.get_volume()[1].cpu().numpy(), ref[1])
    assert np.abs(c.get_volume()[0].cpu().numpy() - ref[0]).max() < 1e-6
    c.reset()
    assert float(c.get_volume()[1].sum()) == 0.0 and float(c.get_volume()[0].min()) == 1.0
