"""Stage breakdown of one cfg4 fragment (NeuConNet.forward): wall time per sub-module with a device
synchronisation before and after each (so host launch overhead is included and nothing overlaps).
    python tools/profile_cfg4_stages.py [n_rounds]"""
import collections
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eprecon_amd.fragment_step import Cfg4Step  # noqa: E402

acc = collections.OrderedDict()


def timed(name, fn):
    def wrap(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize()
        acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0)
        return r
    return wrap


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    step = Cfg4Step(seed=0, device=torch.device("cuda"))
    net = step.net
    for _ in range(step.n_fragments):
        step.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(rounds * step.n_fragments):
        step.run()
    torch.cuda.synchronize()
    free = (time.perf_counter() - t0) / (rounds * step.n_fragments) * 1e3
    net.initialization.forward = timed("occupancy_init", net.initialization.forward)
    for i in range(3):
        net.back_projection[i].forward = timed(f"back_project{i}", net.back_projection[i].forward)
        net.sp_convs[i].forward = timed(f"spvcnn{i}", net.sp_convs[i].forward)
        net.tsdf_preds[i].forward = timed(f"heads{i}", net.tsdf_preds[i].forward)
        net.occ_preds[i].forward = timed(f"heads{i}", net.occ_preds[i].forward)
        net.panoptic_preds[i].forward = timed("panoptic_preds", net.panoptic_preds[i].forward)
        net.gru_fusion.fusion_nets_voxel[i].forward = timed(f"convgru{i}", net.gru_fusion.fusion_nets_voxel[i].forward)
        net.gru_fusion.fusion_nets_img[i].forward = timed(f"convgru{i}", net.gru_fusion.fusion_nets_img[i].forward)
    net.gru_fusion.forward = timed("gru_fusion_total", net.gru_fusion.forward)
    net.panoptic_feat_fusion.generate_mask_features = timed("mask_features",
                                                            net.panoptic_feat_fusion.generate_mask_features)
    net.panoptic.forward = timed("mask_decoder", net.panoptic.forward)
    net.prune_to_ancestors = timed("prune", net.prune_to_ancestors)
    import eprecon_amd.neucon_network as NN
    NN.panoptic_post = timed("panoptic_post", NN.panoptic_post)
    n = rounds * step.n_fragments
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step.run()
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / n * 1e3
    print(f"free-running {free:.2f} ms/fragment; with per-stage syncs {total:.2f} ms/fragment")
    s = 0.0
    for k, v in acc.items():
        ms = v / n * 1e3
        if k != "gru_fusion_total":
            s += ms
        print(f"  {k:18s} {ms:7.3f} ms")
    gru_inner = sum(v for k, v in acc.items() if k.startswith("convgru")) / n * 1e3
    print(f"  gru bookkeeping    {acc['gru_fusion_total'] / n * 1e3 - gru_inner:7.3f} ms (gru_fusion_total - convgru*)")
    print(f"  orchestration rest {total - s - (acc['gru_fusion_total'] / n * 1e3 - gru_inner):7.3f} ms")


if __name__ == "__main__":
    with torch.no_grad():
        main()
