"""Per-stage / per-layer kernel accounting of one cfg4 fragment (NeuConNet.forward, unpipelined).  Run under
    EPRECON_CONV_LOG=<dir>/conv.log rocprofv3 --kernel-trace -d <dir> -o r -- python tools/trace_cfg4_layers.py <dir>
Every sub-module call is bracketed by empty marker launches (eprecon_profile_mark_async: the grid size carries the stage
id); the library appends one line per convolution launch to the log.  tools/summarize_cfg4_layers.py joins the two."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eprecon_amd import _lib  # noqa: E402
from eprecon_amd.fragment_step import Cfg4Step  # noqa: E402

STAGES = []


def marked(name, fn):
    if name not in STAGES:
        STAGES.append(name)
    sid = STAGES.index(name)
    lib = _lib.load()

    def wrap(*a, **k):
        lib.eprecon_profile_mark_async(2 * sid + 10, _lib.current_stream())
        r = fn(*a, **k)
        lib.eprecon_profile_mark_async(2 * sid + 11, _lib.current_stream())
        return r
    return wrap


def main():
    out_dir = sys.argv[1]
    step = Cfg4Step(seed=0, device=torch.device("cuda"))
    net = step.net
    net.initialization.forward = marked("occupancy_init", net.initialization.forward)
    for i in range(3):
        net.back_projection[i].forward = marked(f"back_project{i}", net.back_projection[i].forward)
        net.sp_convs[i].forward = marked(f"spvcnn{i}", net.sp_convs[i].forward)
        net.tsdf_preds[i].forward = marked(f"heads{i}", net.tsdf_preds[i].forward)
        net.occ_preds[i].forward = marked(f"heads{i}", net.occ_preds[i].forward)
        net.panoptic_preds[i].forward = marked("panoptic_preds", net.panoptic_preds[i].forward)
        net.gru_fusion.fusion_nets_voxel[i].forward = marked(f"convgru{i}", net.gru_fusion.fusion_nets_voxel[i].forward)
        net.gru_fusion.fusion_nets_img[i].forward = marked(f"convgru{i}", net.gru_fusion.fusion_nets_img[i].forward)
    net.gru_fusion.forward = marked("gru_fusion", net.gru_fusion.forward)
    net.panoptic_feat_fusion.generate_mask_features = marked("mask_features", net.panoptic_feat_fusion.generate_mask_features)
    net.panoptic.forward = marked("mask_decoder", net.panoptic.forward)
    net.prune_to_ancestors = marked("prune", net.prune_to_ancestors)
    import eprecon_amd.neucon_network as NN
    NN.panoptic_post = marked("panoptic_post", NN.panoptic_post)
    lib = _lib.load()
    for k in range(3 * step.n_fragments):
        lib.eprecon_profile_mark_async(0, _lib.current_stream())      # fragment boundary
        step.run()
    lib.eprecon_profile_mark_async(0, _lib.current_stream())
    torch.cuda.synchronize()
    json.dump(STAGES, open(os.path.join(out_dir, "stages.json"), "w"))


if __name__ == "__main__":
    with torch.no_grad():
        main()
