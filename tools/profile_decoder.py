"""The mask decoder alone at cfg4 size (80 queries, 48 channels, 8 heads, 6 layers; 90k finest voxels and their ancestors):
wall time per call and, with HIP events, the split between the voxel side (mask-logit GEMM, key / value projections, masked
attention) and the replayed query side.
    python tools/profile_decoder.py [n_calls]            # under rocprofv3 --kernel-trace --stats for the per-kernel table"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eprecon_amd.mask3dformer import MultiScaleMaskedTransformerDecoder, panoptic_post  # noqa: E402


def levels(rng, n_fine):
    """a surface-like finest level (a thick spherical shell) and its ancestors at strides 2 and 4 — what ancestor pruning
    leaves of the coarser levels (models/neucon_network.py:516-542)"""
    g = np.stack(np.meshgrid(*[np.arange(96)] * 3, indexing="ij"), -1).reshape(-1, 3)
    r = np.sqrt(((g - 48) ** 2).sum(1))
    fine = g[np.abs(r - 30) < 2.5]
    fine = fine[rng.permutation(len(fine))[:n_fine]]
    fine = fine[np.lexsort((fine[:, 2], fine[:, 1], fine[:, 0]))]
    out = [np.unique(fine // 4 * 4, axis=0), np.unique(fine // 2 * 2, axis=0), fine]
    return [torch.from_numpy(c.astype(np.int32)).cuda() for c in out]


def main():
    n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    dec = MultiScaleMaskedTransformerDecoder(mask_classification=True, num_classes=20, hidden_dim=48, num_queries=80, nheads=8,
                                             dim_feedforward=192, dec_layers=6, pre_norm=False, mask_dim=48).cuda()
    coords = levels(rng, 90000)
    feats = [torch.randn((c.shape[0], 48), device="cuda") for c in coords]
    mask_feat = torch.randn((coords[2].shape[0], 48), device="cuda")
    args = ([f.unsqueeze(0).permute(0, 2, 1) for f in feats], [c[None] for c in coords], mask_feat.unsqueeze(0).permute(0, 2, 1),
            (96, 96, 96))
    for fused in ((True,) if os.environ.get("EPRECON_DECODER_ONLY_FUSED") == "1" else (True, False)):
        dec.use_fused_voxel_side = fused
        for _ in range(4):
            out = dec(*args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_calls):
            out = dec(*args)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n_calls):
            out = dec(*args)
            panoptic_post(out)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"voxel side {'HIP kernels' if fused else 'PyTorch ops'}: decoder {1e3 * (t1 - t0) / n_calls:.3f} ms per call, "
              f"decoder + panoptic_post {1e3 * (t2 - t1) / n_calls:.3f} ms  (levels {[c.shape[0] for c in coords]})")


if __name__ == "__main__":
    with torch.no_grad():
        main()
