"""How many of a sparse convolution's MFMAs multiply zeros, and how many of those a tile-level skip could avoid.
For every distinct 3x3x3 kernel map of a steady-state cfg4 fragment: the share of live (output row, offset) pairs — what the
algorithmic flop count rests on — and the share of (16-row tile, offset) / (32-row wave, offset) groups with AT LEAST ONE live
row: an output-stationary MFMA kernel has to issue the whole 16-row group for an offset as soon as one row has that neighbour.
    python tools/conv_tile_liveness.py > profiles/rNN/conv_tile_liveness.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eprecon_amd import sparse as SP  # noqa: E402
from eprecon_amd.fragment_step import Cfg4Step  # noqa: E402

from eprecon_amd import torchsparse_utils as TU  # noqa: E402

MAPS = []
_orig = SP.VoxelSet.kernel_map
_orig_hier = SP._hierarchy_with_geometry
_orig_pair = TU.register_voxelization_pair


def recording_kernel_map(self, ksize=3):
    fresh = self._k3 is None
    nbr = _orig(self, ksize)
    if fresh:
        MAPS.append((self.stride, nbr))
    return nbr


# (the maps of an SPVCNN pass and of a level's two ConvGRU voxelisations are built inside one library call each: pick them up
# where the call hands them over)
def recording_hierarchy(*a, **k):
    s1, inv, t = _orig_hier(*a, **k)
    s2 = s1._down[0]
    MAPS.extend([(1, s1._k3), (2, s2._k3), (4, s2._down[0]._k3)])
    return s1, inv, t


def recording_pair(*a, **k):
    e1, e2 = _orig_pair(*a, **k)
    MAPS.extend([(1, e1.vset._k3), (1, e2.vset._k3)])
    return e1, e2


def main():
    step = Cfg4Step(seed=0, device=torch.device("cuda"))
    for _ in range(2 * step.n_fragments):
        step.run()
    SP.VoxelSet.kernel_map = recording_kernel_map
    SP._hierarchy_with_geometry = recording_hierarchy
    import eprecon_amd.gru_fusion as GF
    GF.register_voxelization_pair = recording_pair
    for _ in range(step.n_fragments):
        step.run()
    torch.cuda.synchronize()
    print("# cfg4, one pass over the scene's four fragments: every 3x3x3 kernel map built (rows in the order the layers see them)")
    print("# rows | tensor stride | live pairs | live (16-row tile, offset) | live (32-row wave, offset) | executed / live at 16-row granularity")
    tot = [0, 0, 0, 0]
    for stride, nbr in MAPS:
        live = nbr >= 0
        k, n = live.shape
        pad16 = (n + 15) // 16 * 16
        pad32 = (n + 31) // 32 * 32
        l16 = torch.zeros((k, pad16), dtype=torch.bool, device=live.device)
        l16[:, :n] = live
        l32 = torch.zeros((k, pad32), dtype=torch.bool, device=live.device)
        l32[:, :n] = live
        t16 = l16.view(k, -1, 16).any(2)
        t32 = l32.view(k, -1, 32).any(2)
        p, a, b = float(live.float().mean()), float(t16.float().mean()), float(t32.float().mean())
        print(f"{n:8d} {stride:3d}   {p:6.3f}   {a:6.3f}   {b:6.3f}   {a / max(p, 1e-9):5.2f}")
        tot[0] += int(live.sum()); tot[1] += k * n; tot[2] += int(t16.sum()) * 16; tot[3] += int(t32.sum()) * 32
    print(f"# all maps: live pairs {tot[0] / tot[1]:.3f} of the output-stationary work; with (16-row tile, offset) skipping the kernels would "
          f"still execute {tot[2] / tot[1]:.3f} of it ({tot[2] / tot[0]:.2f} x the live pairs), with (32-row, offset) skipping {tot[3] / tot[1]:.3f}")


if __name__ == "__main__":
    with torch.no_grad():
        main()
