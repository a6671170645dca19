"""one layer (dense-grid 48^3 list, 27 offsets, C_in -> C_out) launched a few times: the target of tools/r02_conv_pmc.sh"""
import sys
import numpy as np
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eprecon_amd import sparse as SP
from eprecon_amd import synthetic as S

ci, co = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 32)
dev = torch.device("cuda")
rng = np.random.default_rng(0)
coords = S.dense_coords((96, 96, 96), 2)
keep = np.sort(rng.choice(len(coords), 94000, replace=False))
vs = SP.VoxelSet(torch.from_numpy(np.ascontiguousarray(coords[keep])).to(dev), 2)
nbr = vs.kernel_map(3)
x = torch.randn(vs.n, ci, device=dev)
w = torch.randn(27, ci, co, device=dev) * 0.05
out = torch.empty(vs.n, co, device=dev)
with torch.no_grad():
    for _ in range(6):
        SP.conv_stats(x, w, nbr, out=out)
torch.cuda.synchronize()
