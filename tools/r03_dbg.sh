#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for case in 0 1 2; do
for wv in 2 4; do
EPRECON_D3_WV=$wv EPRECON_CONV_DENSE3D=3 EPRECON_BN_TICKET=1 timeout 60 python - $case <<'PY' 2>&1 | grep -v amdgpu.ids | tail -3
import sys, os, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from eprecon_amd import sparse as SP
from test_dense_conv3d_gpu import dev, grid_set
class O: pass
case = int(sys.argv[1])
dims, fill, cin, cout = [((48,48,48),0.85,32,1), ((20,14,24),0.6,16,16), ((48,48,48),0.85,32,32)][case]
rng = np.random.default_rng(5)
c = grid_set(rng, dims, 2, fill)
vs = SP.VoxelSet(dev(c), 2, dims=dims); dm = SP.DenseMap(vs, dims)
owner = O(); gamma, beta = torch.rand(cout, device='cuda') + 0.5, torch.randn(cout, device='cuda')
for rep in range(3):
    x = torch.randn((len(c), cin), device='cuda'); w = torch.randn((27, cin, cout), device='cuda') / (27*cin)**0.5
    print('kind', dm.kind(x, cin, cout, stats=True, fused=True), flush=True)
    out, partial, aff = SP.conv_stats(x, w, dm, bn=(gamma, beta, 1e-5), owner=owner)
    torch.cuda.synchronize()
    print('case', case, 'wv', os.environ['EPRECON_D3_WV'], 'rep', rep, 'ok', float(partial[:,0,0].sum()), len(c), flush=True)
PY
done
done
