"""run the SubM k3 32->32 conv of the occupancy-init stack N times (for rocprofv3 --pmc)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eprecon_amd import back_project as BP, sparse as SP
from eprecon_amd.fragment_step import Cfg2Step
step = Cfg2Step(seed=0); net = step.init_net
with torch.no_grad():
    f = step.features_init
    f1 = torch.stack([v[2][0] for v in f]); f2 = torch.stack([v[1][0] for v in f]); f4 = torch.stack([v[0][0] for v in f])
    fused = net.feat_fusion_pre(f1, f2, f4).unsqueeze(1)
    res = BP.view_variance(step.coords[2], step.origin, 0.04, fused, step.krcam[1], 2)
    vset = SP.VoxelSet(res["coords"], 2); vset.kernel_map(3)
    x = net.norm0.run(res["var"])
    for _ in range(5): net.subm1.run(x, vset)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(int(os.environ.get("REPS", "20"))): net.subm1.run(x, vset)
    torch.cuda.synchronize()
    print("conv us", (time.perf_counter() - t0) / int(os.environ.get("REPS", "20")) * 1e6, "rows", x.shape[0])
