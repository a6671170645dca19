"""Per-stage wall times of the cfg2 step (sync before/after each stage) + launch counts.
Run on the GPU box:  python tools/profile_cfg2_stages.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eprecon_amd import back_project as BP, grid_ops as GO, sparse as SP
from eprecon_amd.fragment_step import Cfg2Step, LEVELS

step = Cfg2Step(seed=0)
net = step.init_net
for _ in range(3):
    step.run()


def timed(fn, n=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


with torch.no_grad():
    f = step.features_init
    f1 = torch.stack([v[2][0] for v in f]); f2 = torch.stack([v[1][0] for v in f]); f4 = torch.stack([v[0][0] for v in f])
    t, fused = timed(lambda: net.feat_fusion_pre(f1, f2, f4))
    print(f"feat_fusion_pre (direct launches)   {t:7.3f} ms")
    views = [list(t_.unbind(0)) for t_ in (f1, f2, f4)]
    t, fused = timed(lambda: net._fusion_graphed(views))
    print(f"feat_fusion_pre (HIP graph replay)  {t:7.3f} ms")
    fused5 = fused.unsqueeze(1).contiguous()
    t, res = timed(lambda: BP.view_variance(step.coords[2], step.origin, 0.04, fused5, step.krcam[1], 2))
    print(f"view_variance 48^3                  {t:7.3f} ms   n_valid {res['n_valid']}")
    vset = SP.VoxelSet(res["coords"], 2)
    t, _ = timed(lambda: SP.VoxelSet(res["coords"], 2).kernel_map(3))
    print(f"hash build + kernel map             {t:7.3f} ms")
    vset.kernel_map(3)
    t, occ = timed(lambda: net.sparse_stack(res["var"], vset))
    print(f"sparse stack (11 convs + norms)     {t:7.3f} ms")
    x = net.norm0.run(res["var"])
    t, _ = timed(lambda: net.subm1.run(x, vset))
    print(f"  one SubM k3 32->32                {t:7.3f} ms")
    t, _ = timed(lambda: net.similary_1.conv1.conv.run(x, vset))
    print(f"  one SubM k1 32->32                {t:7.3f} ms")
    t, _ = timed(lambda: net.norm1.run(x, residual=x, pre_relu=True))
    print(f"  one rowwise LN                    {t:7.3f} ms")
    t, _ = timed(lambda: net.norm0.run(res['var']))
    print(f"  one BatchNorm                     {t:7.3f} ms")
    t, _ = timed(lambda: GO.init_select(occ, res["coords"], 1, dim=24, cell=4))
    print(f"init_select                         {t:7.3f} ms")
    for name, lvl, interval, mv in LEVELS:
        t, r = timed(lambda: BP.run(step.coords[interval], step.origin, 0.04, step.feats[lvl], step.krcam[lvl], mv))
        print(f"{name}                                {t:7.3f} ms   n_valid {r['n_valid']}")
    t, _ = timed(step.run)
    print(f"whole step                          {t:7.3f} ms")
