"""kernel-level view of the 2D fusion stack (run under rocprofv3 --kernel-trace --stats)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eprecon_amd.fragment_step import Cfg2Step
step = Cfg2Step(seed=0)
net = step.init_net
f = step.features_init
f1 = torch.stack([v[2][0] for v in f]); f2 = torch.stack([v[1][0] for v in f]); f4 = torch.stack([v[0][0] for v in f])
variants = {"default": {}, "no_miopen_bn": {}}
with torch.no_grad():
    for name in ("default", "bn_native", "channels_last"):
        a, b, c = f1, f2, f4
        if name == "channels_last":
            net = net.to(memory_format=torch.channels_last)
            a, b, c = (t.contiguous(memory_format=torch.channels_last) for t in (f1, f2, f4))
        ctx = torch.backends.cudnn.flags(enabled=False) if name == "bn_native" else torch.backends.cudnn.flags(enabled=True)
        with ctx:
            for _ in range(3):
                net.feat_fusion_pre(a, b, c)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20):
                net.feat_fusion_pre(a, b, c)
            torch.cuda.synchronize()
        print(name, (time.perf_counter() - t0) / 20 * 1e3, "ms")
