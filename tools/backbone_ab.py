"""2D feeder A/B (f2): per-view loop (the reference, models/neuralrecon.py:53-54) vs one batched pass, 9 x 640x480"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eprecon_amd.backbone import MnasMulti
torch.manual_seed(0)
net = MnasMulti(1.0).cuda().train()
imgs = [torch.randn(1, 3, 480, 640, device="cuda") * 50 for _ in range(9)]
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    a = timeit(lambda: [net(i) for i in imgs])
    b = timeit(lambda: net.forward_views(imgs))
    out = net.forward_views(imgs)
    print(f"NHWC_ENV={os.environ.get('PYTORCH_MIOPEN_SUGGEST_NHWC')} loop {a:.2f} ms  batched {b:.2f} ms  "
          f"channels_last out: {[o.is_contiguous(memory_format=torch.channels_last) for o in out[0]]}")
