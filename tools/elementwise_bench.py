"""Isolated timing of the HBM-bound BatchNorm passes on the step's tensor sizes (B=16, bf16)."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/gan-aug-pfa_amd')
import torch
from pfa_gan import kernels as K
dt = torch.bfloat16
def timeit(f, reps=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for shape in [(16, 128, 128, 64), (16, 64, 64, 128), (16, 32, 32, 256), (16, 31, 31, 512), (16, 16, 16, 512)]:
    raw = torch.randn(shape, device='cuda').to(dt)
    c = shape[3]
    sc, sh = torch.rand(c, device='cuda') + 0.5, torch.rand(c, device='cuda')
    aL, aR = torch.empty_like(raw), torch.empty_like(raw)
    coef = torch.rand(3 * c, device='cuda')
    dy = torch.randn(shape, device='cuda').to(dt)
    mb = raw.numel() * 2 / 1e6
    t1 = timeit(lambda: K.bn_act(raw, sc, sh, aL, None))
    t2 = timeit(lambda: K.bn_act(raw, sc, sh, aL, aR))
    t3 = timeit(lambda: K.bn_bwd_apply(dy, raw, coef, dy))
    print(f"{str(shape):22s} {mb:6.1f} MB/tensor  bn_act(1 out) {t1:6.1f} us = {2*mb/t1*1e-3:5.2f} TB/s   "
          f"bn_act(2 out) {t2:6.1f} us = {3*mb/t2*1e-3:5.2f} TB/s   bn_bwd_apply {t3:6.1f} us = {3*mb/t3*1e-3:5.2f} TB/s")
