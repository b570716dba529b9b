"""A few launches for `rocprofv3 --pmc`: e2/c2-forward-like stride-2 patch kernel (K = 1024), d3-forward-like transposed
(K = 2048 per class), c4 forward on the stride-1 patch kernel.  tools/pmc_patch2.sh collects and prints the counters."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/gan-aug-pfa_amd')
import torch
from pfa_gan import kernels as K
B = 16; dt = torch.bfloat16; ws = K.Workspace(torch.device('cuda'))
def T(*shape): return torch.randn(shape, device='cuda').to(dt)
cases = [(K.CONV_S2, [T(B, 128, 128, 64)], T(128, 16, 64), 128, (B, 64, 64), (B, 64, 64, 128)),
         (K.TCONV_S2, [T(B, 32, 32, 256), T(B, 32, 32, 256)], T(512, 16, 128), 128, (B, 32, 32), (B, 64, 64, 128)),
         (K.CONV_S1, [T(B, 32, 32, 256)], T(512, 16, 256), 512, (B, 31, 31), (B, 31, 31, 512))]
for mode, srcs, w, n, grid, oshape in cases:
    o = torch.empty(oshape, device='cuda', dtype=dt)
    for _ in range(3):
        K.gather_gemm(mode, srcs, w, n, [o], grid, ws, stats=True)
    torch.cuda.synchronize()
# weight gradients on the patch kernel: c4 (stride 1) and e3/c3 (stride 2)
for (dh, dw_, dc), (gh, gw, gc), stride in (((31, 31, 512), (32, 32, 256), 1), ((32, 32, 256), (64, 64, 128), 2)):
    d = [T(B, dh, dw_, dc)]; g = [T(B, gh, gw, gc)]
    dwt = torch.zeros(dc * 16 * gc, device='cuda')
    for _ in range(3):
        K.wgrad(d, g, dwt, stride, [dc], [gc], ws=ws)
    torch.cuda.synchronize()
