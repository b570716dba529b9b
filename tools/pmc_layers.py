"""Run a few layer launches (for rocprofv3 --pmc) - c4 forward with the default tile, DMA-only and MFMA-only ablations."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/gan-aug-pfa_amd')
import torch
from pfa_gan import kernels as K, _lib
B=16; dt=torch.bfloat16; ws=K.Workspace(torch.device('cuda'))
def T(*shape): return torch.randn(shape, device='cuda').to(dt)
srcs=[T(B,32,32,256)]; wgt=T(512,16,256); o=torch.empty((B,31,31,512),device='cuda',dtype=dt)
for dbg in (0, 4, 8):
    _lib.lib().pfa_set_knob(1, dbg)
    for _ in range(3): K.gather_gemm(K.CONV_S1, srcs, wgt, 512, [o], (B,31,31), ws)
    torch.cuda.synchronize()
_lib.lib().pfa_set_knob(1, 0)
d=[T(B,31,31,512)]; g=[T(B,32,32,256)]; dwt=torch.zeros(512*16*256, device='cuda')
for _ in range(3): K.wgrad(d,g,dwt,1,[512],[256])
torch.cuda.synchronize()
