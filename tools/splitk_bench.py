"""Does slicing K help the launches that give only one workgroup per CU (256 blocks)?"""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/gan-aug-pfa_amd')
import torch
from pfa_gan import kernels as K
import os
from pfa_gan import _lib as _L
_L.lib().pfa_set_knob(0, int(os.environ.get('TILE', '0')))      # 6: 128x64 tile everywhere
B=16; dt=torch.bfloat16; ws=K.Workspace(torch.device('cuda'))
def T(*shape): return torch.randn(shape, device='cuda').to(dt)
L = [("e3/c3 fwd", K.CONV_S2, (64,64,[128]), 256, (32,32), (32,32)),
     ("c4 dgrad",  K.CONV_S1_DGRAD, (31,31,[512]), 256, (32,32), (32,32)),
     ("d4 fwd",    K.TCONV_S2, (16,16,[512,512]), 256, (16,16), (32,32)),
     ("d4 dgrad",  K.CONV_S2, (32,32,[256]), 1024, (16,16), (16,16)),
     ("e4 fwd",    K.CONV_S2, (32,32,[256]), 512, (16,16), (16,16)),
     ("d5 fwd",    K.TCONV_S2, (8,8,[512,512]), 512, (8,8), (16,16)),
     ("c3 dgrad",  K.TCONV_S2, (32,32,[256]), 128, (32,32), (64,64))]
for name, mode, (h,w,cs), n, (hq,wq), (ho,wo) in L:
    srcs=[T(B,h,w,c) for c in cs]; wgt=T(n,16,sum(cs)); o=torch.empty((B,ho,wo,n),device='cuda',dtype=dt)
    row=[]
    for sk in (None, 1, 2, 3, 4, 8):
        f=lambda: K.gather_gemm(mode, srcs, wgt, n, [o], (B,hq,wq), ws, stats=True, splitk=sk)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        row.append((sk, e0.elapsed_time(e1)*1e3/20))
    print(f"{name:10s} " + " ".join(f"[sk={s}: {t:.1f}]" for s,t in row))
