"""Per-layer timing of pfa_wgrad on the BASELINE config-2 shapes (B=16, 256x256, bf16)."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/gan-aug-pfa_amd')
import os
import torch
from pfa_gan import kernels as K
WS = K.Workspace(torch.device('cuda'))
B=16; dt=torch.bfloat16
def T(*shape): return torch.randn(shape, device='cuda').to(dt)
# name, direct (h,w,[c]), gathered (h,w,[c]), stride
L = [("e1", (128,128,[64]), (256,256,[8]), 2, [64], [3]),
     ("e2/c2", (64,64,[128]), (128,128,[64]), 2, [128], [64]),
     ("e3/c3", (32,32,[256]), (64,64,[128]), 2, [256], [128]),
     ("e4", (16,16,[512]), (32,32,[256]), 2, [512], [256]),
     ("e5", (8,8,[512]), (16,16,[512]), 2, [512], [512]),
     ("e6", (4,4,[512]), (8,8,[512]), 2, [512], [512]),
     ("e7", (2,2,[512]), (4,4,[512]), 2, [512], [512]),
     ("c4", (31,31,[512]), (32,32,[256]), 1, [512], [256]),
     ("c5", (30,30,[8]), (31,31,[512]), 1, [1], [512]),
     ("d7", (2,2,[512]), (4,4,[512]), 2, [512], [512]),
     ("d6", (4,4,[512,512]), (8,8,[512]), 2, [512,512], [512]),
     ("d5", (8,8,[512,512]), (16,16,[512]), 2, [512,512], [512]),
     ("d4", (16,16,[512,512]), (32,32,[256]), 2, [512,512], [256]),
     ("d3", (32,32,[256,256]), (64,64,[128]), 2, [256,256], [128]),
     ("d2", (64,64,[128,128]), (128,128,[64]), 2, [128,128], [64]),
     ("d1", (128,128,[64,64]), (256,256,[8]), 2, [64,64], [3])]
from pfa_gan import _lib as _L
_L.lib().pfa_set_knob(4, int(os.environ.get("WG_LOOP", "0")))     # knob 4: 1 reads first, 2 DMA first, 3 ring tile
tot=0
for name,(dh,dw,dcs),(gh,gw,gcs),stride,dreal,greal in L:
    d=[T(B,dh,dw,c) for c in dcs]; g=[T(B,gh,gw,c) for c in gcs]
    dwt=torch.zeros(sum(dreal)*16*sum(greal), device='cuda')
    f=lambda: K.wgrad(d,g,dwt,stride,dreal,greal,ws=WS,fresh=os.environ.get('WG_FRESH','1')=='1')      # (fresh: as the engines call it - a zeroed arena)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)*1e3/20
    flop=2.0*B*dh*dw*sum(dreal)*16*sum(greal)
    tot+=us
    obytes=4.0*sum(dreal)*16*sum(greal)
    print(f"{name:6s} {us:8.1f} us {flop/us/1e6:7.0f} TF/s   dw {obytes/1e6:6.1f} MB = {obytes/us/1e6:5.2f} TB/s")
print("total", tot)
# ablation on two layers
from pfa_gan import _lib
for name,(dh,dw,dcs),(gh,gw,gcs),stride,dreal,greal in [L[2], L[7]]:
    d=[T(B,dh,dw,c) for c in dcs]; g=[T(B,gh,gw,c) for c in gcs]
    dwt=torch.zeros(sum(dreal)*16*sum(greal), device='cuda')
    row=[]
    for dbg in (0,1,4,8,5,9,13):
        _lib.lib().pfa_set_knob(1, dbg)
        f=lambda: K.wgrad(d,g,dwt,stride,dreal,greal)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        row.append((dbg, e0.elapsed_time(e1)*1e3/20))
    _lib.lib().pfa_set_knob(1, 0)
    print(name, " ".join(f"[dbg {d}: {t:.1f}]" for d,t in row), "(1 no atomics, 4 no MFMA, 8 no DMA)")
