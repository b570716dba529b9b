"""Per-layer timing of the MFMA kernels on the BASELINE config-2 shapes (B=16, 256x256), per tile config."""
import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/gan-aug-pfa_amd')
import torch
from pfa_gan import kernels as K, _lib
B = 16
dt = torch.bfloat16
ws = K.Workspace(torch.device('cuda'))
import os
ZERO = os.environ.get('PFA_ZERO') == '1'      # zero operands: the clock the chip holds without data toggling
def T(*shape): return torch.zeros(shape, device='cuda', dtype=dt) if ZERO else torch.randn(shape, device='cuda', dtype=torch.float32).to(dt)
# name, mode, src shape (h,w,c parts), n, grid(hq,wq), out hw, taps
L = [
 ("e2/c2 fwd", K.CONV_S2, (128,128,[64]), 128, (64,64), (64,64)),
 ("e3/c3 fwd", K.CONV_S2, (64,64,[128]), 256, (32,32), (32,32)),
 ("e4 fwd",    K.CONV_S2, (32,32,[256]), 512, (16,16), (16,16)),
 ("e5 fwd",    K.CONV_S2, (16,16,[512]), 512, (8,8), (8,8)),
 ("c4 fwd",    K.CONV_S1, (32,32,[256]), 512, (31,31), (31,31)),
 ("c4 dgrad",  K.CONV_S1_DGRAD, (31,31,[512]), 256, (32,32), (32,32)),
 ("c3 dgrad",  K.TCONV_S2, (32,32,[256]), 128, (32,32), (64,64)),
 ("d5 fwd",    K.TCONV_S2, (8,8,[512,512]), 512, (8,8), (16,16)),
 ("d4 fwd",    K.TCONV_S2, (16,16,[512,512]), 256, (16,16), (32,32)),
 ("d3 fwd",    K.TCONV_S2, (32,32,[256,256]), 128, (32,32), (64,64)),
 ("d4 dgrad",  K.CONV_S2, (32,32,[256]), 1024, (16,16), (16,16)),
 ("d3 dgrad",  K.CONV_S2, (64,64,[128]), 512, (32,32), (32,32)),
 ("d2 dgrad",  K.CONV_S2, (128,128,[64]), 256, (64,64), (64,64)),
 # n = 64 layers
 ("e1 fwd",    K.CONV_S2, (256,256,[8]), 64, (128,128), (128,128)),
 ("d2 fwd",    K.TCONV_S2, (64,64,[128,128]), 64, (64,64), (128,128)),
 ("e2 dgrad",  K.TCONV_S2, (64,64,[128]), 64, (64,64), (128,128)),
 # n <= 16
 ("c5 fwd",    K.CONV_S1, (31,31,[512]), 1, (30,30), (30,30)),
]
def run(cfg):
    tile, loop = cfg
    _lib.lib().pfa_set_knob(0, tile); _lib.lib().pfa_set_knob(7, loop)      # knob 7 = 4: half-step pipelined ring
    out = []
    for name, mode, (h,w,cs), n, (hq,wq), (ho,wo) in L:
        srcs = [T(B,h,w,c) for c in cs]
        ctot = sum(cs)
        wgt = T(n,16,ctot)
        o = torch.empty((B,ho,wo,K.pad_ch(n, dt)), device='cuda', dtype=dt)
        taps = 4 if mode == K.TCONV_S2 else 16
        classes = 4 if mode == K.TCONV_S2 else 1
        flop = 2.0*B*hq*wq*classes*n*taps*ctot
        for _ in range(3): K.gather_gemm(mode, srcs, wgt, n, [o], (B,hq,wq), ws)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps): K.gather_gemm(mode, srcs, wgt, n, [o], (B,hq,wq), ws)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1)*1e3/reps
        out.append((name, us, flop/us/1e6))
    return out
# (tile knob, loop knob): tile 0 auto / 4 = 256x128 / 5 = 256x64 for n <= 64; loop 1 reads-first / 2 DMA-first / 3 DMA spread over the MFMAs
res = {c: run(c) for c in ((0, 0), (0, 6), (0, 7))}      # knob 7: 4 pipelined ring, 6 / 7 staggered ring (step / half-step phases)
print(f"{'layer':12s} " + "  ".join(f"cfg{c}: us   TF/s" for c in res))
for i,(name, *_r) in enumerate(L):
    print(f"{name:12s} " + "  ".join(f"{res[c][i][1]:8.1f} {res[c][i][2]:6.0f}" for c in res))
for c in res:
    print("cfg", c, "total us", sum(r[1] for r in res[c]))
