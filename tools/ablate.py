"""In-kernel ablation of pfa_gather_gemm on a few BASELINE config-2 layers (developer tool).
knob 1 bits: 1 no output stores, 2 no stats, 4 no MFMA, 8 no DMA, 128 no LDS fragment reads (staggered ring only).  knob 0: tile (0 auto, 4 = 256x128); knob 2: K-loop (1 reads-first, 2 DMA-first)."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/gan-aug-pfa_amd')
import torch
from pfa_gan import kernels as K, _lib
B=16; dt=torch.bfloat16; ws=K.Workspace(torch.device('cuda'))
def T(*shape): return torch.randn(shape, device='cuda').to(dt)
L = [("e2 fwd", K.CONV_S2, (128,128,[64]), 128, (64,64), (64,64)),
     ("c4 fwd", K.CONV_S1, (32,32,[256]), 512, (31,31), (31,31)),
     ("d3 dgrad", K.CONV_S2, (64,64,[128]), 512, (32,32), (32,32))]
def timeit(f, reps=50):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e3/reps
for name, mode, (h,w,cs), n, (hq,wq), (ho,wo) in L:
    srcs=[T(B,h,w,c) for c in cs]; wgt=T(n,16,sum(cs)); o=torch.empty((B,ho,wo,n),device='cuda',dtype=dt)
    for cfg in ((0, 0), (0, 6)):
        _lib.lib().pfa_set_knob(0, cfg[0]); _lib.lib().pfa_set_knob(7, cfg[1])       # knob 7 = 6: staggered ring
        row=[]
        for dbg in (0, 8, 0):
            _lib.lib().pfa_set_knob(1, dbg)
            row.append((dbg, timeit(lambda: K.gather_gemm(mode, srcs, wgt, n, [o], (B,hq,wq), ws))))
        _lib.lib().pfa_set_knob(1, 0)
        lab = lambda d: "full" if d == 0 else "-" + ",".join(n for b, n in ((1, "stores"), (2, "stats"), (4, "mfma"), (8, "dma"), (128, "ldsreads"), (32, "burst")) if d & b)
        print(f"{name:9s} cfg{cfg}: " + " ".join(f"[{lab(d)}: {t:.1f}]" for d, t in row))
_lib.lib().pfa_set_knob(0, 0); _lib.lib().pfa_set_knob(7, 0)
