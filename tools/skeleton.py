"""What a pfa_gather_gemm launch costs with the work ablated: knob 1 bits 1 no stores, 2 no stats, 4 no MFMA, 8 no DMA."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/gan-aug-pfa_amd')
import torch
from pfa_gan import kernels as K, _lib
B = 16; dt = torch.bfloat16; ws = K.Workspace(torch.device('cuda'))
def T(*shape): return torch.randn(shape, device='cuda').to(dt)
L = [("e2 fwd (16 K-steps)", K.CONV_S2, (128, 128, [64]), 128, (64, 64)),
     ("c4 fwd (64 K-steps)", K.CONV_S1, (32, 32, [256]), 512, (31, 31)),
     ("d3 fwd (4x8 K-steps)", K.TCONV_S2, (32, 32, [256, 256]), 128, (32, 32))]
def timeit(f, reps=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
empty = torch.empty(1, device='cuda')
print("empty torch kernel back-to-back:", round(timeit(lambda: empty.zero_()), 2), "us")
for name, mode, (h, w, cs), n, (hq, wq) in L:
    srcs = [T(B, h, w, c) for c in cs]; wgt = T(n, 16, sum(cs))
    ho, wo = (2 * hq, 2 * wq) if mode == K.TCONV_S2 else (hq, wq)
    o = torch.empty((B, ho, wo, n), device='cuda', dtype=dt)
    row = []
    for dbg in (0, 1, 3, 12, 13, 15):
        _lib.lib().pfa_set_knob(1, dbg)
        row.append((dbg, timeit(lambda: K.gather_gemm(mode, srcs, wgt, n, [o], (B, hq, wq), ws, stats=True))))
    _lib.lib().pfa_set_knob(1, 0)
    print(f"{name:22s} " + " ".join(f"[{d}: {t:.1f}]" for d, t in row))
