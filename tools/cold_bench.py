"""Hot vs cold-cache time of a few implicit-GEMM layers (library-side HIP-event timing): before each timed launch the
caches are either left alone (hot: the operands of the previous repetition are still in L2 / MALL) or flushed by writing a
1 GiB buffer (cold), or the INPUT alone is re-written by a kernel just before (as in the step: produced by the previous
kernel, weights cold)."""
import ctypes, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/gan-aug-pfa_amd')
import torch
from pfa_gan import kernels as K, _lib
B = 16; dt = torch.bfloat16; ws = K.Workspace(torch.device('cuda'))
def T(*shape): return torch.randn(shape, device='cuda', dtype=torch.float32).to(dt)
L = [("e2 fwd", K.CONV_S2, (128, 128, [64]), 128, (64, 64), (64, 64)),
     ("c3 dgrad", K.TCONV_S2, (32, 32, [256]), 128, (32, 32), (64, 64)),
     ("d2 dgrad", K.CONV_S2, (128, 128, [64]), 256, (64, 64), (64, 64)),
     ("c4 fwd", K.CONV_S1, (32, 32, [256]), 512, (31, 31), (31, 31))]
lib = _lib.lib()
junk = torch.empty(1 << 28, device='cuda', dtype=torch.float32)      # 1 GiB
def timed(f, pre, reps=12):
    lib.pfa_gather_timing(1)
    for _ in range(reps):
        pre()
        f()
    torch.cuda.synchronize()
    lib.pfa_gather_timing(0)
    ms, name, fl, by, sk = ctypes.c_float(), ctypes.c_char_p(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
    ts = []
    for i in range(lib.pfa_gather_timing_count()):
        lib.pfa_gather_timing_get(i, ctypes.byref(ms), ctypes.byref(name), ctypes.byref(fl), ctypes.byref(by), ctypes.byref(sk))
        ts.append(ms.value * 1e3)
    ts = sorted(ts[2:])
    return ts[len(ts) // 2]
for name, mode, (h, w, cs), n, (hq, wq), (ho, wo) in L:
    srcs = [T(B, h, w, c) for c in cs]; wgt = T(n, 16, sum(cs))
    o = torch.empty((B, ho, wo, n), device='cuda', dtype=dt)
    f = lambda: K.gather_gemm(mode, srcs, wgt, n, [o], (B, hq, wq), ws)
    hot = timed(f, lambda: None)
    cold = timed(f, lambda: junk.fill_(1.0))
    def rewrite():
        junk.fill_(1.0)
        for s in srcs: s.mul_(1.0)          # the input produced by the kernel just before, weights cold
    warm_in = timed(f, rewrite)
    print(f"{name:9s} hot {hot:6.1f} us   cold {cold:6.1f} us   input fresh / weights cold {warm_in:6.1f} us")
