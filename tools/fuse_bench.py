"""The c3 input-gradient launch (TCONV, 16x32x32x256 -> 16x64x64x128) with and without the fused BatchNorm-backward
epilogue, isolated, library-side timing; PFA_DBG ablation bits apply."""
import ctypes, os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/gan-aug-pfa_amd')
import torch
from pfa_gan import kernels as K, _lib
B = 16; dt = torch.bfloat16; ws = K.Workspace(torch.device('cuda'))
def T(*shape): return torch.randn(shape, device='cuda', dtype=torch.float32).to(dt)
lib = _lib.lib()
src = T(B, 32, 32, 256); wgt = T(128, 16, 256)
out = torch.empty((B, 64, 64, 128), device='cuda', dtype=dt)
raw = T(B, 64, 64, 128); g2 = T(B, 64, 64, 128)
f32 = lambda n: torch.rand(n, device='cuda') + 0.5
sc, sh, mu, isd = f32(128), f32(128), f32(128), f32(128)
def med(f, reps=15):
    lib.pfa_gather_timing(1)
    for _ in range(reps): f()
    torch.cuda.synchronize(); lib.pfa_gather_timing(0)
    ms, name, fl, by, sk = ctypes.c_float(), ctypes.c_char_p(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
    ts = []
    for i in range(lib.pfa_gather_timing_count()):
        lib.pfa_gather_timing_get(i, ctypes.byref(ms), ctypes.byref(name), ctypes.byref(fl), ctypes.byref(by), ctypes.byref(sk))
        ts.append(ms.value * 1e3)
    ts = sorted(ts[3:]); return ts[len(ts) // 2]
grid = (B, 32, 32)
plain = lambda: K.gather_gemm(K.TCONV_S2, [src], wgt, 128, [out], grid, ws)
act_only = lambda: K.gather_gemm(K.TCONV_S2, [src], wgt, 128, [out], grid, ws, fuse=dict(out=1, raw=raw, act1=K.ACT_LRELU))
bn = lambda: K.gather_gemm(K.TCONV_S2, [src], wgt, 128, [out], grid, ws,
                           fuse=dict(out=1, raw=raw, act1=K.ACT_LRELU, scale=sc, shift=sh, mean=mu, invstd=isd))
bn_g2 = lambda: K.gather_gemm(K.TCONV_S2, [src], wgt, 128, [out], grid, ws,
                              fuse=dict(out=1, raw=raw, act1=K.ACT_LRELU, g2=g2, act2=K.ACT_RELU, scale=sc, shift=sh, mean=mu, invstd=isd))
stats_only = lambda: K.gather_gemm(K.TCONV_S2, [src], wgt, 128, [out], grid, ws, stats=True)
for dbg in (0, 64, 2, 66, 1):
    lib.pfa_set_knob(1, dbg)
    print(f"dbg {dbg:2d}: plain {med(plain):6.1f}  forward-stats {med(stats_only):6.1f}  fused act {med(act_only):6.1f}  "
          f"fused BN {med(bn):6.1f}  fused BN+g2 {med(bn_g2):6.1f} us")
lib.pfa_set_knob(1, 0)
