"""Prototype check + timing: patch-staged transposed conv / stride-2 input gradient (k_tconv_patch in
tools/proto/conv_s1_patch.hip) against the product's pfa_gather_gemm(PFA_TCONV_S2) on shapes of the step."""
import ctypes, os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/gan-aug-pfa_amd')
import torch
from pfa_gan import kernels as K
here = os.path.dirname(os.path.abspath(__file__))
proto = ctypes.CDLL(os.path.join(here, "libpfa_proto.so"))
proto.pfa_proto_tconv_patch.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 5 + [ctypes.c_void_p]
proto.pfa_proto_tconv_patch.restype = ctypes.c_int
dt = torch.bfloat16
ws = K.Workspace(torch.device('cuda'))
torch.manual_seed(0)

def timeit(f, reps=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

for name, nb, h, w, cin, n in (("small", 2, 5, 7, 128, 128), ("c3 dgrad", 16, 32, 32, 256, 128), ("e3 dgrad-like", 16, 16, 16, 512, 256),
                               ("d3 fwd-like", 16, 32, 32, 512, 128), ("d4 fwd-like", 16, 16, 16, 1024, 256)):
    x = (torch.randn(nb, h, w, cin, device='cuda') * 0.5).to(dt)
    wt = (torch.randn(n, 16, cin, device='cuda') * 0.05).to(dt)
    ref = torch.empty(nb, 2 * h, 2 * w, n, device='cuda', dtype=dt)
    out = torch.full_like(ref, 7.0)
    call_ref = lambda: K.gather_gemm(K.TCONV_S2, [x], wt, n, [ref], (nb, h, w), ws, splitk=1)
    call_new = lambda: proto.pfa_proto_tconv_patch(x.data_ptr(), wt.data_ptr(), out.data_ptr(), nb, h, w, cin, n,
                                                   torch.cuda.current_stream().cuda_stream)
    call_ref(); rc = call_new(); torch.cuda.synchronize()
    assert rc == 0, rc
    err = (out.float() - ref.float()).abs().max().item(); scale = ref.float().abs().max().item()
    flop = 2.0 * nb * h * w * 4 * n * 4 * cin
    t_ref, t_new = timeit(call_ref), timeit(call_new)
    print(f"{name:14s} {nb}x{h}x{w}x{cin} -> n={n}: max |diff| {err:.4f} of {scale:.2f} ({'OK' if err <= 2e-2 * scale else 'MISMATCH'})"
          f"   product {t_ref:6.1f} us ({flop / t_ref / 1e6:4.0f} TF/s)   patch prototype {t_new:6.1f} us ({flop / t_new / 1e6:4.0f} TF/s)")
