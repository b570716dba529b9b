"""Prototype check + timing: patch-staged Conv2d(k4,s1,p1) forward (tools/proto/conv_s1_patch.hip) against the product's
pfa_gather_gemm on the discriminator's c4 shape (16 x 32x32 x 256 -> 31x31 x 512, bf16)."""
import ctypes, os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/gan-aug-pfa_amd')
import torch
from pfa_gan import kernels as K
here = os.path.dirname(os.path.abspath(__file__))
proto = ctypes.CDLL(os.path.join(here, "libpfa_proto.so"))
proto.pfa_proto_conv_s1_patch.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 7 + [ctypes.c_void_p]
proto.pfa_proto_conv_s1_patch.restype = ctypes.c_int
dt = torch.bfloat16
ws = K.Workspace(torch.device('cuda'))
torch.manual_seed(0)
for nb, hin, cin, n in ((2, 9, 64, 128), (3, 20, 128, 256), (16, 32, 256, 512)):
    x = (torch.randn(nb, hin, hin, cin, device='cuda') * 0.5).to(dt)
    w = (torch.randn(n, 16, cin, device='cuda') * 0.05).to(dt)
    ref = torch.empty(nb, hin - 1, hin - 1, n, device='cuda', dtype=dt)
    out = torch.full_like(ref, 7.0)
    K.gather_gemm(K.CONV_S1, [x], w, n, [ref], (nb, hin - 1, hin - 1), ws, splitk=1)
    for stag in (0, 1, 2):
        out.fill_(7.0)
        rc = proto.pfa_proto_conv_s1_patch(x.data_ptr(), w.data_ptr(), out.data_ptr(), nb, hin, hin, cin, n, stag, 0,
                                           torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert rc == 0, rc
        err = (out.float() - ref.float()).abs().max().item()
        scale = ref.float().abs().max().item()
        print(f"nb={nb} {hin}x{hin}x{cin} -> n={n} staggered={stag}: max |diff| {err:.4f} of max |ref| {scale:.2f}  ({'OK' if err <= 2e-2 * scale else 'MISMATCH'})")

    def timeit(f, reps=30):
        for _ in range(5): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps
    flop = 2.0 * nb * (hin - 1) ** 2 * n * 16 * cin
    t_ref = timeit(lambda: K.gather_gemm(K.CONV_S1, [x], w, n, [ref], (nb, hin - 1, hin - 1), ws, splitk=1))
    ts = [timeit(lambda st=st: proto.pfa_proto_conv_s1_patch(x.data_ptr(), w.data_ptr(), out.data_ptr(), nb, hin, hin, cin, n, st, 0,
                                                             torch.cuda.current_stream().cuda_stream)) for st in (0, 1, 2)]
    print(f"   product {t_ref:7.1f} us ({flop / t_ref / 1e6:5.0f} TF/s)   patch {ts[0]:7.1f} us ({flop / ts[0] / 1e6:5.0f} TF/s)"
          f"   patch + staggered phases {ts[1]:7.1f} us ({flop / ts[1] / 1e6:5.0f} TF/s)"
          f"   patch, 4 waves of 128x64 {ts[2]:7.1f} us ({flop / ts[2] / 1e6:5.0f} TF/s)")
    if nb == 16:      # ablations of the plain loop on the c4 shape (results are wrong when a bit is set)
        lab = {0: "full", 4: "-dma", 1: "-mfma", 2: "-ldsreads", 5: "-mfma,dma (reads only)", 6: "-reads,dma (mfma only)", 3: "-mfma,reads (dma only)", 7: "empty loop"}
        print("   ablations:", "  ".join(f"[{lab[d]}: " + "%.1f" % timeit(lambda d=d: proto.pfa_proto_conv_s1_patch(
            x.data_ptr(), w.data_ptr(), out.data_ptr(), nb, hin, hin, cin, n, 0, d, torch.cuda.current_stream().cuda_stream)) + "]" for d in lab))
        print("   ablations, 4 waves:", "  ".join(f"[{lab[d]}: " + "%.1f" % timeit(lambda d=d: proto.pfa_proto_conv_s1_patch(
            x.data_ptr(), w.data_ptr(), out.data_ptr(), nb, hin, hin, cin, n, 2, d, torch.cuda.current_stream().cuda_stream)) + "]" for d in lab))
