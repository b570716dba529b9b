"""A/B of the stride-1 patch kernel on the c4 shapes (forward with statistics; input gradient, K-sliced): MFMA 16x16x32 form
(pfa_set_knob(3, 6)), the same two-phase loop on 32x32x16 (knob 3 = 8) and the in-place read-ahead loop (default).  Interleaved rounds, median / min."""
import sys, statistics
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/gan-aug-pfa_amd')
import torch
from pfa_gan import kernels as K, _lib
B = 16; dt = torch.bfloat16; ws = K.Workspace(torch.device('cuda'))
def T(*shape): return torch.randn(shape, device='cuda').to(dt)
lib = _lib.lib()
cases = [("c4 fwd", K.CONV_S1, [T(B, 32, 32, 256)], T(512, 16, 256), 512, (B, 31, 31), (B, 31, 31, 512), True),
         ("c4 dgrad", K.CONV_S1_DGRAD, [T(B, 31, 31, 512)], T(256, 16, 512), 256, (B, 32, 32), (B, 32, 32, 256), False)]
for name, mode, srcs, w, n, grid, oshape, stats in cases:
    o = torch.empty(oshape, device='cuda', dtype=dt)
    times = {6: [], 8: [], 0: []}
    for rnd in range(7):
        for knob in (6, 8, 0):
            lib.pfa_set_knob(3, knob)
            for _ in range(2 if rnd == 0 else 0):
                K.gather_gemm(mode, srcs, w, n, [o], grid, ws, stats=stats)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                K.gather_gemm(mode, srcs, w, n, [o], grid, ws, stats=stats)
            e1.record(); torch.cuda.synchronize()
            times[knob].append(e0.elapsed_time(e1) * 100)
    lib.pfa_set_knob(3, 0)
    flop = 2.0 * B * 31 * 31 * 512 * 16 * 256
    print(f"{name:9s} two-phase 16x16x32: {statistics.median(times[6]):6.1f} us (min {min(times[6]):6.1f})   two-phase 32x32x16: "
          f"{statistics.median(times[8]):6.1f} us (min {min(times[8]):6.1f})   in-place loop: {statistics.median(times[0]):6.1f} us "
          f"(min {min(times[0]):6.1f}) = {flop / statistics.median(times[0]) / 1e6:5.0f} TFLOP/s   [{lib.pfa_last_gather_kernel().decode()}]")
