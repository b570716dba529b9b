"""Ablation of the stride-2 patch kernel's phases on the e2/c2-forward shape (pfa_set_knob(7, 11..17): no MFMA / no fragment
reads / no DMA and their combinations): stamps per workgroup in shader cycles."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/gan-aug-pfa_amd')
import torch
from pfa_gan import kernels as K, _lib
B = 16; dt = torch.bfloat16; ws = K.Workspace(torch.device('cuda'))
def T(*shape): return torch.randn(shape, device='cuda').to(dt)
lib = _lib.lib()
buf = torch.zeros(8192 * 16, dtype=torch.int64, device='cuda')
shapes = [("e2/c2 fwd K=1024", [T(B, 128, 128, 64)], T(128, 16, 64), 128, (B, 64, 64)),
          ("d2-dgrad-like K=1024 n=256", [T(B, 128, 128, 64)], T(256, 16, 64), 256, (B, 64, 64)),
          ("K=4096 (256 ch)", [T(B, 128, 128, 256)], T(128, 16, 256), 128, (B, 64, 64))]
names = {0: "full", 11: "no MFMA", 12: "no reads", 14: "no DMA", 13: "no MFMA, no reads", 15: "no MFMA, no DMA", 16: "no reads, no DMA",
         17: "barriers + waits + addresses only", 18: "full, no counted vmcnt wait in the loop", 19: "full, no vmcnt wait, no barrier"}
for sname, srcs, w, n, grid in shapes:
    o = torch.empty((B, 64, 64, n), device='cuda', dtype=dt)
    print(sname)
    for knob in (0, 18, 19, 11, 12, 14, 13, 15, 16, 17):
        lib.pfa_set_knob(7, knob)
        for _ in range(2):
            K.gather_gemm(K.CONV_S2, srcs, w, n, [o], grid, ws)
        torch.cuda.synchronize()
        buf.zero_()
        lib.pfa_set_debug_buffer(buf.data_ptr(), buf.numel() * 8)
        K.gather_gemm(K.CONV_S2, srcs, w, n, [o], grid, ws)
        torch.cuda.synchronize()
        lib.pfa_set_debug_buffer(None, 0)
        st = buf.view(-1, 16).cpu()
        st = st[st[:, 7] != 0].double()
        loop = float((st[:, 2] - st[:, 1]).median()); tot = float((st[:, 7] - st[:, 0]).median())
        steps = srcs[0].shape[3] // 64 * 16
        print(f"  {names[knob]:36s} loop {loop:8.0f} cycles = {loop / steps:7.0f} per tap step   workgroup {tot:8.0f}")
    lib.pfa_set_knob(7, 0)
