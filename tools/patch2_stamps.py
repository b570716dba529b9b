"""Where a stride-2 patch launch spends its time: four shader-clock stamps per workgroup (pfa_set_debug_buffer) on the
config-2 shapes -> median cycles of prologue / K loop / epilogue, the spread of workgroup start times, and the launch's
event time."""
import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/gan-aug-pfa_amd')
import torch
from pfa_gan import kernels as K, _lib
B = int(os.environ.get("PFA_B", "16"))
dt = torch.bfloat16
ws = K.Workspace(torch.device('cuda'))
def T(*shape): return torch.randn(shape, device='cuda', dtype=torch.float32).to(dt)
L = [
 ("e2/c2 fwd", K.CONV_S2, (128, 128, [64]), 128, (64, 64)),
 ("e3/c3 fwd", K.CONV_S2, (64, 64, [128]), 256, (32, 32)),
 ("d2 dgrad",  K.CONV_S2, (128, 128, [64]), 256, (64, 64)),
 ("d3 dgrad",  K.CONV_S2, (64, 64, [128]), 512, (32, 32)),
 ("d4 dgrad",  K.CONV_S2, (32, 32, [256]), 1024, (16, 16)),
 ("d2 fwd",    K.TCONV_S2, (64, 64, [128, 128]), 64, (64, 64)),
 ("d3 fwd",    K.TCONV_S2, (32, 32, [256, 256]), 128, (32, 32)),
 ("d4 fwd",    K.TCONV_S2, (16, 16, [512, 512]), 256, (16, 16)),
 ("e2/c2 dgrad", K.TCONV_S2, (64, 64, [128]), 64, (64, 64)),
 ("e3/c3 dgrad", K.TCONV_S2, (32, 32, [256]), 128, (32, 32)),
 ("e1 fwd (image)", K.CONV_S2, (256, 256, [8]), 64, (128, 128)),
 ("c1 fwd (image)", K.CONV_S2, (256, 256, [8, 8]), 64, (128, 128)),
]
lib = _lib.lib()
buf = torch.zeros(8192 * 16, dtype=torch.int64, device='cuda')
print(f"{'layer':14s} {'wgs':>5s} {'event us':>9s} | median cycles: prologue    loop  to-LDS    rows shuffle  reduce   drain |   total")
for name, mode, (h, w, cs), n, (hq, wq) in L:
    srcs = [T(B, h, w, c) for c in cs]
    wgt = T(n, 16, sum(cs))
    ho, wo = (2 * hq, 2 * wq) if mode == K.TCONV_S2 else (hq, wq)
    o = torch.empty((B, ho, wo, n), device='cuda', dtype=dt)
    stats = os.environ.get("PFA_STATS") == "1"
    for _ in range(3):
        K.gather_gemm(mode, srcs, wgt, n, [o], (B, hq, wq), ws, stats=stats)
    torch.cuda.synchronize()
    buf.zero_()
    lib.pfa_set_debug_buffer(buf.data_ptr(), buf.numel() * 8)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    K.gather_gemm(mode, srcs, wgt, n, [o], (B, hq, wq), ws, stats=stats)
    e1.record(); torch.cuda.synchronize()
    lib.pfa_set_debug_buffer(None, 0)
    st = buf.view(-1, 16).cpu()
    st = st[st[:, 7] != 0]
    if st.shape[0] == 0:
        print(f"{name:14s} (no stamps: {lib.pfa_last_gather_kernel().decode()} does not write them)")
        continue
    d = st.double()
    seg = [(d[:, k + 1] - d[:, k]) for k in range(7)]
    if not stats:       # stamps 5 is not written without statistics
        seg[4] = d[:, 6] - d[:, 4]; seg[5] = seg[4] * 0
    med = [float(x.median()) for x in seg]
    tot = float((d[:, 7] - d[:, 0]).median())
    rt0, rt1 = d[:, 8], d[:, 9]                   # 100 MHz real-time counter: one time base for the whole chip
    span = float(rt1.max() - rt0.min()) / 100.0
    first = float((rt0 - rt0.min()).median()) / 100.0
    life = float((rt1 - rt0).median()) / 100.0
    mhz = tot / max(life, 1e-9)
    print(f"{name:14s} {st.shape[0]:5d} {e0.elapsed_time(e1)*1e3:9.1f} | " + " ".join(f"{m:7.0f}" for m in med) + f" | {tot:7.0f} | "
          f"first start -> last end {span:6.1f} us, median start offset {first:5.1f} us, median life {life:5.1f} us ({mhz:5.0f} MHz)")
