"""Per-layer A/B of the stride-2 patch kernels (conv_patch2.hip) against the implicit GEMM on the config-2 shapes
(B=16, 256x256, bf16, random operands, hot caches): interleaved rounds in one process, median and min per arm."""
import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/gan-aug-pfa_amd')
import statistics
import torch
from pfa_gan import kernels as K, _lib
B = int(os.environ.get("PFA_B", "16"))
dt = torch.bfloat16
ws = K.Workspace(torch.device('cuda'))
def T(*shape): return torch.randn(shape, device='cuda', dtype=torch.float32).to(dt)
# name, mode, src (h, w, channel parts), n, position grid, launches per step
L = [
 ("e2/c2 fwd", K.CONV_S2, (128, 128, [64]), 128, (64, 64), 4),
 ("e3/c3 fwd", K.CONV_S2, (64, 64, [128]), 256, (32, 32), 4),
 ("e4 fwd",    K.CONV_S2, (32, 32, [256]), 512, (16, 16), 1),
 ("d2 dgrad",  K.CONV_S2, (128, 128, [64]), 256, (64, 64), 1),
 ("d3 dgrad",  K.CONV_S2, (64, 64, [128]), 512, (32, 32), 1),
 ("d4 dgrad",  K.CONV_S2, (32, 32, [256]), 1024, (16, 16), 1),
 ("d2 fwd",    K.TCONV_S2, (64, 64, [128, 128]), 64, (64, 64), 1),
 ("d3 fwd",    K.TCONV_S2, (32, 32, [256, 256]), 128, (32, 32), 1),
 ("d4 fwd",    K.TCONV_S2, (16, 16, [512, 512]), 256, (16, 16), 1),
 ("e2/c2 dgrad", K.TCONV_S2, (64, 64, [128]), 64, (64, 64), 4),
 ("e3/c3 dgrad", K.TCONV_S2, (32, 32, [256]), 128, (32, 32), 4),
 ("e4 dgrad",  K.TCONV_S2, (16, 16, [512]), 256, (16, 16), 1),
]
lib = _lib.lib()
arms = [("ring", {3: 1}), ("patch2", {3: 0})]
if os.environ.get("PFA_ARMS"):          # e.g. PFA_ARMS="dp0=7:0,dp1=7:11,dp2=7:12"
    arms = [(a.split("=")[0], {int(kv.split(":")[0]): int(kv.split(":")[1]) for kv in a.split("=")[1].split(";")})
            for a in os.environ["PFA_ARMS"].split(",")]
rows = []
for name, mode, (h, w, cs), n, (hq, wq), per_step in L:
    srcs = [T(B, h, w, c) for c in cs]
    ctot = sum(cs)
    wgt = T(n, 16, ctot)
    classes = 4 if mode == K.TCONV_S2 else 1
    ho, wo = (2 * hq, 2 * wq) if mode == K.TCONV_S2 else (hq, wq)
    o = torch.empty((B, ho, wo, n), device='cuda', dtype=dt)
    flop = 2.0 * B * hq * wq * classes * n * (4 if mode == K.TCONV_S2 else 16) * ctot
    times = {a: [] for a, _ in arms}
    kern = {}
    for rnd in range(7):
        for arm, knobs in arms:
            for kid in (3, 7):
                lib.pfa_set_knob(kid, knobs.get(kid, 0))
            for _ in range(2 if rnd == 0 else 0):
                K.gather_gemm(mode, srcs, wgt, n, [o], (B, hq, wq), ws)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 10
            e0.record()
            for _ in range(reps):
                K.gather_gemm(mode, srcs, wgt, n, [o], (B, hq, wq), ws)
            e1.record(); torch.cuda.synchronize()
            times[arm].append(e0.elapsed_time(e1) * 1e3 / reps)
            kern[arm] = lib.pfa_last_gather_kernel().decode()
    lib.pfa_set_knob(3, 0)
    lib.pfa_set_knob(7, 0)
    row = [name]
    for arm, _ in arms:
        med, mn = statistics.median(times[arm]), min(times[arm])
        row.append((arm, med, mn, flop / med / 1e6, kern[arm]))
    rows.append((row, per_step))
print(f"{'layer':14s}" + "".join(f"{a:>10s} med us  min us   TF/s" for a, _ in arms))
tot = {a: 0.0 for a, _ in arms}
for row, per_step in rows:
    print(f"{row[0]:14s}" + "".join(f"{'':10s}{r[1]:7.1f} {r[2]:7.1f} {r[3]:6.0f}" for r in row[1:]) + "   " + " | ".join(r[4] for r in row[1:]))
    for r in row[1:]:
        tot[r[0]] += r[1] * per_step
print("per step (us, launches weighted):", {a: round(v, 1) for a, v in tot.items()})
