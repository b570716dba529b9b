"""Per-layer A/B of the weight-gradient kernels on the config-2 shapes (B=16, 256x256, bf16, random operands): patch kernel
(wgrad_patch.hip, knob 4 = 0) against the 128x128-tile kernel (wgrad.hip, knob 4 = 7); interleaved rounds, median / min.
Times include the reduce kernel of a slab launch."""
import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/gan-aug-pfa_amd')
import statistics
import torch
from pfa_gan import kernels as K, _lib
B = 16
dt = torch.bfloat16
ws = K.Workspace(torch.device('cuda'))
def T(*shape): return torch.randn(shape, device='cuda', dtype=torch.float32).to(dt)
# name, stride, direct (h, w, channel parts), gathered (h, w, channels), launches per step
L = [
 ("e2/c2", 2, (64, 64, [128]), (128, 128, 64), 3),
 ("e3/c3", 2, (32, 32, [256]), (64, 64, 128), 3),
 ("e4",    2, (16, 16, [512]), (32, 32, 256), 1),
 ("c4",    1, (31, 31, [512]), (32, 32, 256), 2),
 ("d4",    2, (16, 16, [512, 512]), (32, 32, 256), 1),
 ("d3",    2, (32, 32, [256, 256]), (64, 64, 128), 1),
 ("d2",    2, (64, 64, [128, 128]), (128, 128, 64), 1),
]
lib = _lib.lib()
arms = [("wgrad.hip", 7), ("patch", 0)]
tot = {a: 0.0 for a, _ in arms}
print(f"{'layer':8s}" + "".join(f"{a:>12s} med us  min us   TF/s" for a, _ in arms))
for name, stride, (hq, wq, dparts), (hg, wg, gc), per_step in L:
    direct = [T(B, hq, wq, c) for c in dparts]
    gathered = [T(B, hg, wg, gc)]
    dsum = sum(dparts)
    dw = torch.zeros((dsum, 16, gc), device='cuda', dtype=torch.float32)
    flop = 2.0 * B * hq * wq * 16 * dsum * gc
    times = {a: [] for a, _ in arms}
    for rnd in range(7):
        for arm, knob in arms:
            lib.pfa_set_knob(4, knob)
            for _ in range(2 if rnd == 0 else 0):
                K.wgrad(direct, gathered, dw, stride, dparts, [gc], ws=ws)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 10
            e0.record()
            for _ in range(reps):
                K.wgrad(direct, gathered, dw, stride, dparts, [gc], ws=ws)
            e1.record(); torch.cuda.synchronize()
            times[arm].append(e0.elapsed_time(e1) * 1e3 / reps)
    lib.pfa_set_knob(4, 0)
    row = f"{name:8s}"
    for arm, _ in arms:
        med, mn = statistics.median(times[arm]), min(times[arm])
        row += f"{'':12s}{med:7.1f} {mn:7.1f} {flop / med / 1e6:6.0f}"
        tot[arm] += med * per_step
    print(row)
print("per step (us, launches weighted):", {a: round(v, 1) for a, v in tot.items()})
