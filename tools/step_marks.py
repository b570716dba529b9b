"""Where the three streams of the training step are, without a profiler: timing events (GanTrainer.marks) at the phase
boundaries of the step, averaged over the steps after a warm-up.  Times are microseconds after the step's first mark on
the main stream; a negative time is work of this step that ran under the tail of the previous one.

    python tools/step_marks.py [--batch 16] [--size 256] [--steps 20]
(eager enqueue only: a launch list re-issues library calls, not these events)
"""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gan-aug-pfa_amd"))
import torch
import pfa_gan

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=20)
ap.add_argument("--no-early-real", action="store_true")
args = ap.parse_args()

torch.manual_seed(0)
gen = pfa_gan.UNetGenerator(3, 3).cuda()
disc = pfa_gan.NLayerDiscriminator(6).cuda()
tr = pfa_gan.GanTrainer(gen, disc, precision="bf16", use_graph=False,
                        inputs_resident=not args.no_early_real)
g = torch.Generator(device="cuda").manual_seed(1)
pool = []
for _ in range(4):
    a = torch.rand((args.batch, 3, args.size, args.size), device="cuda", generator=g) * 2 - 1
    b = torch.rand((args.batch, 3, args.size, args.size), device="cuda", generator=g) * 2 - 1
    pool.append((tr.pack(a), tr.pack(b)))
torch.cuda.synchronize()
for i in range(args.warmup):
    tr.step_packed(*pool[i % 4])
tr.marks = []
for i in range(args.steps):
    tr.step_packed(*pool[i % 4])
torch.cuda.synchronize()
marks, tr.marks = tr.marks, None
starts = [i for i, (n, _) in enumerate(marks) if n == "step"]
per = len(marks) // len(starts)
names = [n for n, _ in marks[:per]]
acc = [0.0] * per
wall = 0.0
n = 0
for k in range(1, len(starts) - 1):                      # (skip the first and last marked steps)
    s0 = marks[starts[k]][1]
    for j in range(per):
        acc[j] += s0.elapsed_time(marks[starts[k] + j][1]) * 1e3
    wall += s0.elapsed_time(marks[starts[k + 1]][1]) * 1e3
    n += 1
print(f"batch {args.batch} {args.size}x{args.size}  step {wall / n:8.1f} us  ({n} steps)")
for j in sorted(range(per), key=lambda j: acc[j]):
    print(f"  {acc[j] / n:9.1f}  {names[j]}")
