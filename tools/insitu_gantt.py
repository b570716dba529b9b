#!/usr/bin/env python3
"""In-situ timeline of the MFMA kernels of the eager THREE-STREAM step: the library's per-launch HIP events
(pfa_gather_timing) give every launch's start (relative to the step's first timed launch) and duration while the step runs
as it does in bench.py - no profiler, whose host cost distorts a multi-stream trace.  Prints launches in start order with
the number of MFMA launches running at that moment, and per phase the sum of durations against the wall time."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gan-aug-pfa_amd"))
import torch

import pfa_gan
from pfa_gan import _lib

torch.manual_seed(0)
gen = pfa_gan.UNetGenerator(3, 3).cuda()
disc = pfa_gan.NLayerDiscriminator(6).cuda()
tr = pfa_gan.GanTrainer(gen, disc, precision="bf16", inputs_resident=True)
g = torch.Generator(device="cuda").manual_seed(1234)
pool = []
for _ in range(4):
    a = torch.rand((16, 3, 256, 256), device="cuda", generator=g) * 2 - 1
    b = torch.rand((16, 3, 256, 256), device="cuda", generator=g) * 2 - 1
    pool.append((tr.pack(a), tr.pack(b)))
for i in range(6):
    tr.step_packed(*pool[i % 4])
torch.cuda.synchronize()
lib = _lib.lib()
STEPS = 3
lib.pfa_gather_timing(1)
for i in range(STEPS):
    tr.step_packed(*pool[i % 4])
torch.cuda.synchronize()
lib.pfa_gather_timing(0)
n = lib.pfa_gather_timing_count() // STEPS
ms, st, name, flop, byts, sk = ctypes.c_float(), ctypes.c_float(), ctypes.c_char_p(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
base = n          # the second timed step
rows = []
for i in range(base, base + n):
    _lib.check(lib.pfa_gather_timing_get(i, ctypes.byref(ms), ctypes.byref(name), ctypes.byref(flop), ctypes.byref(byts), ctypes.byref(sk)))
    _lib.check(lib.pfa_gather_timing_start(i, base, ctypes.byref(st)))
    rows.append((st.value * 1e3, ms.value * 1e3, name.value.decode(), flop.value, i - base))
t0 = min(r[0] for r in rows)
rows = sorted((r[0] - t0, r[1], r[2], r[3], r[4]) for r in rows)
print(f"{n} MFMA launches per step; issue index = program order")
busy = 0.0
for s, d, nm, fl, idx in rows:
    conc = sum(1 for s2, d2, _, _, _ in rows if s2 <= s < s2 + d2)
    print(f"{s:8.1f} +{d:6.1f}  x{conc}  #{idx:3d}  {nm:38s} {fl / max(d, 1e-3) / 1e6:6.0f} TF/s")
end = max(s + d for s, d, *_ in rows)
# time covered by at least one / at least two MFMA launches
edges = sorted([(s, 1) for s, d, *_ in rows] + [(s + d, -1) for s, d, *_ in rows])
cur, last, c1, c2 = 0, 0.0, 0.0, 0.0
for t, e in edges:
    if cur >= 1:
        c1 += t - last
    if cur >= 2:
        c2 += t - last
    cur += e
    last = t
print(f"span {end:.0f} us, sum of MFMA kernel durations {sum(r[1] for r in rows):.0f} us, >=1 running {c1:.0f} us, >=2 running {c2:.0f} us")
