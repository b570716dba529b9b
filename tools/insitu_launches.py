"""Every implicit-GEMM launch of one training step, timed in situ by the library (pfa_gather_timing), one stream."""
import ctypes, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/gan-aug-pfa_amd')
import torch, pfa_gan
from pfa_gan import _lib
torch.manual_seed(0)
gen = pfa_gan.UNetGenerator(3, 3).cuda(); disc = pfa_gan.NLayerDiscriminator(6).cuda()
tr = pfa_gan.GanTrainer(gen, disc, precision="bf16", overlap=False)
a = torch.rand(16, 3, 256, 256, device='cuda') * 2 - 1; b = torch.rand(16, 3, 256, 256, device='cuda') * 2 - 1
xa, xb = tr.pack(a), tr.pack(b)
for _ in range(4): tr.step_packed(xa, xb)
torch.cuda.synchronize()
lib = _lib.lib()
import os
lib.pfa_set_knob(1, int(os.environ.get('PFA_DBG', '0')))      # ablation bits (results are wrong when set)
REPS = 5
lib.pfa_gather_timing(1)
for _ in range(REPS): tr.step_packed(xa, xb)
torch.cuda.synchronize()
lib.pfa_gather_timing(0)
n = lib.pfa_gather_timing_count() // REPS
ms, name, flop, byts, sk = ctypes.c_float(), ctypes.c_char_p(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
rows = []
for i in range(n):
    t = 0.0
    for r in range(REPS):
        lib.pfa_gather_timing_get(r * n + i, ctypes.byref(ms), ctypes.byref(name), ctypes.byref(flop), ctypes.byref(byts), ctypes.byref(sk))
        t += ms.value
    rows.append((i, name.value.decode().replace("k_gather_gemm", ""), sk.value, t / REPS * 1e3, flop.value, byts.value))
tot = sum(r[3] for r in rows)
print(f"{n} launches, {tot:.0f} us of implicit-GEMM kernel time per step")
for i, nm, s, us, fl, by in rows:
    print(f"{i:3d} {nm:24s} sk={s:2d} {us:7.1f} us {fl / us / 1e6:6.0f} TF/s  {fl / 1e9:6.1f} GF {by / 1e6:6.1f} MB {by / us / 1e3:6.0f} GB/s(unique)")
