"""Stress of the in-kernel K-slice reduction (developer tool): the same workspace (slab + tickets) is reused by launches
that alternate between two inputs, so a stale partial from the previous launch would show as a mismatch against the
slab + finish-pass result of the same input.  Prints the number of mismatching launches."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/gan-aug-pfa_amd')
import torch
from pfa_gan import kernels as K, _lib
lib = _lib.lib()
dt = torch.bfloat16
ws = K.Workspace(torch.device('cuda'))
torch.manual_seed(0)
def T(*shape): return torch.randn(shape, device='cuda').to(dt)
CASES = [  # mode, nb, h, w, c, n, grid, out hw, splitk
    (K.CONV_S1, 3, 16, 16, [256], 512, (15, 15), (15, 15), 16),
    (K.CONV_S1_DGRAD, 3, 15, 15, [512], 256, (16, 16), (16, 16), 16),
    (K.CONV_S2, 3, 16, 16, [128], 256, (8, 8), (8, 8), 4),
    (K.CONV_S2, 16, 32, 32, [256], 512, (16, 16), (16, 16), 8),
    (K.TCONV_S2, 16, 16, 16, [512, 512], 256, (16, 16), (32, 32), 4),
    (K.CONV_S1, 16, 32, 32, [256], 512, (31, 31), (31, 31), 2),
]
lib.pfa_set_knob(6, 1)
import os
INTERLEAVE = os.environ.get('INTERLEAVE') == '1'
bad_total = 0
for mode, nb, h, w, cs, n, (hq, wq), (ho, wo), sk in CASES:
    ins = [[T(nb, h, w, c) for c in cs] for _ in range(2)]
    wgt = T(n, 16, sum(cs))
    ref = []
    lib.pfa_set_knob(3, 0)
    for srcs in ins:
        o = torch.empty((nb, ho, wo, n), device='cuda', dtype=dt)
        K.gather_gemm(mode, srcs, wgt, n, [o], (nb, hq, wq), ws, splitk=sk)
        ref.append(o.clone())
    lib.pfa_set_knob(3, 1)
    bad = 0
    o = torch.empty((nb, ho, wo, n), device='cuda', dtype=dt)
    stat_ref = [None, None]
    o2 = torch.empty((nb, ho, wo, n), device='cuda', dtype=dt)
    for it in range(300):
        if INTERLEAVE:      # a slab + finish-pass launch (plain stores and loads on the same slab) between two in-kernel ones
            lib.pfa_set_knob(3, 0)
            K.gather_gemm(mode, ins[(it + 1) & 1], wgt, n, [o2], (nb, hq, wq), ws, splitk=sk)
            lib.pfa_set_knob(3, 1)
        psum, psq, nt = K.gather_gemm(mode, ins[it & 1], wgt, n, [o], (nb, hq, wq), ws, splitk=sk, stats=True)
        st = torch.cat([psum[: nt * n], psq[: nt * n]]).clone()
        if stat_ref[it & 1] is None:
            stat_ref[it & 1] = st
        if not torch.equal(o, ref[it & 1]) or not torch.equal(st, stat_ref[it & 1]):
            bad += 1
    name = lib.pfa_last_gather_kernel().decode()
    print(f"mode {mode} M={nb*hq*wq} n={n} splitk={sk} {name}: {bad} of 300 launches differ from the finish-pass result")
    bad_total += bad
lib.pfa_set_knob(6, 0); lib.pfa_set_knob(3, 0)
print("tickets nonzero:", int(ws.get("tickets", 1, torch.int32, zero=True).abs().max()))
sys.exit(1 if bad_total else 0)
