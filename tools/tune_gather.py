"""Record every distinct pfa_gather_gemm launch of one fused step (B=16, 256x256, bf16) and time it under the tile
knob {0: automatic, 4: 256x128 for n >= 128} x split-K {1,2,4,8,16}.  Prints the best combination per launch and the
step totals (automatic choice vs per-launch optimum)."""
import sys, collections
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/gan-aug-pfa_amd')
import torch
import pfa_gan
from pfa_gan import kernels as K, _lib, engine as E
torch.manual_seed(0)
gen = pfa_gan.UNetGenerator(3, 3).cuda(); disc = pfa_gan.NLayerDiscriminator(6).cuda()
tr = pfa_gan.GanTrainer(gen, disc, precision="bf16", overlap=False)
a = torch.rand(16, 3, 256, 256, device='cuda') * 2 - 1; b = torch.rand(16, 3, 256, 256, device='cuda') * 2 - 1
xa, xb = tr.pack(a), tr.pack(b)
tr.step_packed(xa, xb)
real = K.gather_gemm
calls = collections.OrderedDict()
def rec(mode, srcs, w, n, outs, grid, ws, **kw):
    key = (mode, tuple(tuple(s.shape) for s in srcs), n, tuple(tuple(o.shape) for o in outs), grid, kw.get('stats', False), kw.get('act', 0), kw.get('bias') is not None)
    if key not in calls: calls[key] = [0, (mode, srcs, w, n, outs, grid, ws, dict(kw))]
    calls[key][0] += 1
    return real(mode, srcs, w, n, outs, grid, ws, **kw)
K.gather_gemm = rec; E.K.gather_gemm = rec
tr.step_packed(xa, xb)
K.gather_gemm = real; E.K.gather_gemm = real
torch.cuda.synchronize()
def timeit(f, reps=15):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
tot_auto = tot_best = 0
names = {K.CONV_S2: 'conv2', K.CONV_S1: 'conv1', K.TCONV_S2: 'tconv', K.CONV_S1_DGRAD: 'dg_s1'}
for key, (cnt, (mode, srcs, w, n, outs, grid, ws, kw)) in calls.items():
    kw.pop('splitk', None)
    M = grid[0] * grid[1] * grid[2]; ctot = sum(s.shape[3] for s in srcs)
    _lib.lib().pfa_set_knob(0, 0)
    t_auto = timeit(lambda: real(mode, srcs, w, n, outs, grid, ws, **kw))
    res = {}
    for cfg in ((0, 4) if n > 64 else (0,)):
        _lib.lib().pfa_set_knob(0, cfg)
        for sk in (1, 2, 4, 8, 16):
            if sk > 1 and (16 if mode != K.TCONV_S2 else 4) * (ctot // 8) // 8 < 2 * sk: continue
            res[(cfg, sk)] = timeit(lambda: real(mode, srcs, w, n, outs, grid, ws, splitk=sk, **kw))
    _lib.lib().pfa_set_knob(0, 0)
    best = min(res, key=res.get)
    tot_auto += cnt * t_auto; tot_best += cnt * res[best]
    print(f"x{cnt} {names[mode]:5s} M={M:7d} n={n:5d} C={ctot:5d} auto={t_auto:7.1f} best={best} {res[best]:7.1f}  " +
          " ".join(f"{k}:{v:.0f}" for k, v in sorted(res.items())))
print("per step: auto", tot_auto, "best", tot_best)
