"""Run the discriminator engine's train-mode forward repeatedly on one input and report which named buffers differ from the
first run (developer tool)."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/gan-aug-pfa_amd')
import torch
import pfa_gan
from pfa_gan import kernels as K
n, h, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
torch.manual_seed(0)
D = pfa_gan.NLayerDiscriminator(6).cuda().set_precision("bf16")
eng = D.engine()
eng.refresh_shadows(force=True)
x = [K.pack_nhwc(torch.rand(n, 3, h, h, device="cuda") * 2 - 1, eng.dtype) for _ in range(2)]
ctx = eng.new_ctx()
first, diff = None, {}
for it in range(reps):
    eng.forward(x, ctx, train=True)
    torch.cuda.synchronize()
    snap = {k: v.detach().float().clone() for k, v in ctx._t.items()}
    if first is None:
        first = snap
        continue
    for k, v in snap.items():
        if not torch.equal(v, first[k]):
            diff.setdefault(k, []).append((it, float((v - first[k]).abs().max())))
for k in first:
    if k in diff:
        print(k, len(diff[k]), "runs differ; first:", diff[k][:3])
print("done;", "no differences" if not diff else f"{len(diff)} buffers differ")
