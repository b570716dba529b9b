"""Repeat one fused step from the same state many times and report run-to-run deviations (developer tool): a stale or
racy hand-off shows as an outlier.  PFA_KNOBS selects the code path."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/gan-aug-pfa_amd')
import numpy as np, torch
import pfa_gan
n, h, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
prec = sys.argv[4] if len(sys.argv) > 4 else "bf16"
overlap = (sys.argv[5] != "0") if len(sys.argv) > 5 else True
torch.manual_seed(21)                                    # one seeded start state (the modules' default init) and one batch
G = pfa_gan.UNetGenerator(3, 3).cuda(); D = pfa_gan.NLayerDiscriminator(6).cuda()
gst = {k: v.detach().clone() for k, v in G.state_dict().items()}
dst = {k: v.detach().clone() for k, v in D.state_dict().items()}
gen = torch.Generator(device="cuda").manual_seed(22)
a = torch.rand((n, 3, h, h), device="cuda", generator=gen) * 2 - 1
b = torch.rand((n, 3, h, h), device="cuda", generator=gen) * 2 - 1
def load(m, st): m.load_state_dict({k: v.clone() for k, v in st.items()}, strict=True)
res = []
snaps = []
for it in range(reps):
    load(G, gst); load(D, dst)
    tr = pfa_gan.GanTrainer(G, D, precision=prec, overlap=overlap)
    losses = tr.step(a, b).cpu().numpy()
    fake = pfa_gan.kernels.unpack_nchw(tr.gctx.v["fake"], 3).float()
    w = G.state_dict()["model.model.1.model.3.model.3.model.1.weight"].float()
    res.append((losses.copy(), fake.clone(), w.clone()))
    snap = {}
    for an, av in vars(tr).items():
        if hasattr(av, '_t') and hasattr(av, 'v'):
            for bn_, bt in av._t.items():
                snap[an + ':' + bn_] = bt.detach().float().clone()
    for on, opt in (("opt_g", tr.opt_g), ("opt_d", tr.opt_d)):             # parameters, gradient arenas, Adam moments
        fl = opt.flat()
        snap[on + ":params"] = fl.data.detach().clone()
        snap[on + ":grads"] = fl.grad.detach().clone()
        snap[on + ":m"] = opt.m.detach().clone()
        snap[on + ":v"] = opt.v.detach().clone()
        if getattr(opt, "_grad2", None) is not None:
            snap[on + ":grads2"] = opt._grad2.detach().clone()
    for mn, mod in (("G", G), ("D", D)):                                     # BatchNorm running buffers
        for bn_, bt in mod.named_buffers():
            snap[mn + ":" + bn_] = bt.detach().float().clone()
    snaps.append(snap)
l0, f0, w0 = res[0]
for it, (l, f, w) in enumerate(res):
    dl = np.abs(l - l0).max(); df = float((f - f0).abs().max()); dw = float((w - w0).abs().max())
    flag = "  <-- outlier" if (dl > 1e-3 or df > 1e-2 or not np.isfinite(l).all()) else ""
    if flag or it < 3: print(it, l, "dloss %.2e dfake %.2e dw %.2e" % (dl, df, dw), flag)
print("max dloss", max(np.abs(l - l0).max() for l, _, _ in res), "max dfake", max(float((f - f0).abs().max()) for _, f, _ in res))

names = list(snaps[0].keys())
diff = {}
for it in range(1, len(snaps)):
    for k in names:
        if k in snaps[it] and snaps[it][k].shape == snaps[0][k].shape and not torch.equal(snaps[it][k], snaps[0][k]):
            diff[k] = diff.get(k, 0) + 1
print("buffers that differ from run 0 in at least one run (count of runs):")
for k in names:
    if k in diff: print("  ", k, diff[k], tuple(snaps[0][k].shape))
