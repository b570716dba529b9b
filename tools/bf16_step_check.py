#!/usr/bin/env python3
"""Per-layer relative L2 distance of the bf16 trainer's raw gradients from the bf16-storage step oracle (oracle/bf16.py) and
from the float32 oracle, for the cases of tests/test_gpu_networks.py::test_bf16_step_matches_the_bf16_storage_step_oracle."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gan-aug-pfa_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import pfa_gan
from oracle import bf16 as o16, fill, nets, step as ostep


def load(module, st):
    module.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in st.items()}, strict=True)


def rel(a, b):
    return np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-20)


for name, nd, ngf, ndf, n, h in (("small", 5, 8, 8, 2, 32), ("wide", 5, 16, 16, 2, 64), ("wide4", 5, 16, 16, 4, 64)):
    gst = fill.fill_state(nets.g_state_shapes(3, 3, nd, ngf), seed=5, bn_noise=False)
    dst = fill.fill_state(nets.d_state_shapes(6, ndf, 3), seed=6, bn_noise=False)
    a = fill.uniform(f"bf16step.{name}.a", (n, 3, h, h))
    b = fill.uniform(f"bf16step.{name}.b", (n, 3, h, h))
    G = pfa_gan.UNetGenerator(3, 3, num_downs=nd, ngf=ngf).cuda()
    D = pfa_gan.NLayerDiscriminator(6, ndf=ndf).cuda()
    load(G, gst)
    load(D, dst)
    unfused = os.environ.get("PFA_FUSE_BWD", "1") == "0"       # the engines then take the un-fused route everywhere
    tr = pfa_gan.GanTrainer(G, D, precision="bf16", overlap=os.environ.get("OVERLAP", "1") != "0")
    losses = tr.step(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()).cpu().numpy()
    torch.cuda.synchronize()
    res = {}
    for tag, mod in (("bf16", o16), ("fp32", ostep)):
        og, od = {k: v.copy() for k, v in gst.items()}, {k: v.copy() for k, v in dst.items()}
        kw = dict(unfused=unfused) if mod is o16 else {}
        res[tag] = mod.gan_step(og, od, ostep.new_adam_state(og), ostep.new_adam_state(od), a, b, nd, 3, **kw)
    print(f"== {name}: losses gpu {losses}  bf16 oracle {res['bf16'][:2]}  fp32 oracle {res['fp32'][:2]}")
    ex = res["bf16"][2]
    for k, p in G.named_parameters():
        if p.dim() == 4:
            print(f"  G     {k:55s} vs bf16 {rel(p.grad.cpu().numpy(), ex['g_grads'][k]):.2e}   vs fp32 {rel(p.grad.cpu().numpy(), res['fp32'][2]['g_grads'][k]):.2e}")
    flat_d, g2 = tr.opt_d.flat(), tr.opt_d.grad2()
    for k, p in D.named_parameters():
        if p.dim() == 4:
            o, cnt = flat_d.slots[id(p)]
            real = flat_d._view(g2, p, o, cnt).cpu().numpy()
            fake = p.grad.cpu().numpy()
            print(f"  D     {k:20s} real vs bf16 {rel(real, ex['d_grads_real'][k]):.2e}  fake vs bf16 {rel(fake, ex['d_grads_fake'][k]):.2e}  "
                  f"sum vs bf16 {rel(real + fake, ex['d_grads'][k]):.2e}  sum vs fp32 {rel(real + fake, res['fp32'][2]['d_grads'][k]):.2e}")
