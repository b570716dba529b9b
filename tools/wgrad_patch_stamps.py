"""Where a patch weight-gradient launch spends its time: shader-clock stamps per workgroup (pfa_set_debug_buffer):
prologue, K loop (cycles per 64-position step), slab epilogue; workgroup life against the launch's span (100 MHz counter)."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/gan-aug-pfa_amd')
import torch
from pfa_gan import kernels as K, _lib
B = 16; dt = torch.bfloat16; ws = K.Workspace(torch.device('cuda'))
def T(*shape): return torch.randn(shape, device='cuda').to(dt)
L = [("c4", 1, (31, 31, [512]), (32, 32, 256)), ("e3/c3", 2, (32, 32, [256]), (64, 64, 128)), ("e2/c2", 2, (64, 64, [128]), (128, 128, 64)),
     ("e4", 2, (16, 16, [512]), (32, 32, 256)), ("d3", 2, (32, 32, [256, 256]), (64, 64, 128)), ("d2", 2, (64, 64, [128, 128]), (128, 128, 64))]
lib = _lib.lib()
buf = torch.zeros(8192 * 8, dtype=torch.int64, device='cuda')
import os
nky = int(os.environ.get("WG_NKY", "1"))                 # 2: two tap rows per workgroup (knob 13)
lib.pfa_set_knob(13, nky)
knobs = [0] + ([11, 12, 14, 13, 15, 16, 17] if os.environ.get("PFA_ABLATE") else [])
names = {0: "full", 11: "no MFMA", 12: "no reads", 14: "no DMA", 13: "no MFMA, no reads", 15: "no MFMA, no DMA", 16: "no reads, no DMA",
         17: "barriers + waits + addresses"}
for knob in knobs:
  lib.pfa_set_knob(7, knob)
  print("----", names[knob])
  for name, stride, (hq, wq, dparts), (hg, wg, gc) in (L if knob == 0 else [L[1], L[4]]):
      d = [T(B, hq, wq, c) for c in dparts]; g = [T(B, hg, wg, gc)]
      dw = torch.zeros(sum(dparts) * 16 * gc, device='cuda')
      for _ in range(2):
          K.wgrad(d, g, dw, stride, dparts, [gc], ws=ws)
      torch.cuda.synchronize()
      buf.zero_()
      lib.pfa_set_debug_buffer(buf.data_ptr(), buf.numel() * 8)
      K.wgrad(d, g, dw, stride, dparts, [gc], ws=ws)
      torch.cuda.synchronize()
      lib.pfa_set_debug_buffer(None, 0)
      st = buf.view(-1, 8).cpu()
      st = st[st[:, 3] != 0].double()
      if st.shape[0] == 0:
          print(f"{name:8s} (no stamps)"); continue
      pro, loop, epi = (st[:, 1] - st[:, 0]).median(), (st[:, 2] - st[:, 1]).median(), (st[:, 3] - st[:, 2]).median()
      ty = 2 if wq >= 24 else 4
      steps_total = B * ((hq + ty - 1) // ty) * ((wq + 64 // ty - 1) // (64 // ty))
      nwg = st.shape[0]
      tiles = (sum(dparts) // 128) * (4 // nky) * (gc // 64)
      ms = nwg // tiles
      sper = (steps_total + ms - 1) // ms
      life = float((st[:, 5] - st[:, 4]).median()) / 100.0
      span = float(st[:, 5].max() - st[:, 4].min()) / 100.0
      print(f"{name:8s} {nwg:4d} workgroups ({ms} slices x {sper} steps): prologue {pro:7.0f}  loop {loop:8.0f} = {loop / sper:6.0f} cycles per step  "
            f"epilogue {epi:7.0f} | life {life:6.1f} us, first start -> last end {span:6.1f} us")

lib.pfa_set_knob(7, 0)
lib.pfa_set_knob(13, 0)
