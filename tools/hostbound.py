"""Is the step bound by host launch overhead?  Enqueue time vs completion time for K steps: eager launches from Python
(three streams, one stream) and the same launches re-issued from C as launch lists (GanTrainer(use_graph="replay"))."""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/gan-aug-pfa_amd')
import torch, pfa_gan
from pfa_gan import _lib
torch.manual_seed(0)
gen = pfa_gan.UNetGenerator(3, 3).cuda(); disc = pfa_gan.NLayerDiscriminator(6).cuda()
for overlap, mode in ((True, False), (True, "replay"), (False, False), (False, "replay")):
    tr = pfa_gan.GanTrainer(gen, disc, precision="bf16", overlap=overlap, use_graph=mode)
    a = torch.rand(16, 3, 256, 256, device='cuda') * 2 - 1; b = torch.rand(16, 3, 256, 256, device='cuda') * 2 - 1
    xa, xb = tr.pack(a), tr.pack(b)
    for _ in range(5): tr.step_packed(xa, xb)
    torch.cuda.synchronize()
    n0 = _lib.launch_count()
    t0 = time.perf_counter()
    for _ in range(20): tr.step_packed(xa, xb)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"overlap={overlap} launch={'python' if not mode else 'pfa_replay'}: enqueue {1e3*(t1-t0)/20:.2f} ms/step, "
          f"complete {1e3*(t2-t0)/20:.2f} ms/step, pfa launches/step {(_lib.launch_count()-n0)/20:.0f}")
