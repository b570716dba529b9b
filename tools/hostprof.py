"""cProfile of the Python side of the fused step (enqueue only; B=16, 256x256, bf16)."""
import cProfile, pstats, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/gan-aug-pfa_amd')
import torch, pfa_gan
torch.manual_seed(0)
gen = pfa_gan.UNetGenerator(3, 3).cuda(); disc = pfa_gan.NLayerDiscriminator(6).cuda()
tr = pfa_gan.GanTrainer(gen, disc, precision="bf16", inputs_resident=True)
a = torch.rand(16, 3, 256, 256, device='cuda') * 2 - 1; b = torch.rand(16, 3, 256, 256, device='cuda') * 2 - 1
xa, xb = tr.pack(a), tr.pack(b)
for _ in range(5): tr.step_packed(xa, xb)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20): tr.step_packed(xa, xb)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(22)
