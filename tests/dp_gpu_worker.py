"""Two-rank data-parallel worker for tests/test_gpu_networks.py (launched with torch.distributed.run).

Both ranks sit on cuda:0 and reduce through gloo: a one-GPU box has no second device for RCCL, but the trainer's
control flow is the same - rank-0 broadcast, D-gradient all-reduce with the second arena folded in, bucketed
asynchronous all-reduce of the generator arena issued during the backward - and every tensor involved is a device
tensor produced by the HIP engine.  Checks (each rank):
  * parameters and Adam moments are bit-identical on both ranks after every step (same reduced gradient everywhere);
  * the update equals the one a single process computes from the MEAN of the two ranks' local gradients
    (train_gan.py:55-75 per rank, gradients averaged), recomputed here without a process group.
PFA_DP_PAYLOAD=bf16: the gradient ranges travel as bfloat16 (GanTrainer(grad_payload="bf16")); the checks are the same - Adam's
first steps move a parameter by ~lr * sign(g), which bf16 rounding of g does not change beyond the stated tolerance.
PFA_DP_RESIDENT=1: three pre-packed steps enqueued back to back with GanTrainer(inputs_resident=True) - the next step's
D(real) pass starts under the current step's tail - compared after the last one.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gan-aug-pfa_amd"))

import pfa_gan  # noqa: E402
from oracle import fill, nets  # noqa: E402


def build(seed_g, seed_d, nd, ngf, ndf):
    G = pfa_gan.UNetGenerator(3, 3, num_downs=nd, ngf=ngf).cuda()
    D = pfa_gan.NLayerDiscriminator(6, ndf=ndf).cuda()
    G.load_state_dict({k: torch.from_numpy(v) for k, v in fill.fill_state(nets.g_state_shapes(3, 3, nd, ngf), seed=seed_g, bn_noise=False).items()})
    D.load_state_dict({k: torch.from_numpy(v) for k, v in fill.fill_state(nets.d_state_shapes(6, ndf, 3), seed=seed_d, bn_noise=False).items()})
    return G, D


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    nd, ngf, ndf, n, h = 5, 8, 8, 2, 32
    overlap = os.environ.get("PFA_DP_OVERLAP", "1") == "1"
    resident = os.environ.get("PFA_DP_RESIDENT", "0") == "1"      # pre-packed batches + GanTrainer(inputs_resident=True)
    # rank r starts from DIFFERENT weights; the broadcast must make rank 0's the job's
    G, D = build(21 + rank, 31 + rank, nd, ngf, ndf)
    tr = pfa_gan.GanTrainer(G, D, precision="fp32", process_group=dist.group.WORLD, overlap=overlap,
                            inputs_resident=resident, grad_payload=os.environ.get("PFA_DP_PAYLOAD") or None)
    pfa_gan.broadcast_model(G, dist.group.WORLD)
    pfa_gan.broadcast_model(D, dist.group.WORLD)
    tr.G.refresh_shadows(force=True)
    tr.Dn.refresh_shadows(force=True)

    # the single-process restatement: one trainer per rank's data, no process group, gradients averaged by hand
    refs = []
    for r in range(world):
        Gr, Dr = build(21, 31, nd, ngf, ndf)
        refs.append(pfa_gan.GanTrainer(Gr, Dr, precision="fp32", overlap=False))

    def data(step, r):
        a = torch.from_numpy(fill.uniform(f"dp.a{step}.{r}", (n, 3, h, h))).cuda()
        b = torch.from_numpy(fill.uniform(f"dp.b{step}.{r}", (n, 3, h, h))).cuda()
        return a, b

    nsteps = 3 if resident else 2
    packed_mine = [tuple(tr.pack(t) for t in data(st, rank)) for st in range(nsteps)]
    torch.cuda.synchronize()
    if resident:                      # enqueue every step before any check: consecutive steps overlap on the device
        for st in range(nsteps):
            tr.step_packed(*packed_mine[st])
    for step in range(nsteps):
        if not resident:
            tr.step(*data(step, rank))
        # --- reference: D phase on each rank's data, mean gradient, one Adam step; then the G phase likewise
        packed = [(t.pack(a), t.pack(b)) for t, (a, b) in zip(refs, [data(step, r) for r in range(world)])]
        for t, (xa, xb) in zip(refs, packed):
            t._d_phase(xa, xb, False)
        gd = sum(t.opt_d.flat().grad + t.opt_d.grad2() for t in refs) / world       # (an arena per D pass: real | fake)
        for t in refs:
            t.opt_d.flat().grad.copy_(gd)
            t.opt_d.grad2().zero_()
        for t, (xa, xb) in zip(refs, packed):
            t._d_update_g_phase(xa, xb, False)
        gg = sum(t.opt_g.flat().grad for t in refs) / world
        for t in refs:
            t.opt_g.flat().grad.copy_(gg)
            t._g_update(False)
        torch.cuda.synchronize()
        if resident and step < nsteps - 1:
            continue                  # every step is already enqueued: only the final state can be compared
        for name, mine, ref in (("G", tr.opt_g, refs[rank].opt_g), ("D", tr.opt_d, refs[rank].opt_d)):
            p, q = mine.flat().data, ref.flat().data
            # Adam's first steps move every parameter by ~lr * sign(g): an element whose gradient is at round-off level
            # may flip, so the criterion is the one of _check_step_state (a few step sizes at most, almost all exact)
            err = (p - q).abs()
            assert (err > 2e-5).float().mean().item() < 0.02 and err.max().item() < 5e-4, \
                f"rank {rank} step {step}: {name} parameters differ from the mean-gradient update (max {err.max().item()})"
            # identical on every rank
            gathered = [torch.empty_like(p) for _ in range(world)]
            dist.all_gather(gathered, p)
            assert all(torch.equal(gathered[0], g) for g in gathered), f"{name} parameters diverged across ranks at step {step}"
            m = [torch.empty_like(mine.m) for _ in range(world)]
            dist.all_gather(m, mine.m)
            assert all(torch.equal(m[0], g) for g in m), f"{name} Adam moments diverged across ranks at step {step}"
    losses = tr.losses.cpu().numpy()
    assert np.isfinite(losses).all()
    dist.barrier()
    if rank == 0:
        print("DP_GPU_WORKER_OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
