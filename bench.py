#!/usr/bin/env python3
"""Headline benchmark: GAN img-pairs/s of the complete Pix2Pix G+D training step (train_gan.py:52-71:
G fwd, 3x D fwd, both backwards, both Adam updates) at 256x256x3, batch 16 per GPU, bf16 operands.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

One JSON line on rank 0.  `value` = N * batch * K / (max over ranks of the timed K steps), inputs resident in
HBM (a pool of pre-packed synthetic batches, seed 1234 + rank).  Extra objects:
  roofline      dominant kernel = the bf16 128x128-tile implicit-GEMM (k_gather_gemm<bf16,256,128,4,2,3>: 256x128 tile, 8 waves, 3-stage LDS ring): HIP-event
                time of every launch of it over a pass of the same steps, against its algorithmic conv FLOPs
                (2 * pixels * Cout * taps * Cin_real per launch; SURVEY 8d counts the same MACs)
  cpu_baseline  the numpy oracle's gan_step (oracle/step.py, "port") on the host cores, bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "gan-aug-pfa_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0        # dense MFMA bf16, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_F32_TFLOPS = 157.3
FLOP_PER_PAIR = {256: 86.79e9, 512: 351.38e9}      # minimal schedule, conv MACs x2


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="img-pairs per GPU (BASELINE config 2: 16)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--auto-graph", action="store_true", help="GanTrainer(use_graph='auto'): hipGraph replay for tiny batches")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as three captured hipGraphs (default: eager launches on two HIP streams, which "
                         "measured faster: the graph executor does not run the forked side-stream branch concurrently)")
    ap.add_argument("--no-graph", action="store_true", help=argparse.SUPPRESS)      # former default spelling, accepted
    ap.add_argument("--no-overlap", action="store_true", help="single stream: no wgrad / D(real) overlap")
    ap.add_argument("--no-early-real", action="store_true",
                    help="D(real) of a step waits for the whole previous step (A/B switch for inputs_resident)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--mode", default="train", choices=["train", "infer"],
                    help="infer = BASELINE config 5: generator-only eval forward (generate_synthetic_data.py:55-68), tiles/s")
    return ap.parse_args()


def kernel_roofline(trainer, xa, xb, steps, precision):
    """Live measurement of the dominant kernel: libpfa_hip brackets every implicit-GEMM kernel launch with HIP events on
    its launch stream (pfa_gather_timing; a split-K call's finish kernel stays outside) over `steps` eager one-stream
    steps of the same workload; the instantiation with the largest total time is the dominant kernel and ALL its launches
    count - the same population `rocprofv3 --kernel-trace --stats` averages for that kernel name
    (profiles/r01_bench_b16_256_bf16_onestream_kernel_stats.csv is the same mode)."""
    import ctypes
    from pfa_gan import _lib
    lib = _lib.lib()
    was_graph, was_side = trainer.use_graph, trainer.side
    trainer.use_graph, trainer.side = False, None      # one stream: nothing else shares the GPU with the timed launches
    lib.pfa_gather_timing(1)
    try:
        for _ in range(steps):
            trainer.step_packed(xa, xb)
        torch.cuda.synchronize()
    finally:
        lib.pfa_gather_timing(0)
        trainer.use_graph, trainer.side = was_graph, was_side
    per = {}
    ms, name, flop, byts, sk = ctypes.c_float(), ctypes.c_char_p(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
    for i in range(lib.pfa_gather_timing_count()):
        _lib.check(lib.pfa_gather_timing_get(i, ctypes.byref(ms), ctypes.byref(name), ctypes.byref(flop), ctypes.byref(byts),
                                             ctypes.byref(sk)), "pfa_gather_timing_get")
        r = per.setdefault(name.value.decode(), [0.0, 0.0, 0.0, 0, 0])
        r[0] += ms.value; r[1] += flop.value; r[2] += byts.value; r[3] += 1; r[4] += sk.value > 1
    kernel, (tms, tflop, tbytes, n, nsplit) = max(per.items(), key=lambda kv: kv[1][0])
    peak = PEAK_BF16_TFLOPS if precision == "bf16" else PEAK_F32_TFLOPS
    achieved = tflop / (tms * 1e-3) / 1e12
    traffic, traffic_src = None, None
    tfile = os.path.join(ROOT, "profiles", "r01_traffic_gather_ring.json")
    if precision == "bf16" and xa.shape[0] == 16 and xa.shape[1] == 256 and os.path.exists(tfile):
        # HBM bytes per launch from the committed rocprofv3 PMC passes of this same workload (tools/pmc_traffic.py);
        # counters cannot be read from inside the process being measured
        t = json.load(open(tfile))
        if t.get("kernel") == kernel:
            traffic, traffic_src = round(t["traffic_bytes_per_launch"]), "profiles/r01_traffic_gather_ring.json"
    total_ms = sum(v[0] for v in per.values())
    # every MFMA convolution kernel of the step, largest time first (the fields above `kernels` describe the first one)
    others = [dict(kernel=k, launches_per_step=v[3] // steps, avg_launch_us=round(1e3 * v[0] / v[3], 2),
                   achieved=round(v[1] / (v[0] * 1e-3) / 1e12, 2), frac=round(v[1] / (v[0] * 1e-3) / 1e12 / peak, 4),
                   share_of_implicit_gemm_time=round(v[0] / total_ms, 3))
              for k, v in sorted(per.items(), key=lambda kv: -kv[1][0])]
    return dict(bound="mfma", achieved=round(achieved, 2), peak=peak, unit="TFLOP/s", frac=round(achieved / peak, 4),
                traffic=traffic, traffic_source=traffic_src, algorithmic_bytes_per_launch=round(tbytes / n),
                kernel=kernel, launches_per_step=n // steps, split_k_launches_per_step=nsplit // steps,
                avg_launch_us=round(1e3 * tms / n, 2), flop_per_launch=round(tflop / n),
                flop_share_of_step=round(tflop / steps / (FLOP_PER_PAIR.get(xa.shape[1], 0) * xa.shape[0] or 1), 3),
                share_of_implicit_gemm_time=round(tms / total_ms, 3), kernels=others)


def cpu_baseline(size, steps):
    """The CPU restatement of the same step (oracle/step.py) on this host's cores: 1 warm-up + `steps` timed."""
    import numpy as np
    from oracle import fill, nets, step as ostep
    b = 1
    gst = fill.fill_state(nets.g_state_shapes(), seed=0, bn_noise=False)
    dst = fill.fill_state(nets.d_state_shapes(), seed=1, bn_noise=False)
    go, do = ostep.new_adam_state(gst), ostep.new_adam_state(dst)
    a = fill.uniform("cpu.a", (b, 3, size, size))
    bb = fill.uniform("cpu.b", (b, 3, size, size))
    ostep.gan_step(gst, dst, go, do, a, bb)
    t0 = time.perf_counter()
    for _ in range(steps):
        ostep.gan_step(gst, dst, go, do, a, bb)
    dt = time.perf_counter() - t0
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    return dict(value=round(b * steps / dt, 4), unit="img-pairs/s", cores=cores, kind="port",
                sample=f"oracle/step.py gan_step (numpy {np.__version__}, BLAS threads = cores), default nets, "
                       f"{size}x{size}, batch {b}, {steps} steps after 1 warm-up, fp32")


def infer_main(args):
    """BASELINE config 5: eval-mode generator forward over synthetic 256x256 tiles, hipGraph-captured, one GPU."""
    import pfa_gan
    from pfa_gan import kernels as K
    torch.manual_seed(0)
    gen = pfa_gan.UNetGenerator(3, 3).cuda().set_precision(args.precision).eval()
    eng = gen.engine()
    eng.refresh_shadows(force=True)
    ctx = eng.new_ctx()
    x = torch.rand((args.batch, 3, args.size, args.size), device="cuda") * 2 - 1
    xh = K.pack_nhwc(x, eng.dtype)
    out_u8 = None
    for _ in range(3):
        eng.forward(xh, ctx, train=False)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        fake = eng.forward(xh, ctx, train=False)
    for _ in range(args.warmup):
        graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        graph.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tiles = args.batch * args.steps / dt
    peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else PEAK_F32_TFLOPS
    print(json.dumps({
        "metric": "generator-only eval forward, 256x256 tiles/s", "value": round(tiles, 1), "unit": "tiles/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "config": {"workload": f"UNetGenerator eval forward, {args.size}x{args.size}, batch {args.batch}, hipGraph replay"},
        "step_mfma": {"algorithmic_tflops_per_gpu": round(tiles * 12.046e9 / 1e12, 2),
                      "frac_of_peak": round(tiles * 12.046e9 / 1e12 / peak, 4)}}), flush=True)


def main():
    args = parse()
    if args.mode == "infer":
        torch.cuda.set_device(0)
        return infer_main(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus} "
                             f"bench.py ...` (one process per GPU)")
        args.gpus = world
    # Rehearsal of the N-rank control flow on a one-GPU box (tests/test_gpu_networks.py): PFA_BENCH_BACKEND=gloo with
    # PFA_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and reduces through gloo.  The measured configuration is RCCL.
    backend = os.environ.get("PFA_BENCH_BACKEND", "nccl")
    if os.environ.get("PFA_BENCH_SHARE_GPU") == "1":
        local = 0
    torch.cuda.set_device(local)
    pg = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))   # nccl == RCCL on ROCm
        else:
            torch.distributed.init_process_group(backend)
        pg = torch.distributed.group.WORLD

    import pfa_gan
    from pfa_gan import _lib
    _lib.lib()                                       # fail loudly if the HIP extension is missing
    torch.manual_seed(0)                             # identical default init on every rank (reference init, a12)
    gen = pfa_gan.UNetGenerator(3, 3).cuda()
    disc = pfa_gan.NLayerDiscriminator(6).cuda()
    # inputs_resident: the batches below are packed and complete in HBM before the first step (the bench contract)
    trainer = pfa_gan.GanTrainer(gen, disc, precision=args.precision, process_group=pg, use_graph="auto" if args.auto_graph else args.graph,
                                 overlap=not args.no_overlap, inputs_resident=not args.no_early_real)
    if world > 1:                                    # rank 0's parameters and buffers are the job's
        for t in [trainer.opt_g.flat().data, trainer.opt_d.flat().data] + list(gen.buffers()) + list(disc.buffers()):
            torch.distributed.broadcast(t, 0)
        trainer.G.refresh_shadows(force=True)
        trainer.Dn.refresh_shadows(force=True)

    gsrc = torch.Generator(device="cuda").manual_seed(1234 + rank)
    pool = []
    for _ in range(8):                               # distinct batches, already packed: resident in HBM
        a = torch.rand((args.batch, 3, args.size, args.size), device="cuda", generator=gsrc) * 2 - 1
        b = torch.rand((args.batch, 3, args.size, args.size), device="cuda", generator=gsrc) * 2 - 1
        pool.append((trainer.pack(a), trainer.pack(b)))
    torch.cuda.synchronize()                         # resident: every batch is complete before the first step

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    launches0 = _lib.launch_count()
    for i in range(args.warmup):
        trainer.step_packed(*pool[i % len(pool)])
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        trainer.step_packed(*pool[i % len(pool)])
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device="cuda", dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax)
    losses = trainer.losses.tolist()
    assert all(l == l and abs(l) < 1e6 for l in losses), f"non-finite loss {losses}"
    assert _lib.launch_count() > launches0

    # every rank takes part in the roofline pass: its steps carry the same gradient all-reduces as the timed ones
    roof = None if args.no_roofline else kernel_roofline(trainer, *pool[0], max(2, min(args.steps, 5)), args.precision)
    if rank == 0:
        pairs = world * args.batch * args.steps
        value = pairs / dt
        flop_pair = FLOP_PER_PAIR.get(args.size)
        peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else PEAK_F32_TFLOPS
        out = {
            "metric": "GAN img-pairs/sec (G+D step, 256x256x3)" if args.size == 256 else f"GAN img-pairs/sec (G+D step, {args.size}x{args.size}x3)",
            "value": round(value, 2), "unit": "img-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"{args.size}x{args.size} RGB Pix2Pix G+D step (UNetGenerator num_downs=7 ngf=64 + "
                                   f"NLayerDiscriminator ndf=64, BatchNorm, BCE+100*L1, 2x Adam), batch {args.batch}/GPU",
                       "global_batch": world * args.batch, "parallelism": f"dp{world}",
                       "launch": "hipGraph x3 segments" if trainer.use_graph is True else
                                 ("eager, 3 HIP streams" if not args.no_overlap else "eager, 1 HIP stream"),
                       "last_losses": [round(l, 4) for l in losses]},
        }
        if flop_pair:
            tf = value * flop_pair / 1e12 / world
            out["step_mfma"] = {"algorithmic_tflops_per_gpu": round(tf, 2), "frac_of_peak": round(tf / peak, 4)}
        if roof is not None:
            out["roofline"] = roof
        if world == 1 and not args.no_cpu_baseline and args.size == 256:
            out["cpu_baseline"] = cpu_baseline(args.size, args.cpu_steps)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
