#!/usr/bin/env python3
"""Headline benchmark: GAN img-pairs/s of the complete Pix2Pix G+D training step (train_gan.py:52-71:
G fwd, 3x D fwd, both backwards, both Adam updates) at 256x256x3, batch 16 per GPU, bf16 operands.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a torch.distributed environment: this process starts `python -m torch.distributed.run` with N ranks as a
CHILD process (before anything here touches a GPU), relays rank 0's JSON line and exits with the child's code.

One JSON line on rank 0.  `value` = N * batch * K / (max over ranks of the K timed steps), inputs resident in HBM (a pool
of pre-packed synthetic batches, seed 1234 + rank).  The K-step measurement is repeated `--repeats` times (default 5,
each bracketed by barrier + device synchronise) and the MEDIAN repeat is reported; all repeats are listed.
Extra objects:
  roofline      the MFMA kernel with the largest total time: HIP-event time of every launch of it (pfa_gather_timing,
                recorded by the library on the launch stream) over a pass of the same steps on one stream, against its
                algorithmic conv FLOPs (2 x MACs on real channels and taps: SURVEY 8d's count); `kernels` lists every
                MFMA kernel family of the step, weight gradients included
  cpu_baseline  the build's PyTorch-CPU restatement of the same step (pfa_gan.train_gan_one_epoch on CPU modules with
                torch.optim.Adam, pinned to the reference by tests/test_host_cpu.py), B=1 and B=4, on the host cores
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "gan-aug-pfa_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

PEAK_BF16_TFLOPS = 2500.0        # dense MFMA bf16, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
FLOP_PER_PAIR = {256: 86.79e9, 512: 351.38e9}      # minimal schedule, conv MACs x2
FLOP_PER_TILE = {256: 12.046e9, 512: 48.18e9}      # generator-only forward
# SURVEY.md 8(d) "algorithmic bytes per img-pair" (ideal fusion: every conv pass touches input, output and weights once, bf16
# activations, fp32 weight gradients, Adam 28 B/parameter), at the batch per GPU it is quoted for
BYTES_PER_PAIR = {(256, 16): 237.5e6, (512, 8): 747.3e6}
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r04_traffic.json")   # rocprofv3 --pmc passes of this workload (tools/pmc_traffic.py)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=5, help="timed repeats of --steps steps; the median is reported")
    ap.add_argument("--batch", type=int, default=16, help="img-pairs per GPU (BASELINE config 2: 16)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--auto-graph", action="store_true", help="GanTrainer(use_graph='auto'): hipGraph replay for tiny batches")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as three captured hipGraphs (default: eager launches on three HIP streams, which "
                         "measured faster: the graph executor does not run the forked side-stream branch concurrently)")
    ap.add_argument("--replay", action="store_true",
                    help="GanTrainer(use_graph='replay'): the step's launches recorded once as launch lists and re-issued "
                         "from C (pfa_replay) on the same three streams")
    ap.add_argument("--collectives", choices=["torch", "c"], default=None,
                    help="N > 1: transport of the gradient all-reduces - torch.distributed (default) or the library's RCCL "
                         "communicator (pfa_comm_* / pfa_allreduce; GanTrainer(collectives='c'))")
    ap.add_argument("--grad-bf16", action="store_true",
                    help="N > 1: the gradient all-reduces carry bfloat16 (GanTrainer(grad_payload='bf16')); default float32")
    ap.add_argument("--no-graph", action="store_true", help=argparse.SUPPRESS)      # former default spelling, accepted
    ap.add_argument("--no-overlap", action="store_true", help="single stream: no wgrad / D(real) overlap")
    ap.add_argument("--plan-shared", action="store_true",
                    help="with --no-overlap: keep the launch plans of the three-stream step (profiles of the timed configuration's kernels)")
    ap.add_argument("--no-early-real", action="store_true",
                    help="D(real) of a step waits for the whole previous step (A/B switch for inputs_resident)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra legs of the training line: fp32 parity mode and generator-only inference (config 5)")
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="time budget of the cpu_baseline leg")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)   # the child process of cpu_baseline_isolated
    ap.add_argument("--mode", default="train", choices=["train", "infer"],
                    help="infer = BASELINE config 5: generator-only eval forward (generate_synthetic_data.py:55-68), tiles/s")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: start the N ranks as a child process.  Nothing in
    this process has touched a GPU (torch is not even imported yet); the child's rank 0 prints the JSON line."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    for l in proc.stdout.splitlines():
        if l not in lines:
            print(l, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    raise SystemExit(proc.returncode if proc.returncode else (0 if lines else 1))


# ----------------------------------------------------------------------------------------------------- roofline
def _timed_kernels(lib, _lib):
    """{kernel name: [ms, flop, bytes, launches, sliced launches]} from the library's timing store."""
    import ctypes
    per = {}
    ms, name, flop, byts, sk = ctypes.c_float(), ctypes.c_char_p(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
    nwg = ctypes.c_int()
    for i in range(lib.pfa_gather_timing_count()):
        _lib.check(lib.pfa_gather_timing_get(i, ctypes.byref(ms), ctypes.byref(name), ctypes.byref(flop), ctypes.byref(byts),
                                             ctypes.byref(sk)), "pfa_gather_timing_get")
        _lib.check(lib.pfa_gather_timing_grid(i, ctypes.byref(nwg)), "pfa_gather_timing_grid")
        r = per.setdefault(name.value.decode(), [0.0, 0.0, 0.0, 0, 0])
        r[0] += ms.value; r[1] += flop.value; r[2] += byts.value; r[3] += 1; r[4] += sk.value > 1
        # the launches of one kernel by operand shape (forward and input-gradient launches of a layer share the kernel)
        sh = _SHAPES.setdefault(name.value.decode(), {}).setdefault((round(flop.value), round(byts.value), sk.value, nwg.value), [0.0, 0])
        sh[0] += ms.value; sh[1] += 1
    return per


_SHAPES = {}


def _roofline_object(per, steps, peak, unit_flop_total, workload_key):
    """The bench line's `roofline`: the kernel with the largest total time first, every MFMA kernel family in `kernels`."""
    kernel, (tms, tflop, tbytes, n, nsplit) = max(per.items(), key=lambda kv: kv[1][0])
    achieved = tflop / (tms * 1e-3) / 1e12
    traffic = traffic_src = None
    if os.path.exists(TRAFFIC_FILE):
        # HBM bytes per launch from committed rocprofv3 PMC passes of this same workload: counters cannot be read from
        # inside the process being measured.  The file names the commit and date it was taken at.
        t = json.load(open(TRAFFIC_FILE))
        ent = t.get("workloads", {}).get(workload_key, {}).get(kernel)
        if ent:
            traffic = round(ent["traffic_bytes_per_launch"])
            traffic_src = f"profiles/{os.path.basename(TRAFFIC_FILE)} ({t.get('commit', '?')}, {t.get('date', '?')})"
    total_ms = sum(v[0] for v in per.values())
    others = [dict(kernel=k, launches_per_step=v[3] // steps, avg_launch_us=round(1e3 * v[0] / v[3], 2),
                   achieved=round(v[1] / (v[0] * 1e-3) / 1e12, 2), frac=round(v[1] / (v[0] * 1e-3) / 1e12 / peak, 4),
                   flop_share_of_step=round(v[1] / steps / unit_flop_total, 3) if unit_flop_total else None,
                   share_of_mfma_kernel_time=round(v[0] / total_ms, 3))
              for k, v in sorted(per.items(), key=lambda kv: -kv[1][0])]
    shapes = []
    for (fl, by, sk, wg), (ms, c) in sorted(_SHAPES.get(kernel, {}).items(), key=lambda kv: -kv[1][0]):
        ach = fl / (ms / c * 1e-3) / 1e12
        e = dict(launches_per_step=c // steps, workgroups=wg or None, avg_launch_us=round(1e3 * ms / c, 2), flop_per_launch=fl,
                 algorithmic_bytes=by, k_slices=sk, achieved=round(ach, 2), frac=round(ach / peak, 4))
        if wg and wg < 256:      # one 144-KiB workgroup per CU: the launch occupies wg of the 256 CUs by plan
            e["frac_of_the_cus_it_occupies"] = round(ach / (peak * wg / 256), 4)
        shapes.append(e)
    return dict(bound="mfma", achieved=round(achieved, 2), peak=peak, unit="TFLOP/s", frac=round(achieved / peak, 4),
                launch_shapes=shapes,
                traffic=traffic, traffic_source=traffic_src, algorithmic_bytes_per_launch=round(tbytes / n),
                kernel=kernel, launches_per_step=n // steps, sliced_launches_per_step=nsplit // steps,
                avg_launch_us=round(1e3 * tms / n, 2), flop_per_launch=round(tflop / n),
                flop_share_of_step=round(tflop / steps / unit_flop_total, 3) if unit_flop_total else None,
                share_of_mfma_kernel_time=round(tms / total_ms, 3),
                all_mfma_kernels=dict(ms_per_step=round(total_ms / steps, 3),
                                      achieved=round(sum(v[1] for v in per.values()) / (total_ms * 1e-3) / 1e12, 2)),
                kernels=others)


def kernel_roofline(trainer, xa, xb, steps, precision):
    """Live measurement: libpfa_hip brackets every MFMA kernel launch (implicit GEMM, patch kernels, weight gradients) with
    HIP events on its launch stream over `steps` eager one-stream steps of the same workload; ALL launches of a kernel
    instantiation count - the population `rocprofv3 --kernel-trace --stats` averages for that kernel name
    (profiles/r02_*_onestream_* is the same mode)."""
    import torch
    from pfa_gan import _lib
    lib = _lib.lib()
    was_graph, was_side, was_side2, was_plan = trainer.use_graph, trainer.side, trainer.side2, trainer.plan_shared
    trainer.use_graph, trainer.side, trainer.side2 = False, None, None   # one stream: nothing shares the GPU with a timed launch
    trainer.plan_shared = was_side is not None                           # ... and every launch keeps the plan it has in the timed step
    lib.pfa_gather_timing(1)
    _SHAPES.clear()
    try:
        for _ in range(steps):
            trainer.step_packed(xa, xb)
        torch.cuda.synchronize()
    finally:
        lib.pfa_gather_timing(0)
        trainer.use_graph, trainer.side, trainer.side2, trainer.plan_shared = was_graph, was_side, was_side2, was_plan
    per = _timed_kernels(lib, _lib)
    peak = PEAK_BF16_TFLOPS if precision == "bf16" else PEAK_F32_TFLOPS
    unit = FLOP_PER_PAIR.get(xa.shape[1], 0) * xa.shape[0]
    roof = _roofline_object(per, steps, peak, unit, f"train_b{xa.shape[0]}_{xa.shape[1]}_{precision}")
    roof["mode"] = ("one HIP stream: every launch alone on the GPU, with the launch plan it has in the timed three-stream step "
                    "(e.g. c4's input gradient K-sliced on 128-channel tiles in the D phase, where it shares the GPU, and "
                    "un-sliced on 64-channel tiles in the generator phase, where it runs alone).  launch_shapes: the dominant "
                    "kernel's launches by operand shape - in the default workload c4's forward (256 workgroups, the whole chip) "
                    "and its D-phase input gradient (128 workgroups of twice the K: half the chip by plan, the weight "
                    "gradients of the other streams take the rest); `achieved` averages over both as rocprofv3 --stats does")
    return roof


# ----------------------------------------------------------------------------------------------------- CPU baseline
def _cpu_info():
    model, phys = "unknown", set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # a container's CPU quota (cgroup v2 cpu.max / v1 cfs quota) bounds what the threads can really use
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = int(q) / int(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        usable = max(1, min(usable, int(quota + 0.5)))
    return model, len(phys) or None, os.cpu_count(), usable


def cpu_baseline(size, budget_s, infer=False):
    """The reference's CPU path restated: pfa_gan modules on the CPU are plain PyTorch sub-modules (ATen / oneDNN, fp32)
    and pfa_gan.train_gan_one_epoch then runs the reference's own sequence of calls (train_gan.py:52-71) with
    torch.optim.Adam - what BASELINE.md section 5 specifies.  Batch 1 (the reference's BATCH_SIZE, train_gan.py:26) and
    batch 4, 1 warm-up + 3 timed steps each, bounded by `budget_s`.  Thread count: every usable core, unless a quick
    probe shows fewer threads are faster on this host (stated in `sample`)."""
    import torch
    import pfa_gan
    model, phys, logical, usable = _cpu_info()
    t_start = time.perf_counter()
    torch.manual_seed(0)
    gen = pfa_gan.UNetGenerator(3, 3)
    disc = pfa_gan.NLayerDiscriminator(6)
    og = torch.optim.Adam(gen.parameters(), lr=1e-4, betas=(0.5, 0.999))
    od = torch.optim.Adam(disc.parameters(), lr=1e-4, betas=(0.5, 0.999))
    g = torch.Generator().manual_seed(1234)

    def batches(b, n):
        return [dict(image1=torch.rand((b, 3, size, size), generator=g) * 2 - 1,
                     image2=torch.rand((b, 3, size, size), generator=g) * 2 - 1) for _ in range(n)]

    def run(b, n):
        if infer:
            gen.eval()
            xs = [bt["image1"] for bt in batches(b, n)]
            t0 = time.perf_counter()
            with torch.no_grad():
                for x in xs:
                    gen(x)
            return (time.perf_counter() - t0) / n
        bs = batches(b, n)
        t0 = time.perf_counter()
        pfa_gan.train_gan_one_epoch(gen, disc, bs, og, od)
        return (time.perf_counter() - t0) / n

    # thread count: all usable cores unless fewer are measurably faster (small convolutions on
    # very many threads spend their time in the thread pool)
    cands = sorted({min(usable, 16), min(usable, 32), min(usable, 64), usable})      # ascending: stop when more threads lose
    best_t, best_s = cands[0], None
    probed = []
    for t in cands:
        torch.set_num_threads(t)
        run(4, 1)                                     # warm-up at this thread count
        s = run(4, 1)
        probed.append(t)
        if best_s is None or s < best_s:
            best_t, best_s = t, s
        elif s > 1.1 * best_s:
            break
        if time.perf_counter() - t_start > 0.4 * budget_s:
            break
    torch.set_num_threads(best_t)
    res = {}
    for b in (4, 1):
        run(b, 1)
        n = 3
        s = run(b, n)
        res[b] = (b / s, n)
        if time.perf_counter() - t_start > budget_s:
            break
    unit = "tiles/s" if infer else "img-pairs/s"
    b_best = max(res, key=lambda b: res[b][0])
    what = "generator eval forward (no_grad)" if infer else "train_gan_one_epoch on CPU modules + torch.optim.Adam"
    return dict(value=round(res[b_best][0], 3), unit=unit, cores=best_t, kind="port",
                by_batch={str(b): round(v[0], 3) for b, v in sorted(res.items())},
                cpu_model=model, physical_cores=phys, logical_cores=logical, usable_cores=usable, threads=best_t,
                torch=torch.__version__,
                sample=f"pfa_gan {what}, default nets, {size}x{size}, fp32, batch "
                       f"{' and '.join(str(b) for b in sorted(res))}: 1 warm-up + 3 timed steps each; value = batch {b_best}; "
                       f"threads probed {probed}, fastest {best_t}")


def cpu_baseline_isolated(size, budget_s, infer=False):
    """cpu_baseline() in a child process with a hard time limit: the CPU leg can never hang or starve the GPU process
    (its thread pools stay out of this process), and the JSON line is printed whatever happens to it."""
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--size", str(size), "--cpu-seconds", str(budget_s),
           "--mode", "infer" if infer else "train"]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=3 * budget_s + 60)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode == 0 and lines:
            return json.loads(lines[-1])
        return dict(value=None, unit="tiles/s" if infer else "img-pairs/s", cores=None, kind="port",
                    sample=f"cpu_baseline child failed (rc {r.returncode}): {r.stderr[-300:]}")
    except subprocess.TimeoutExpired:
        return dict(value=None, unit="tiles/s" if infer else "img-pairs/s", cores=None, kind="port",
                    sample=f"cpu_baseline child exceeded {3 * budget_s + 60:.0f} s")


# ----------------------------------------------------------------------------------------------------- N > 1: what the collectives cost
def comm_diagnostics(trainer, pool, args, world, barrier, ms_with):
    """After the headline measurement, on every rank: (1) the same timed loop with the gradient all-reduces switched off
    (GanTrainer.skip_collectives: the ranks' parameters diverge, which no longer matters) - the difference to the headline
    ms/step is what the collectives cost the step, exposed or through contention; (2) each gradient range of the step
    all-reduced alone, back to back: its time and bus bandwidth (2 (n-1)/n x bytes / time - above one xGMI link's ~153 GB/s
    only a multi-link algorithm can be at work); (3) the bucket layout.  -> dict for config["comm"] (rank 0's view; times are
    the max over ranks)."""
    import torch
    import torch.distributed as dist

    def tmax(x):
        t = torch.tensor([x], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t)

    g_total, d_total = trainer.opt_g.flat().total, trainer.opt_d.flat().total
    buckets, whens = trainer.G.grad_buckets()
    d_lo = trainer.Dn.grad_bucket()[0]
    esize = 2 if trainer.grad_payload else 4
    ranges = [("G " + "/".join(str(x) for x in w), hi - lo) for (lo, hi), w in zip(buckets, whens)]
    ranges += [("D(real) arena", d_total), ("D(fake) c4..c5", d_total - d_lo), ("D(fake) c1..c3", d_lo)]
    probe = []
    reduce_ = trainer.ccomm.allreduce if trainer.ccomm is not None else dist.all_reduce     # the step's own transport
    for name, n in ranges:
        if n <= 0:
            continue
        buf = torch.zeros(n, dtype=torch.bfloat16 if trainer.grad_payload else torch.float32, device="cuda")
        for _ in range(2):
            reduce_(buf)
        barrier()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            reduce_(buf)
        torch.cuda.synchronize()
        ms = tmax(1e3 * (time.perf_counter() - t0) / reps)
        probe.append(dict(range=name, mbytes=round(n * esize / 1e6, 2), ms=round(ms, 3),
                          busbw_gbs=round(2 * (world - 1) / world * n * esize / (ms * 1e-3) / 1e9, 3)))   # (3 decimals: a 0.5 MB range over a slow transport - the gloo rehearsal - must not round to 0)
        del buf
    trainer.skip_collectives = True
    try:
        for i in range(3):
            trainer.step_packed(*pool[i % len(pool)])
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            trainer.step_packed(*pool[i % len(pool)])
        barrier()
        ms_without = tmax(1e3 * (time.perf_counter() - t0) / args.steps)
    finally:
        trainer.skip_collectives = False
    return dict(ms_per_step_without_collectives=round(ms_without, 3), comm_cost_ms_per_step=round(ms_with - ms_without, 3),
                payload="bf16" if trainer.grad_payload else "f32",
                bytes_per_step_mb=round((g_total + 2 * d_total) * esize / 1e6, 1),
                allreduce_alone=probe, backend=dist.get_backend(),
                transport="pfa_allreduce" if trainer.ccomm is not None else "torch.distributed.all_reduce",
                note="comm_cost = headline ms/step - the same loop without the gradient all-reduces; allreduce_alone: each "
                     "gradient range of the step reduced by itself (5 back-to-back calls)")


# ----------------------------------------------------------------------------------------------------- extra legs
def fp32_leg(args, pool32_src, steps=5, warmup=2):
    """The parity mode (exact-f32 MFMA: the mode that meets the north star's 1e-3 / 1e-2 gate, and the reference's own
    arithmetic, train_gan.py:46-75) on the SAME workload, a short pass after the headline measurement: fresh default networks,
    the same synthetic batches, eager on three streams.  -> dict for the line's "fp32" key."""
    import torch
    import pfa_gan
    torch.manual_seed(0)
    gen = pfa_gan.UNetGenerator(3, 3).cuda()
    disc = pfa_gan.NLayerDiscriminator(6).cuda()
    tr = pfa_gan.GanTrainer(gen, disc, precision="fp32", overlap=not args.no_overlap, inputs_resident=True)
    pool = [(tr.pack(a), tr.pack(b)) for a, b in pool32_src]
    torch.cuda.synchronize()
    for i in range(warmup):
        tr.step_packed(*pool[i % len(pool)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        tr.step_packed(*pool[i % len(pool)])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    losses = tr.losses.tolist()
    assert all(l == l and abs(l) < 1e6 for l in losses), f"fp32 leg: non-finite loss {losses}"
    value = args.batch * steps / dt
    flop_pair = FLOP_PER_PAIR.get(args.size, 0)
    return dict(value=round(value, 2), unit="img-pairs/s", ms_per_step=round(1e3 * dt / steps, 3), steps=steps, warmup=warmup,
                dtype="fp32", algorithmic_tflops=round(value * flop_pair / 1e12, 2), peak=PEAK_F32_TFLOPS,
                frac_of_peak=round(value * flop_pair / 1e12 / PEAK_F32_TFLOPS, 4),
                note="parity mode: v_mfma_f32_16x16x4_f32 everywhere, float32 storage; same workload and launch mode")


def measure_infer(batch, size, precision, steps, warmup, repeats):
    """BASELINE config 5's core: eval-mode generator forward (BatchNorm as an affine epilogue), one captured hipGraph,
    inputs resident.  -> (tiles/s, seconds per replay, repeats, engine, ctx, packed input)"""
    import torch
    import pfa_gan
    from pfa_gan import kernels as K
    torch.manual_seed(0)
    gen = pfa_gan.UNetGenerator(3, 3).cuda().set_precision(precision).eval()
    eng = gen.engine()
    eng.refresh_shadows(force=True)
    ctx = eng.new_ctx()
    ctx.pin()
    x = torch.rand((batch, 3, size, size), device="cuda") * 2 - 1
    xh = K.pack_nhwc(x, eng.dtype)
    for _ in range(3):
        eng.forward(xh, ctx, train=False)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        eng.forward(xh, ctx, train=False)
    for _ in range(warmup):
        graph.replay()
    torch.cuda.synchronize()
    reps = []
    for _ in range(max(1, repeats)):
        t0 = time.perf_counter()
        for _ in range(steps):
            graph.replay()
        torch.cuda.synchronize()
        reps.append(time.perf_counter() - t0)
    dt = statistics.median(reps)
    return batch * steps / dt, dt / steps, reps, eng, ctx, xh


def infer_leg(args, batch=64, steps=20, warmup=5):
    """-> dict for the training line's "infer" key (config 5 at batch 64; `--mode infer` is the full line)."""
    tiles, per, _, _, _, _ = measure_infer(batch, args.size, args.precision, steps, warmup, 3)
    peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else PEAK_F32_TFLOPS
    flop_tile = FLOP_PER_TILE.get(args.size, 0)
    return dict(value=round(tiles, 1), unit="tiles/s", batch=batch, ms_per_batch=round(1e3 * per, 3), steps=steps,
                dtype=args.precision, algorithmic_tflops=round(tiles * flop_tile / 1e12, 2),
                frac_of_peak=round(tiles * flop_tile / 1e12 / peak, 4),
                workload=f"UNetGenerator eval forward (generate_synthetic_data.py:55-68), {args.size}x{args.size}, hipGraph "
                         f"replay, inputs resident")


# ----------------------------------------------------------------------------------------------------- inference (config 5)
def infer_main(args):
    """BASELINE config 5: eval-mode generator forward over synthetic 256x256 tiles, hipGraph-captured, one GPU."""
    import torch
    import pfa_gan
    from pfa_gan import _lib, kernels as K
    torch.cuda.set_device(0)
    lib = _lib.lib()
    tiles, dt_step, reps, eng, ctx, xh = measure_infer(args.batch, args.size, args.precision, args.steps, args.warmup, args.repeats)
    dt = dt_step * args.steps
    peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else PEAK_F32_TFLOPS
    flop_tile = FLOP_PER_TILE.get(args.size, 0)
    out = {
        "metric": "generator-only eval forward, 256x256 tiles/s", "value": round(tiles, 1), "unit": "tiles/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
        "repeats_ms_per_step": [round(1e3 * r / args.steps, 3) for r in reps],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "config": {"workload": f"UNetGenerator eval forward (generate_synthetic_data.py:55-68), {args.size}x{args.size}, "
                               f"batch {args.batch}, hipGraph replay"},
        "step_mfma": {"algorithmic_tflops_per_gpu": round(tiles * flop_tile / 1e12, 2),
                      "frac_of_peak": round(tiles * flop_tile / 1e12 / peak, 4)}}
    if not args.no_roofline:
        n = max(2, min(args.steps, 5))
        lib.pfa_gather_timing(1)
        try:
            for _ in range(n):
                eng.forward(xh, ctx, train=False)
            torch.cuda.synchronize()
        finally:
            lib.pfa_gather_timing(0)
        out["roofline"] = _roofline_object(_timed_kernels(lib, _lib), n, peak, flop_tile * args.batch,
                                           f"infer_b{args.batch}_{args.size}_{args.precision}")
    if not args.no_cpu_baseline and args.size == 256:
        out["cpu_baseline"] = cpu_baseline_isolated(args.size, args.cpu_seconds, infer=True)
    print(json.dumps(out), flush=True)


def main():
    args = parse()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args.size, args.cpu_seconds, infer=args.mode == "infer")), flush=True)
        return
    if args.mode == "infer":
        return infer_main(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)                                # never returns
    if world != args.gpus:
        args.gpus = world
    import torch
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
    trainer = pfa_gan.GanTrainer(gen, disc, precision=args.precision, process_group=pg, use_graph="auto" if args.auto_graph else ("replay" if args.replay else args.graph),
                                 overlap=not args.no_overlap, inputs_resident=not args.no_early_real,
                                 grad_payload="bf16" if args.grad_bf16 else None, collectives=args.collectives)
    if world > 1:                                    # rank 0's parameters and buffers are the job's
        for t in [trainer.opt_g.flat().data, trainer.opt_d.flat().data] + list(gen.buffers()) + list(disc.buffers()):
            torch.distributed.broadcast(t, 0)
        trainer.G.refresh_shadows(force=True)
        trainer.Dn.refresh_shadows(force=True)

    gsrc = torch.Generator(device="cuda").manual_seed(1234 + rank)
    pool, pool_src = [], []
    for _ in range(8):                               # distinct batches, already packed: resident in HBM
        a = torch.rand((args.batch, 3, args.size, args.size), device="cuda", generator=gsrc) * 2 - 1
        b = torch.rand((args.batch, 3, args.size, args.size), device="cuda", generator=gsrc) * 2 - 1
        pool.append((trainer.pack(a), trainer.pack(b)))
        if len(pool_src) < 2:
            pool_src.append((a, b))                  # (float32 sources of two batches: the fp32 leg packs them itself)
    torch.cuda.synchronize()                         # resident: every batch is complete before the first step

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if args.plan_shared:
        trainer.plan_shared = True
    launches0 = _lib.launch_count()
    if os.environ.get("PFA_MAIN_HIGH"):              # A/B switch: the step's main stream at high priority (side streams: normal)
        hp = torch.cuda.Stream(priority=-1)
        hp.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(hp)
    for i in range(args.warmup):
        trainer.step_packed(*pool[i % len(pool)])
    reps = []
    for r in range(max(1, args.repeats)):
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):                  # EXACTLY --steps steps between the two barriers
            trainer.step_packed(*pool[i % len(pool)])
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], device="cuda", dtype=torch.float64)
            torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
            dt = float(tmax)
        reps.append(dt)
    dt = statistics.median(reps)
    losses = trainer.losses.tolist()
    assert all(l == l and abs(l) < 1e6 for l in losses), f"non-finite loss {losses}"
    assert _lib.launch_count() > launches0

    # every rank takes part in the roofline pass: its steps carry the same gradient all-reduces as the timed ones
    roof = None if args.no_roofline else kernel_roofline(trainer, *pool[0], max(2, min(args.steps, 5)), args.precision)
    comm = comm_diagnostics(trainer, pool, args, world, barrier, 1e3 * dt / args.steps) if world > 1 else None
    if rank == 0:
        pairs = world * args.batch * args.steps
        value = pairs / dt
        flop_pair = FLOP_PER_PAIR.get(args.size)
        peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else PEAK_F32_TFLOPS
        out = {
            "metric": "GAN img-pairs/sec (G+D step, 256x256x3)" if args.size == 256 else f"GAN img-pairs/sec (G+D step, {args.size}x{args.size}x3)",
            "value": round(value, 2), "unit": "img-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3),
            "repeats_ms_per_step": [round(1e3 * r / args.steps, 3) for r in reps],
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"{args.size}x{args.size} RGB Pix2Pix G+D step (UNetGenerator num_downs=7 ngf=64 + "
                                   f"NLayerDiscriminator ndf=64, BatchNorm, BCE+100*L1, 2x Adam), batch {args.batch}/GPU",
                       "global_batch": world * args.batch, "parallelism": f"dp{world}",
                       "grad_payload": "bf16" if trainer.grad_payload else "f32",
                       "collectives": None if world == 1 else ("pfa_allreduce (library communicator)" if trainer.ccomm is not None
                                                              else "torch.distributed"),
                       "launch": "hipGraph x3 segments" if trainer.use_graph is True else
                                 (("launch lists (pfa_replay), " if trainer.use_graph == "replay" else "eager, ") +
                                  ("3 HIP streams" if not args.no_overlap else "1 HIP stream")),
                       "timing": f"median of {len(reps)} repeats of {args.steps} steps",
                       "last_losses": [round(l, 4) for l in losses]},
        }
        if flop_pair:
            tf = value * flop_pair / 1e12 / world
            out["step_mfma"] = {"algorithmic_tflops_per_gpu": round(tf, 2), "frac_of_peak": round(tf / peak, 4)}
        bpp = BYTES_PER_PAIR.get((args.size, args.batch))
        if bpp and args.precision == "bf16":
            gbs = value / world * bpp / 1e9
            out["step_hbm"] = {"algorithmic_bytes_per_step": round(bpp * args.batch), "achieved": round(gbs, 1), "unit": "GB/s",
                               "peak": PEAK_HBM_GBS, "frac": round(gbs / PEAK_HBM_GBS, 4),
                               "note": "SURVEY 8(d) ideal-fusion bytes per img-pair x pairs/s: the step sits on the ridge, "
                                       "so both the MFMA and the HBM fraction are reported"}
        if roof is not None:
            out["roofline"] = roof
        if comm is not None:
            out["config"]["comm"] = comm
        if world == 1 and not args.no_extras and args.precision == "bf16" and args.size == 256:
            # extra legs, AFTER the headline's timed region and roofline pass, each bounded to a fraction of a second of GPU
            # time: the parity mode on the same workload, and BASELINE config 5
            del trainer
            import gc
            gc.collect()                             # the trainer's reference cycles: collected NOW, not in the middle of a leg
            torch.cuda.synchronize()                 # (freeing its buffers synchronises the device: a leg measured 40 % low)
            try:
                out["fp32"] = fp32_leg(args, pool_src)
            except Exception as e:                   # an extra leg never costs the headline line
                out["fp32"] = {"value": None, "error": repr(e)[:300]}
            gc.collect()                             # (the fp32 leg's trainer)
            torch.cuda.synchronize()
            try:
                out["infer"] = infer_leg(args)
            except Exception as e:
                out["infer"] = {"value": None, "error": repr(e)[:300]}
        if world == 1 and not args.no_cpu_baseline and args.size == 256:
            out["cpu_baseline"] = cpu_baseline_isolated(args.size, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
