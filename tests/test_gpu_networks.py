"""-m gpu: whole-network and whole-step parity of the HIP engines against reference outputs (tests/golden)
and the numpy oracle.  fp32 mode carries the north-star gate: 1e-3 abs / 1e-2 rel on the forward."""
import io

import numpy as np
import pytest
import torch

import pfa_gan
from conftest import assert_close
from make_golden import NET_CASES, STEP_CASES
from oracle import fill, nets, step as ostep
from pfa_gan import _lib

pytestmark = pytest.mark.gpu

ABS, REL = 1e-3, 1e-2          # the north star's forward gate


def load(module, st):
    module.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in st.items()}, strict=True)


def grad_close(got, want, what, rel_l2=2e-3):
    """Gradient parity that tolerates the rare ReLU knife-edge element (tests/make_golden.py:_pick_seed)."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    scale = max(np.abs(want).max(), 1e-12)
    err = np.abs(got - want) / scale
    l2 = np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-12)
    assert np.median(err) < 2e-4 and l2 < rel_l2, f"{what}: rel L2 {l2:.2e}, median {np.median(err):.2e}, max {err.max():.2e}"


@pytest.mark.parametrize("case", NET_CASES, ids=[c[0] for c in NET_CASES])
def test_generator_fp32_vs_reference(golden, case):
    name, nd, ngf, ndf, n, h = case
    z = golden("nets")
    gst = fill.fill_state(nets.g_state_shapes(3, 3, nd, ngf), seed=1)
    G = pfa_gan.UNetGenerator(3, 3, num_downs=nd, ngf=ngf).cuda()
    load(G, gst)
    seed = int(z[f"{name}.g.seed"])
    x = torch.from_numpy(fill.uniform(f"{name}.x", (n, 3, h, h), seed=seed)).cuda()
    G.train()
    before = _lib.launch_count()
    y = G(x)
    assert _lib.launch_count() - before > 20, "forward must run on libpfa_hip kernels"
    assert_close(y.detach().cpu().numpy(), z[f"{name}.g.y"], ABS, REL, "G train forward")
    dy = torch.from_numpy(fill.uniform(f"{name}.g.dy", tuple(y.shape), seed=seed)).cuda()
    for p in G.parameters():
        p.grad = None
    y.backward(dy)
    for k, p in G.named_parameters():
        grad_close(fill.compact(p.grad.cpu().numpy()), z[f"{name}.g.grad.{k}"], f"G grad {k}")
    for k, b in G.named_buffers():
        assert_close(b.cpu().numpy(), z[f"{name}.g.buf.{k}"], 1e-5, 1e-3, f"G buffer {k}")
    G.eval()
    with torch.no_grad():
        ye = G(x)
    assert_close(ye.cpu().numpy(), z[f"{name}.g.y_eval"], ABS, REL, "G eval forward")


@pytest.mark.parametrize("case", NET_CASES, ids=[c[0] for c in NET_CASES])
def test_discriminator_fp32_vs_reference(golden, case):
    name, nd, ngf, ndf, n, h = case
    z = golden("nets")
    dst = fill.fill_state(nets.d_state_shapes(6, ndf, 3), seed=2)
    D = pfa_gan.NLayerDiscriminator(6, ndf=ndf).cuda()
    load(D, dst)
    seed = int(z[f"{name}.d.seed"])
    x = torch.from_numpy(fill.uniform(f"{name}.xd", (n, 6, h, h), seed=seed)).cuda().requires_grad_(True)
    D.train()
    y = D(x)
    assert_close(y.detach().cpu().numpy(), z[f"{name}.d.y"], ABS, REL, "D train forward")
    dy = torch.from_numpy(fill.uniform(f"{name}.d.dy", tuple(y.shape), seed=seed)).cuda()
    for p in D.parameters():
        p.grad = None
    y.backward(dy)
    grad_close(x.grad.cpu().numpy(), z[f"{name}.d.dx"], "D dx")
    for k, p in D.named_parameters():
        grad_close(fill.compact(p.grad.cpu().numpy()), z[f"{name}.d.grad.{k}"], f"D grad {k}")
    for k, b in D.named_buffers():
        assert_close(b.cpu().numpy(), z[f"{name}.d.buf.{k}"], 1e-5, 1e-3, f"D buffer {k}")
    D.eval()
    with torch.no_grad():
        ye = D(x.detach())
    assert_close(ye.cpu().numpy(), z[f"{name}.d.y_eval"], ABS, REL, "D eval forward")


@pytest.mark.parametrize("case", NET_CASES, ids=[c[0] for c in NET_CASES])
def test_bf16_networks_match_the_bf16_storage_oracle(case):
    """The throughput mode, gated tightly at network level: the bf16 train-mode forward of both networks against the
    oracle restated with a bf16 rounding at the engine's storage points (oracle/nets.py: generator_fwd_bf16 /
    discriminator_fwd_bf16).  What is left is the order of the float32 sums and rounding ties: a handful of elements one
    bf16 ulp apart (2^-8 relative), against the 4.7e-2 a bf16 evaluation drifts from the float32 reference."""
    name, nd, ngf, ndf, n, h = case
    gst = fill.fill_state(nets.g_state_shapes(3, 3, nd, ngf), seed=1)
    G = pfa_gan.UNetGenerator(3, 3, num_downs=nd, ngf=ngf).cuda().set_precision("bf16")
    load(G, gst)
    x = fill.uniform(f"{name}.x16", (n, 3, h, h))
    G.train()
    with torch.no_grad():
        y = G(torch.from_numpy(x).cuda()).cpu().numpy()
    want = nets.generator_fwd_bf16(gst, x, nd)
    err = np.abs(y - want)
    assert err.max() <= 2.0 ** -6 and np.mean(err > 2.0 ** -9) < 0.02 and err.mean() < 2e-4, \
        f"G: max {err.max():.2e}, mean {err.mean():.2e}, frac > 2^-9 {np.mean(err > 2.0 ** -9):.3%}"
    dst = fill.fill_state(nets.d_state_shapes(6, ndf, 3), seed=2)
    D = pfa_gan.NLayerDiscriminator(6, ndf=ndf).cuda().set_precision("bf16")
    load(D, dst)
    xd = fill.uniform(f"{name}.xd16", (n, 6, h, h))
    D.train()
    with torch.no_grad():
        yd = D(torch.from_numpy(xd).cuda()).cpu().numpy()
    wantd = nets.discriminator_fwd_bf16(dst, xd)
    errd = np.abs(yd - wantd)
    scale = max(np.abs(wantd).max(), 1.0)
    assert errd.max() <= 2.0 ** -6 * scale and errd.mean() < 5e-4 * scale, f"D: max {errd.max():.2e}, mean {errd.mean():.2e}"


def _digest_ok(z, name, a, atol, rtol):
    flat = np.asarray(a).reshape(-1)
    stride = int(z[name + ".stride"])
    assert flat.size == int(z[name + ".size"])
    assert_close(flat[::stride][: z[name + ".sample"].size], z[name + ".sample"], atol, rtol, name + ".sample")
    a64 = flat.astype(np.float64)
    assert_close(np.abs(a64).sum(), z[name + ".abs_sum"], 0, 1e-3, name + ".abs_sum")
    assert_close((a64 * a64).sum(), z[name + ".sq_sum"], 0, 1e-3, name + ".sq_sum")


@pytest.mark.parametrize("tag,n,h", [("b2_128", 2, 128), ("b1_256", 1, 256)])
def test_full_size_fp32_vs_reference(golden, tag, n, h):
    """Default networks (num_downs=7, ngf=64; 41.8 M + 2.77 M parameters) at the reference's 256x256."""
    z = golden("full")
    G = pfa_gan.UNetGenerator(3, 3).cuda()
    gst = fill.fill_state(nets.g_state_shapes(), seed=3)
    load(G, gst)
    x = torch.from_numpy(fill.uniform(f"full.{tag}.x", (n, 3, h, h))).cuda()
    G.train()
    with torch.no_grad():
        y = G(x).cpu().numpy()
    _digest_ok(z, f"full.{tag}.g.y", y, ABS, REL)
    assert_close(y.astype(np.float64).mean(axis=(0, 2, 3)), z[f"full.{tag}.g.y_chan_mean"], 1e-4, 1e-3)
    run = np.concatenate([b.cpu().numpy().reshape(-1) for k, b in G.named_buffers() if "running" in k])
    _digest_ok(z, f"full.{tag}.g.running_cat", run, 1e-5, 1e-3)
    load(G, gst)
    G.eval()
    with torch.no_grad():
        ye = G(x).cpu().numpy()
    _digest_ok(z, f"full.{tag}.g.y_eval", ye, ABS, REL)
    D = pfa_gan.NLayerDiscriminator(6).cuda()
    load(D, fill.fill_state(nets.d_state_shapes(), seed=4))
    xd = torch.from_numpy(fill.uniform(f"full.{tag}.xd", (n, 6, h, h))).cuda()
    D.train()
    with torch.no_grad():
        yd = D(xd).cpu().numpy()
    assert_close(yd, z[f"full.{tag}.d.y"], ABS, REL, "D logits")


def _check_step_state(z, name, G, D, tol_param=2e-5):
    for prefix, m in (("g", G), ("d", D)):
        for k, v in m.state_dict().items():
            want = z[f"step.{name}.{prefix}.{k}"]
            v = v.cpu().numpy()
            if k.endswith("num_batches_tracked"):
                assert int(v) == int(want), k
            elif ostep.is_param(k):
                got = fill.compact(v)
                frac_bad = np.mean(np.abs(got - want) > tol_param)
                assert frac_bad < 0.02, f"{prefix}.{k}: {frac_bad:.2%} of parameters off by > {tol_param}"
                assert np.abs(got - want).max() < 5e-4, f"{prefix}.{k}"
            else:
                assert_close(v, want, 1e-5, 2e-3, f"{prefix} buffer {k}")


@pytest.mark.parametrize("use_graph", [False, True, "replay"], ids=["eager", "hipgraph", "launch_lists"])
@pytest.mark.parametrize("case", STEP_CASES, ids=[c[0] for c in STEP_CASES])
def test_fused_step_fp32_vs_reference(golden, case, use_graph):
    """Two consecutive train_gan_one_epoch iterations of the reference vs GanTrainer (fused schedule): launched from
    Python, replayed as hipGraphs, and re-issued from C as launch lists (pfa_replay) on the three streams."""
    name, nd, ngf, ndf, n, h, steps = case
    z = golden("step")
    G = pfa_gan.UNetGenerator(3, 3, num_downs=nd, ngf=ngf).cuda()
    D = pfa_gan.NLayerDiscriminator(6, ndf=ndf).cuda()
    load(G, fill.fill_state(nets.g_state_shapes(3, 3, nd, ngf), seed=5, bn_noise=False))
    load(D, fill.fill_state(nets.d_state_shapes(6, ndf, 3), seed=6, bn_noise=False))
    tr = pfa_gan.GanTrainer(G, D, precision="fp32", use_graph=use_graph)
    seed = int(z[f"step.{name}.seed"])
    for i in range(steps):
        a = torch.from_numpy(fill.uniform(f"step.{name}.a{i}", (n, 3, h, h), seed=seed)).cuda()
        b = torch.from_numpy(fill.uniform(f"step.{name}.b{i}", (n, 3, h, h), seed=seed)).cuda()
        losses = tr.step(a, b).cpu().numpy()
        assert_close(losses[0], z[f"step.{name}.loss_d{i}"], 1e-4, 1e-4, f"loss_d step {i}")
        assert_close(losses[1], z[f"step.{name}.loss_g{i}"], 1e-3, 1e-4, f"loss_g step {i}")
        if not use_graph:
            for k, p in G.named_parameters():
                grad_close(fill.compact(p.grad.cpu().numpy()), z[f"step.{name}.ggrad{i}.{k}"], f"G grad {k} step {i}", 5e-3)
    _check_step_state(z, name, G, D)
    ld, lg = tr.epoch_losses()
    assert_close(ld, np.mean([z[f"step.{name}.loss_d{i}"] for i in range(steps)]), 1e-4, 1e-4)


def test_reference_style_loop_on_gpu_modules(golden):
    """The reference's own call sequence (G(x).detach(), D(cat), loss.backward(), optim.Adam.step) on the
    drop-in modules: every forward/backward is the HIP engine behind one autograd node per network."""
    name, nd, ngf, ndf, n, h, steps = STEP_CASES[0]
    z = golden("step")
    G = pfa_gan.UNetGenerator(3, 3, num_downs=nd, ngf=ngf).cuda()
    D = pfa_gan.NLayerDiscriminator(6, ndf=ndf).cuda()
    load(G, fill.fill_state(nets.g_state_shapes(3, 3, nd, ngf), seed=5, bn_noise=False))
    load(D, fill.fill_state(nets.d_state_shapes(6, ndf, 3), seed=6, bn_noise=False))
    og = torch.optim.Adam(G.parameters(), lr=1e-4, betas=(0.5, 0.999))
    od = torch.optim.Adam(D.parameters(), lr=1e-4, betas=(0.5, 0.999))
    seed = int(z[f"step.{name}.seed"])
    for i in range(steps):
        batch = dict(image1=torch.from_numpy(fill.uniform(f"step.{name}.a{i}", (n, 3, h, h), seed=seed)),
                     image2=torch.from_numpy(fill.uniform(f"step.{name}.b{i}", (n, 3, h, h), seed=seed)))
        ld, lg = pfa_gan.train_gan_one_epoch(G, D, [batch], og, od)
        assert_close(ld, z[f"step.{name}.loss_d{i}"], 1e-4, 1e-4, f"loss_d step {i}")
        assert_close(lg, z[f"step.{name}.loss_g{i}"], 1e-3, 1e-4, f"loss_g step {i}")
    _check_step_state(z, name, G, D)


def test_checkpoint_layout_roundtrip():
    """generator_epoch_*.pth contract (train_gan.py:149-153, generate_synthetic_data.py:46-48): plain
    state_dict, reference key order, float32 NCHW-contiguous tensors; strict reload both ways."""
    G = pfa_gan.UNetGenerator(3, 3, num_downs=5, ngf=8).cuda()
    x = torch.rand(2, 3, 32, 32, device="cuda") * 2 - 1
    G(x)                                            # binds the flat arena, updates BN buffers
    buf = io.BytesIO()
    torch.save(G.state_dict(), buf)
    buf.seek(0)
    sd = torch.load(buf, map_location="cpu")
    assert list(sd.keys()) == list(nets.g_state_shapes(3, 3, 5, 8).keys())
    for k, v in sd.items():
        assert v.is_contiguous(), k
        assert v.dtype == (torch.int64 if k.endswith("num_batches_tracked") else torch.float32)
    G2 = pfa_gan.UNetGenerator(3, 3, num_downs=5, ngf=8)
    G2.load_state_dict(sd, strict=True)             # CPU consumer (plain torch modules)
    G.eval(), G2.eval()
    with torch.no_grad():
        np.testing.assert_allclose(G(x).cpu().numpy(), G2(x.cpu()).numpy(), atol=ABS, rtol=REL)
    G3 = pfa_gan.UNetGenerator(3, 3, num_downs=5, ngf=8).cuda()
    G3(x)
    G3.load_state_dict(sd, strict=True)             # into an already-bound GPU model
    G3.eval()
    with torch.no_grad():
        np.testing.assert_array_equal(G(x).cpu().numpy(), G3(x).cpu().numpy())


def test_bf16_mode_tracks_fp32(golden):
    """Throughput mode: bf16 storage/operands, f32 accumulation.  The whole network cannot meet the fp32
    gate (4.7e-2 max-abs for bf16 autocast on the reference itself); stated tolerance here:
    max-abs 8e-2, mean-abs 1e-2 on the tanh output, losses within 2 %."""
    name, nd, ngf, ndf, n, h = NET_CASES[1]
    z = golden("nets")
    G = pfa_gan.UNetGenerator(3, 3, num_downs=nd, ngf=ngf).cuda().set_precision("bf16")
    load(G, fill.fill_state(nets.g_state_shapes(3, 3, nd, ngf), seed=1))
    seed = int(z[f"{name}.g.seed"])
    x = torch.from_numpy(fill.uniform(f"{name}.x", (n, 3, h, h), seed=seed)).cuda()
    G.train()
    with torch.no_grad():
        y = G(x).cpu().numpy()
    err = np.abs(y - z[f"{name}.g.y"])
    assert err.max() < 8e-2 and err.mean() < 1e-2, (err.max(), err.mean())
    sname, nd, ngf, ndf, n, h, steps = STEP_CASES[1]
    zs = golden("step")
    G = pfa_gan.UNetGenerator(3, 3, num_downs=nd, ngf=ngf).cuda()
    D = pfa_gan.NLayerDiscriminator(6, ndf=ndf).cuda()
    load(G, fill.fill_state(nets.g_state_shapes(3, 3, nd, ngf), seed=5, bn_noise=False))
    load(D, fill.fill_state(nets.d_state_shapes(6, ndf, 3), seed=6, bn_noise=False))
    tr = pfa_gan.GanTrainer(G, D, precision="bf16")
    seed = int(zs[f"step.{sname}.seed"])
    for i in range(steps):
        a = torch.from_numpy(fill.uniform(f"step.{sname}.a{i}", (n, 3, h, h), seed=seed)).cuda()
        b = torch.from_numpy(fill.uniform(f"step.{sname}.b{i}", (n, 3, h, h), seed=seed)).cuda()
        losses = tr.step(a, b).cpu().numpy()
        assert_close(losses[0], zs[f"step.{sname}.loss_d{i}"], 0, 2e-2, f"bf16 loss_d step {i}")
        assert_close(losses[1], zs[f"step.{sname}.loss_g{i}"], 0, 2e-2, f"bf16 loss_g step {i}")
    for k, v in G.state_dict().items():
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(zs[f"step.{sname}.g.{k}"])


BF16_STEP_CASES = [  # name, num_downs, ngf, ndf, N, H
    ("small", 5, 8, 8, 2, 32),
    ("wide", 5, 16, 16, 2, 64),
]


@pytest.mark.parametrize("route", ["fused", "unfused"])
@pytest.mark.parametrize("case", BF16_STEP_CASES, ids=[c[0] for c in BF16_STEP_CASES])
def test_bf16_step_matches_the_bf16_storage_step_oracle(case, route, monkeypatch):
    """The throughput mode's WHOLE STEP, gated tightly: GanTrainer(precision="bf16") against oracle/bf16.py - forward,
    backward and Adam with a bf16 rounding at the engines' storage points (anchored to the pinned float32 oracle by
    tests/test_oracle_golden.py; the float32 reference itself is 10-20 % away from any bf16 evaluation of these gradients,
    SURVEY F4).  Both routes of the backward are held: "unfused" (PFA_FUSE_BWD=0: every input gradient is stored before the
    activation backward of its node) and the default, where launches that can carry the fused backward epilogue do.
      small   every kernel of the step sums in the oracle's order: the un-fused route agrees to float32 round-off (1e-5) -
              losses, every weight gradient; a rounding point the engine has and the oracle lacks (or the reverse) cannot hide
      wide    the stride-1 patch kernel takes over c4 (another order of the float32 sums: a few last-bit differences in the
              forward).  Generator gradients <= 1.5e-2 in relative L2; the discriminator's <= 5e-2 per pass: at the default
              initialisation its logit gradients are nearly constant over the map, so the BatchNorm backward of c4 subtracts
              almost all of the stored (bf16) dy - what is left carries the rounding noise of the stored tensor amplified
              ~10-20x, in the engine and in the oracle alike, with a different noise pattern as soon as one bit differs
    Losses <= 1e-3 relative (observed <= 5e-5), BatchNorm buffers and counters, and Adam - whose first step moves a
    parameter by +-lr whatever the gradient's size, so one flipped sign is a whole step - moving >= 99 % of the parameters
    alike (train_gan.py:52-71)."""
    from oracle import bf16 as o16
    from pfa_gan import kernels as K
    name, nd, ngf, ndf, n, h = case
    unfused = route == "unfused"
    if unfused:
        monkeypatch.setattr(K, "FUSE_BWD", False)
    exact = unfused and name == "small"
    gst = fill.fill_state(nets.g_state_shapes(3, 3, nd, ngf), seed=5, bn_noise=False)
    dst = fill.fill_state(nets.d_state_shapes(6, ndf, 3), seed=6, bn_noise=False)
    a = fill.uniform(f"bf16step.{name}.a", (n, 3, h, h))
    b = fill.uniform(f"bf16step.{name}.b", (n, 3, h, h))
    G = pfa_gan.UNetGenerator(3, 3, num_downs=nd, ngf=ngf).cuda()
    D = pfa_gan.NLayerDiscriminator(6, ndf=ndf).cuda()
    load(G, gst)
    load(D, dst)
    tr = pfa_gan.GanTrainer(G, D, precision="bf16")
    losses = tr.step(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()).cpu().numpy()
    og, od = {k: v.copy() for k, v in gst.items()}, {k: v.copy() for k, v in dst.items()}
    ld, lg, ex = o16.gan_step(og, od, ostep.new_adam_state(og), ostep.new_adam_state(od), a, b, nd, 3, unfused=unfused)
    ltol = 2e-6 if exact else 1e-3
    assert abs(losses[0] - ld) <= ltol * abs(ld), (losses, ld)
    assert abs(losses[1] - lg) <= ltol * abs(lg), (losses, lg)
    fake = pfa_gan.kernels.unpack_nchw(tr.gctx.v["fake"], 3).cpu().numpy()
    dfake = np.abs(fake - ex["fake_b"])
    # (a tanh output near +-1 has a bf16 ulp of 2^-8: a last-bit difference upstream moves many of them by one ulp)
    assert dfake.mean() < (1e-6 if exact else 1e-3) and dfake.max() <= 2.0 ** -6, (dfake.mean(), dfake.max())
    torch.cuda.synchronize()
    g_tol, d_tol = (1e-5, 1e-5) if exact else (1.5e-2, 5e-2)
    for k, p in G.named_parameters():
        want = ex["g_grads"][k]
        if want.ndim == 4:
            rel = np.linalg.norm(p.grad.cpu().numpy().astype(np.float64) - want) / max(np.linalg.norm(want), 1e-20)
            assert rel <= g_tol, f"G grad {k}: rel L2 {rel:.2e}"
    # (the discriminator's two passes one by one: their gradients largely cancel in the sum Adam takes)
    flat_d, g2 = tr.opt_d.flat(), tr.opt_d.grad2()
    for k, p in D.named_parameters():
        o, cnt = flat_d.slots[id(p)]
        for what, got, want in (("fake", p.grad, ex["d_grads_fake"][k]), ("real", flat_d._view(g2, p, o, cnt), ex["d_grads_real"][k])):
            if want.ndim == 4:
                rel = np.linalg.norm(got.cpu().numpy().astype(np.float64) - want) / max(np.linalg.norm(want), 1e-20)
                assert rel <= d_tol, f"D({what}) grad {k}: rel L2 {rel:.2e}"
    for mod, ref in ((G, og), (D, od)):
        alike = total = 0
        for k, v in mod.state_dict().items():
            got = v.cpu().numpy()
            if k.endswith("num_batches_tracked"):
                assert int(got) == int(ref[k]), k
            elif ostep.is_param(k):
                alike += int(np.sum(np.abs(got - ref[k]) < 0.5 * ostep.LR))
                total += ref[k].size
            else:
                assert_close(got, ref[k], 1e-4, 2e-3, f"buffer {k}")
        assert alike >= (0.9999 if exact else 0.99) * total, f"only {alike / total:.2%} of the parameters moved as in the bf16 step oracle"


def test_gpu_input_without_library_path_is_an_error(monkeypatch):
    """No silent fallback: if the .so cannot be found the GPU path raises."""
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libpfa_hip.so")
    with pytest.raises(_lib.PfaError, match="mandatory"):
        pfa_gan.kernels.pack_nhwc(torch.zeros(1, 3, 4, 4, device="cuda"), torch.float32)


def test_data_parallel_code_path_single_rank_rccl(golden):
    """The data-parallel trainer path (rank-0 broadcast, D-gradient all-reduce with the second arena folded in, bucketed
    asynchronous all-reduce of the generator arena issued from the side stream during the backward) on a 1-rank RCCL
    group: every collective call is the real one, and the result must equal the single-process step exactly in
    schedule (same goldens).  Multi-rank semantics are covered on gloo in tests/test_host_cpu.py."""
    import os
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(29600 + os.getpid() % 1000)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        name, nd, ngf, ndf, n, h, steps = STEP_CASES[0]
        z = golden("step")
        G = pfa_gan.UNetGenerator(3, 3, num_downs=nd, ngf=ngf).cuda()
        D = pfa_gan.NLayerDiscriminator(6, ndf=ndf).cuda()
        load(G, fill.fill_state(nets.g_state_shapes(3, 3, nd, ngf), seed=5, bn_noise=False))
        load(D, fill.fill_state(nets.d_state_shapes(6, ndf, 3), seed=6, bn_noise=False))
        tr = pfa_gan.GanTrainer(G, D, precision="fp32", process_group=dist.group.WORLD)
        pfa_gan.broadcast_model(G, dist.group.WORLD)
        pfa_gan.broadcast_model(D, dist.group.WORLD)
        tr.G.refresh_shadows(force=True)
        tr.Dn.refresh_shadows(force=True)
        buckets, _ = tr.G.grad_buckets()
        assert sorted(buckets)[0][0] == 0 and sorted(buckets)[-1][1] == tr.G.flat.total
        assert all(a[1] == b[0] for a, b in zip(sorted(buckets), sorted(buckets)[1:])), "buckets must tile the arena"
        seed = int(z[f"step.{name}.seed"])
        for i in range(steps):
            a = torch.from_numpy(fill.uniform(f"step.{name}.a{i}", (n, 3, h, h), seed=seed)).cuda()
            b = torch.from_numpy(fill.uniform(f"step.{name}.b{i}", (n, 3, h, h), seed=seed)).cuda()
            losses = tr.step(a, b).cpu().numpy()
            assert_close(losses[0], z[f"step.{name}.loss_d{i}"], 1e-4, 1e-4, f"loss_d step {i}")
            assert_close(losses[1], z[f"step.{name}.loss_g{i}"], 1e-3, 1e-4, f"loss_g step {i}")
        _check_step_state(z, name, G, D)
    finally:
        dist.destroy_process_group()


def test_library_communicator_single_rank(golden):
    """The exchange step behind the C boundary (pfa_comm_* / pfa_allreduce, include/pfa_hip.h) on the one rank a one-GPU
    box can hold: RCCL is found and bound, a communicator comes up, float32 and bfloat16 ranges are reduced in place on the
    calling stream (one rank: the sum is the range itself), a launch list holds the call, and GanTrainer(collectives="c") -
    every gradient range of the step through pfa_allreduce on the stream that produced it, eagerly and with the step's
    segments replayed as launch lists - reproduces the single-process goldens.  More than one rank is unmeasured here."""
    import os
    import torch.distributed as dist
    from pfa_gan import kernels as K
    comm = pfa_gan.Comm.create(pfa_gan.Comm.unique_id(), 1, 0)
    assert _lib.lib().pfa_comm_size(comm.handle) == 1 and _lib.lib().pfa_comm_rank(comm.handle) == 0
    x = torch.randn(100003, device="cuda")
    xb = torch.randn(4099, device="cuda").to(torch.bfloat16)
    x0, xb0 = x.clone(), xb.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        comm.allreduce(x)
        comm.allreduce(xb)
        comm.allreduce(x[:0])                      # an empty range is no call at all
    torch.cuda.synchronize()
    assert torch.equal(x, x0) and torch.equal(xb, xb0)
    with pytest.raises(_lib.PfaError, match="contiguous"):
        comm.allreduce(x[::2])
    with pytest.raises(TypeError):
        comm.allreduce(x.double())
    y = torch.zeros_like(x)
    with _lib.Recorder() as rec:
        K.axpy(y, x, 1.0)
        comm.allreduce(y)
    ll = rec.finish()
    assert ll.calls == 2
    ll.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, 2 * x)
    comm.close()
    comm.close()                                   # idempotent

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(29700 + os.getpid() % 1000)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        name, nd, ngf, ndf, n, h, steps = STEP_CASES[0]
        z = golden("step")
        seed = int(z[f"step.{name}.seed"])
        for use_graph in (False, "replay"):
            G = pfa_gan.UNetGenerator(3, 3, num_downs=nd, ngf=ngf).cuda()
            D = pfa_gan.NLayerDiscriminator(6, ndf=ndf).cuda()
            load(G, fill.fill_state(nets.g_state_shapes(3, 3, nd, ngf), seed=5, bn_noise=False))
            load(D, fill.fill_state(nets.d_state_shapes(6, ndf, 3), seed=6, bn_noise=False))
            tr = pfa_gan.GanTrainer(G, D, precision="fp32", process_group=dist.group.WORLD, collectives="c",
                                    use_graph=use_graph)
            assert tr.ccomm is not None and tr.ccomm.nranks == 1
            calls = []
            inner = tr.ccomm.allreduce
            tr.ccomm.allreduce = lambda t: (calls.append(t.numel()), inner(t))[1]
            for i in range(steps):
                a = torch.from_numpy(fill.uniform(f"step.{name}.a{i}", (n, 3, h, h), seed=seed)).cuda()
                b = torch.from_numpy(fill.uniform(f"step.{name}.b{i}", (n, 3, h, h), seed=seed)).cuda()
                losses = tr.step(a, b).cpu().numpy()
                assert_close(losses[0], z[f"step.{name}.loss_d{i}"], 1e-4, 1e-4, f"loss_d step {i}")
                assert_close(losses[1], z[f"step.{name}.loss_g{i}"], 1e-3, 1e-4, f"loss_g step {i}")
            _check_step_state(z, name, G, D)
            assert sum(calls) >= steps * (tr.G.flat.total + tr.Dn.flat.total), "every gradient range went through pfa_allreduce"
            tr.ccomm.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["three-streams", "one-stream", "resident-inputs", "bf16-payload"])
def test_data_parallel_stream_order_under_a_slow_collective(monkeypatch, mode):
    """What neither gloo (host-blocking waits) nor a 1-rank RCCL group (an all-reduce that changes nothing) can show: that
    every consumer of a gradient range is ordered behind that range's collective, and every collective behind the range's
    last producer, by STREAM dependencies alone.  torch.distributed.all_reduce is replaced by an emulation with NCCL's
    stream semantics - the collective runs on a communicator stream that waits for the calling stream; work.wait() makes
    the then-current stream wait, never the host - which is slow (a matrix product in front of it) and mixes the range
    (t += roll(t) / 2: "the other rank's gradient"; not a pure scaling, which Adam would not notice).  The same steps with
    a collective that synchronises the whole device before and after itself are ordered trivially; the two runs must
    end in the same parameters, bit for bit.  An update that ran ahead of its collective, or a collective that ran ahead of
    a weight gradient, ends elsewhere."""
    import torch.distributed as dist
    comm = torch.cuda.Stream()
    ballast = torch.randn(2048, 2048, device="cuda")
    state = {"trusted": False, "calls": []}

    class _Work:
        def __init__(self, ev):
            self.ev = ev

        def wait(self):
            if self.ev is not None:
                torch.cuda.current_stream().wait_event(self.ev)
            return True

    def mix(t):
        t.add_(torch.roll(t, 1), alpha=0.5)

    def all_reduce(t, op=None, group=None, async_op=False):
        state["calls"].append((t.numel(), bool(async_op)))
        if state["trusted"]:
            torch.cuda.synchronize()
            mix(t)
            torch.cuda.synchronize()
            return _Work(None) if async_op else None
        cur = torch.cuda.current_stream()
        comm.wait_stream(cur)
        with torch.cuda.stream(comm):
            torch.mm(ballast, ballast)                     # ~0.2 ms: the collective is still running when the host moves on
            mix(t)
            ev = torch.cuda.Event()
            ev.record(comm)
        if async_op:
            return _Work(ev)
        cur.wait_event(ev)
        return None

    monkeypatch.setattr(dist, "all_reduce", all_reduce)
    monkeypatch.setattr(dist, "get_world_size", lambda g=None: 2)
    name, nd, ngf, ndf, n, h, _ = STEP_CASES[0]
    steps = 3
    batches = [(torch.from_numpy(fill.uniform(f"order.a{i}", (n, 3, h, h), seed=77)).cuda(),
                torch.from_numpy(fill.uniform(f"order.b{i}", (n, 3, h, h), seed=77)).cuda()) for i in range(steps)]

    def run(trusted):
        state["trusted"], state["calls"] = trusted, []
        G = pfa_gan.UNetGenerator(3, 3, num_downs=nd, ngf=ngf).cuda()
        D = pfa_gan.NLayerDiscriminator(6, ndf=ndf).cuda()
        load(G, fill.fill_state(nets.g_state_shapes(3, 3, nd, ngf), seed=5, bn_noise=False))
        load(D, fill.fill_state(nets.d_state_shapes(6, ndf, 3), seed=6, bn_noise=False))
        tr = pfa_gan.GanTrainer(G, D, precision="fp32", process_group=object(), overlap=mode != "one-stream",
                                inputs_resident=mode == "resident-inputs",
                                grad_payload="bf16" if mode == "bf16-payload" else None)
        assert tr.world == 2
        torch.cuda.synchronize()
        if mode == "resident-inputs":
            packed = [(tr.pack(a), tr.pack(b)) for a, b in batches]
            torch.cuda.synchronize()
            out = [tr.step_packed(a, b) for a, b in packed]        # no host synchronisation between the steps
        else:
            out = [tr.step(a, b) for a, b in batches]
        losses = torch.stack([l.clone() for l in out]).cpu()
        torch.cuda.synchronize()
        sd = {f"{p}.{k}": v.detach().cpu().clone() for p, m in (("g", G), ("d", D)) for k, v in m.state_dict().items()}
        return losses, sd, list(state["calls"]), (tr.G.flat.total, tr.Dn.flat.total)

    l0, s0, c0, (gtot, dtot) = run(True)
    l1, s1, c1, _ = run(False)
    assert c0 == c1, "both runs issue the same collectives"
    # every gradient element crosses the ranks once per step; the discriminator's two arenas (D(real)'s leaves early, under
    # the generator's forward) travel separately where the step runs on three streams
    assert sum(c[0] for c in c1) in (steps * (gtot + dtot), steps * (gtot + 2 * dtot))
    if mode != "one-stream":
        assert sum(1 for c in c1 if c[1]) >= steps * 3, "the bucketed collectives are asynchronous"
    assert torch.isfinite(l1).all()
    for k in s0:
        assert torch.equal(s0[k], s1[k]), f"{k}: differs from the trivially ordered run by {(s0[k].float() - s1[k].float()).abs().max():.3e}"


@pytest.mark.parametrize("precision,n,h", [("fp32", 1, 128), ("bf16", 1, 256), ("bf16", 3, 128), ("fp32", 2, 256)],
                         ids=lambda v: str(v))
def test_full_size_step_vs_oracle(precision, n, h):
    """Default networks (41.8 M + 2.77 M parameters), one fused step against oracle/step.py on the same weights and data:
    the reference's own batch size 1 (train_gan.py:26), a ragged batch of 3, and 256x256."""
    gst = fill.fill_state(nets.g_state_shapes(), seed=21, bn_noise=False)
    dst = fill.fill_state(nets.d_state_shapes(), seed=22, bn_noise=False)
    a = fill.uniform(f"fs.a.{n}.{h}", (n, 3, h, h))
    b = fill.uniform(f"fs.b.{n}.{h}", (n, 3, h, h))
    G = pfa_gan.UNetGenerator(3, 3).cuda()
    D = pfa_gan.NLayerDiscriminator(6).cuda()
    load(G, gst)
    load(D, dst)
    tr = pfa_gan.GanTrainer(G, D, precision=precision)
    losses = tr.step(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()).cpu().numpy()
    og, od = {k: v.copy() for k, v in gst.items()}, {k: v.copy() for k, v in dst.items()}
    ld, lg, ex = ostep.gan_step(og, od, ostep.new_adam_state(og), ostep.new_adam_state(od), a, b)
    rel = 2e-4 if precision == "fp32" else 2e-2
    assert_close(losses[0], ld, 0, rel, f"loss_d {precision}")
    assert_close(losses[1], lg, 0, rel, f"loss_g {precision}")
    fake = pfa_gan.kernels.unpack_nchw(tr.gctx.v["fake"], 3).cpu().numpy()
    if precision == "fp32":
        assert_close(fake, ex["fake_b"], ABS, REL, "fake_B")
        for k in ("model.model.1.model.2.running_mean", "model.model.1.model.3.model.3.model.6.running_var"):
            assert_close(G.state_dict()[k].cpu().numpy(), og[k], 1e-5, 1e-3, k)
        # Adam's first step moves every parameter by ~lr in the direction of -sign(g): compare the moved weights
        for k in ("model.model.0.weight", "model.model.3.weight", "model.model.1.model.3.model.3.model.1.weight"):
            got = fill.compact(G.state_dict()[k].cpu().numpy())
            want = fill.compact(og[k])
            assert np.mean(np.abs(got - want) > 2e-5) < 0.02, k
    else:
        assert np.abs(fake - ex["fake_b"]).mean() < 2e-2
    assert int(G.state_dict()["model.model.1.model.2.num_batches_tracked"]) == 2
    assert int(D.state_dict()["model.3.num_batches_tracked"]) == 3
    for t in list(G.state_dict().values()) + list(D.state_dict().values()):
        assert torch.isfinite(t.float()).all()


def _torchrun(script_args, env_extra, timeout=600):
    import os
    import subprocess
    import sys
    env = dict(os.environ)
    env.update(env_extra)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import socket
    with socket.socket() as sk:             # a free port per call (a fixed one can still be held by the previous run's store)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args
    return subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout)


@pytest.mark.parametrize("overlap", ["1", "0", "resident", "bf16"],
                         ids=["three-streams", "one-stream", "resident-inputs", "bf16-payload"])
def test_data_parallel_two_ranks_on_device(overlap):
    """Two ranks, device tensors, the HIP engine: parameters stay identical across ranks and equal the single-process
    update from the mean gradient (tests/dp_gpu_worker.py; gloo transport because the box has one GPU).  bf16-payload: the
    gradient ranges cross the ranks as bfloat16 (GanTrainer(grad_payload="bf16"))."""
    import os
    env = {"PFA_DP_OVERLAP": "1", "PFA_DP_RESIDENT": "1"} if overlap == "resident" else \
        {"PFA_DP_OVERLAP": "1", "PFA_DP_PAYLOAD": "bf16"} if overlap == "bf16" else {"PFA_DP_OVERLAP": overlap}
    r = _torchrun([os.path.join(os.path.dirname(__file__), "dp_gpu_worker.py")], env)
    assert r.returncode == 0 and "DP_GPU_WORKER_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("spelling", ["plain", "torchrun"])
def test_bench_two_ranks_contract(spelling):
    """bench.py at two ranks, both ways the driver may start it: plain `python bench.py --gpus 2` (bench.py itself starts
    torch.distributed.run as a child process before touching a GPU) and under `torch.distributed.run --nproc-per-node 2`:
    both ranks finish, rank 0 prints exactly one JSON line with the whole-job rate and the roofline object."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    argv = [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--repeats", "2", "--batch", "2",
            "--size", "128"]
    extra = {"PFA_BENCH_BACKEND": "gloo", "PFA_BENCH_SHARE_GPU": "1"}
    if spelling == "plain":
        env = dict(os.environ, **extra)
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
            env.pop(k, None)
        r = subprocess.run([sys.executable] + argv, env=env, capture_output=True, text=True, timeout=900)
    else:
        r = _torchrun(argv, extra)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 4 and out["scaling"] == "weak"
    assert out["value"] > 0 and abs(out["value"] - 4 * 3 / (out["ms_per_step"] * 3e-3)) < 0.02 * out["value"]
    assert len(out["repeats_ms_per_step"]) == 2
    # (two processes time-slice ONE GPU here: a launch's event pair spans the other rank's slices, so the rate itself is
    # meaningless in this test - the object, its kernel and its per-launch work are what the contract needs)
    rf = out["roofline"]
    assert rf["avg_launch_us"] > 0 and rf["flop_per_launch"] > 0 and rf["achieved"] >= 0 and "cpu_baseline" not in out
    assert any(k["kernel"].startswith("k_wgrad") for k in out["roofline"]["kernels"])
    # N > 1 diagnostics: what the collectives cost the step, every gradient range reduced alone, the bucket layout
    comm = out["config"]["comm"]
    assert comm["ms_per_step_without_collectives"] > 0 and comm["payload"] == "f32" and comm["backend"] == "gloo"
    assert abs(comm["comm_cost_ms_per_step"] - (out["ms_per_step"] - comm["ms_per_step_without_collectives"])) < 2e-3
    names = [p["range"] for p in comm["allreduce_alone"]]
    assert sum(n.startswith("G ") for n in names) >= 3 and "D(real) arena" in names
    assert all(p["ms"] > 0 and p["busbw_gbs"] > 0 and p["mbytes"] > 0 for p in comm["allreduce_alone"])
    g_mb = sum(p["mbytes"] for p in comm["allreduce_alone"] if p["range"].startswith("G "))
    d_mb = [p["mbytes"] for p in comm["allreduce_alone"] if p["range"] == "D(real) arena"][0]
    assert abs(g_mb + 2 * d_mb - comm["bytes_per_step_mb"]) < 0.1            # the generator's buckets tile its arena


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_generate_uint8_equals_denormalised_forward(precision):
    """generate_synthetic_data.py:61-71 on the device: the uint8 image equals (G(x)*0.5+0.5).mul(255).byte() of the same
    eval-mode forward, byte for byte."""
    torch.manual_seed(3)
    G = pfa_gan.UNetGenerator(3, 3, num_downs=5, ngf=8).cuda().set_precision(precision).eval()
    x = torch.rand((2, 3, 64, 32), device="cuda") * 2 - 1
    with torch.no_grad():
        want = (G(x) * 0.5 + 0.5).mul(255).byte().permute(0, 2, 3, 1).cpu()
    got = G.generate_uint8(x).cpu()
    assert got.dtype == torch.uint8 and got.shape == (2, 64, 32, 3)
    assert torch.equal(got, want)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_tile_pipeline_equals_eager_generation(precision):
    """generate_synthetic_data.py:61-71 as the captured-graph, double-buffered pipeline (pfa_gan.TilePipeline): byte for
    byte the eager generate_uint8 of every batch, in order, including a ragged last batch and slot reuse."""
    torch.manual_seed(4)
    G = pfa_gan.UNetGenerator(3, 3, num_downs=5, ngf=8).cuda().set_precision(precision).eval()
    with torch.no_grad():                     # non-trivial running statistics
        for m in G.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.uniform_(-0.2, 0.2)
                m.running_var.uniform_(0.5, 1.5)
    gen = torch.Generator().manual_seed(8)
    batches = [torch.rand((4, 3, 64, 64), generator=gen) * 2 - 1 for _ in range(5)] + [torch.rand((3, 3, 64, 64), generator=gen) * 2 - 1]
    tiles = pfa_gan.TilePipeline(G, 4, 64)
    before = _lib.launch_count()
    got = list(tiles(iter(batches)))
    assert [k for k, _ in got] == list(range(6))
    assert _lib.launch_count() - before < 60, "the fixed-shape batches must be graph replays, not eager launches"
    for (k, u8), x in zip(got, batches):
        want = G.generate_uint8(x.cuda()).cpu().numpy()
        assert u8.dtype == np.uint8 and u8.shape == want.shape
        assert np.array_equal(u8, want), f"batch {k}"
    again = list(tiles(iter(batches[:3])))    # the pipeline is reusable
    assert all(np.array_equal(a[1], b[1]) for a, b in zip(again, got))


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_eval_affine_epilogue_matches_separate_normalisation(precision, monkeypatch):
    """Eval-mode generator (generate_synthetic_data.py:55): the BatchNorm-in-epilogue launches (fuse_out = 3) against
    the same forward with the normalisation as a separate pass (the un-split launches take the fused route)."""
    from pfa_gan import engine as E, kernels as K
    torch.manual_seed(11)
    G = pfa_gan.UNetGenerator(3, 3, num_downs=6, ngf=32).cuda().set_precision(precision)
    with torch.no_grad():                     # non-trivial running statistics and affine parameters
        for m in G.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.uniform_(-0.3, 0.3)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
    G.eval()
    x = torch.rand((8, 3, 128, 128), device="cuda") * 2 - 1
    fused_launches = []
    real = K.gather_gemm

    def spy(*a, **kw):
        if kw.get("affine") is not None:
            fused_launches.append(kw["affine"][2])
        return real(*a, **kw)

    monkeypatch.setattr(E.K, "gather_gemm", spy)
    with torch.no_grad():
        y_fused = G(x).float().cpu()
    assert len(fused_launches) >= 2, "the large (un-split) layers take the affine epilogue"
    monkeypatch.setattr(E._Engine, "_affine_ok", staticmethod(lambda *a: False))
    with torch.no_grad():
        y_plain = G(x).float().cpu()
    tol = 2e-5 if precision == "fp32" else 2e-2
    assert float((y_fused - y_plain).abs().max()) <= tol, float((y_fused - y_plain).abs().max())


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_fused_shadow_launch_equals_per_layer_kernels(precision):
    """pfa_weight_shadow_batch (one launch: 64x64 tiles, both layouts) against pfa_weight_shadow layer by layer, bit for
    bit, on networks with ragged channel counts (3/6-channel image layers, virtual-concat decoder inputs, a 1-channel
    output conv)."""
    from pfa_gan import kernels as K
    torch.manual_seed(5)
    G = pfa_gan.UNetGenerator(3, 3, num_downs=5, ngf=24).cuda().set_precision(precision)
    D = pfa_gan.NLayerDiscriminator(6, ndf=40).cuda().set_precision(precision)
    for eng in (G.engine(), D.engine(in_split=[3, 3])):
        for L in eng.layers:
            L.same.fill_(7.0)
            L.tr.fill_(7.0)
        eng.refresh_shadows(force=True)
        for L in eng.layers:
            (c1r, c1p), (c2r, c2p) = L.cparts[0], (L.cparts[1] if len(L.cparts) > 1 else (0, 0))
            same, tr = torch.full_like(L.same, 5.0), torch.full_like(L.tr, 5.0)
            K.weight_shadow(eng.flat.seg(eng.flat.data, L.w), L.rows, c1r, c1p, c2r, c2p, L.rows_p, same, tr)
            torch.cuda.synchronize()
            assert torch.equal(L.same, same), f"{L.name if hasattr(L, 'name') else L}: same"
            assert torch.equal(L.tr, tr), f"{L.name if hasattr(L, 'name') else L}: tr"


@pytest.mark.parametrize("case", STEP_CASES, ids=[c[0] for c in STEP_CASES])
def test_fused_step_with_resident_inputs_matches_reference(golden, case):
    """GanTrainer(inputs_resident=True): D(real) of step i+1 starts as soon as the discriminator is free in step i (under
    the generator's Adam / shadow refresh) instead of after it.  Same goldens as the default schedule: losses of every
    step (the real-pass loss travels through its own slot) and the final state."""
    name, nd, ngf, ndf, n, h, steps = case
    z = golden("step")
    G = pfa_gan.UNetGenerator(3, 3, num_downs=nd, ngf=ngf).cuda()
    D = pfa_gan.NLayerDiscriminator(6, ndf=ndf).cuda()
    load(G, fill.fill_state(nets.g_state_shapes(3, 3, nd, ngf), seed=5, bn_noise=False))
    load(D, fill.fill_state(nets.d_state_shapes(6, ndf, 3), seed=6, bn_noise=False))
    tr = pfa_gan.GanTrainer(G, D, precision="fp32", inputs_resident=True)
    seed = int(z[f"step.{name}.seed"])
    batches = []
    for i in range(steps):
        a = torch.from_numpy(fill.uniform(f"step.{name}.a{i}", (n, 3, h, h), seed=seed)).cuda()
        b = torch.from_numpy(fill.uniform(f"step.{name}.b{i}", (n, 3, h, h), seed=seed)).cuda()
        batches.append((tr.pack(a), tr.pack(b)))
    torch.cuda.synchronize()
    got = [tr.step_packed(xa, xb).clone() for xa, xb in batches]      # enqueue every step before looking at any result
    torch.cuda.synchronize()
    for i, l in enumerate(got):
        losses = l.cpu().numpy()
        assert_close(losses[0], z[f"step.{name}.loss_d{i}"], 1e-4, 1e-4, f"loss_d step {i}")
        assert_close(losses[1], z[f"step.{name}.loss_g{i}"], 1e-3, 1e-4, f"loss_g step {i}")
    _check_step_state(z, name, G, D)


@pytest.mark.parametrize("case", STEP_CASES, ids=[c[0] for c in STEP_CASES])
def test_epoch_loop_with_input_pipeline_matches_reference(golden, case):
    """pfa_gan.train_gan_one_epoch(G, D, loader, FusedAdam, FusedAdam) - the drop-in for train_gan.py:46-75 - over a
    loader of CPU batches: the next batch is staged (H2D + pack) on its own stream while a step runs.  One call over all
    the golden batches: mean losses (train_gan.py:75) and the final state."""
    name, nd, ngf, ndf, n, h, steps = case
    z = golden("step")
    G = pfa_gan.UNetGenerator(3, 3, num_downs=nd, ngf=ngf).cuda().set_precision("fp32")
    D = pfa_gan.NLayerDiscriminator(6, ndf=ndf).cuda().set_precision("fp32")
    load(G, fill.fill_state(nets.g_state_shapes(3, 3, nd, ngf), seed=5, bn_noise=False))
    load(D, fill.fill_state(nets.d_state_shapes(6, ndf, 3), seed=6, bn_noise=False))
    og, od = pfa_gan.FusedAdam(G.parameters()), pfa_gan.FusedAdam(D.parameters())
    seed = int(z[f"step.{name}.seed"])
    loader = [dict(image1=torch.from_numpy(fill.uniform(f"step.{name}.a{i}", (n, 3, h, h), seed=seed)).pin_memory(),
                   image2=torch.from_numpy(fill.uniform(f"step.{name}.b{i}", (n, 3, h, h), seed=seed)).pin_memory())
              for i in range(steps)]
    loader.insert(1, None)          # a batch the collate function dropped (collate_fn_skip_none)
    ld, lg = pfa_gan.train_gan_one_epoch(G, D, loader, og, od)
    want_d = np.mean([float(z[f"step.{name}.loss_d{i}"]) for i in range(steps)])
    want_g = np.mean([float(z[f"step.{name}.loss_g{i}"]) for i in range(steps)])
    assert_close(ld, want_d, 1e-4, 1e-4, "mean loss_d")
    assert_close(lg, want_g, 1e-3, 1e-4, "mean loss_g")
    _check_step_state(z, name, G, D)


# ---------------------------------------------------------------------------- non-default constructor options
def _fill_for(module, seed):
    return fill.fill_state({k: tuple(v.shape) for k, v in module.state_dict().items()}, seed=seed)


def _variant_grad_ok(got, want, what):
    """As grad_close; a gradient that is exactly zero in exact arithmetic (a bias in front of an InstanceNorm) only has
    to be round-off."""
    want = np.asarray(want, np.float64)
    if np.abs(want).max() < 1e-4:
        assert np.abs(np.asarray(got, np.float64)).max() < 1e-3, what
        return
    grad_close(got, want, what, rel_l2=5e-3)


def test_instance_norm_generator_layerwise_vs_reference(golden):
    """UNetGenerator(norm_layer=nn.InstanceNorm2d) (models.py:151,171-174) on the GPU: the module graph runs with every
    convolution on libpfa_hip (pfa_gan.layerwise); train forward and all parameter gradients against the reference."""
    z = golden("variants")
    G = pfa_gan.UNetGenerator(3, 3, num_downs=5, ngf=8, norm_layer=torch.nn.InstanceNorm2d).cuda().set_precision("fp32")
    assert not G.fused()
    load(G, _fill_for(G, 21))
    seed = int(z["var.g_in.seed"])
    x = torch.from_numpy(fill.uniform("var.g_in.x", (2, 3, 64, 64), seed=seed)).cuda()
    G.train()
    before = _lib.launch_count()
    y = G(x)
    assert _lib.launch_count() - before >= 2 * 10, "every convolution must be a libpfa_hip launch"
    assert_close(y.detach().cpu().numpy(), z["var.g_in.y"], ABS, REL, "IN generator forward")
    y.backward(torch.from_numpy(fill.uniform("var.g_in.dy", tuple(y.shape), seed=seed)).cuda())
    for k, p in G.named_parameters():
        _variant_grad_ok(fill.compact(p.grad.cpu().numpy()), z[f"var.g_in.grad.{k}"], f"IN G grad {k}")


def test_instance_norm_discriminator_layerwise_vs_reference(golden):
    z = golden("variants")
    D = pfa_gan.NLayerDiscriminator(6, ndf=8, norm_layer=torch.nn.InstanceNorm2d).cuda().set_precision("fp32")
    load(D, _fill_for(D, 22))
    seed = int(z["var.d_in.seed"])
    x = torch.from_numpy(fill.uniform("var.d_in.x", (2, 6, 64, 64), seed=seed)).cuda().requires_grad_(True)
    D.train()
    y = D(x)
    assert_close(y.detach().cpu().numpy(), z["var.d_in.y"], ABS, REL, "IN discriminator forward")
    y.backward(torch.from_numpy(fill.uniform("var.d_in.dy", tuple(y.shape), seed=seed)).cuda())
    grad_close(x.grad.cpu().numpy(), z["var.d_in.dx"], "IN D input gradient", rel_l2=5e-3)
    for k, p in D.named_parameters():
        _variant_grad_ok(fill.compact(p.grad.cpu().numpy()), z[f"var.d_in.grad.{k}"], f"IN D grad {k}")


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_dropout_generator_layerwise(golden, precision):
    """UNetGenerator(use_dropout=True) (models.py:197-198): same keys as the reference; eval mode (Dropout = identity)
    matches it; train mode runs, drops activations (two forwards differ) and back-propagates finite gradients."""
    z = golden("variants")
    G = pfa_gan.UNetGenerator(3, 3, num_downs=6, ngf=8, use_dropout=True).cuda().set_precision(precision)
    assert list(G.state_dict().keys()) == str(z["var.g_do.keys"]).split("\n") and not G.fused()
    load(G, _fill_for(G, 23))
    x = torch.from_numpy(fill.uniform("var.g_do.x", (2, 3, 64, 64))).cuda()
    G.eval()
    with torch.no_grad():
        ye = G(x).cpu().numpy()
    tol = (ABS, REL) if precision == "fp32" else (3e-2, 3e-2)
    assert_close(ye, z["var.g_do.y_eval"], *tol, "dropout generator, eval forward")
    G.train()
    y1, y2 = G(x), G(x)
    assert float((y1 - y2).detach().abs().max()) > 1e-3, "training-mode dropout must change the output between calls"
    y1.square().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in G.parameters())


# ---------------------------------------------------------------------------- BASELINE config 2 (B=16, 256x256) by properties
def _one_step(precision, a, b, gst, dst):
    G = pfa_gan.UNetGenerator(3, 3).cuda()
    D = pfa_gan.NLayerDiscriminator(6).cuda()
    load(G, gst)
    load(D, dst)
    tr = pfa_gan.GanTrainer(G, D, precision=precision)
    losses = tr.step(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()).cpu().numpy()
    fake = pfa_gan.kernels.unpack_nchw(tr.gctx.v["fake"], 3).cpu().numpy()
    return losses, fake, {k: v.cpu().numpy() for k, v in G.state_dict().items()}, \
        {k: v.cpu().numpy() for k, v in D.state_dict().items()}


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_config2_step_on_16_copies_equals_step_on_one_pair(precision):
    """The oracle cannot run B=16 at 256x256 in test time, so the bench configuration is pinned by a size-independent
    property: a batch made of 16 copies of one pair has the batch statistics, mean losses and mean gradients of that pair
    alone, hence the same fake image, losses and Adam update as the B=1 step (which test_full_size_step_vs_oracle checks
    against the oracle) - while running the launch configurations of the bench (256x128 ring tile, K slices, work-list
    orders, slab reductions of the weight gradients).  Only running_var differs (unbiased factor n/(n-1))."""
    gst = fill.fill_state(nets.g_state_shapes(), seed=31, bn_noise=False)
    dst = fill.fill_state(nets.d_state_shapes(), seed=32, bn_noise=False)
    a1 = fill.uniform("copies.a", (1, 3, 256, 256))
    b1 = fill.uniform("copies.b", (1, 3, 256, 256))
    l1, f1, g1, d1 = _one_step(precision, a1, b1, gst, dst)
    l16, f16, g16, d16 = _one_step(precision, np.repeat(a1, 16, 0), np.repeat(b1, 16, 0), gst, dst)
    rel = 1e-4 if precision == "fp32" else 1e-2
    assert_close(l16[0], l1[0], 0, rel, "loss_d")
    assert_close(l16[1], l1[1], 0, rel, "loss_g")
    assert np.abs(f16 - f16[:1]).max() == 0.0, "the 16 copies must produce one and the same fake image"
    if precision == "fp32":
        assert_close(f16[:1], f1, ABS, REL, "fake_B")
    else:
        assert np.abs(f16[:1] - f1).mean() < 1e-2
    lr = 1e-4
    for got, want, who in ((g16, g1, "G"), (d16, d1, "D")):
        for k, v in want.items():
            if k.endswith("num_batches_tracked"):
                assert int(got[k]) == int(v), k
            elif k.endswith("running_var"):
                continue
            elif k.endswith("running_mean"):
                assert_close(got[k], v, 1e-4 if precision == "fp32" else 2e-2, 1e-2, f"{who} {k}")
            else:
                # Adam's first step moves every parameter by lr * g / (|g| + eps): equal up to the elements whose
                # gradient is within rounding of zero
                off = np.mean(np.abs(got[k] - v) > 0.2 * lr)
                assert off < (0.02 if precision == "fp32" else 0.10), f"{who} {k}: {off:.2%} of the parameters moved differently"


def test_config2_step_with_the_meeting_finish_equals_step_on_one_pair(monkeypatch):
    """PFA_COOP=1 (kernels.COOP): the generator's 8x8 and 16x16 levels finish their K slices in one launch whose workgroups
    meet inside it (csrc/finish_coop.hip), forward and backward.  The bench configuration with it on, pinned like the default
    route: 16 copies of one pair against that pair's B = 1 step (which the oracle test covers)."""
    gst = fill.fill_state(nets.g_state_shapes(), seed=31, bn_noise=False)
    dst = fill.fill_state(nets.d_state_shapes(), seed=32, bn_noise=False)
    a1 = fill.uniform("copies.a", (1, 3, 256, 256))
    b1 = fill.uniform("copies.b", (1, 3, 256, 256))
    l1, f1, g1, d1 = _one_step("bf16", a1, b1, gst, dst)
    monkeypatch.setattr(pfa_gan.kernels, "COOP", True)
    before = _lib.launch_count()
    l16, f16, g16, d16 = _one_step("bf16", np.repeat(a1, 16, 0), np.repeat(b1, 16, 0), gst, dst)
    with_coop = _lib.launch_count() - before
    monkeypatch.setattr(pfa_gan.kernels, "COOP", False)
    before = _lib.launch_count()
    m16 = _one_step("bf16", np.repeat(a1, 16, 0), np.repeat(b1, 16, 0), gst, dst)
    without = _lib.launch_count() - before
    assert with_coop <= without - 6, (with_coop, without)        # the launch pairs it folds (four forward, at least two backward)
    assert_close(l16[0], l1[0], 0, 1e-2, "loss_d")
    assert_close(l16[1], l1[1], 0, 1e-2, "loss_g")
    assert_close(l16[0], m16[0][0], 0, 2e-3, "loss_d vs the two-launch route")
    assert_close(l16[1], m16[0][1], 0, 2e-3, "loss_g vs the two-launch route")
    assert np.abs(f16 - f16[:1]).max() == 0.0, "the 16 copies must produce one and the same fake image"
    assert np.abs(f16[:1] - f1).mean() < 1e-2
    lr = 1e-4
    for got, want, who in ((g16, g1, "G"), (d16, d1, "D")):
        for k, v in want.items():
            if k.endswith("num_batches_tracked"):
                assert int(got[k]) == int(v), k
            elif k.endswith("running_var"):
                continue
            elif k.endswith("running_mean"):
                assert_close(got[k], v, 2e-2, 1e-2, f"{who} {k}")
            else:
                off = np.mean(np.abs(got[k] - v) > 0.2 * lr)
                assert off < 0.10, f"{who} {k}: {off:.2%} of the parameters moved differently"


def test_512_forward_fp32_vs_reference(golden):
    """BASELINE config 4's tile size: default networks at 512x512 against the reference's forward (tests/golden/full512.npz):
    every map twice as wide as at 256x256, the discriminator's logit map 62x62 (stride-1 patch kernel tiled in x)."""
    z = golden("full512")
    G = pfa_gan.UNetGenerator(3, 3).cuda()
    load(G, fill.fill_state(nets.g_state_shapes(), seed=3))
    x = torch.from_numpy(fill.uniform("full512.x", (1, 3, 512, 512))).cuda()
    G.train()
    with torch.no_grad():
        y = G(x).cpu().numpy()
    _digest_ok(z, "full512.g.y", y, ABS, REL)
    assert_close(y.astype(np.float64).mean(axis=(0, 2, 3)), z["full512.g.y_chan_mean"], 1e-4, 1e-3)
    D = pfa_gan.NLayerDiscriminator(6).cuda()
    load(D, fill.fill_state(nets.d_state_shapes(), seed=4))
    xd = torch.from_numpy(fill.uniform("full512.xd", (1, 6, 512, 512))).cuda()
    D.train()
    with torch.no_grad():
        yd = D(xd).cpu().numpy()
    assert yd.shape == (1, 1, 62, 62)
    assert_close(yd, z["full512.d.y"], ABS, REL, "D logits 512")


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_config4_step_on_8_copies_equals_step_on_one_pair(precision):
    """BASELINE config 4 (512x512, 8 pairs per GPU) by the replicated-batch property of the config-2 test: a step on 8
    copies of one 512x512 pair must give the fake image, losses, BatchNorm means and Adam update of the B=1 step, and
    runs the launch configurations of that shape (4x the rows per layer of 256x256, 62x62 logits)."""
    gst = fill.fill_state(nets.g_state_shapes(), seed=41, bn_noise=False)
    dst = fill.fill_state(nets.d_state_shapes(), seed=42, bn_noise=False)
    a1 = fill.uniform("copies512.a", (1, 3, 512, 512))
    b1 = fill.uniform("copies512.b", (1, 3, 512, 512))
    l1, f1, g1, d1 = _one_step(precision, a1, b1, gst, dst)
    l8, f8, g8, d8 = _one_step(precision, np.repeat(a1, 8, 0), np.repeat(b1, 8, 0), gst, dst)
    rel = 1e-4 if precision == "fp32" else 1e-2
    assert_close(l8[0], l1[0], 0, rel, "loss_d")
    assert_close(l8[1], l1[1], 0, rel, "loss_g")
    assert np.abs(f8 - f8[:1]).max() == 0.0, "the 8 copies must produce one and the same fake image"
    if precision == "fp32":
        assert_close(f8[:1], f1, ABS, REL, "fake_B")
    else:
        assert np.abs(f8[:1] - f1).mean() < 1e-2
    lr = 1e-4
    for got, want, who in ((g8, g1, "G"), (d8, d1, "D")):
        for k, v in want.items():
            if k.endswith("num_batches_tracked"):
                assert int(got[k]) == int(v), k
            elif k.endswith("running_var"):
                continue
            elif k.endswith("running_mean"):
                assert_close(got[k], v, 1e-4 if precision == "fp32" else 2e-2, 1e-2, f"{who} {k}")
            else:
                off = np.mean(np.abs(got[k] - v) > 0.2 * lr)
                assert off < (0.02 if precision == "fp32" else 0.10), f"{who} {k}: {off:.2%} of the parameters moved differently"


def test_fused_adam_counts_one_step_sequence_across_device_and_host_counters():
    """FusedAdam keeps ONE step count: launches that read it from device memory (graph_safe=True: what a captured hipGraph
    replays) and eager launches that pass it by value must see the same sequence 1, 2, 3, ... whatever the mix.  Against
    torch.optim.Adam on the same gradients (the reference's optimizer, train_gan.py:140-141)."""
    torch.manual_seed(3)
    net = pfa_gan.NLayerDiscriminator(6, ndf=8).cuda()
    net.engine("fp32")
    ref = [p.detach().clone().contiguous().requires_grad_(True) for p in net.parameters()]
    opt = pfa_gan.FusedAdam(net.parameters())
    topt = torch.optim.Adam(ref, lr=1e-4, betas=(0.5, 0.999))
    gen = torch.Generator(device="cuda").manual_seed(5)
    for graph_safe in (True, False, True, True, False, True):      # e.g. replay, ragged eager batch, replay, ...
        for p, r in zip(net.parameters(), ref):
            g = torch.randn(r.shape, device="cuda", generator=gen)
            p.grad.copy_(g)
            r.grad = g.clone()
        opt.step(graph_safe=graph_safe)
        topt.step()
    assert opt.steps == 6 and int(opt.step_dev.item()) <= 6
    for p, r in zip(net.parameters(), ref):
        # a count off by one at step 3 scales that update by ~5 %: 5e-6 of the 1e-4 step
        assert float((p.detach() - r.detach()).abs().max()) < 2e-7


def test_graph_replays_and_ragged_eager_steps_share_one_trainer():
    """A batch of another shape between two hipGraph replays runs eagerly (the short batch collate_fn_skip_none yields when
    a sample fails to load, dataset.py:13-18) on buffers of its own - the captured graph keeps its buffers - and with the
    host step count; the next replay reads the same count from the device.  Graph, ragged eager, graph, graph against
    the same four steps run eagerly: equal up to the parameters whose gradient is rounding noise (Adam turns those into
    +-lr moves), like the other whole-step comparisons of this file."""
    def run(use_graph):
        torch.manual_seed(9)
        G = pfa_gan.UNetGenerator(3, 3, num_downs=5, ngf=8).cuda()
        D = pfa_gan.NLayerDiscriminator(6, ndf=8).cuda()
        tr = pfa_gan.GanTrainer(G, D, precision="fp32", use_graph=use_graph, overlap=False)
        gen = torch.Generator(device="cuda").manual_seed(77)
        out = []
        for n in (2, 1, 2, 2):
            a = torch.rand((n, 3, 64, 64), device="cuda", generator=gen) * 2 - 1
            b = torch.rand((n, 3, 64, 64), device="cuda", generator=gen) * 2 - 1
            losses = tr.step(a, b).clone()
            out.append((losses, tr.opt_g.flat().data.clone(), tr.opt_d.flat().data.clone()))
        torch.cuda.synchronize()
        assert tr.opt_g.steps == 4 and tr.opt_d.steps == 4
        assert int(tr.opt_g.step_dev.item()) == (4 if use_graph else 0)
        return out

    eager, graph = run(False), run(True)
    lr = 1e-4
    for i, ((le, ge, de), (lg, gg, dg)) in enumerate(zip(eager, graph)):
        assert_close(lg.cpu().numpy(), le.cpu().numpy(), 1e-4, 1e-4, f"losses of step {i}")
        for e, g, who in ((ge, gg, "G"), (de, dg, "D")):
            d = (e - g).abs()
            assert float(d.median()) < 0.002 * lr, f"{who} parameters after step {i}: median {float(d.median()):.2e}"
            assert float((d > 0.02 * lr).float().mean()) < 0.02, f"{who} parameters after step {i}"


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_config5_eval_batch_equals_per_tile_forwards(precision):
    """Eval-mode BatchNorm couples nothing across a batch: a 16-tile generator forward (the inference configuration, large-M
    launches) must reproduce the single-tile forwards, which the goldens pin."""
    G = pfa_gan.UNetGenerator(3, 3).cuda()
    load(G, fill.fill_state(nets.g_state_shapes(), seed=33))
    G.set_precision(precision)
    G.eval()
    x = torch.from_numpy(fill.uniform("tiles.x", (16, 3, 256, 256))).cuda()
    with torch.no_grad():
        y = G(x)
        for i in (0, 7, 15):
            yi = G(x[i:i + 1])
            if precision == "fp32":
                assert_close(y[i:i + 1].cpu().numpy(), yi.cpu().numpy(), 1e-5, 1e-4, f"tile {i}")
            else:
                assert float((y[i:i + 1] - yi).abs().mean()) < 2e-3, f"tile {i}"


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_eval_caches_follow_parameter_and_buffer_writes(precision):
    """Eval-mode forwards reuse the kernel-side weight copies and the folded BatchNorm coefficients until something
    writes the masters: torch in-place ops (version counters), load_state_dict, the library's own train-mode forward
    (running statistics) and fused Adam (parameters, written behind torch's back).  After each kind of write the eval
    output must equal that of a fresh module built from the current state_dict."""
    def fresh_eval(state, x):
        m = pfa_gan.UNetGenerator(3, 3, num_downs=5, ngf=8).cuda().set_precision(precision)
        m.load_state_dict(state)
        m.eval()
        with torch.no_grad():
            return m(x)

    torch.manual_seed(5)
    G = pfa_gan.UNetGenerator(3, 3, num_downs=5, ngf=8).cuda().set_precision(precision)
    D = pfa_gan.NLayerDiscriminator(6, ndf=8).cuda().set_precision(precision)
    x = torch.rand(2, 3, 64, 64, device="cuda") * 2 - 1
    b = torch.rand(2, 3, 64, 64, device="cuda") * 2 - 1
    G.eval()
    with torch.no_grad():
        y0 = G(x)
        launches = _lib.launch_count()
        y0b = G(x)
        per_forward = _lib.launch_count() - launches
        assert torch.equal(y0, y0b)
        launches = _lib.launch_count()
        G(x)
        assert _lib.launch_count() - launches == per_forward
        # 1. a torch in-place write to a buffer and to a parameter
        bn = [m for m in G.modules() if isinstance(m, torch.nn.BatchNorm2d)][1]
        bn.running_mean.add_(0.25)
        conv = [m for m in G.modules() if isinstance(m, torch.nn.Conv2d)][1]
        conv.weight.mul_(1.5)
        y1 = G(x)
        assert not torch.equal(y1, y0)
        assert torch.equal(y1, fresh_eval(G.state_dict(), x))
    # 2. a fused training step (running statistics and parameters written by the library's kernels)
    G.train()
    tr = pfa_gan.GanTrainer(G, D, precision=precision)
    tr.step(x, b)
    G.eval()
    with torch.no_grad():
        y2 = G(x)
        assert not torch.equal(y2, y1)
        assert torch.equal(y2, fresh_eval(G.state_dict(), x))
    # 3. a train-mode module forward (running statistics move) followed by FusedAdam on torch autograd gradients
    G.train()
    opt = pfa_gan.FusedAdam(G.parameters(), lr=1e-3)
    G(x).abs().mean().backward()
    opt.step()
    G.eval()
    with torch.no_grad():
        y3 = G(x)
        assert not torch.equal(y3, y2)
        assert torch.equal(y3, fresh_eval(G.state_dict(), x))
        # 4. load_state_dict back to an earlier state
        sd = {k: v.clone() for k, v in G.state_dict().items()}
        G.load_state_dict(sd)
        assert torch.equal(G(x), y3)


@pytest.mark.parametrize("n_layers", [1, 2, 3])
def test_discriminator_gradient_ranges_tile_the_arena(n_layers):
    """Data parallel: the discriminator's gradients cross the ranks as [grad_bucket, total) and [0, grad_bucket) - together
    exactly the arena, whatever the depth (models.py:214 n_layers), and the early range holds the two innermost convolutions."""
    D = pfa_gan.NLayerDiscriminator(6, ndf=8, n_layers=n_layers).cuda()
    eng = D.engine(torch.bfloat16, in_split=(3, 3))
    lo, level = eng.grad_bucket()
    total = eng.flat.total
    assert 0 < lo < total and level == len(eng.layers) - 2
    inner = [eng.flat.slots[id(L.w)] for L in eng.layers[-2:]]
    assert all(lo <= o and o + n <= total for o, n in inner)
    assert all(o + n <= lo for o, n in (eng.flat.slots[id(L.w)] for L in eng.layers[:-2]))
    assert lo % 4 == 0                                # 16-byte aligned: a range Adam and the collective can take


def test_eager_contexts_free_replaced_buffers_and_pinned_ones_shelve_them():
    """engine.Ctx: an eager context (eval forwards on tiles of changing size: generate_uint8's scratch context) hands a
    replaced buffer back to the allocator - memory stays bounded however often two shapes alternate; a pinned context (a
    hipGraph / launch list holds its addresses, or side streams use it) keeps one buffer per shape and takes it up again."""
    torch.manual_seed(0)
    G = pfa_gan.UNetGenerator(3, 3, num_downs=5, ngf=8).cuda().eval()
    xs = [torch.rand((2, 3, 64, 64), device="cuda") * 2 - 1, torch.rand((1, 3, 96, 32), device="cuda") * 2 - 1]
    ref = [G.generate_uint8(x).clone() for x in xs]
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    for i in range(12):
        out = G.generate_uint8(xs[i % 2])
        assert torch.equal(out, ref[i % 2])
    torch.cuda.synchronize()
    del out
    assert torch.cuda.memory_allocated() <= base + (1 << 20), "an eager context must not accumulate replaced buffers"
    ctx = G.engine().new_ctx()
    ctx.pin()
    a = ctx.buf("t", (4, 8), torch.float32)
    b = ctx.buf("t", (2, 8), torch.float32)
    assert ctx.buf("t", (4, 8), torch.float32).data_ptr() == a.data_ptr()      # the shelved buffer of that shape
    assert ctx.buf("t", (2, 8), torch.float32).data_ptr() == b.data_ptr()
    assert len(ctx._shelf) == 1


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_adam_and_weight_copies_in_one_pass_equal_the_two_launches(precision):
    """pfa_adam_shadow_batch (optim.Adam of train_gan.py:140-141 applied by the launch that emits the kernel-side weight
    copies, BatchNorm parameters and biases by a small launch of their own) against pfa_adam + pfa_weight_shadow_batch on
    the same state: parameters, both moments and every layer's two copies bit for bit - whole arena, two sub-ranges, the
    discriminator's two gradient arenas - over two steps."""
    def build():
        torch.manual_seed(11)
        G = pfa_gan.UNetGenerator(3, 3, num_downs=5, ngf=8).cuda()
        D = pfa_gan.NLayerDiscriminator(6, ndf=8).cuda()
        eg, ed = G.engine(precision), D.engine(precision, in_split=(3, 3))
        og, od = pfa_gan.FusedAdam(G.parameters()), pfa_gan.FusedAdam(D.parameters())
        for o in (og, od):
            o.flat()
        gen = torch.Generator(device="cuda").manual_seed(5)

        def real(flat):          # 1 on parameter elements, 0 on the alignment padding between them (gradients there are zero)
            m = torch.zeros(flat.total, device="cuda")
            for o, n in flat.slots.values():
                m[o:o + n] = 1
            return m
        og.flat().grad.copy_(torch.randn(og.flat().total, device="cuda", generator=gen) * 1e-2 * real(og.flat()))
        od.flat().grad.copy_(torch.randn(od.flat().total, device="cuda", generator=gen) * 1e-2 * real(od.flat()))
        od.grad2().copy_(torch.randn(od.flat().total, device="cuda", generator=gen) * 1e-2 * real(od.flat()))
        return eg, ed, og, od
    eg1, ed1, og1, od1 = build()
    eg2, ed2, og2, od2 = build()
    buckets, _ = eg1.grad_buckets()
    for step in range(2):
        # generator: bucket by bucket, as the trainer does it
        og1.step_begin()
        og2.step_begin()
        for lo, hi in buckets:
            eg1.adam_refresh_range(og1, lo, hi, grad_scale=0.5)
            og2.step_range(lo, hi, grad_scale=0.5)
            eg2.refresh_shadows_range(lo, hi)
        # discriminator: whole arena, two gradient arenas
        od1.step_begin()
        ed1.adam_refresh_range(od1, 0, od1.flat().total, use_grad2=True)
        od2.step(use_grad2=True)
        ed2.refresh_shadows(force=True)
        torch.cuda.synchronize()
        for (e1, o1), (e2, o2) in (((eg1, og1), (eg2, og2)), ((ed1, od1), (ed2, od2))):
            assert torch.equal(o1.flat().data, o2.flat().data), f"step {step}: parameters differ"
            assert torch.equal(o1.m, o2.m) and torch.equal(o1.v, o2.v), f"step {step}: moments differ"
            assert o1.steps == o2.steps == step + 1
            for L1, L2 in zip(e1.layers, e2.layers):
                assert torch.equal(L1.same, L2.same) and torch.equal(L1.tr, L2.tr), f"step {step}: copies of {L1.name} differ"
    assert float((og1.flat().data - build()[2].flat().data).abs().max()) > 0        # (the update did something)


def test_launch_list_records_calls_events_and_replays():
    """pfa_replay: library calls on two streams with an event between them, recorded once (they also run), re-issued by one
    C call; a struct argument is deep-copied; a failing op reports its index."""
    from pfa_gan import kernels as K
    main, side = torch.cuda.current_stream(), torch.cuda.Stream()
    x = torch.arange(1024, dtype=torch.float32, device="cuda")
    y = torch.zeros(1024, dtype=torch.float32, device="cuda")
    z = torch.zeros(1024, dtype=torch.float32, device="cuda")
    ev = torch.cuda.Event()
    before = _lib.launch_count()
    with _lib.Recorder() as rec:
        K.wait_stream(side, main)
        with torch.cuda.stream(side):
            K.fill_zero(y)
            K.axpy(y, x, 2.0)                    # y = 2x on the side stream
            K.record_event(ev, side)
        K.wait_event(main, ev)
        K.axpy(z, y, 0.5)                        # z += x on the main stream, after the side stream's work
        assert K.ch_of(torch.bfloat16) == 8      # (a non-launching call passes through)
    ll = rec.finish()
    assert ll.calls == 3 and ll.n == 3 + 4       # fill, 2 axpy; wait_stream = record + wait, record, wait
    torch.cuda.synchronize()
    assert torch.equal(z, x)                     # the recording run executed
    for _ in range(3):
        ll.replay()
    torch.cuda.synchronize()
    assert torch.equal(z, 4 * x) and torch.equal(y, 2 * x)
    assert _lib.launch_count() - before == 2 * 4         # (fill_zero is a memset, not a kernel launch)
    # a bad op: the list stops there and says where
    bad = _lib.LaunchList([_lib.ReplayOp(_lib.OP_CALL, 9999, 1, None)], [])
    with pytest.raises(_lib.PfaError, match="op 0"):
        bad.replay()


def test_launch_lists_take_the_batches_own_buffers():
    """GanTrainer(use_graph="replay"): one recorded launch list serves every batch of the recorded shape - the input
    addresses in its argument blocks are rewritten per step (LaunchList.replay(subst=)), nothing is copied into static
    buffers - and computes exactly what the eager three-stream step does on the same sequence of batches."""
    def run(mode):
        torch.manual_seed(3)
        G = pfa_gan.UNetGenerator(3, 3, num_downs=5, ngf=8).cuda()
        D = pfa_gan.NLayerDiscriminator(6, ndf=8).cuda()
        tr = pfa_gan.GanTrainer(G, D, precision="bf16", inputs_resident=True, use_graph=mode)
        g = torch.Generator(device="cuda").manual_seed(7)
        pool = []
        for _ in range(3):
            a = torch.rand((2, 3, 64, 64), device="cuda", generator=g) * 2 - 1
            b = torch.rand((2, 3, 64, 64), device="cuda", generator=g) * 2 - 1
            pool.append((tr.pack(a), tr.pack(b)))
        torch.cuda.synchronize()
        out = [tr.step_packed(*pool[i % 3]).clone() for i in range(7)]
        torch.cuda.synchronize()
        return tr, pool, torch.stack(out), tr.opt_g.flat().data.clone(), tr.opt_d.flat().data.clone()
    tr, pool, l_list, pg_list, pd_list = run("replay")
    _, _, l_eager, pg_eager, pd_eager = run(False)
    assert torch.equal(l_list, l_eager) and torch.equal(pg_list, pg_eager) and torch.equal(pd_list, pd_eager)
    assert len({p[0].data_ptr() for p in pool}) == 3                       # three distinct input buffers were stepped on
    sa, sb = tr._static
    assert tr._graphs[0].sites(sa.data_ptr()) >= 3 and tr._graphs[0].sites(sb.data_ptr()) >= 1     # D(real), D(fake), G read A
    assert tr._graphs[1].sites(sa.data_ptr()) >= 1 and tr._graphs[1].sites(sb.data_ptr()) >= 1     # G-phase D pass; the L1 term
    # the static buffers were only ever read by the recording run: a replay that still used them would have trained on
    # batch 0 seven times - different losses from the eager run's (checked above) - and poisoning them changes nothing now
    sa.fill_(float("nan")); sb.fill_(float("nan"))
    l8 = tr.step_packed(*pool[1]).clone()
    torch.cuda.synchronize()
    assert torch.isfinite(l8).all()


def test_launch_lists_fall_back_to_the_static_inputs_behind_a_substituted_batch():
    """A batch the lists cannot take as it is (not contiguous) is copied into the static inputs, and the replay must read
    THEM - not the buffers the previous, substituted replay left in the argument blocks (round-3 advisor finding): contiguous,
    strided, contiguous batches give what the eager step gives on the same tensors."""
    def run(mode):
        torch.manual_seed(5)
        G = pfa_gan.UNetGenerator(3, 3, num_downs=5, ngf=8).cuda()
        D = pfa_gan.NLayerDiscriminator(6, ndf=8).cuda()
        tr = pfa_gan.GanTrainer(G, D, precision="bf16", inputs_resident=True, use_graph=mode)
        g = torch.Generator(device="cuda").manual_seed(11)
        packs = []
        for _ in range(3):
            a = torch.rand((2, 3, 64, 64), device="cuda", generator=g) * 2 - 1
            b = torch.rand((2, 3, 64, 64), device="cuda", generator=g) * 2 - 1
            packs.append((tr.pack(a), tr.pack(b)))
        # the middle batch lives in every second row of a larger buffer: same values, not contiguous
        wide = [torch.empty((x.shape[0] * 2,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device) for x in packs[1]]
        for w, x in zip(wide, packs[1]):
            w[::2].copy_(x)
        seq = [packs[0], (wide[0][::2], wide[1][::2]), packs[2], packs[0]]
        assert not seq[1][0].is_contiguous() and torch.equal(seq[1][0], packs[1][0])
        torch.cuda.synchronize()
        out = [tr.step_packed(*xs).clone() for xs in seq]
        torch.cuda.synchronize()
        return torch.stack(out), tr.opt_g.flat().data.clone(), tr.opt_d.flat().data.clone()
    l_list, pg_list, pd_list = run("replay")
    l_eager, pg_eager, pd_eager = run(False)
    assert torch.equal(l_list, l_eager) and torch.equal(pg_list, pg_eager) and torch.equal(pd_list, pd_eager)


def test_step_marks_follow_the_phases_of_the_step():
    """GanTrainer.marks (tools/step_marks.py): timing events at the phase boundaries of the three-stream step - one set per
    step, in program order on the main stream, D(real) complete before the discriminator's gradients are - and the marked
    steps compute what unmarked steps do."""
    def run(marked):
        torch.manual_seed(3)
        G = pfa_gan.UNetGenerator(3, 3, num_downs=5, ngf=8).cuda()
        D = pfa_gan.NLayerDiscriminator(6, ndf=8).cuda()
        tr = pfa_gan.GanTrainer(G, D, precision="bf16", inputs_resident=True)
        g = torch.Generator(device="cuda").manual_seed(7)
        a = torch.rand((2, 3, 64, 64), device="cuda", generator=g) * 2 - 1
        b = torch.rand((2, 3, 64, 64), device="cuda", generator=g) * 2 - 1
        xa, xb = tr.pack(a), tr.pack(b)
        torch.cuda.synchronize()
        if marked:
            tr.marks = []
        out = [tr.step_packed(xa, xb).clone() for _ in range(3)]
        torch.cuda.synchronize()
        return tr, torch.stack(out), tr.opt_g.flat().data.clone()
    tr, marked, pm = run(True)
    _, plain, pp = run(False)
    assert torch.equal(pm, pp) and torch.equal(marked, plain)         # trained state and loss scalars: the same bits
    names = [n for n, _ in tr.marks]
    per = len(names) // 3
    assert len(names) == 3 * per and names[:per] == names[per:2 * per] == names[2 * per:]
    assert names[0] == "step" and names[per - 1].startswith("G gradients")
    at = {n: e for n, e in tr.marks[per:2 * per]}                      # the second step
    main = [n for n in names[:per] if not n.startswith("side:")]
    t = [at["step"].elapsed_time(at[n]) for n in main]
    assert all(t[i] <= t[i + 1] for i in range(len(t) - 1)), list(zip(main, t))
    assert at["side: D(real) backward done"].elapsed_time(at["D gradients complete"]) >= 0
