"""Pins the numpy oracle (oracle/) against outputs of the reference itself (tests/golden).

CPU-only.  The fixtures were produced by tests/make_golden.py importing
/root/reference/models.py and train_gan.py; tolerances here are fp32 round-off
(torch CPU/oneDNN vs numpy/BLAS summation order), far inside the 1e-3 abs / 1e-2 rel
parity gate of the north star.
"""
import numpy as np
import pytest

from conftest import assert_close
from oracle import fill, nets, ops, step
from make_golden import CONV_CASES, CONVT_CASES, NET_CASES, STEP_CASES

ATOL, RTOL = 2e-5, 2e-4


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv(golden, case):
    name, n, ci, h, w, co, stride, bias = case
    z = golden("ops")
    wv = fill.uniform(name + ".w", (co, ci, 4, 4), -0.25, 0.25)
    bv = fill.uniform(name + ".b", (co,), -0.5, 0.5) if bias else None
    x = fill.uniform(name + ".x", (n, ci, h, w))
    y = ops.conv2d_fwd(x, wv, bv, stride)
    assert_close(y, z[name + ".y"], ATOL, RTOL, name + ".y")
    dy = fill.uniform(name + ".dy", y.shape)
    dx, dw, db = ops.conv2d_bwd(x, wv, dy, stride)
    assert_close(dx, z[name + ".dx"], ATOL, RTOL, name + ".dx")
    assert_close(dw, z[name + ".dw"], ATOL, RTOL, name + ".dw")
    if bias:
        assert_close(db, z[name + ".db"], ATOL, RTOL, name + ".db")


@pytest.mark.parametrize("case", CONVT_CASES, ids=[c[0] for c in CONVT_CASES])
def test_convT(golden, case):
    name, n, ci, h, w, co, bias = case
    z = golden("ops")
    wv = fill.uniform(name + ".w", (ci, co, 4, 4), -0.25, 0.25)
    bv = fill.uniform(name + ".b", (co,), -0.5, 0.5) if bias else None
    x = fill.uniform(name + ".x", (n, ci, h, w))
    y = ops.convT2d_fwd(x, wv, bv)
    assert_close(y, z[name + ".y"], ATOL, RTOL, name + ".y")
    dy = fill.uniform(name + ".dy", y.shape)
    dx, dw, db = ops.convT2d_bwd(x, wv, dy)
    assert_close(dx, z[name + ".dx"], ATOL, RTOL, name + ".dx")
    assert_close(dw, z[name + ".dw"], ATOL, RTOL, name + ".dw")
    if bias:
        assert_close(db, z[name + ".db"], ATOL, RTOL, name + ".db")


def test_batchnorm(golden):
    z = golden("ops")
    g = fill.uniform("bn.g", (8,), 0.5, 1.5)
    b = fill.uniform("bn.b", (8,), -0.5, 0.5)
    x = fill.uniform("bn.x", (4, 8, 5, 5), -2, 3)
    rm, rv, nbt = np.zeros(8, np.float32), np.ones(8, np.float32), 0
    y, cache, rm1, rv1, nbt1 = ops.bn_train_fwd(x, g, b, rm, rv, nbt)
    assert_close(y, z["bn.y"], ATOL, RTOL, "bn.y")
    assert_close(rm1, z["bn.rm1"], 1e-6, 1e-5, "bn.rm1")
    assert_close(rv1, z["bn.rv1"], 1e-6, 1e-5, "bn.rv1")
    dy = fill.uniform("bn.dy", (4, 8, 5, 5))
    dx, dg, db = ops.bn_train_bwd(dy, g, cache)
    assert_close(dx, z["bn.dx"], ATOL, RTOL, "bn.dx")
    assert_close(dg, z["bn.dg"], ATOL, RTOL, "bn.dg")
    assert_close(db, z["bn.db"], ATOL, RTOL, "bn.db")
    # the F5 closed form: two updates with the same batch statistics == two forwards
    _, _, rm2, rv2, nbt2 = ops.bn_train_fwd(x, g, b, rm, rv, nbt, repeats=2)
    assert_close(rm2, z["bn.rm2"], 1e-6, 1e-5, "bn.rm2")
    assert_close(rv2, z["bn.rv2"], 1e-6, 1e-5, "bn.rv2")
    assert nbt2 == int(z["bn.nbt2"]) == 2
    assert_close(ops.bn_eval_fwd(x, g, b, rm2, rv2), z["bn.y_eval"], ATOL, RTOL, "bn.y_eval")


def test_activations_losses_adam(golden):
    z = golden("ops")
    x = fill.uniform("act.x", (2, 4, 3, 3), -2, 2)
    dy = fill.uniform("act.dy", (2, 4, 3, 3))
    assert_close(ops.leaky_relu(x), z["act.lrelu.y"], 0, 1e-7)
    assert_close(ops.leaky_relu_bwd(x, dy), z["act.lrelu.dx"], 0, 1e-7)
    assert_close(ops.relu(x), z["act.relu.y"], 0, 0)
    assert_close(ops.relu_bwd(x, dy), z["act.relu.dx"], 0, 0)
    assert_close(np.tanh(x), z["act.tanh.y"], 1e-7, 1e-6)
    x = fill.uniform("bce.x", (2, 1, 6, 6), -4, 4)
    for tgt in (1.0, 0.0):
        loss, dx = ops.bce_logits(x, tgt)
        assert_close(loss, z[f"bce.t{int(tgt)}.loss"], 1e-6, 1e-6)
        assert_close(dx, z[f"bce.t{int(tgt)}.dx"], 1e-8, 1e-5)
    a = fill.uniform("l1.a", (2, 3, 8, 8))
    b = fill.uniform("l1.b", (2, 3, 8, 8))
    loss, da = ops.l1_loss(a, b)
    assert_close(loss, z["l1.loss"], 1e-6, 1e-6)
    assert_close(da, z["l1.da"], 0, 1e-6)
    p = fill.uniform("adam.p", (64,))
    m = np.zeros_like(p)
    v = np.zeros_like(p)
    for i in range(3):
        g = fill.uniform(f"adam.g{i}", (64,), -1e-2, 1e-2)
        p, m, v = ops.adam_step(p, g, m, v, i + 1)
        assert_close(p, z[f"adam.p{i + 1}"], 2e-7, 1e-6, f"adam.p{i + 1}")


@pytest.mark.parametrize("case", NET_CASES, ids=[c[0] for c in NET_CASES])
def test_networks(golden, case):
    name, nd, ngf, ndf, n, h = case
    z = golden("nets")
    gst = fill.fill_state(nets.g_state_shapes(3, 3, nd, ngf), seed=1)
    seed = int(z[f"{name}.g.seed"])
    x = fill.uniform(f"{name}.x", (n, 3, h, h), seed=seed)
    y, cache, nb = nets.generator_fwd(gst, x, nd, train=True)
    assert_close(y, z[f"{name}.g.y"], 5e-5, 1e-3, "G train fwd")
    dy = fill.uniform(f"{name}.g.dy", y.shape, seed=seed)
    grads = nets.generator_bwd(gst, cache, dy, nd)
    for k, g in grads.items():
        want = z[f"{name}.g.grad.{k}"]
        scale = max(np.abs(want).max(), 1e-3)
        assert_close(fill.compact(g), want, 2e-4 * scale, 2e-3, f"G grad {k}")
    for k, v in nb.items():
        assert_close(v, z[f"{name}.g.buf.{k}"], 1e-5, 1e-4, f"G buffer {k}")
    # the golden eval forward ran after the train forward had updated the running buffers
    y_eval, _, _ = nets.generator_fwd({**gst, **nb}, x, nd, train=False)
    assert_close(y_eval, z[f"{name}.g.y_eval"], 5e-5, 1e-3, "G eval fwd")

    dst = fill.fill_state(nets.d_state_shapes(6, ndf, 3), seed=2)
    seed = int(z[f"{name}.d.seed"])
    xd = fill.uniform(f"{name}.xd", (n, 6, h, h), seed=seed)
    y, cache, nb = nets.discriminator_fwd(dst, xd, 3, train=True)
    assert_close(y, z[f"{name}.d.y"], 5e-5, 1e-3, "D train fwd")
    dy = fill.uniform(f"{name}.d.dy", y.shape, seed=seed)
    grads, dx = nets.discriminator_bwd(dst, cache, dy, need_dx=True)
    assert_close(dx, z[f"{name}.d.dx"], 2e-5 * max(np.abs(z[f"{name}.d.dx"]).max(), 1), 2e-3, "D dx")
    for k, g in grads.items():
        want = z[f"{name}.d.grad.{k}"]
        scale = max(np.abs(want).max(), 1e-3)
        assert_close(fill.compact(g), want, 2e-4 * scale, 2e-3, f"D grad {k}")
    for k, v in nb.items():
        assert_close(v, z[f"{name}.d.buf.{k}"], 1e-5, 1e-4, f"D buffer {k}")
    y_eval, _, _ = nets.discriminator_fwd({**dst, **nb}, xd, 3, train=False)
    assert_close(y_eval, z[f"{name}.d.y_eval"], 5e-5, 1e-3, "D eval fwd")


def _check_digest(z, name, a, atol, rtol):
    flat = np.asarray(a).reshape(-1)
    stride = int(z[name + ".stride"])
    assert flat.size == int(z[name + ".size"])
    assert_close(flat[::stride][: z[name + ".sample"].size], z[name + ".sample"], atol, rtol, name + ".sample")
    a64 = flat.astype(np.float64)
    assert_close(np.abs(a64).sum(), z[name + ".abs_sum"], 0, 1e-4, name + ".abs_sum")
    assert_close((a64 * a64).sum(), z[name + ".sq_sum"], 0, 1e-4, name + ".sq_sum")


def test_full_size_networks(golden):
    """Default num_downs=7 / ngf=64 networks at 128^2 B=2 (the 256^2 case runs in the gpu suite)."""
    z = golden("full")
    gst = fill.fill_state(nets.g_state_shapes(), seed=3)
    dst = fill.fill_state(nets.d_state_shapes(), seed=4)
    tag, n, h = "b2_128", 2, 128
    x = fill.uniform(f"full.{tag}.x", (n, 3, h, h))
    y, _, nb = nets.generator_fwd(gst, x, 7, train=True)
    _check_digest(z, f"full.{tag}.g.y", y, 1e-4, 1e-3)
    assert_close(y.astype(np.float64).mean(axis=(0, 2, 3)), z[f"full.{tag}.g.y_chan_mean"], 1e-5, 1e-4)
    run = np.concatenate([v.reshape(-1) for k, v in nb.items() if "running" in k])
    _check_digest(z, f"full.{tag}.g.running_cat", run, 1e-5, 1e-4)
    ye, _, _ = nets.generator_fwd(gst, x, 7, train=False)
    _check_digest(z, f"full.{tag}.g.y_eval", ye, 1e-4, 1e-3)
    xd = fill.uniform(f"full.{tag}.xd", (n, 6, h, h))
    yd, _, _ = nets.discriminator_fwd(dst, xd, 3, train=True)
    assert_close(yd, z[f"full.{tag}.d.y"], 1e-4, 1e-3, "D logits")


@pytest.mark.parametrize("case", STEP_CASES, ids=[c[0] for c in STEP_CASES])
def test_gan_step(golden, case):
    """Two consecutive train_gan_one_epoch iterations: losses, raw grads, params, BN buffers."""
    name, nd, ngf, ndf, n, h, steps = case
    z = golden("step")
    gst = fill.fill_state(nets.g_state_shapes(3, 3, nd, ngf), seed=5, bn_noise=False)
    dst = fill.fill_state(nets.d_state_shapes(6, ndf, 3), seed=6, bn_noise=False)
    gopt, dopt = step.new_adam_state(gst), step.new_adam_state(dst)
    seed = int(z[f"step.{name}.seed"])
    for i in range(steps):
        a = fill.uniform(f"step.{name}.a{i}", (n, 3, h, h), seed=seed)
        b = fill.uniform(f"step.{name}.b{i}", (n, 3, h, h), seed=seed)
        ld, lg, ex = step.gan_step(gst, dst, gopt, dopt, a, b, nd, 3)
        assert_close(ld, z[f"step.{name}.loss_d{i}"], 1e-5, 1e-5, f"loss_d step {i}")
        assert_close(lg, z[f"step.{name}.loss_g{i}"], 1e-4, 1e-5, f"loss_g step {i}")
        for k, g in ex["d_grads"].items():
            want = z[f"step.{name}.dgrad{i}.{k}"]
            assert_close(fill.compact(g), want, 5e-4 * max(np.abs(want).max(), 1e-6), 5e-3, f"D grad {k} step {i}")
        for k, g in ex["g_grads"].items():
            want = z[f"step.{name}.ggrad{i}.{k}"]
            assert_close(fill.compact(g), want, 5e-4 * max(np.abs(want).max(), 1e-6), 5e-3, f"G grad {k} step {i}")
    for prefix, st in (("g", gst), ("d", dst)):
        for k, v in st.items():
            want = z[f"step.{name}.{prefix}.{k}"]
            if k.endswith("num_batches_tracked"):
                assert int(v) == int(want), k          # G: +2 per step, D: +3 per step
            elif step.is_param(k):
                # Adam's first steps move every weight by ~lr; a sign flip on a ~0 gradient costs 2e-4
                got = fill.compact(v)
                frac_bad = np.mean(np.abs(got - want) > 2e-5)
                assert frac_bad < 0.01, f"{prefix}.{k}: {frac_bad:.3%} of params off by >2e-5"
                assert np.abs(got - want).max() < 5e-4, f"{prefix}.{k}"
            else:
                assert_close(v, want, 1e-5, 1e-3, f"{prefix} buffer {k}")


def test_bf16_storage_oracle_tracks_the_pinned_fp32_oracle():
    """The bf16-storage restatement (oracle/nets.py, the checker of the bf16 network gate) is the pinned fp32 oracle plus
    roundings: it must stay within bf16's drift of it (a few 1e-2 on a train-mode generator) and be exactly
    bf16-representable."""
    nd, ngf, ndf, n, h = 5, 8, 8, 2, 32
    gst = fill.fill_state(nets.g_state_shapes(3, 3, nd, ngf), seed=1)
    x = fill.uniform("bf16or.x", (n, 3, h, h))
    y16 = nets.generator_fwd_bf16(gst, x, nd)
    y32, _, _ = nets.generator_fwd(gst, x, nd, train=True)
    assert np.abs(y16 - y32).max() < 6e-2 and np.abs(y16 - y32).mean() < 6e-3
    assert np.array_equal(fill.bf16_round(y16), y16)
    dst = fill.fill_state(nets.d_state_shapes(6, ndf, 3), seed=2)
    xd = fill.uniform("bf16or.xd", (n, 6, h, h))
    d16 = nets.discriminator_fwd_bf16(dst, xd)
    d32, _, _ = nets.discriminator_fwd(dst, xd)
    assert np.abs(d16 - d32).max() < 3e-2 * max(1.0, np.abs(d32).max())


def test_bf16_step_oracle_tracks_the_pinned_fp32_oracle():
    """oracle/bf16.py - the checker of the throughput mode's whole-step gate (forward, backward and Adam with a bf16 rounding
    at the engines' storage points) - is the pinned float32 step (oracle/step.py, held to the reference's goldens above)
    plus roundings: losses within 2 %, every weight gradient within bf16's drift in the L2 sense, the same BatchNorm
    counters, and Adam moving (almost) every parameter the same way."""
    from oracle import bf16 as o16, step as ostep
    nd, ngf, ndf, n, h = 5, 8, 8, 2, 32
    a = fill.uniform("bf16step.a", (n, 3, h, h))
    b = fill.uniform("bf16step.b", (n, 3, h, h))
    g0 = fill.fill_state(nets.g_state_shapes(3, 3, nd, ngf), seed=5, bn_noise=False)
    d0 = fill.fill_state(nets.d_state_shapes(6, ndf, 3), seed=6, bn_noise=False)
    g32, d32 = {k: v.copy() for k, v in g0.items()}, {k: v.copy() for k, v in d0.items()}
    g16, d16 = {k: v.copy() for k, v in g0.items()}, {k: v.copy() for k, v in d0.items()}
    ld32, lg32, ex32 = ostep.gan_step(g32, d32, ostep.new_adam_state(g32), ostep.new_adam_state(d32), a, b, nd, 3)
    ld16, lg16, ex16 = o16.gan_step(g16, d16, ostep.new_adam_state(g16), ostep.new_adam_state(d16), a, b, nd, 3)
    assert abs(ld16 - ld32) <= 2e-2 * abs(ld32) and abs(lg16 - lg32) <= 2e-2 * abs(lg32)
    assert np.array_equal(fill.bf16_round(ex16["fake_b"]), ex16["fake_b"])
    assert np.abs(ex16["fake_b"] - ex32["fake_b"]).mean() < 1e-2
    for name, g16s, g32s in (("G", ex16["g_grads"], ex32["g_grads"]), ("D", ex16["d_grads"], ex32["d_grads"])):
        assert set(g16s) == set(g32s)
        for k in g32s:
            if g32s[k].ndim == 4:
                rel = np.linalg.norm(g16s[k].astype(np.float64) - g32s[k]) / max(np.linalg.norm(g32s[k]), 1e-12)
                assert rel < 0.15, f"{name} grad {k}: rel L2 {rel:.3f}"
    for st16, st32 in ((g16, g32), (d16, d32)):
        alike = total = 0
        for k in st32:
            if k.endswith("num_batches_tracked"):
                assert int(st16[k]) == int(st32[k])
            elif ostep.is_param(k):
                alike += int(np.sum(np.abs(st16[k] - st32[k]) < 0.5 * ostep.LR))      # the first Adam step moves by +-lr
                total += st32[k].size
            else:
                assert np.allclose(st16[k], st32[k], rtol=5e-2, atol=5e-3), k
        assert alike > 0.9 * total, f"only {alike / total:.1%} of the parameters moved alike"
