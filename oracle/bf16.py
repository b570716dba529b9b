"""bf16-storage restatement of one train_gan_one_epoch iteration: the checker of the throughput mode's WHOLE-STEP gate.

TEST INFRASTRUCTURE (see oracle/__init__.py).  oracle/step.py (pinned to the reference by tests/golden) is the float32
arithmetic of train_gan.py:52-71.  The HIP engines' throughput mode keeps every tensor that lives in HBM in bfloat16 and
everything inside a kernel in float32; this module restates the same step with a round-to-nearest-even bf16 at the
engines' storage points, forward AND backward, so that GanTrainer(precision="bf16") can be held to 1e-3-class tolerances
instead of the 2 % / 10 % a comparison with the float32 oracle allows:

  forward    operands of every convolution (activations and the bf16 weight copies), the stored conv output `raw` of a
             BatchNorm layer (statistics come from the float32 accumulators), the activated tensors, `fake`, the logits
  backward   dlogits, dpre, every stored gradient tensor: the fused epilogue of an input-gradient launch forms
             dy = act'(y) g from the float32 accumulators (BatchNorm sums from those float32 values) and stores bf16; the
             BatchNorm backward then maps the stored dy and raw to the stored conv-output gradient
  float32    weight / bias / BatchNorm gradients (accumulated from the bf16 operands), master parameters, Adam

Where the engine has a choice of route (a gradient that passes through a bf16 tensor before its activation backward on
shapes whose launch is not fused) this file takes the fused one: the difference is one rounding of 2^-9 relative on that
tensor, inside the tolerances of the tests that use it.  tests/test_oracle_golden.py anchors this module to the pinned
float32 oracle.
"""
import numpy as np

from . import nets, ops
from .fill import bf16_round
from .step import BETAS, LAMBDA_L1, LR, is_param

F32 = np.float32


def r16(x):
    return bf16_round(np.asarray(x, F32))


def _stats(acc):
    a = acc.astype(np.float64)
    mean = a.mean(axis=(0, 2, 3))
    var = a.var(axis=(0, 2, 3))
    return mean, var


class _BN:
    """Training-mode BatchNorm2d of one conv output: statistics of the float32 accumulators, applied to the bf16 `raw`."""

    def __init__(self, acc, gamma, beta):
        self.cnt = acc.shape[0] * acc.shape[2] * acc.shape[3]
        mean, var = _stats(acc)
        self.mean64, self.var64 = mean, var
        self.invstd = (1.0 / np.sqrt(var + ops.BN_EPS)).astype(F32)
        self.mean = mean.astype(F32)
        self.gamma = gamma
        self.scale = (gamma.astype(np.float64) / np.sqrt(var + ops.BN_EPS)).astype(F32)
        self.shift = (beta.astype(np.float64) - mean * (gamma.astype(np.float64) / np.sqrt(var + ops.BN_EPS))).astype(F32)
        self.raw = r16(acc)

    def y(self):
        return (self.raw * self.scale[None, :, None, None] + self.shift[None, :, None, None]).astype(F32)

    def xhat(self):
        return ((self.raw.astype(np.float64) - self.mean.astype(np.float64)[None, :, None, None])
                * self.invstd.astype(np.float64)[None, :, None, None])

    def running(self, st, prefix, new_state, repeats):
        unb = self.var64 * (self.cnt / max(self.cnt - 1, 1))
        rm = st[prefix + ".running_mean"].astype(np.float64)
        rv = st[prefix + ".running_var"].astype(np.float64)
        for _ in range(repeats):
            rm = (1 - ops.BN_MOMENTUM) * rm + ops.BN_MOMENTUM * self.mean64
            rv = (1 - ops.BN_MOMENTUM) * rv + ops.BN_MOMENTUM * unb
        new_state[prefix + ".running_mean"] = rm.astype(F32)
        new_state[prefix + ".running_var"] = rv.astype(F32)
        new_state[prefix + ".num_batches_tracked"] = np.asarray(st[prefix + ".num_batches_tracked"] + repeats, dtype=np.int64)

    def backward(self, dy32):
        """dy32: float32 gradient of the normalised output's pre-activation as the fused epilogue holds it.
        -> (stored conv-output gradient bf16, dgamma, dbeta)"""
        d64 = dy32.astype(np.float64)
        xh = self.xhat()
        dbeta = d64.sum(axis=(0, 2, 3))
        dgamma = (d64 * xh).sum(axis=(0, 2, 3))
        dy = r16(dy32).astype(np.float64)                       # stored, then read back by the apply pass
        s = (self.gamma.astype(np.float64) * self.invstd.astype(np.float64))[None, :, None, None]
        dr = s * (dy - dbeta[None, :, None, None] / self.cnt - xh * dgamma[None, :, None, None] / self.cnt)
        return r16(dr), dgamma.astype(F32), dbeta.astype(F32)


# ----------------------------------------------------------------------------------------------- generator
def generator_fwd(st, x, num_downs=7, bn_repeats=2):
    """-> (fake bf16, cache, new BatchNorm buffers)"""
    lv = nets.g_levels(num_downs)
    w = lambda k: r16(st[k])
    nb = {}
    c = dict(lv=lv, a_in=[], bn_down=[], aL=[], aR=[], src_up=[None] * num_downs, bn_up=[None] * num_downs)
    a = r16(x)
    for k in range(num_downs):
        c["a_in"].append(a)
        acc = ops.conv2d_fwd(a, w(lv[k]["down"] + ".weight"), None, 2)
        if k == num_downs - 1:
            c["bn_down"].append(None)
            c["aL"].append(None)
            c["aR"].append(r16(ops.relu(acc)))
            break
        if lv[k]["down_bn"]:
            bn = _BN(acc, st[lv[k]["down_bn"] + ".weight"], st[lv[k]["down_bn"] + ".bias"])
            bn.running(st, lv[k]["down_bn"], nb, bn_repeats)
            y = bn.y()
        else:
            bn, y = None, acc
        c["bn_down"].append(bn)
        c["aL"].append(r16(ops.leaky_relu(y)))
        c["aR"].append(r16(ops.relu(y)))
        a = c["aL"][k]
    u = None
    for k in range(num_downs - 1, 0, -1):
        src = c["aR"][k] if u is None else np.concatenate([c["aR"][k], u], axis=1)
        c["src_up"][k] = src
        acc = ops.convT2d_fwd(src, w(lv[k]["up"] + ".weight"), None)
        bn = _BN(acc, st[lv[k]["up_bn"] + ".weight"], st[lv[k]["up_bn"] + ".bias"])
        bn.running(st, lv[k]["up_bn"], nb, bn_repeats)
        c["bn_up"][k] = bn
        u = r16(ops.relu(bn.y()))
    src = np.concatenate([c["aR"][0], u], axis=1)
    c["src_up"][0] = src
    acc = ops.convT2d_fwd(src, w(lv[0]["up"] + ".weight"), st[lv[0]["up"] + ".bias"])
    fake = r16(np.tanh(acc).astype(F32))
    c["fake"] = fake
    return fake, c, nb


def generator_bwd(st, c, dpre, num_downs=7, unfused=False):
    """dpre: stored (bf16) gradient w.r.t. the pre-Tanh output.  -> parameter gradients (float32) by state_dict key.
    unfused: every input gradient is stored (bf16) before the activation backward of the node it belongs to - the engines'
    route for launches whose epilogue does not carry the fused backward (kernels.fuse_ok)."""
    rg = r16 if unfused else (lambda t: t)
    lv = c["lv"]
    w = lambda k: r16(st[k])
    grads = {}
    g_skip = [None] * num_downs
    d = dpre
    dr_inner = None
    for k in range(num_downs):
        src = c["src_up"][k]
        da, dw, db = ops.convT2d_bwd(src, w(lv[k]["up"] + ".weight"), d)
        grads[lv[k]["up"] + ".weight"] = dw
        if lv[k]["up"] + ".bias" in st:
            grads[lv[k]["up"] + ".bias"] = db
        cs = c["aR"][k].shape[1]
        if k == num_downs - 1:
            dr_inner = r16(np.where(c["aR"][k] > 0, rg(da), 0))              # ReLU' from the stored activation's sign
            break
        g_skip[k] = r16(da[:, :cs])
        bn = c["bn_up"][k + 1]
        d, dg, dbt = bn.backward(np.where(bn.y() > 0, rg(da[:, cs:]), 0).astype(F32))
        grads[lv[k + 1]["up_bn"] + ".weight"] = dg
        grads[lv[k + 1]["up_bn"] + ".bias"] = dbt
    dr = dr_inner
    for k in range(num_downs - 1, -1, -1):
        dx, dw, _ = ops.conv2d_bwd(c["a_in"][k], w(lv[k]["down"] + ".weight"), dr, 2, need_dx=(k > 0))
        grads[lv[k]["down"] + ".weight"] = dw
        if k == 0:
            break
        bn = c["bn_down"][k - 1]
        if bn is not None:
            y = bn.y()
            dx = rg(dx)
            dy = np.where(y > 0, dx, dx * F32(0.2)) + np.where(y > 0, g_skip[k - 1], 0)
            dr, dg, dbt = bn.backward(dy.astype(F32))
            grads[lv[k - 1]["down_bn"] + ".weight"] = dg
            grads[lv[k - 1]["down_bn"] + ".bias"] = dbt
        else:                                                                 # outermost level: sign from the stored aL
            aL = c["aL"][k - 1]
            dx = rg(dx)
            dr = r16(np.where(aL > 0, dx, dx * F32(0.2)) + np.where(aL > 0, g_skip[k - 1], 0))
    return grads


# ----------------------------------------------------------------------------------------------- discriminator
def discriminator_fwd(st, x, n_layers=3):
    """x: bf16-representable [N, 6, H, W].  -> (logits bf16, cache, new BatchNorm buffers)"""
    layers = nets.d_layers(n_layers)
    nb = {}
    c = dict(layers=layers, a_in=[], bn=[], act=[])
    h = r16(x)
    for i, (ck, bk, s, has_b) in enumerate(layers):
        c["a_in"].append(h)
        acc = ops.conv2d_fwd(h, r16(st[ck + ".weight"]), st[ck + ".bias"] if has_b else None, s)
        bn = None
        if bk:
            bn = _BN(acc, st[bk + ".weight"], st[bk + ".bias"])
            bn.running(st, bk, nb, 1)
            h = r16(ops.leaky_relu(bn.y()))
        elif i < len(layers) - 1:
            h = r16(ops.leaky_relu(acc))
        else:
            h = r16(acc)
        c["bn"].append(bn)
        c["act"].append(h)
    return h, c, nb


def discriminator_bwd(st, c, dlogits, param_grads=True, need_dx=False, unfused=False):
    """dlogits: stored (bf16).  -> (parameter gradients or {}, stored dx or None); unfused: see generator_bwd."""
    rg = r16 if unfused else (lambda t: t)
    layers = c["layers"]
    grads = {}
    dr = dlogits
    dx_out = None
    for i in range(len(layers) - 1, -1, -1):
        ck, bk, s, has_b = layers[i]
        want_dx = i > 0 or need_dx
        dx, dw, db = ops.conv2d_bwd(c["a_in"][i], r16(st[ck + ".weight"]), dr, s, need_dx=want_dx)
        if param_grads:
            grads[ck + ".weight"] = dw
            if has_b:
                grads[ck + ".bias"] = db
        if i == 0:
            dx_out = r16(dx) if need_dx else None
            break
        bn = c["bn"][i - 1]
        dx = rg(dx)
        if bn is not None:
            dr, dg, dbt = bn.backward(np.where(bn.y() > 0, dx, dx * F32(0.2)).astype(F32))
            if param_grads:
                grads[layers[i - 1][1] + ".weight"] = dg
                grads[layers[i - 1][1] + ".bias"] = dbt
        else:
            a = c["act"][i - 1]
            dr = r16(np.where(a > 0, dx, dx * F32(0.2)))
    return grads, dx_out


# ----------------------------------------------------------------------------------------------- losses, Adam, the step
def _bce(logits, target, lscale, gscale):
    x = logits.astype(np.float64)
    loss = lscale * np.mean(np.maximum(x, 0) - x * target + np.log1p(np.exp(-np.abs(x))))
    grad = gscale * (1.0 / (1.0 + np.exp(-x)) - target) / x.size
    return F32(loss), r16(grad)


def _adam(st, grads, opt):
    for k, g in grads.items():
        s = opt[k]
        s["t"] += 1
        st[k], s["m"], s["v"] = ops.adam_step(st[k], g.astype(F32), s["m"], s["v"], s["t"], LR, BETAS[0], BETAS[1])


def gan_step(g_st, d_st, g_opt, d_opt, real_a, real_b, num_downs=7, n_layers=3, unfused=False):
    """One batch of train_gan_one_epoch with bf16 storage (see the module header); same contract as oracle.step.gan_step.
    unfused: the backward passes take the un-fused route at every node (generator_bwd)."""
    xa, xb = r16(real_a), r16(real_b)
    fake, gc, g_nb = generator_fwd(g_st, xa, num_downs, bn_repeats=2)
    pred_real, c_real, nb1 = discriminator_fwd(d_st, np.concatenate([xa, xb], 1), n_layers)
    d_st.update(nb1)
    l_real, dl_real = _bce(pred_real, 1.0, 0.5, 0.5)
    pred_fake, c_fake, nb2 = discriminator_fwd(d_st, np.concatenate([xa, fake], 1), n_layers)
    d_st.update(nb2)
    l_fake, dl_fake = _bce(pred_fake, 0.0, 0.5, 0.5)
    loss_d = F32(l_real + l_fake)
    gr, _ = discriminator_bwd(d_st, c_real, dl_real, unfused=unfused)
    gf, _ = discriminator_bwd(d_st, c_fake, dl_fake, unfused=unfused)
    d_grads = {k: (gr[k] + gf[k]).astype(F32) for k in gr}
    _adam(d_st, d_grads, d_opt)
    g_st.update(g_nb)
    pred_g, c_g, nb3 = discriminator_fwd(d_st, np.concatenate([xa, fake], 1), n_layers)
    d_st.update(nb3)
    l_gan, dl_gan = _bce(pred_g, 1.0, 1.0, 1.0)
    _, d_in = discriminator_bwd(d_st, c_g, dl_gan, param_grads=False, need_dx=True, unfused=unfused)
    dgan = d_in[:, real_a.shape[1]:]
    diff = fake.astype(np.float64) - xb.astype(np.float64)
    l_l1 = LAMBDA_L1 * np.mean(np.abs(diff))
    loss_g = F32(l_gan + F32(l_l1))
    dpre = r16((LAMBDA_L1 * np.sign(diff) / diff.size + dgan.astype(np.float64)) * (1.0 - fake.astype(np.float64) ** 2))
    g_grads = generator_bwd(g_st, gc, dpre, num_downs, unfused=unfused)
    _adam(g_st, g_grads, g_opt)
    return loss_d, loss_g, dict(fake_b=fake, d_grads=d_grads, d_grads_real=gr, d_grads_fake=gf, g_grads=g_grads,
                                pred_real=pred_real, pred_fake=pred_fake, pred_g=pred_g)
