"""numpy restatement of the reference's UNetGenerator and NLayerDiscriminator.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Parameters and buffers travel as a
dict keyed by the reference's own state_dict names, so a
fixture, the oracle and the product modules all speak the same layout.
Default norm layer only (nn.BatchNorm2d, models.py:151,214), no dropout.
"""
import numpy as np

from . import ops

F32 = np.float32


# ----------------------------------------------------------------------- generator
def g_levels(num_downs=7):
    """Key prefixes per U-Net level, outermost (0) .. innermost (num_downs-1).

    Mirrors the nn.Sequential indices of models.py:183-200: outermost
    [0 conv, 1 sub, 2 ReLU, 3 convT, 4 Tanh]; middle [0 LReLU, 1 conv, 2 BN,
    3 sub, 4 ReLU, 5 convT, 6 BN]; innermost [0 LReLU, 1 conv, 2 ReLU, 3 convT, 4 BN].
    """
    lv = [dict(down="model.model.0", down_bn=None, up="model.model.3", up_bn=None)]
    pre = "model.model.1"
    for k in range(1, num_downs):
        if k == num_downs - 1:
            lv.append(dict(down=pre + ".model.1", down_bn=None, up=pre + ".model.3", up_bn=pre + ".model.4"))
        else:
            lv.append(dict(down=pre + ".model.1", down_bn=pre + ".model.2", up=pre + ".model.5", up_bn=pre + ".model.6"))
        pre = pre + ".model.3"
    return lv


def g_channels(input_nc=3, output_nc=3, num_downs=7, ngf=64):
    """(encoder out-channels per level, decoder out-channels per level) - models.py:155-161."""
    enc = [ngf * min(2 ** k, 8) for k in range(num_downs)]
    dec = [output_nc] + enc[: num_downs - 1]
    return enc, dec


def g_state_shapes(input_nc=3, output_nc=3, num_downs=7, ngf=64):
    """Ordered {key: shape} of the generator state_dict."""
    enc, dec = g_channels(input_nc, output_nc, num_downs, ngf)
    lv = g_levels(num_downs)
    shapes = {}

    def bn(prefix, c):
        shapes[prefix + ".weight"] = (c,)
        shapes[prefix + ".bias"] = (c,)
        shapes[prefix + ".running_mean"] = (c,)
        shapes[prefix + ".running_var"] = (c,)
        shapes[prefix + ".num_batches_tracked"] = ()

    def walk(k):
        cin = input_nc if k == 0 else enc[k - 1]
        shapes[lv[k]["down"] + ".weight"] = (enc[k], cin, 4, 4)
        if lv[k]["down_bn"]:
            bn(lv[k]["down_bn"], enc[k])
        if k + 1 < num_downs:
            walk(k + 1)
        up_in = enc[k] if k == num_downs - 1 else 2 * enc[k]
        shapes[lv[k]["up"] + ".weight"] = (up_in, dec[k], 4, 4)
        if k == 0:
            shapes[lv[k]["up"] + ".bias"] = (dec[k],)
        if lv[k]["up_bn"]:
            bn(lv[k]["up_bn"], dec[k])

    walk(0)
    return shapes


def _bn(x, st, prefix, train, new_state, repeats):
    g, b = st[prefix + ".weight"], st[prefix + ".bias"]
    rm, rv, nbt = st[prefix + ".running_mean"], st[prefix + ".running_var"], st[prefix + ".num_batches_tracked"]
    if train:
        y, cache, rm2, rv2, nbt2 = ops.bn_train_fwd(x, g, b, rm, rv, nbt, repeats)
        new_state[prefix + ".running_mean"] = rm2
        new_state[prefix + ".running_var"] = rv2
        new_state[prefix + ".num_batches_tracked"] = np.asarray(nbt2, dtype=np.int64)
        return y, cache
    return ops.bn_eval_fwd(x, g, b, rm, rv), None


def generator_fwd(st, x, num_downs=7, train=True, bn_repeats=1):
    """UNetGenerator.forward (models.py:163-164, blocks :204-208).

    Returns (out, cache, new_buffers).  `new_buffers` holds the BN buffers the
    forward would have written (empty in eval mode).
    """
    lv = g_levels(num_downs)
    nb = {}
    c = dict(lv=lv, x=x, skip=[], down_in=[], down_bn=[], up_in=[], up_bn=[None] * num_downs)
    h = x
    for k in range(num_downs):
        if k > 0:
            h = ops.leaky_relu(h)          # in-place LeakyReLU: also what gets concatenated
            c["skip"].append(h)
        c["down_in"].append(h)
        r = ops.conv2d_fwd(h, st[lv[k]["down"] + ".weight"], None, 2)
        if lv[k]["down_bn"]:
            r, bc = _bn(r, st, lv[k]["down_bn"], train, nb, bn_repeats)
            c["down_bn"].append(bc)
        else:
            c["down_bn"].append(None)
        h = r
    # h = innermost conv output (pre-ReLU); walk back up
    u = h
    for k in range(num_downs - 1, -1, -1):
        a = ops.relu(u)
        c["up_in"].insert(0, (u, a))
        bias = st.get(lv[k]["up"] + ".bias")
        u = ops.convT2d_fwd(a, st[lv[k]["up"] + ".weight"], bias)
        if lv[k]["up_bn"]:
            u, bc = _bn(u, st, lv[k]["up_bn"], train, nb, bn_repeats)
            c["up_bn"][k] = bc
        if k > 0:
            u = np.concatenate([c["skip"][k - 1], u], axis=1)   # cat([x, model(x)], 1)  models.py:208
    out = np.tanh(u).astype(F32)
    c["out"] = out
    return out, c, nb


def generator_bwd(st, cache, dout, num_downs=7):
    """Gradients of generator_fwd w.r.t. every parameter (dict by state_dict key)."""
    lv = cache["lv"]
    grads = {}
    d = (dout * (1 - cache["out"].astype(np.float64) ** 2)).astype(F32)   # Tanh
    dskip = [None] * num_downs
    # up path, outermost -> innermost
    for k in range(num_downs):
        if k > 0:
            cs = cache["skip"][k - 1].shape[1]
            dskip[k] = d[:, :cs]           # gradient w.r.t. the concatenated (LeakyReLU'd) skip
            d = d[:, cs:]
        if lv[k]["up_bn"]:
            d, dg, db = ops.bn_train_bwd(d, st[lv[k]["up_bn"] + ".weight"], cache["up_bn"][k])
            grads[lv[k]["up_bn"] + ".weight"] = dg
            grads[lv[k]["up_bn"] + ".bias"] = db
        u, a = cache["up_in"][k]
        da, dw, dbias = ops.convT2d_bwd(a, st[lv[k]["up"] + ".weight"], d)
        grads[lv[k]["up"] + ".weight"] = dw
        if lv[k]["up"] + ".bias" in st:
            grads[lv[k]["up"] + ".bias"] = dbias
        d = ops.relu_bwd(u, da)            # ReLU before the transposed conv
    # d = gradient w.r.t. innermost conv output; down path, innermost -> outermost
    for k in range(num_downs - 1, -1, -1):
        if lv[k]["down_bn"]:
            d, dg, db = ops.bn_train_bwd(d, st[lv[k]["down_bn"] + ".weight"], cache["down_bn"][k])
            grads[lv[k]["down_bn"] + ".weight"] = dg
            grads[lv[k]["down_bn"] + ".bias"] = db
        xin = cache["down_in"][k]
        dx, dw, _ = ops.conv2d_bwd(xin, st[lv[k]["down"] + ".weight"], d, 2, need_dx=(k > 0))
        grads[lv[k]["down"] + ".weight"] = dw
        if k > 0:
            # xin = LeakyReLU(y_{k-1}) feeds both this conv and the skip concat
            dz = dx + dskip[k]
            d = np.where(xin > 0, dz, dz * F32(0.2)).astype(F32)   # LeakyReLU' from its own output sign
    return grads


# ------------------------------------------------------------------- discriminator
def d_layers(n_layers=3):
    """(conv key, bn key or None, stride, has_bias) per conv - Sequential indices of models.py:223-243."""
    out = [("model.0", None, 2, True)]
    idx = 2
    for _ in range(1, n_layers):
        out.append((f"model.{idx}", f"model.{idx + 1}", 2, False))
        idx += 3
    out.append((f"model.{idx}", f"model.{idx + 1}", 1, False))
    idx += 3
    out.append((f"model.{idx}", None, 1, True))
    return out


def d_state_shapes(input_nc=6, ndf=64, n_layers=3):
    shapes = {}
    chans = [ndf * min(2 ** n, 8) for n in range(n_layers + 1)] + [1]
    cin = input_nc
    for (ck, bk, _, has_b), co in zip(d_layers(n_layers), chans):
        shapes[ck + ".weight"] = (co, cin, 4, 4)
        if has_b:
            shapes[ck + ".bias"] = (co,)
        if bk:
            shapes[bk + ".weight"] = (co,)
            shapes[bk + ".bias"] = (co,)
            shapes[bk + ".running_mean"] = (co,)
            shapes[bk + ".running_var"] = (co,)
            shapes[bk + ".num_batches_tracked"] = ()
        cin = co
    return shapes


def discriminator_fwd(st, x, n_layers=3, train=True):
    """NLayerDiscriminator.forward (models.py:246-247). Returns (logits, cache, new_buffers)."""
    layers = d_layers(n_layers)
    nb = {}
    c = dict(layers=layers, conv_in=[], bn=[], pre_act=[])
    h = x
    for i, (ck, bk, s, has_b) in enumerate(layers):
        c["conv_in"].append(h)
        r = ops.conv2d_fwd(h, st[ck + ".weight"], st[ck + ".bias"] if has_b else None, s)
        bc = None
        if bk:
            r, bc = _bn(r, st, bk, train, nb, 1)
        c["bn"].append(bc)
        c["pre_act"].append(r)
        h = ops.leaky_relu(r) if i < len(layers) - 1 else r
    return h, c, nb


def discriminator_bwd(st, cache, dout, need_dx=False):
    """Gradients of discriminator_fwd: (param grads dict, dx or None)."""
    layers = cache["layers"]
    grads = {}
    d = dout
    dx = None
    for i in range(len(layers) - 1, -1, -1):
        ck, bk, s, has_b = layers[i]
        if i < len(layers) - 1:
            d = ops.leaky_relu_bwd(cache["pre_act"][i], d)
        if bk:
            d, dg, db = ops.bn_train_bwd(d, st[bk + ".weight"], cache["bn"][i])
            grads[bk + ".weight"] = dg
            grads[bk + ".bias"] = db
        want_dx = need_dx or i > 0
        dxi, dw, dbias = ops.conv2d_bwd(cache["conv_in"][i], st[ck + ".weight"], d, s, need_dx=want_dx)
        grads[ck + ".weight"] = dw
        if has_b:
            grads[ck + ".bias"] = dbias
        d = dxi
        if i == 0:
            dx = dxi
    return grads, dx


# ----------------------------------------------------------------------- bf16 storage emulation
# The throughput mode of the HIP engines keeps every tensor that lives in HBM in bfloat16 and everything inside a kernel
# in float32: conv accumulators, the BatchNorm statistics (taken from the float32 accumulators), the normalisation
# arithmetic.  These two functions restate the forward with a round-to-nearest-even bf16 at exactly those storage
# points (gan-aug-pfa_amd/pfa_gan/engine.py: `raw`, `aL`/`aR`, `uR`, `fake`; the bf16 weight shadows), so that a bf16
# network-level parity test can be as tight as the per-kernel ones: what remains is the order of the float32 sums and
# the occasional rounding tie that falls the other way.
def _bf16(x):
    from .fill import bf16_round
    return bf16_round(np.asarray(x, F32))


def _bn_coeffs_train(acc, gamma, beta):
    a64 = acc.astype(np.float64)
    mean = a64.mean(axis=(0, 2, 3))
    var = a64.var(axis=(0, 2, 3))
    scale = (gamma.astype(np.float64) / np.sqrt(var + ops.BN_EPS)).astype(F32)
    shift = (beta.astype(np.float64) - mean * scale.astype(np.float64)).astype(F32)
    return scale, shift


def _affine(raw, scale, shift):
    return (raw.astype(F32) * scale[None, :, None, None] + shift[None, :, None, None]).astype(F32)


def generator_fwd_bf16(st, x, num_downs=7):
    """Train-mode UNetGenerator forward with the engine's bf16 storage points (see above).  Returns the tanh output."""
    lv = g_levels(num_downs)
    w = lambda k: _bf16(st[k])
    a_l = _bf16(x)                                   # operand of the next down-conv
    relu_skip = []                                   # ReLU view of every encoder activation (decoder operand)
    for k in range(num_downs):
        acc = ops.conv2d_fwd(a_l, w(lv[k]["down"] + ".weight"), None, 2)
        if k == num_downs - 1:                       # innermost: consumed through ReLU only
            relu_skip.append(_bf16(ops.relu(acc)))
            break
        if lv[k]["down_bn"]:
            scale, shift = _bn_coeffs_train(acc, st[lv[k]["down_bn"] + ".weight"], st[lv[k]["down_bn"] + ".bias"])
            y = _affine(_bf16(acc), scale, shift)    # the raw conv output is stored (bf16), then normalised
        else:
            y = acc                                  # outermost level: both views straight from the accumulators
        relu_skip.append(_bf16(ops.relu(y)))
        a_l = _bf16(ops.leaky_relu(y))
    u = None
    for k in range(num_downs - 1, 0, -1):
        src = relu_skip[k] if u is None else np.concatenate([relu_skip[k], u], axis=1)
        acc = ops.convT2d_fwd(src, w(lv[k]["up"] + ".weight"), None)
        scale, shift = _bn_coeffs_train(acc, st[lv[k]["up_bn"] + ".weight"], st[lv[k]["up_bn"] + ".bias"])
        u = _bf16(ops.relu(_affine(_bf16(acc), scale, shift)))
    src = np.concatenate([relu_skip[0], u], axis=1)
    acc = ops.convT2d_fwd(src, w(lv[0]["up"] + ".weight"), st[lv[0]["up"] + ".bias"])
    return _bf16(np.tanh(acc).astype(F32))


def discriminator_fwd_bf16(st, x, n_layers=3):
    """Train-mode NLayerDiscriminator forward with the engine's bf16 storage points.  Returns the logits."""
    layers = d_layers(n_layers)
    h = _bf16(x)
    for i, (ck, bk, s, has_b) in enumerate(layers):
        acc = ops.conv2d_fwd(h, _bf16(st[ck + ".weight"]), st[ck + ".bias"] if has_b else None, s)
        if bk:
            scale, shift = _bn_coeffs_train(acc, st[bk + ".weight"], st[bk + ".bias"])
            h = _bf16(ops.leaky_relu(_affine(_bf16(acc), scale, shift)))
        elif i < len(layers) - 1:
            h = _bf16(ops.leaky_relu(acc))
        else:
            h = _bf16(acc)
    return h
