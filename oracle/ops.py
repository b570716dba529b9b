"""numpy restatement of the PyTorch operators the reference hot path calls.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Layout is the reference's:
NCHW float32 activations, OIHW conv weights, IOHW transposed-conv weights.
Every function cites the reference call site whose arithmetic it restates.
All kernels in the reference path are 4x4, padding 1 (models.py:177,184,223).
"""
import numpy as np
from numpy.lib.stride_tricks import sliding_window_view

F32 = np.float32


def _windows(xp, stride, ho, wo):
    """[N,C,Hp,Wp] -> view [N,C,ho,wo,4,4] of 4x4 windows at the given stride."""
    w = sliding_window_view(xp, (4, 4), axis=(2, 3))
    return w[:, :, : stride * ho : stride, : stride * wo : stride]


# --------------------------------------------------------------------------- conv
def conv2d_fwd(x, w, b=None, stride=2):
    """nn.Conv2d(k=4, s=stride, p=1) forward (models.py:177 s=2; :223,:230 s=2; :238,:243 s=1).

    Cross-correlation: out[n,o,y,x] = sum_{c,ky,kx} in[n,c,s*y-1+ky,s*x-1+kx] * w[o,c,ky,kx].
    """
    n, c, h, wd = x.shape
    co = w.shape[0]
    ho = (h + 2 - 4) // stride + 1
    wo = (wd + 2 - 4) // stride + 1
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))
    cols = _windows(xp, stride, ho, wo).transpose(0, 2, 3, 1, 4, 5).reshape(n * ho * wo, c * 16)
    out = cols @ w.reshape(co, c * 16).T
    if b is not None:
        out = out + b[None, :]
    return np.ascontiguousarray(out.reshape(n, ho, wo, co).transpose(0, 3, 1, 2)).astype(F32)


def conv2d_bwd(x, w, dy, stride=2, need_dx=True):
    """Gradients of conv2d_fwd: (dx, dw, db) - what autograd computes for train_gan.py:62,70."""
    n, c, h, wd = x.shape
    co = w.shape[0]
    _, _, ho, wo = dy.shape
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))
    cols = _windows(xp, stride, ho, wo).transpose(0, 2, 3, 1, 4, 5).reshape(n * ho * wo, c * 16)
    dyf = dy.transpose(0, 2, 3, 1).reshape(n * ho * wo, co)
    dw = (dyf.T @ cols).reshape(co, c, 4, 4).astype(F32)
    db = dyf.sum(axis=0).astype(F32)
    dx = None
    if need_dx:
        dcols = (dyf @ w.reshape(co, c * 16)).reshape(n, ho, wo, c, 4, 4)
        dxp = np.zeros((n, c, h + 2, wd + 2), dtype=F32)
        for ky in range(4):
            for kx in range(4):
                dxp[:, :, ky : ky + stride * ho : stride, kx : kx + stride * wo : stride] += \
                    dcols[:, :, :, :, ky, kx].transpose(0, 3, 1, 2)
        dx = np.ascontiguousarray(dxp[:, :, 1:-1, 1:-1])
    return dx, dw, db


# ------------------------------------------------------------------ transposed conv
def convT2d_fwd(x, w, b=None):
    """nn.ConvTranspose2d(k=4, s=2, p=1) forward (models.py:184,189,194); w is [Ci,Co,4,4].

    out[n,o,2*iy-1+ky,2*ix-1+kx] += in[n,c,iy,ix] * w[c,o,ky,kx].
    """
    n, ci, h, wd = x.shape
    co = w.shape[1]
    xf = x.transpose(0, 2, 3, 1).reshape(n * h * wd, ci)
    cols = (xf @ w.reshape(ci, co * 16)).reshape(n, h, wd, co, 4, 4)
    outp = np.zeros((n, co, 2 * h + 2, 2 * wd + 2), dtype=F32)
    for ky in range(4):
        for kx in range(4):
            outp[:, :, ky : ky + 2 * h : 2, kx : kx + 2 * wd : 2] += cols[:, :, :, :, ky, kx].transpose(0, 3, 1, 2)
    out = outp[:, :, 1:-1, 1:-1]
    if b is not None:
        out = out + b[None, :, None, None]
    return np.ascontiguousarray(out).astype(F32)


def convT2d_bwd(x, w, dy, need_dx=True):
    """Gradients of convT2d_fwd: (dx, dw, db)."""
    n, ci, h, wd = x.shape
    co = w.shape[1]
    dyp = np.pad(dy, ((0, 0), (0, 0), (1, 1), (1, 1)))
    # dcols[n,iy,ix,o,ky,kx] = dy[n,o,2iy-1+ky,2ix-1+kx]
    dcols = _windows(dyp, 2, h, wd).transpose(0, 2, 3, 1, 4, 5).reshape(n * h * wd, co * 16)
    xf = x.transpose(0, 2, 3, 1).reshape(n * h * wd, ci)
    dw = (xf.T @ dcols).reshape(ci, co, 4, 4).astype(F32)
    db = dy.sum(axis=(0, 2, 3)).astype(F32)
    dx = None
    if need_dx:
        dx = (dcols @ w.reshape(ci, co * 16).T).reshape(n, h, wd, ci).transpose(0, 3, 1, 2)
        dx = np.ascontiguousarray(dx).astype(F32)
    return dx, dw, db


# ---------------------------------------------------------------------- batch norm
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def bn_train_fwd(x, gamma, beta, running_mean, running_var, nbt, repeats=1):
    """nn.BatchNorm2d training forward (models.py:179,181,231,239; eps 1e-5, momentum 0.1).

    Per-channel mean and BIASED variance over N*H*W normalise the batch; the
    running buffers take the UNBIASED variance.  `repeats` applies the buffer
    update that many times with the same batch statistics - the reference runs
    the generator forward twice per step on unchanged weights
    (train_gan.py:56,65).
    Returns (y, cache, new_running_mean, new_running_var, new_nbt).
    """
    n, c, h, w = x.shape
    cnt = n * h * w
    x64 = x.astype(np.float64)
    mean = x64.mean(axis=(0, 2, 3))
    var = x64.var(axis=(0, 2, 3))
    inv = 1.0 / np.sqrt(var + BN_EPS)
    xhat = ((x64 - mean[None, :, None, None]) * inv[None, :, None, None]).astype(F32)
    y = (xhat * gamma[None, :, None, None] + beta[None, :, None, None]).astype(F32)
    unb = var * (cnt / max(cnt - 1, 1))
    rm, rv = running_mean.astype(np.float64), running_var.astype(np.float64)
    for _ in range(repeats):
        rm = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean
        rv = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * unb
    cache = (xhat, inv.astype(F32))
    return y, cache, rm.astype(F32), rv.astype(F32), nbt + repeats


def bn_train_bwd(dy, gamma, cache):
    """Backward of training-mode BatchNorm2d: (dx, dgamma, dbeta)."""
    xhat, inv = cache
    n, c, h, w = dy.shape
    cnt = n * h * w
    dy64 = dy.astype(np.float64)
    xh64 = xhat.astype(np.float64)
    dbeta = dy64.sum(axis=(0, 2, 3))
    dgamma = (dy64 * xh64).sum(axis=(0, 2, 3))
    g = (gamma.astype(np.float64) * inv.astype(np.float64))[None, :, None, None]
    dx = g * (dy64 - dbeta[None, :, None, None] / cnt - xh64 * dgamma[None, :, None, None] / cnt)
    return dx.astype(F32), dgamma.astype(F32), dbeta.astype(F32)


def bn_eval_fwd(x, gamma, beta, running_mean, running_var):
    """nn.BatchNorm2d eval forward (generate_synthetic_data.py:55; train_gan.py:79)."""
    inv = 1.0 / np.sqrt(running_var.astype(np.float64) + BN_EPS)
    sc = (gamma * inv).astype(F32)
    sh = (beta - running_mean * gamma * inv).astype(F32)
    return (x * sc[None, :, None, None] + sh[None, :, None, None]).astype(F32)


# --------------------------------------------------------------------- activations
def leaky_relu(x, slope=0.2):
    """nn.LeakyReLU(0.2) (models.py:178,223,232,240)."""
    return np.where(x >= 0, x, x * F32(slope)).astype(F32)


def leaky_relu_bwd(x, dy, slope=0.2):
    return np.where(x > 0, dy, dy * F32(slope)).astype(F32)


def relu(x):
    """nn.ReLU (models.py:180)."""
    return np.maximum(x, 0).astype(F32)


def relu_bwd(x, dy):
    return np.where(x > 0, dy, 0).astype(F32)


# -------------------------------------------------------------------------- losses
def bce_logits(x, target):
    """nn.BCEWithLogitsLoss() mean reduction with a constant target (train_gan.py:42,58,60,67).

    loss = mean(max(x,0) - x*t + log1p(exp(-|x|)));  dloss/dx = (sigmoid(x) - t)/numel.
    """
    x64 = x.astype(np.float64)
    loss = np.mean(np.maximum(x64, 0) - x64 * target + np.log1p(np.exp(-np.abs(x64))))
    grad = (1.0 / (1.0 + np.exp(-x64)) - target) / x.size
    return F32(loss), grad.astype(F32)


def l1_loss(a, b):
    """nn.L1Loss() mean reduction (train_gan.py:43,68); dloss/da = sign(a-b)/numel."""
    d = a.astype(np.float64) - b.astype(np.float64)
    return F32(np.mean(np.abs(d))), (np.sign(d) / a.size).astype(F32)


# ---------------------------------------------------------------------------- Adam
def adam_step(p, g, m, v, t, lr=1e-4, b1=0.5, b2=0.999, eps=1e-8):
    """optim.Adam(lr=1e-4, betas=(0.5,0.999)) single-tensor update (train_gan.py:28-30,140-141).

    torch form: denom = sqrt(v)/sqrt(1-b2^t) + eps; p -= lr/(1-b1^t) * m/denom.
    `t` is the 1-based step count after this update.  Returns (p, m, v).
    """
    m = (b1 * m + (1 - b1) * g).astype(F32)
    v = (b2 * v + (1 - b2) * g * g).astype(F32)
    bc1 = 1 - b1 ** t
    bc2 = 1 - b2 ** t
    denom = np.sqrt(v) / F32(np.sqrt(bc2)) + F32(eps)
    p = (p - F32(lr / bc1) * (m / denom)).astype(F32)
    return p, m, v
