"""Helpers shared by the -m gpu parity tests: NCHW numpy <-> padded NHWC device tensors, weight shadows."""
import numpy as np
import torch

from oracle import fill
from pfa_gan import kernels as K

DTYPES = [torch.float32, torch.bfloat16]
DT_IDS = ["f32", "bf16"]


def tol(dtype):
    """(atol relative to the tensor's max magnitude, rtol).  f32: summation order only.
    bf16: operands are bf16-representable so products are exact in f32; the stored result is
    rounded to bf16 (2^-9 relative)."""
    return (2e-5, 2e-4) if dtype == torch.float32 else (6e-3, 1e-2)


def rnd(x, dtype):
    return fill.bf16_round(x) if dtype == torch.bfloat16 else np.asarray(x, np.float32)


def to_nhwc(x, dtype):
    """numpy NCHW -> cuda NHWC padded tensor, through the product's own pack kernel."""
    return K.pack_nhwc(torch.from_numpy(np.ascontiguousarray(x, np.float32)).cuda(), dtype)


def from_nhwc(t, c):
    torch.cuda.synchronize()
    return K.unpack_nchw(t, c).cpu().numpy()


def close(got, want, dtype, what="", scale=None):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    a, r = tol(dtype)
    s = scale if scale is not None else max(np.abs(want).max(), 1e-6)
    err = np.abs(got - want)
    lim = a * s + r * np.abs(want)
    if (err > lim).any():
        i = np.unravel_index(np.argmax(err - lim), err.shape)
        raise AssertionError(f"{what}: {(err > lim).sum()}/{err.size} outside tol; worst at {i}: "
                             f"got {got[i]:.6g} want {want[i]:.6g}; max err {err.max():.3e}, scale {s:.3g}")


def master_of(w):
    """OIHW / IOHW numpy weight -> float32 cuda master [rows][16][c] (channels-last memory of the parameter)."""
    r, c = w.shape[0], w.shape[1]
    return torch.from_numpy(np.ascontiguousarray(w.transpose(0, 2, 3, 1).reshape(r, 16, c))).cuda()


def shadows(w, dtype, cparts=None, want=("same", "tr")):
    """Device shadows of a master weight. cparts: [(real, padded), ...] split of the c axis."""
    m = master_of(w)
    rows, c = w.shape[0], w.shape[1]
    if cparts is None:
        cparts = [(c, K.pad_ch(c, dtype))]
    (c1r, c1p), (c2r, c2p) = cparts[0], (cparts[1] if len(cparts) > 1 else (0, 0))
    cp = c1p + c2p
    rows_p = K.pad_ch(rows, dtype)
    same = torch.empty((rows, 16, cp), dtype=dtype, device="cuda") if "same" in want else None
    tr = torch.empty((cp, 16, rows_p), dtype=dtype, device="cuda") if "tr" in want else None
    K.weight_shadow(m, rows, c1r, c1p, c2r, c2p, rows_p, same, tr)
    return same, tr
