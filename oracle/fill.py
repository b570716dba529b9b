"""Deterministic, torch-independent tensor filler (test infrastructure).

Full-size weights and inputs are never committed (the default generator's
state_dict is 167 MB): both the golden generator (which runs the reference) and
the parity tests regenerate them from a name and a seed with this SplitMix64
counter hash, and only expected outputs / digests are stored in tests/golden.
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def uniform(name, shape, lo=-1.0, hi=1.0, seed=0):
    """float32 array of `shape`, i.i.d. U[lo, hi), a pure function of (name, seed)."""
    n = int(np.prod(shape)) if len(shape) else 1
    key = np.uint64(zlib.crc32(name.encode()) | (int(seed) << 32))
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) + _splitmix64(key)
        bits = _splitmix64(ctr)
    # 24 random mantissa bits -> exactly representable float32 in [0, 1)
    u = (bits >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))
    out = (np.float32(lo) + u * np.float32(hi - lo)).astype(np.float32)
    return out.reshape(shape)


def bf16_round(x):
    """Round float32 -> nearest-even bfloat16, returned as float32 (bf16-representable)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + np.uint64(0x7FFF) + ((u >> np.uint64(16)) & np.uint64(1))) >> np.uint64(16)) << np.uint64(16)
    return (r & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.float32).reshape(x.shape)


def fill_state(shapes, seed=0, bn_noise=True):
    """Fill a reference-layout state dict {key: shape} -> {key: array}.

    Conv / ConvTranspose weights and biases get U(+-1/sqrt(fan_in)) with
    fan_in = weight.shape[1]*16 (what torch's default init uses for both,
    SURVEY.md a12); BN weight ~ U(0.5,1.5), BN bias ~ U(-0.5,0.5) when
    `bn_noise` (so gamma/beta paths are exercised), running_mean 0,
    running_var 1, num_batches_tracked 0.
    """
    out = {}
    fan = {}
    for k, shp in shapes.items():
        if k.endswith(".weight") and len(shp) == 4:
            fan[k[: -len(".weight")]] = shp[1] * 16
    for k, shp in shapes.items():
        mod = k.rsplit(".", 1)[0]
        if k.endswith("num_batches_tracked"):
            out[k] = np.zeros((), dtype=np.int64)
        elif k.endswith("running_mean"):
            out[k] = np.zeros(shp, dtype=np.float32)
        elif k.endswith("running_var"):
            out[k] = np.ones(shp, dtype=np.float32)
        elif mod in fan:
            b = 1.0 / np.sqrt(fan[mod])
            out[k] = uniform(k, shp, -b, b, seed)
        elif k.endswith(".weight"):
            out[k] = uniform(k, shp, 0.5, 1.5, seed) if bn_noise else np.ones(shp, np.float32)
        else:
            out[k] = uniform(k, shp, -0.5, 0.5, seed) if bn_noise else np.zeros(shp, np.float32)
    return out


def compact(a, full_max=4096, nsample=2048):
    """Fixture-size policy shared by the golden generator and the tests: arrays up to
    `full_max` elements are kept whole, larger ones as a strided sample of `nsample`."""
    a = np.asarray(a)
    if a.size <= full_max:
        return a.copy()
    return a.reshape(-1)[:: a.size // nsample][:nsample].copy()
