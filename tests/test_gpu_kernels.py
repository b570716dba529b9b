"""-m gpu: every libpfa_hip kernel against the numpy oracle (oracle/ops.py), through the C ABI.

f32 mode must sit far inside the north-star gate (1e-3 abs / 1e-2 rel); bf16 mode is checked on
bf16-representable operands (products exact in f32, fp32 accumulation, one bf16 rounding of the result).
"""
import numpy as np
import pytest
import torch

from oracle import fill, ops
from pfa_gan import _lib, kernels as K
from gpu_util import DTYPES, DT_IDS, close, from_nhwc, master_of, rnd, shadows, to_nhwc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ws():
    return K.Workspace(torch.device("cuda"))


def test_library_is_the_path():
    assert _lib.lib().pfa_version() >= 100
    before = _lib.launch_count()
    x = torch.zeros(1, 3, 4, 4, device="cuda")
    K.pack_nhwc(x, torch.bfloat16)
    assert _lib.launch_count() == before + 1


@pytest.mark.parametrize("dtype", DTYPES, ids=DT_IDS)
def test_pack_unpack_and_shadows(dtype):
    x = rnd(fill.uniform("pk.x", (2, 3, 5, 7)), dtype)
    t = to_nhwc(x, dtype)
    assert t.shape == (2, 5, 7, K.pad_ch(3, dtype))
    assert float(t[..., 3:].abs().max()) == 0.0
    np.testing.assert_array_equal(from_nhwc(t, 3), x)
    w = rnd(fill.uniform("pk.w", (10, 6, 4, 4)), dtype)
    cp = K.pad_ch(3, dtype)
    same, tr = shadows(w, dtype, cparts=[(3, cp), (3, cp)])
    torch.cuda.synchronize()
    ref = np.zeros((10, 16, 2 * cp), np.float32)
    m = w.transpose(0, 2, 3, 1).reshape(10, 16, 6)
    ref[:, :, :3] = m[:, :, :3]
    ref[:, :, cp:cp + 3] = m[:, :, 3:]
    np.testing.assert_array_equal(same.float().cpu().numpy(), ref)
    rp = K.pad_ch(10, dtype)
    ref_tr = np.zeros((2 * cp, 16, rp), np.float32)
    ref_tr[:, :, :10] = ref.transpose(2, 1, 0)
    np.testing.assert_array_equal(tr.float().cpu().numpy(), ref_tr)


# ---------------------------------------------------------------------------- forward convolutions
CONV_FWD = [  # name, N, cin parts, H, W, Co, stride, bias, act, splitk
    ("e1_like", 2, [3], 16, 16, 16, 2, False, K.ACT_NONE, None),
    ("c1_like_cat", 2, [3, 3], 16, 16, 64, 2, True, K.ACT_LRELU, None),
    ("mid_16_32", 2, [16], 8, 8, 32, 2, False, K.ACT_NONE, None),
    ("bn128_tiles", 2, [64], 32, 32, 128, 2, False, K.ACT_NONE, 1),
    ("bn128_edge", 3, [32], 10, 14, 136, 2, False, K.ACT_NONE, 1),
    ("splitk4", 2, [64], 16, 16, 64, 2, False, K.ACT_RELU, 4),
    ("auto_split_deep", 2, [128], 4, 4, 128, 2, False, K.ACT_NONE, None),
    ("s1_c4_like", 2, [32], 9, 9, 64, 1, False, K.ACT_NONE, None),
    ("s1_odd31", 1, [16], 32, 32, 32, 1, False, K.ACT_NONE, 1),
    ("s1_c5_like", 2, [64], 7, 6, 1, 1, True, K.ACT_NONE, None),
    ("one_by_one", 2, [16], 2, 2, 16, 2, False, K.ACT_NONE, None),
]


@pytest.mark.parametrize("dtype", DTYPES, ids=DT_IDS)
@pytest.mark.parametrize("case", CONV_FWD, ids=[c[0] for c in CONV_FWD])
def test_conv_forward(case, dtype, ws):
    name, n, cins, h, w, co, stride, bias, act, splitk = case
    ci = sum(cins)
    x = rnd(fill.uniform(name + ".x", (n, ci, h, w)), dtype)
    wt = rnd(fill.uniform(name + ".w", (co, ci, 4, 4), -0.25, 0.25), dtype)
    b = fill.uniform(name + ".b", (co,), -0.5, 0.5) if bias else None
    want = ops.conv2d_fwd(x, wt, b, stride)
    raw = want.copy()
    if act == K.ACT_LRELU:
        want = ops.leaky_relu(want)
    elif act == K.ACT_RELU:
        want = ops.relu(want)
    parts, srcs, o = [], [], 0
    for c in cins:
        srcs.append(to_nhwc(x[:, o:o + c], dtype))
        parts.append((c, K.pad_ch(c, dtype)))
        o += c
    same, _ = shadows(wt, dtype, cparts=parts, want=("same",))
    ho, wo = want.shape[2:]
    out = torch.full((n, ho, wo, K.pad_ch(co, dtype)), 7.0, dtype=dtype, device="cuda")
    bt = torch.from_numpy(b).cuda() if bias else None
    before = _lib.launch_count()
    st = K.gather_gemm(K.CONV_S2 if stride == 2 else K.CONV_S1, srcs, same, co, [out], (n, ho, wo), ws, bias=bt,
                       act=act, stats=True, splitk=splitk)
    assert _lib.launch_count() > before
    close(from_nhwc(out, co), want, dtype, name)
    if K.pad_ch(co, dtype) > co:
        assert float(out[..., co:].float().abs().max()) == 0.0, "pad channels must be written as zeros"
    psum, psq, nt = st
    torch.cuda.synchronize()
    s = psum[: nt * co].view(nt, co).double().sum(0).cpu().numpy()
    q = psq[: nt * co].view(nt, co).double().sum(0).cpu().numpy()
    close(s, raw.astype(np.float64).sum(axis=(0, 2, 3)), torch.float32, name + ".psum",
          scale=np.abs(raw).sum(axis=(0, 2, 3)).max())
    close(q, (raw.astype(np.float64) ** 2).sum(axis=(0, 2, 3)), torch.float32, name + ".psq")


@pytest.mark.parametrize("dtype", DTYPES, ids=DT_IDS)
@pytest.mark.parametrize("splitk", [1, 2], ids=["gemm_epilogue", "finish_pass"])
def test_relu_epilogue_of_an_overflowed_sum_is_zero(dtype, splitk, ws):
    """ReLU(-inf) = 0 as torch.relu gives it (models.py:180), not -inf * 0 = NaN: a conv output that overflowed must not
    poison what reads it.  LeakyReLU(-inf) stays -inf."""
    n, ci, h, w, co = 1, 64, 8, 8, 16
    x = np.abs(rnd(fill.uniform("inf.x", (n, ci, h, w)), dtype)) + 0.5
    x[0, 3, 2, 5] = -np.inf
    wt = np.abs(rnd(fill.uniform("inf.w", (co, ci, 4, 4), -0.25, 0.25), dtype)) + 0.125
    raw = ops.conv2d_fwd(x, wt, None, 2)
    assert np.isneginf(raw).any() and np.isfinite(raw).any()
    same, _ = shadows(wt, dtype, want=("same",))
    for act, want in ((K.ACT_RELU, ops.relu(raw)), (K.ACT_LRELU, np.where(raw > 0, raw, raw * np.float32(K.SLOPE)))):
        out = torch.full((n, h // 2, w // 2, K.pad_ch(co, dtype)), 7.0, dtype=dtype, device="cuda")
        K.gather_gemm(K.CONV_S2, [to_nhwc(x, dtype)], same, co, [out], (n, h // 2, w // 2), ws, act=act, splitk=splitk)
        got = from_nhwc(out, co)
        assert not np.isnan(got).any(), f"act {act}: NaN in the output"
        assert np.array_equal(np.isneginf(got), np.isneginf(want))
        fin = np.isfinite(want)
        close(got[fin], want[fin], dtype, f"act {act}")


PATCH_CASES = [  # name, N, Ci, H, W, Co, K slices
    ("c4_like", 2, 64, 17, 19, 128, 1),       # ragged patch rows (16 -> two 8-row tiles) and columns (18 of 32)
    ("full_width", 3, 128, 12, 33, 256, 1),   # 32 output columns: the whole patch width; two channel chunks, two n-tiles
    ("one_row", 1, 256, 2, 5, 128, 1),        # a single output row
    ("tall", 1, 64, 40, 9, 128, 1),           # five row tiles, the last one ragged
    ("sliced", 2, 256, 10, 13, 128, 2),       # two K slices of two channel chunks each -> slab + finish pass
    ("wide", 1, 64, 11, 71, 128, 1),          # 70 output columns: three patch columns, the last one ragged (512x512 inputs)
]


@pytest.mark.parametrize("grad", [False, True], ids=["forward", "input_gradient"])
def test_conv_s1_inplace_64_channel_tiles(grad, ws):
    """The stride-1 patch kernel on 64-channel tiles (round 4: c4's input gradient un-sliced on 256 workgroups instead of
    K-sliced through float32 partial planes; pfa_set_knob(3, 7) takes them for these small shapes): forward and input
    gradient against the oracle and against the implicit GEMM, ragged tiles, two channel chunks, three 64-channel tiles."""
    dtype = torch.bfloat16
    n, ci, h, w, co = 2, 128, 17, 35, 192
    lib = _lib.lib()
    if not grad:
        x = rnd(fill.uniform("s64.x", (n, ci, h, w)), dtype)
        wt = rnd(fill.uniform("s64.w", (co, ci, 4, 4), -0.25, 0.25), dtype)
        want = ops.conv2d_fwd(x, wt, None, 1)
        wsh, _ = shadows(wt, dtype, cparts=[(ci, ci)], want=("same",))
        src, mode, cout = to_nhwc(x, dtype), K.CONV_S1, co
    else:
        x0 = fill.uniform("s64.gx", (n, ci, h, w))
        wt = rnd(fill.uniform("s64.wg", (co, ci, 4, 4), -0.25, 0.25), dtype)
        dy = rnd(fill.uniform("s64.dy", (n, co, h - 1, w - 1)), dtype)
        want, _, _ = ops.conv2d_bwd(x0, wt, dy, 1)
        _, wsh = shadows(wt, dtype, want=("tr",))
        src, mode, cout = to_nhwc(dy, dtype), K.CONV_S1_DGRAD, ci
    ho, wo = want.shape[2:]
    outs = []
    for knob in (7, 1):
        out = torch.full((n, ho, wo, cout), 7.0, dtype=dtype, device="cuda")
        lib.pfa_set_knob(3, knob)
        try:
            K.gather_gemm(mode, [src], wsh, cout, [out], (n, ho, wo), ws)
            kernel = lib.pfa_last_gather_kernel().decode()
        finally:
            lib.pfa_set_knob(3, 0)
        assert kernel == ("k_conv_s1_inplace<bf16,8x32,64>" if knob == 7 else kernel) and (knob == 7 or kernel.startswith("k_gather_gemm")), kernel
        outs.append(from_nhwc(out, cout))
    close(outs[0], want, dtype, "s1 64-channel tiles")
    close(outs[0], outs[1], dtype, "s1 64-channel tiles vs implicit GEMM", scale=np.abs(want).max())


@pytest.mark.parametrize("case", PATCH_CASES, ids=[c[0] for c in PATCH_CASES])
def test_conv_s1_patch_kernel(case, ws):
    """Stride-1 forward convolutions (the discriminator's c4) take the patch kernel of conv_patch.hip when they are large
    enough; pfa_set_knob(3, 2) routes these small ones there too.  Same oracle, same statistics contract as the
    implicit GEMM, and the implicit GEMM's result up to the order of the float32 sums."""
    name, n, ci, h, w, co, slices = case
    dtype = torch.bfloat16
    x = rnd(fill.uniform(name + ".px", (n, ci, h, w)), dtype)
    wt = rnd(fill.uniform(name + ".pw", (co, ci, 4, 4), -0.25, 0.25), dtype)
    want = ops.conv2d_fwd(x, wt, None, 1)
    same, _ = shadows(wt, dtype, cparts=[(ci, ci)], want=("same",))
    ho, wo = want.shape[2:]
    lib = _lib.lib()
    outs, stats = [], []
    for knob in (2, 1):                      # patch kernel, then the implicit GEMM
        out = torch.full((n, ho, wo, co), 7.0, dtype=dtype, device="cuda")
        lib.pfa_set_knob(3, knob)
        try:
            psum, psq, nt = K.gather_gemm(K.CONV_S1, [to_nhwc(x, dtype)], same, co, [out], (n, ho, wo), ws, stats=True,
                                          splitk=slices)
            kernel = lib.pfa_last_gather_kernel().decode()
        finally:
            lib.pfa_set_knob(3, 0)
        assert kernel.startswith("k_conv_s1_") == (knob == 2), kernel
        torch.cuda.synchronize()
        outs.append(out)
        stats.append((psum[: nt * co].view(nt, co).double().sum(0).cpu().numpy(), psq[: nt * co].view(nt, co).double().sum(0).cpu().numpy()))
    close(from_nhwc(outs[0], co), want, dtype, name)
    close(from_nhwc(outs[0], co), from_nhwc(outs[1], co), dtype, name + " vs implicit GEMM", scale=np.abs(want).max())
    close(stats[0][0], want.astype(np.float64).sum(axis=(0, 2, 3)), torch.float32, name + ".psum", scale=np.abs(want).sum(axis=(0, 2, 3)).max())
    close(stats[0][1], (want.astype(np.float64) ** 2).sum(axis=(0, 2, 3)), torch.float32, name + ".psq")


PATCH_GRAD_CASES = [  # name, N, Ci (columns of the result), H, W, Co (contraction), K slices
    ("c4_grad_like", 2, 128, 17, 19, 64, 1),
    ("full_width", 2, 128, 12, 32, 128, 2),     # 32 result columns; two K slices
    ("c4_grad_sliced", 3, 256, 9, 8, 256, 4),
    ("wide", 1, 128, 10, 40, 64, 1),            # 40 result columns: two patch columns
]


@pytest.mark.parametrize("case", PATCH_GRAD_CASES, ids=[c[0] for c in PATCH_GRAD_CASES])
def test_conv_s1_patch_kernel_input_gradient(case, ws):
    """The input gradient of a stride-1 convolution on the patch kernel: the same walk over the (pad 2) patch with the taps
    reversed; K slices go through the slab + finish pass."""
    name, n, ci, h, w, co, slices = case
    dtype = torch.bfloat16
    x = fill.uniform(name + ".gx", (n, ci, h, w))
    wt = rnd(fill.uniform(name + ".gw", (co, ci, 4, 4), -0.25, 0.25), dtype)
    dy = rnd(fill.uniform(name + ".gdy", (n, co, h - 1, w - 1)), dtype)
    want, _, _ = ops.conv2d_bwd(x, wt, dy, 1)
    _, tr = shadows(wt, dtype, want=("tr",))
    lib = _lib.lib()
    outs = []
    for knob in (2, 1):
        out = torch.full((n, h, w, ci), 7.0, dtype=dtype, device="cuda")
        lib.pfa_set_knob(3, knob)
        try:
            K.gather_gemm(K.CONV_S1_DGRAD, [to_nhwc(dy, dtype)], tr, ci, [out], (n, h, w), ws, splitk=slices)
            kernel = lib.pfa_last_gather_kernel().decode()
        finally:
            lib.pfa_set_knob(3, 0)
        assert kernel.startswith("k_conv_s1_") == (knob == 2), kernel
        outs.append(out)
    close(from_nhwc(outs[0], ci), want, dtype, name)
    close(from_nhwc(outs[0], ci), from_nhwc(outs[1], ci), dtype, name + " vs implicit GEMM", scale=np.abs(want).max())


IMAGE_CASES = [  # name, N, real channels per source, H, W, bias + LeakyReLU, two outputs through the identity affine map
    ("e1_like", 2, [3], 64, 64, False, True),            # generator's first conv: 3 -> 64, LeakyReLU and ReLU views
    ("e1_ragged", 1, [3], 36, 80, False, True),          # output 18 x 40: ragged tile rows and columns
    ("c1_like", 2, [3, 3], 32, 64, True, False),         # discriminator's first conv: (A | B) -> 64, bias, LeakyReLU
    ("c1_ragged", 3, [3, 3], 20, 36, True, False),
]


@pytest.mark.parametrize("case", IMAGE_CASES, ids=[c[0] for c in IMAGE_CASES])
def test_conv_image_kernel(case, ws):
    """The image-side stride-2 convolutions (models.py:183, :223) on conv_image.hip: the oracle, and the implicit GEMM's
    result (pfa_set_knob(3, 1)) up to the order of the float32 sums."""
    name, n, cins, h, w, biased, dual = case
    dtype = torch.bfloat16
    ci, co = sum(cins), 64
    x = rnd(fill.uniform(name + ".ix", (n, ci, h, w)), dtype)
    wt = rnd(fill.uniform(name + ".iw", (co, ci, 4, 4), -0.25, 0.25), dtype)
    b = fill.uniform(name + ".ib", (co,), -0.5, 0.5) if biased else None
    raw = ops.conv2d_fwd(x, wt, b, 2)
    srcs, o = [], 0
    for c in cins:
        srcs.append(to_nhwc(x[:, o:o + c], dtype))
        o += c
    wsh, _ = shadows(wt, dtype, cparts=[(c, K.pad_ch(c, dtype)) for c in cins], want=("same",))
    ho, wo = raw.shape[2:]
    lib = _lib.lib()
    res = []
    for knob in (0, 1):
        outs = [torch.full((n, ho, wo, co), 7.0, dtype=dtype, device="cuda") for _ in range(2 if dual else 1)]
        lib.pfa_set_knob(3, knob)
        try:
            if dual:
                one, zero = torch.ones(co, device="cuda"), torch.zeros(co, device="cuda")
                K.gather_gemm(K.CONV_S2, srcs, wsh, co, outs, (n, ho, wo), ws, act=K.ACT_LRELU, affine=(one, zero, K.ACT_RELU))
            else:
                K.gather_gemm(K.CONV_S2, srcs, wsh, co, outs, (n, ho, wo), ws, bias=torch.from_numpy(b).cuda(), act=K.ACT_LRELU)
            kernel = lib.pfa_last_gather_kernel().decode()
        finally:
            lib.pfa_set_knob(3, 0)
        assert kernel.startswith("k_conv_image") == (knob == 0), kernel
        res.append([from_nhwc(t, co) for t in outs])
    close(res[0][0], ops.leaky_relu(raw), dtype, name + " lrelu", scale=np.abs(raw).max())
    if dual:
        close(res[0][1], ops.relu(raw), dtype, name + " relu", scale=np.abs(raw).max())
    for a, bb in zip(res[0], res[1]):
        close(a, bb, dtype, name + " vs implicit GEMM", scale=np.abs(raw).max())


PATCH2_CASES = [  # name, kind, N, cin parts, H, W (input), Co, K slices, knob 6 (workgroups wanted), kernel, output split
    ("tconv_cat", "tconv", 2, [64, 64], 9, 33, 128, 1, 1, "k_conv_patch2<tconv,128,8x32>", None),       # ragged 8x32 tiles, virtual concat
    ("tconv_16", "tconv", 2, [128], 16, 16, 64, 1, 0, "k_conv_patch2<tconv,64,16x16>", None),          # 16x16 tile, 64-channel tile
    ("tconv_17", "tconv", 1, [64], 17, 20, 128, 1, 1, "k_conv_patch2<tconv,128,16x16>", None),         # ragged 16x16 tiles
    ("tconv_sliced", "tconv", 2, [128, 64], 16, 16, 64, 3, 0, "k_conv_patch2<tconv,64,16x16>", None),  # three K slices -> slab + finish
    ("tconv_wide", "tconv", 1, [64], 8, 70, 64, 1, 0, "k_conv_patch2<tconv,64,8x32>", None),           # three tile columns
    # four parity classes per workgroup (conv_tconv4.hip; knob 3 = 5 takes it whatever the launch size)
    ("t4_cat", "tconv", 2, [64, 64], 9, 33, 64, 1, 0, "k_tconv4<64,8x32>", None),                      # ragged tiles, two sources, 2 chunks
    ("t4_two_ntiles", "tconv", 1, [64], 8, 40, 128, 1, 0, "k_tconv4<64,8x32>", None),                  # two 64-channel tiles, one chunk
    ("t4_chunks3", "tconv", 1, [128, 64], 11, 32, 64, 1, 0, "k_tconv4<64,8x32>", None),                # an odd number of chunks
    ("conv_e2", "conv", 1, [64], 34, 70, 128, 1, 1, "k_conv_patch2<conv_s2,128,8x32>", None),          # out 17x35: ragged both ways
    ("conv_16", "conv", 2, [128], 32, 32, 192, 1, 0, "k_conv_patch2<conv_s2,64,16x16>", None),         # 3 n-tiles of 64
    ("conv_split_out", "conv", 2, [64], 32, 64, 128, 1, 1, "k_conv_patch2<conv_s2,128,8x32>", 64),    # two outputs (decoder dgrad)
    ("conv_sliced", "conv", 1, [128], 32, 32, 64, 4, 0, "k_conv_patch2<conv_s2,64,16x16>", None),      # 8 super-steps in 4 slices
    ("conv_cat", "conv", 1, [64, 128], 36, 32, 64, 1, 0, "k_conv_patch2<conv_s2,64,16x16>", None),     # two sources of unequal width
    # 8 x 8 position maps: four images per tile (the generator's fifth level, e5 / d5 and their input gradients)
    ("conv_img4", "conv", 6, [64], 16, 16, 128, 1, 1, "k_conv_patch2<conv_s2,128,4x8x8>", None),       # 6 images: a ragged second tile
    ("conv_img4_sliced", "conv", 4, [128], 16, 16, 128, 4, 0, "k_conv_patch2<conv_s2,128,4x8x8>", None),   # 8 super-steps in 4 slices
    ("conv_img4_64", "conv", 5, [64, 64], 16, 16, 64, 1, 1, "k_conv_patch2<conv_s2,64,4x8x8>", None),  # 64-channel tile, two sources
    ("conv_img4_split", "conv", 4, [64], 16, 16, 256, 2, 0, "k_conv_patch2<conv_s2,128,4x8x8>", 128),  # two outputs (d5's input gradient)
    ("tconv_img4", "tconv", 5, [64, 64], 8, 8, 128, 1, 1, "k_conv_patch2<tconv,128,4x8x8>", None),     # virtual concat, ragged tile
    ("tconv_img4_sliced", "tconv", 4, [128, 64], 8, 8, 128, 3, 0, "k_conv_patch2<tconv,128,4x8x8>", None),   # 3 chunks in 3 slices, unequal sources
]


@pytest.mark.parametrize("case", PATCH2_CASES, ids=[c[0] for c in PATCH2_CASES])
def test_conv_patch2_kernels(case, ws):
    """Stride-2 convolutions and transposed convolutions on the patch kernels of conv_patch2.hip (pfa_set_knob(3, 2) routes
    launches this small there): the oracle, the statistics contract, and the implicit GEMM's result (knob 3 = 1) up to
    the order of the float32 sums - ragged tiles, both tile shapes, both channel tiles, K slices, the virtual concat
    and the split output of a decoder input gradient."""
    name, kind, n, cins, h, w, co, slices, knob6, kernel_want, n1 = case
    dtype = torch.bfloat16
    ci = sum(cins)
    x = rnd(fill.uniform(name + ".x", (n, ci, h, w)), dtype)
    srcs, o = [], 0
    for c in cins:
        srcs.append(to_nhwc(x[:, o:o + c], dtype))
        o += c
    if kind == "conv":
        wt = rnd(fill.uniform(name + ".w", (co, ci, 4, 4), -0.25, 0.25), dtype)
        want = ops.conv2d_fwd(x, wt, None, 2)
        wsh, _ = shadows(wt, dtype, cparts=[(c, c) for c in cins], want=("same",))
        mode, grid = K.CONV_S2, (n,) + want.shape[2:]
    else:
        wt = rnd(fill.uniform(name + ".w", (ci, co, 4, 4), -0.25, 0.25), dtype)
        want = ops.convT2d_fwd(x, wt, None)
        _, wsh = shadows(wt, dtype, want=("tr",))
        mode, grid = K.TCONV_S2, (n, h, w)
    lib = _lib.lib()
    res, stats = [], []
    for knob3 in (5 if kernel_want.startswith("k_tconv4") else 2, 1):
        shape = (n,) + want.shape[2:]
        outs = [torch.full(shape + (co,), 7.0, dtype=dtype, device="cuda")] if n1 is None else \
            [torch.full(shape + (n1,), 7.0, dtype=dtype, device="cuda"), torch.full(shape + (co - n1,), 7.0, dtype=dtype, device="cuda")]
        lib.pfa_set_knob(3, knob3)
        lib.pfa_set_knob(6, knob6)
        try:
            st = K.gather_gemm(mode, srcs, wsh, co, outs, grid, ws, stats=n1 is None, splitk=slices)
            kernel = lib.pfa_last_gather_kernel().decode()
        finally:
            lib.pfa_set_knob(3, 0)
            lib.pfa_set_knob(6, 0)
        assert (kernel == kernel_want) if knob3 != 1 else kernel.startswith("k_gather_gemm"), kernel
        got = from_nhwc(outs[0], outs[0].shape[3]) if n1 is None else \
            np.concatenate([from_nhwc(outs[0], n1), from_nhwc(outs[1], co - n1)], axis=1)
        res.append(got)
        if st is not None:
            psum, psq, nt = st
            torch.cuda.synchronize()
            stats.append((psum[: nt * co].view(nt, co).double().sum(0).cpu().numpy(),
                          psq[: nt * co].view(nt, co).double().sum(0).cpu().numpy()))
    close(res[0], want, dtype, name)
    close(res[0], res[1], dtype, name + " vs implicit GEMM", scale=np.abs(want).max())
    if stats:
        close(stats[0][0], want.astype(np.float64).sum(axis=(0, 2, 3)), torch.float32, name + ".psum",
              scale=np.abs(want).sum(axis=(0, 2, 3)).max())
        close(stats[0][1], (want.astype(np.float64) ** 2).sum(axis=(0, 2, 3)), torch.float32, name + ".psq")


CONVT_FWD = [  # name, N, cin parts, H, W, Co, bias, act, splitk
    ("d7_like", 2, [16], 1, 1, 16, False, K.ACT_NONE, None),
    ("mid_cat", 2, [16, 16], 4, 4, 8, False, K.ACT_NONE, None),
    ("d1_like", 2, [16, 16], 8, 8, 3, True, K.ACT_TANH, None),
    ("rect", 1, [8], 3, 5, 24, False, K.ACT_NONE, 1),
    ("big", 2, [64, 64], 16, 16, 64, False, K.ACT_NONE, 1),
    ("deep_split", 2, [128], 2, 2, 128, False, K.ACT_NONE, None),
]


@pytest.mark.parametrize("dtype", DTYPES, ids=DT_IDS)
@pytest.mark.parametrize("case", CONVT_FWD, ids=[c[0] for c in CONVT_FWD])
def test_convT_forward(case, dtype, ws):
    name, n, cins, h, w, co, bias, act, splitk = case
    ci = sum(cins)
    x = rnd(fill.uniform(name + ".x", (n, ci, h, w)), dtype)
    wt = rnd(fill.uniform(name + ".w", (ci, co, 4, 4), -0.25, 0.25), dtype)
    b = fill.uniform(name + ".b", (co,), -0.5, 0.5) if bias else None
    raw = ops.convT2d_fwd(x, wt, b)
    want = np.tanh(raw) if act == K.ACT_TANH else raw
    srcs, o = [], 0
    for c in cins:
        srcs.append(to_nhwc(x[:, o:o + c], dtype))
        o += c
    # master [Ci][16][Co]; the forward operand is its transpose [Co_p][16][Ci]
    _, tr = shadows(wt, dtype, want=("tr",))
    out = torch.full((n, 2 * h, 2 * w, K.pad_ch(co, dtype)), 7.0, dtype=dtype, device="cuda")
    bt = torch.from_numpy(b).cuda() if bias else None
    st = K.gather_gemm(K.TCONV_S2, srcs, tr, co, [out], (n, h, w), ws, bias=bt, act=act, stats=True, splitk=splitk)
    close(from_nhwc(out, co), want, dtype, name)
    psum, psq, nt = st
    torch.cuda.synchronize()
    s = psum[: nt * co].view(nt, co).double().sum(0).cpu().numpy()
    close(s, raw.astype(np.float64).sum(axis=(0, 2, 3)), torch.float32, name + ".psum",
          scale=np.abs(raw).sum(axis=(0, 2, 3)).max())


RING_LOOPS = [  # knob 7 value, K-loop template parameter of the 256x128 tile
    (0, 3),     # the default 3-stage ring
    (4, 4),     # half-step software-pipelined ring
    (6, 5),     # staggered ring: the two waves of a SIMD in opposite read / MFMA phases
    (7, 6),     # staggered ring, one phase pair per 32-wide half
]
RING_CASES = [  # name, kind, N, cin parts, H, W, Co, stride, splitk
    ("conv_s2", "conv", 2, [64], 32, 32, 128, 2, 1),
    ("conv_s2_splitk_ragged", "conv", 3, [128], 18, 22, 200, 2, 2),
    ("conv_s1", "conv", 2, [64], 17, 19, 128, 1, 1),
    ("tconv_cat", "tconv", 2, [64, 64], 9, 16, 136, 2, 1),
]


@pytest.mark.parametrize("loop", RING_LOOPS, ids=[f"loop{e}" for _, e in RING_LOOPS])
@pytest.mark.parametrize("case", RING_CASES, ids=[c[0] for c in RING_CASES])
def test_ring_loops_of_the_256x128_tile(case, loop, ws):
    """Every K-loop variant of the dominant kernel (the 8-wave 256x128 tile with three LDS stages) on shapes with ragged
    rows / columns, K slices and the virtual concat; pfa_set_knob(6, 1) admits the tile on launches this small."""
    name, kind, n, cins, h, w, co, stride, splitk = case
    knob7, early = loop
    dtype = torch.bfloat16
    ci = sum(cins)
    x = rnd(fill.uniform(name + ".x", (n, ci, h, w)), dtype)
    srcs, o = [], 0
    for c in cins:
        srcs.append(to_nhwc(x[:, o:o + c], dtype))
        o += c
    if kind == "conv":
        wt = rnd(fill.uniform(name + ".w", (co, ci, 4, 4), -0.25, 0.25), dtype)
        want = ops.conv2d_fwd(x, wt, None, stride)
        wsh, _ = shadows(wt, dtype, cparts=[(c, K.pad_ch(c, dtype)) for c in cins], want=("same",))
        mode, grid = (K.CONV_S2 if stride == 2 else K.CONV_S1), (n,) + want.shape[2:]
    else:
        wt = rnd(fill.uniform(name + ".w", (ci, co, 4, 4), -0.25, 0.25), dtype)
        want = ops.convT2d_fwd(x, wt, None)
        _, wsh = shadows(wt, dtype, want=("tr",))
        mode, grid = K.TCONV_S2, (n, h, w)
    out = torch.full((n,) + want.shape[2:] + (K.pad_ch(co, dtype),), 7.0, dtype=dtype, device="cuda")
    lib = _lib.lib()
    lib.pfa_set_knob(6, 1)
    lib.pfa_set_knob(7, knob7)
    try:
        st = K.gather_gemm(mode, srcs, wsh, co, [out], grid, ws, stats=True, splitk=splitk)
        assert lib.pfa_last_gather_kernel().decode() == f"k_gather_gemm<bf16,256,128,4,2,{early}>"
    finally:
        lib.pfa_set_knob(6, 0)
        lib.pfa_set_knob(7, 0)
    close(from_nhwc(out, co), want, dtype, name)
    psum, _, nt = st
    torch.cuda.synchronize()
    s = psum[: nt * co].view(nt, co).double().sum(0).cpu().numpy()
    close(s, want.astype(np.float64).sum(axis=(0, 2, 3)), torch.float32, name + ".psum",
          scale=np.abs(want).sum(axis=(0, 2, 3)).max())


ROWS_CASES = [  # name, kind, N, cin parts, H, W (input), Co, stride, K slices
    ("e6_like", "conv", 16, [128], 8, 8, 128, 2, 4),             # 256 rows = one tile, 32 K-steps in 4 slices
    ("e7_like", "conv", 5, [64], 4, 4, 128, 2, 2),               # 20 rows of 256 (2 x 2 maps), taps mostly outside
    ("one_by_one", "conv", 3, [64], 2, 2, 128, 2, 1),            # 1 x 1 output maps: only the four centre taps see pixels
    ("d6_like", "tconv", 4, [128, 64], 4, 4, 192, 2, 3),         # four parity classes, virtual concat of unequal widths, 3 column tiles
    ("d7_like", "tconv", 6, [64], 2, 2, 128, 2, 1),              # 2 x 2 source maps
    ("s1", "conv", 2, [64], 9, 11, 128, 1, 1),                   # stride 1
    ("two_row_tiles", "conv", 2, [64, 64], 32, 32, 128, 2, 5),   # 512 rows = two tiles, concat, 32 K-steps in 5 slices of 7 (the last: 4)
]


@pytest.mark.parametrize("case", ROWS_CASES, ids=[c[0] for c in ROWS_CASES])
def test_conv_rows_kernel(case, ws):
    """The implicit GEMM over gathered rows with the patch kernels' K loop (conv_rows.hip, round 4: the small maps of the
    generator's inner levels and whatever the patch planners turn down; opt-in: knob 1 bit 16384): the oracle, the statistics
    contract, and the ring tile of k_gather_gemm up to the order of the float32 sums.  pfa_set_knob(6, 1) admits the 256-row
    tile on launches this small."""
    name, kind, n, cins, h, w, co, stride, splitk = case
    dtype = torch.bfloat16
    ci = sum(cins)
    x = rnd(fill.uniform(name + ".rx", (n, ci, h, w)), dtype)
    srcs, o = [], 0
    for c in cins:
        srcs.append(to_nhwc(x[:, o:o + c], dtype))
        o += c
    if kind == "conv":
        wt = rnd(fill.uniform(name + ".rw", (co, ci, 4, 4), -0.25, 0.25), dtype)
        want = ops.conv2d_fwd(x, wt, None, stride)
        wsh, _ = shadows(wt, dtype, cparts=[(c, K.pad_ch(c, dtype)) for c in cins], want=("same",))
        mode, grid = (K.CONV_S2 if stride == 2 else K.CONV_S1), (n,) + want.shape[2:]
    else:
        wt = rnd(fill.uniform(name + ".rw", (ci, co, 4, 4), -0.25, 0.25), dtype)
        want = ops.convT2d_fwd(x, wt, None)
        _, wsh = shadows(wt, dtype, want=("tr",))
        mode, grid = K.TCONV_S2, (n, h, w)
    lib = _lib.lib()
    res, sums = [], []
    for knob1, kernel_want in ((16384, "k_conv_rows<bf16,256,64>"), (0, "k_gather_gemm<bf16,256,128,4,2,")):
        out = torch.full((n,) + want.shape[2:] + (K.pad_ch(co, dtype),), 7.0, dtype=dtype, device="cuda")
        lib.pfa_set_knob(6, 1)
        lib.pfa_set_knob(1, knob1)
        try:
            st = K.gather_gemm(mode, srcs, wsh, co, [out], grid, ws, stats=True, splitk=splitk)
            kernel = lib.pfa_last_gather_kernel().decode()
        finally:
            lib.pfa_set_knob(6, 0)
            lib.pfa_set_knob(1, 0)
        assert kernel.startswith(kernel_want), kernel
        res.append(from_nhwc(out, co))
        psum, _, nt = st
        torch.cuda.synchronize()
        sums.append(psum[: nt * co].view(nt, co).double().sum(0).cpu().numpy())
    close(res[0], want, dtype, name)
    close(res[0], res[1], dtype, name + " vs the ring tile", scale=np.abs(want).max())
    close(sums[0], want.astype(np.float64).sum(axis=(0, 2, 3)), torch.float32, name + ".psum",
          scale=np.abs(want).sum(axis=(0, 2, 3)).max())


def test_conv_rows_kernel_stride1_input_gradient(ws):
    """The input gradient of a stride-1 convolution (PFA_CONV_S1_DGRAD: 16 taps walked backwards) on the row kernel."""
    dtype = torch.bfloat16
    n, ci, h, w, co, slices = 2, 128, 9, 8, 64, 2
    x = fill.uniform("rows.gx", (n, ci, h, w))
    wt = rnd(fill.uniform("rows.gw", (co, ci, 4, 4), -0.25, 0.25), dtype)
    dy = rnd(fill.uniform("rows.gdy", (n, co, h - 1, w - 1)), dtype)
    want, _, _ = ops.conv2d_bwd(x, wt, dy, 1)
    _, tr = shadows(wt, dtype, want=("tr",))
    lib = _lib.lib()
    out = torch.full((n, h, w, ci), 7.0, dtype=dtype, device="cuda")
    lib.pfa_set_knob(6, 1)
    lib.pfa_set_knob(1, 16384)
    try:
        K.gather_gemm(K.CONV_S1_DGRAD, [to_nhwc(dy, dtype)], tr, ci, [out], (n, h, w), ws, splitk=slices)
        kernel = lib.pfa_last_gather_kernel().decode()
    finally:
        lib.pfa_set_knob(6, 0)
        lib.pfa_set_knob(1, 0)
    assert kernel == "k_conv_rows<bf16,256,64>", kernel
    close(from_nhwc(out, ci), want, dtype, "rows s1 dgrad")


AFFINE_CASES = [  # name, mode, N, cin parts, H, W, Co, two outputs, K slices
    ("enc_small", "conv", 2, [8], 16, 16, 8, True, 1),             # narrow staging, 256x16 tile
    ("enc_mid", "conv", 2, [64], 32, 32, 96, True, 1),             # 128x128 tile, ragged columns
    ("enc_e1", "conv", 2, [3], 32, 32, 64, True, 1),               # 3-channel image layer, 128x64 tile
    ("dec", "tconv", 2, [64, 64], 8, 8, 48, False, 1),             # transposed conv, virtual concat, one output
    ("enc_sliced", "conv", 2, [64], 8, 8, 128, True, 4),           # K-sliced: the affine map moves to the finish pass
    ("dec_sliced", "tconv", 2, [64, 64], 4, 4, 64, False, 3),
    ("p2_enc", "conv", 2, [64], 32, 32, 128, True, 1),              # p2_*: on the patch kernels (bf16, pfa_set_knob(3, 2))
    ("p2_dec", "tconv", 2, [64, 64], 16, 16, 64, False, 1),
]


@pytest.mark.parametrize("dtype", DTYPES, ids=DT_IDS)
@pytest.mark.parametrize("case", AFFINE_CASES, ids=[c[0] for c in AFFINE_CASES])
def test_affine_epilogue(case, dtype, ws):
    """fuse_out = 3: out1 = LeakyReLU/ReLU(scale*conv + shift), out2 = ReLU(scale*conv + shift) - the eval-mode
    BatchNorm of generate_synthetic_data.py:55 applied inside the launch."""
    name, mode, n, cins, h, w, co, dual, slices = case
    ci = sum(cins)
    x = rnd(fill.uniform(name + ".ax", (n, ci, h, w)), dtype)
    scale = fill.uniform(name + ".asc", (co,), 0.5, 1.5)
    shift = fill.uniform(name + ".ash", (co,), -0.5, 0.5)
    parts, srcs, o = [], [], 0
    for c in cins:
        srcs.append(to_nhwc(x[:, o:o + c], dtype))
        parts.append((c, K.pad_ch(c, dtype)))
        o += c
    if mode == "conv":
        wt = rnd(fill.uniform(name + ".aw", (co, ci, 4, 4), -0.25, 0.25), dtype)
        raw = ops.conv2d_fwd(x, wt, None, 2)
        wdev, _ = shadows(wt, dtype, cparts=parts, want=("same",))
        gmode, grid = K.CONV_S2, (n, h // 2, w // 2)
    else:
        wt = rnd(fill.uniform(name + ".aw", (ci, co, 4, 4), -0.25, 0.25), dtype)
        raw = ops.convT2d_fwd(x, wt, None)
        _, wdev = shadows(wt, dtype, want=("tr",))
        gmode, grid = K.TCONV_S2, (n, h, w)
    y = raw * scale[None, :, None, None] + shift[None, :, None, None]
    ho, wo = raw.shape[2:]
    cp = K.pad_ch(co, dtype)
    outs = [torch.full((n, ho, wo, cp), 7.0, dtype=dtype, device="cuda") for _ in range(2 if dual else 1)]
    sc, sh = torch.from_numpy(scale).cuda(), torch.from_numpy(shift).cuda()
    act1 = K.ACT_LRELU if dual else K.ACT_RELU
    force = name.startswith("p2_")
    if force:
        _lib.lib().pfa_set_knob(3, 2)
    try:
        K.gather_gemm(gmode, srcs, wdev, co, outs, grid, ws, act=act1, splitk=slices, affine=(sc, sh, K.ACT_RELU))
        kernel = _lib.lib().pfa_last_gather_kernel().decode()
    finally:
        _lib.lib().pfa_set_knob(3, 0)
    if name.startswith("img_"):
        assert kernel.startswith("k_conv_image") == (dtype == torch.bfloat16), kernel
    else:
        assert kernel.startswith("k_conv_patch2") == (force and dtype == torch.bfloat16), kernel
    close(from_nhwc(outs[0], co), ops.leaky_relu(y) if dual else ops.relu(y), dtype, name + " out1", scale=np.abs(y).max())
    if dual:
        close(from_nhwc(outs[1], co), ops.relu(y), dtype, name + " out2", scale=np.abs(y).max())
    for t in outs:
        if cp > co:
            assert float(t[..., co:].float().abs().max()) == 0.0, "pad channels must be written as zeros"


FUSED_BWD_CASES = [  # name, N, cin, H, W, Co (columns), K slices, BatchNorm, second gradient, two outputs
    ("act_only", 2, [64], 16, 16, 64, 1, False, False, False),
    ("bn", 2, [64], 16, 16, 128, 1, True, False, False),
    ("bn_g2", 2, [64], 16, 16, 128, 1, True, True, False),
    ("bn_sliced", 2, [64], 8, 8, 128, 4, True, False, False),            # K-sliced: the finish pass applies it
    ("bn_g2_sliced", 3, [128], 6, 10, 64, 3, True, True, False),
    ("act_sliced", 2, [64], 8, 8, 64, 2, False, False, False),
    ("two_outputs_sliced", 2, [64], 8, 8, 128, 4, True, False, True),    # [plain 64 | fused 64] like the decoder's dgrad
    ("two_outputs", 2, [64], 16, 16, 128, 1, True, False, True),
    # p2_*: 16x16 output maps, routed to the patch kernels of conv_patch2.hip in bf16 (pfa_set_knob(3, 2)): the shared epilogue
    ("p2_bn_g2", 2, [64], 32, 32, 128, 1, True, True, False),
    ("p2_two_outputs", 2, [64], 32, 32, 128, 1, True, False, True),
    ("p2_act_sliced", 2, [128], 32, 32, 64, 2, False, False, False),
    # the input gradient of the generator's output layer: a 3-channel source (8 padded), 128 columns = [skip gradient 64 |
    # fused BatchNorm backward 64]; in bf16 on the image-side kernel (conv_image.hip), two 64-channel tiles per patch
    ("img_two_outputs", 2, [3], 32, 64, 128, 1, True, False, True),
    ("img_act_only", 1, [3], 36, 80, 64, 1, False, False, False),
]


@pytest.mark.parametrize("dtype", DTYPES, ids=DT_IDS)
@pytest.mark.parametrize("case", FUSED_BWD_CASES, ids=[c[0] for c in FUSED_BWD_CASES])
def test_fused_backward_epilogue(case, dtype, ws):
    """fuse_out = 1|2: the launch computes g = conv(...) and stores dy = LeakyReLU'(y)*g (+ ReLU'(y)*g2) with
    y = scale*raw + shift, plus the per-column sums of dy and dy*xhat - in the GEMM epilogue, or in the finish pass
    when the launch is K-sliced."""
    name, n, cins, h, w, co, slices, bn, with_g2, two = case
    ci = sum(cins)
    x = rnd(fill.uniform(name + ".fx", (n, ci, h, w)), dtype)
    wt = rnd(fill.uniform(name + ".fw", (co, ci, 4, 4), -0.25, 0.25), dtype)
    g = ops.conv2d_fwd(x, wt, None, 2)                                  # [n, co, ho, wo]: the incoming gradient
    ho, wo = g.shape[2:]
    wsh, _ = shadows(wt, dtype, cparts=[(c, K.pad_ch(c, dtype)) for c in cins], want=("same",))
    srcs = [to_nhwc(x, dtype)]
    c0 = co // 2 if two else 0                                          # first fused column
    cf = co - c0                                                         # fused columns
    raw = rnd(fill.uniform(name + ".fraw", (n, cf, ho, wo), -1.0, 1.0), dtype)
    g2 = rnd(fill.uniform(name + ".fg2", (n, cf, ho, wo), -1.0, 1.0), dtype) if with_g2 else None
    scale = fill.uniform(name + ".fsc", (cf,), 0.5, 1.5) if bn else np.ones(cf, np.float32)
    shift = fill.uniform(name + ".fsh", (cf,), -0.3, 0.3) if bn else np.zeros(cf, np.float32)
    mean = fill.uniform(name + ".fmu", (cf,), -0.2, 0.2)
    invstd = fill.uniform(name + ".fis", (cf,), 0.5, 2.0)
    y = raw * scale[None, :, None, None] + shift[None, :, None, None]
    gf = g[:, c0:]
    want = np.where(y >= 0, 1.0, 0.2) * gf
    if with_g2:
        want = want + (y > 0) * g2
    outs = [torch.full((n, ho, wo, c0 if two else co), 7.0, dtype=dtype, device="cuda")]
    if two:
        outs.append(torch.full((n, ho, wo, cf), 7.0, dtype=dtype, device="cuda"))
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    spec = dict(out=2 if two else 1, raw=to_nhwc(raw, dtype), act1=K.ACT_LRELU,
                g2=to_nhwc(g2, dtype) if with_g2 else None, act2=K.ACT_RELU if with_g2 else K.ACT_NONE)
    if bn:
        spec.update(scale=dev(scale), shift=dev(shift), mean=dev(mean), invstd=dev(invstd))
    before = _lib.launch_count()
    force = name.startswith("p2_")
    if force:
        _lib.lib().pfa_set_knob(3, 2)
    try:
        st = K.gather_gemm(K.CONV_S2, srcs, wsh, co, outs, (n, ho, wo), ws, splitk=slices, fuse=spec)
        kernel = _lib.lib().pfa_last_gather_kernel().decode()
    finally:
        _lib.lib().pfa_set_knob(3, 0)
    if name.startswith("img_"):
        assert kernel.startswith("k_conv_image") == (dtype == torch.bfloat16), kernel
    else:
        assert kernel.startswith("k_conv_patch2") == (force and dtype == torch.bfloat16), kernel
    launches = _lib.launch_count() - before
    assert launches == (2 if slices > 1 else 1)
    amax = np.abs(g).max()
    close(from_nhwc(outs[-1], cf), want, dtype, name + ".dy", scale=amax)
    if two:
        close(from_nhwc(outs[0], c0), g[:, :c0], dtype, name + ".plain", scale=amax)
    if bn:
        part, _, nt = st
        torch.cuda.synchronize()
        p = part[: 2 * nt * cf].view(2, nt, cf).double().sum(1).cpu().numpy()
        got_dy = from_nhwc(outs[-1], cf).astype(np.float64)            # sums of what was stored, before its rounding
        xhat = (raw - mean[None, :, None, None]) * invstd[None, :, None, None]
        close(p[0], want.astype(np.float64).sum(axis=(0, 2, 3)), dtype, name + ".sum_dy", scale=np.abs(want).sum(axis=(0, 2, 3)).max())
        close(p[1], (want.astype(np.float64) * xhat).sum(axis=(0, 2, 3)), dtype, name + ".sum_dy_xhat",
              scale=np.abs(want * xhat).sum(axis=(0, 2, 3)).max())
        assert np.isfinite(got_dy).all()
    else:
        assert st is None


LOGIT_CASES = [  # name, N, H, W of the 512-channel input
    ("c5_256", 2, 31, 31),          # the layer's input at 256 x 256 tiles
    ("c5_ragged", 1, 9, 37),        # two column tiles, the second 5 wide; odd row count
    ("c5_wide", 1, 4, 63),          # the input width at 512 x 512 tiles
    ("c5_tiny", 2, 2, 2),           # one logit per image
]


@pytest.mark.parametrize("case", LOGIT_CASES, ids=[c[0] for c in LOGIT_CASES])
def test_logit_layer_kernels(case, ws):
    """The discriminator's Conv2d(512, 1, 4, 1, 1) (models.py:243) on conv_logit.hip: forward, input gradient fused with
    the LeakyReLU + BatchNorm backward of the layer below (values and partial sums), weight and bias gradient - against
    the oracle's convolution."""
    name, n, h, w = case
    dtype, c = torch.bfloat16, 512
    x = rnd(fill.uniform(name + ".lx", (n, c, h, w)), dtype)
    wt = rnd(fill.uniform(name + ".lw", (1, c, 4, 4), -0.25, 0.25), dtype)
    b = fill.uniform(name + ".lb", (1,), -0.5, 0.5)
    want = ops.conv2d_fwd(x, wt, b, 1)                                   # [n, 1, h-1, w-1]
    ho, wo = want.shape[2:]
    same, _ = shadows(wt, dtype, want=("same",))
    xd = to_nhwc(x, dtype)
    assert K.logit_ok(xd, 8)
    logits = torch.full((n, ho, wo, 8), 7.0, dtype=dtype, device="cuda")
    before = _lib.launch_count()
    K.logit_fwd(xd, same, torch.from_numpy(b).cuda(), logits)
    assert _lib.launch_count() - before == 1
    close(from_nhwc(logits, 1), want, dtype, name + ".logits")
    assert float(logits[..., 1:].abs().max()) == 0.0                    # pad channels are written as zeros
    # the same launch with BCEWithLogitsLoss in its epilogue (train_gan.py:58,60,67): logits, dlogits and - through
    # pfa_loss_finish - the loss of the two-launch route (pfa_logit_fwd, pfa_bce_logits_parts): dlogits bit for bit
    for target, ls, gs in ((1.0, 0.5, 0.5), (0.0, 1.0, 1.0)):
        lg2 = torch.full_like(logits, 5.0)
        dl_f = torch.full_like(logits, 5.0)
        parts = torch.full((4096,), 3.0, device="cuda")
        before = _lib.launch_count()
        np_f = K.logit_fwd_bce(xd, same, torch.from_numpy(b).cuda(), lg2, target, ls, gs, parts, dl_f)
        assert _lib.launch_count() - before == 1 and 0 < np_f <= 4096
        assert torch.equal(lg2, logits)
        dl_t = torch.full_like(logits, 9.0)
        p2 = torch.zeros(64, device="cuda")
        np_t = K.bce_logits_parts(logits, target, ls, gs, p2, dl_t)
        assert torch.equal(dl_f, dl_t), "dlogits of the fused and the two-launch route differ"
        o1, o2 = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
        K.loss_finish(parts, [(0, np_f), (0, 0)], o1)
        K.loss_finish(p2, [(0, np_t), (0, 0)], o2)
        torch.cuda.synchronize()
        lref, _ = ops.bce_logits(from_nhwc(logits, 1), target)
        assert abs(float(o1[0]) - float(o2[0])) <= 2e-6 * abs(float(o2[0])) and abs(float(o1[0]) - ls * lref) <= 1e-5 * abs(ls * lref)

    # backward: dlogits -> (dw, db) and dy = LeakyReLU'(scale*raw + shift) * dx with the BatchNorm partial sums
    dl = rnd(fill.uniform(name + ".ldl", (n, 1, ho, wo), -1.0, 1.0), dtype)
    dld = to_nhwc(dl, dtype)
    dx, dw, db = ops.conv2d_bwd(x, wt, dl, 1)
    scale, shift = fill.uniform(name + ".lsc", (c,), 0.5, 1.5), fill.uniform(name + ".lsh", (c,), -0.3, 0.3)
    mean, invstd = fill.uniform(name + ".lmu", (c,), -0.2, 0.2), fill.uniform(name + ".lis", (c,), 0.5, 2.0)
    y = x * scale[None, :, None, None] + shift[None, :, None, None]
    want_dy = np.where(y > 0, 1.0, 0.2) * dx
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    for bn in (True, False):
        dy = torch.full((n, h, w, c), 7.0, dtype=dtype, device="cuda")
        args = (dev(scale), dev(shift), dev(mean), dev(invstd)) if bn else (None, None, None, None)
        part, nt = K.logit_dgrad(dld, same, xd, *args, K.ACT_LRELU, dy, ws)
        wdy = want_dy if bn else np.where(x > 0, 1.0, 0.2) * dx
        close(from_nhwc(dy, c), wdy, dtype, name + ".dy", scale=np.abs(dx).max())
        if bn:
            torch.cuda.synchronize()
            ps = part[: 2 * nt * c].view(2, nt, c).double().sum(1).cpu().numpy()
            xhat = (x - mean[None, :, None, None]) * invstd[None, :, None, None]
            close(ps[0], wdy.astype(np.float64).sum(axis=(0, 2, 3)), dtype, name + ".sum_dy",
                  scale=np.abs(wdy).sum(axis=(0, 2, 3)).max())
            close(ps[1], (wdy.astype(np.float64) * xhat).sum(axis=(0, 2, 3)), dtype, name + ".sum_dy_xhat",
                  scale=np.abs(wdy * xhat).sum(axis=(0, 2, 3)).max())
        else:
            assert part is None
    gw = torch.full((1, 16, c), 0.5, dtype=torch.float32, device="cuda")      # gradients accumulate
    gb = torch.full((1,), 0.25, dtype=torch.float32, device="cuda")
    K.logit_wgrad(dld, xd, gw, gb)
    torch.cuda.synchronize()
    got_w = (gw - 0.5).cpu().numpy().reshape(1, 4, 4, c).transpose(0, 3, 1, 2)
    close(got_w, dw, torch.float32, name + ".dw", scale=np.abs(dw).max())
    close((gb - 0.25).cpu().numpy(), db, torch.float32, name + ".db", scale=max(np.abs(dl).sum(), 1.0))
    # with a workspace: partial rows + ordered reduce - the same values, and the same bits on every run
    runs = []
    for _ in range(4):
        gw2 = torch.full((1, 16, c), 0.5, dtype=torch.float32, device="cuda")
        gb2 = torch.full((1,), 0.25, dtype=torch.float32, device="cuda")
        before = _lib.launch_count()
        K.logit_wgrad(dld, xd, gw2, gb2, ws=ws)
        assert _lib.launch_count() - before == 2
        runs.append((gw2.clone(), gb2.clone()))
    torch.cuda.synchronize()
    close((runs[0][0] - 0.5).cpu().numpy().reshape(1, 4, 4, c).transpose(0, 3, 1, 2), dw, torch.float32, name + ".dw (slab)",
          scale=np.abs(dw).max())
    close((runs[0][1] - 0.25).cpu().numpy(), db, torch.float32, name + ".db (slab)", scale=max(np.abs(dl).sum(), 1.0))
    assert all(torch.equal(runs[0][0], r[0]) and torch.equal(runs[0][1], r[1]) for r in runs[1:]), name + ": not reproducible"
    K.logit_wgrad(dld, xd, gw, None, ws=ws)                                     # (no bias gradient wanted)


def test_logit_layer_support():
    lib = _lib.lib()
    assert lib.pfa_logit_supported(K.dt_code(torch.bfloat16), 512, 8, 31, 31) == 1
    assert lib.pfa_logit_supported(K.dt_code(torch.float32), 512, 4, 31, 31) == 0      # float32 keeps the general kernels
    assert lib.pfa_logit_supported(K.dt_code(torch.bfloat16), 256, 8, 31, 31) == 0
    # the widest map ALL THREE kernels take (the weight gradient's LDS budget bounds it): the engine routes forward, input
    # gradient and weight gradient on this one predicate
    assert lib.pfa_logit_supported(K.dt_code(torch.bfloat16), 512, 8, 4, 1129) == 1
    assert lib.pfa_logit_supported(K.dt_code(torch.bfloat16), 512, 8, 4, 1130) == 0
    wide = torch.zeros((1, 4, 1129, 512), dtype=torch.bfloat16, device="cuda")
    K.logit_wgrad(torch.zeros((1, 3, 1128, 8), dtype=torch.bfloat16, device="cuda"), wide,
                  torch.zeros(16 * 512, device="cuda"), torch.zeros(1, device="cuda"))
    torch.cuda.synchronize()
    x = torch.zeros((1, 4, 4, 256), dtype=torch.bfloat16, device="cuda")
    with pytest.raises(_lib.PfaError):
        K.logit_fwd(x, x, None, torch.zeros((1, 3, 3, 8), dtype=torch.bfloat16, device="cuda"))


SCATTER_CASES = [  # name, N, cin parts, H, W, Co, bias, act
    ("d1", 2, [64, 64], 8, 8, 3, True, K.ACT_TANH),              # the generator's output layer: two 64-channel sources
    ("d1_ragged", 1, [64, 64], 7, 33, 3, True, K.ACT_TANH),      # H not a multiple of the 6-row tile, W crosses a tile
    ("one_tile", 2, [64], 6, 30, 3, False, K.ACT_NONE),          # exactly one interior tile, single source
    ("c1_dgrad_like", 2, [64], 13, 31, 6, False, K.ACT_NONE),    # 6 real of 8 columns per tap
    ("n8_relu", 1, [128], 5, 61, 8, True, K.ACT_RELU),
    ("n1", 1, [64], 3, 4, 1, True, K.ACT_NONE),
    ("1x1", 3, [64], 1, 1, 3, True, K.ACT_TANH),
]


@pytest.mark.parametrize("dtype", DTYPES, ids=DT_IDS)
@pytest.mark.parametrize("case", SCATTER_CASES, ids=[c[0] for c in SCATTER_CASES])
def test_tconv_scatter(case, dtype, ws):
    """pfa_tconv_scatter (pointwise GEMM + col2im) against the oracle's ConvTranspose2d, and against the implicit-GEMM
    path on the same operands."""
    name, n, cins, h, w, co, bias, act = case
    ci = sum(cins)
    x = rnd(fill.uniform(name + ".sx", (n, ci, h, w)), dtype)
    wt = rnd(fill.uniform(name + ".sw", (ci, co, 4, 4), -0.25, 0.25), dtype)
    b = fill.uniform(name + ".sb", (co,), -0.5, 0.5) if bias else None
    raw = ops.convT2d_fwd(x, wt, b)
    want = np.tanh(raw) if act == K.ACT_TANH else np.maximum(raw, 0) if act == K.ACT_RELU else raw
    srcs, o = [], 0
    for c in cins:
        srcs.append(to_nhwc(x[:, o:o + c], dtype))
        o += c
    _, tr = shadows(wt, dtype, want=("tr",))
    cp = K.pad_ch(co, dtype)
    out = torch.full((n, 2 * h, 2 * w, cp), 7.0, dtype=dtype, device="cuda")
    bt = torch.from_numpy(b).cuda() if bias else None
    assert K.tconv_scatter_ok(srcs, co, out)
    K.tconv_scatter(srcs, tr, co, out, bias=bt, act=act)
    close(from_nhwc(out, co), want, dtype, name)
    torch.cuda.synchronize()
    assert float(out[..., co:].float().abs().max()) == 0.0 if cp > co else True, "pad lanes must be exact zeros"
    ref = torch.full_like(out, 7.0)
    K.gather_gemm(K.TCONV_S2, srcs, tr, co, [ref], (n, h, w), ws, bias=bt, act=act)
    close(from_nhwc(out, co), from_nhwc(ref, co), dtype, name + " vs implicit GEMM")


@pytest.mark.parametrize("dtype", DTYPES, ids=DT_IDS)
@pytest.mark.parametrize("cins", [[64, 64], [64], [128, 64]], ids=["d1_like", "one_source", "uneven"])
def test_tconv_scatter_relu_on_the_first_source(cins, dtype):
    """pfa_scatter_args.relu_src1: the first source enters through max(x, 0) (the eval forward hands the last layer the
    LeakyReLU copy of e1's activation) - the same bits as the launch on a materialised ReLU copy; the second source untouched."""
    n, h, w, co = 2, 9, 35, 3
    ci = sum(cins)
    y = rnd(fill.uniform("relu1.y", (n, ci, h, w), -2, 2), dtype)
    y[0, 0, 0, :4] = [-0.0, 0.0, -1e-30, 1e-30]
    wt = rnd(fill.uniform("relu1.w", (ci, co, 4, 4), -0.25, 0.25), dtype)
    _, tr = shadows(wt, dtype, want=("tr",))
    lre = np.where(y > 0, y, 0.2 * y).astype(np.float32)                       # what the encoder stores: LeakyReLU(y)
    srcs, srcs_relu, o = [], [], 0
    for i, c in enumerate(cins):
        part = rnd(lre[:, o:o + c], dtype)
        srcs.append(to_nhwc(part, dtype))
        srcs_relu.append(to_nhwc(np.maximum(part, 0) if i == 0 else part, dtype))
        o += c
    cp = K.pad_ch(co, dtype)
    a = torch.full((n, 2 * h, 2 * w, cp), 7.0, dtype=dtype, device="cuda")
    b = torch.full_like(a, 5.0)
    K.tconv_scatter(srcs, tr, co, a, act=K.ACT_TANH, relu_src1=True)
    K.tconv_scatter(srcs_relu, tr, co, b, act=K.ACT_TANH)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    x = np.concatenate([np.maximum(lre[:, :cins[0]], 0)] + ([lre[:, cins[0]:]] if len(cins) > 1 else []), axis=1)
    close(from_nhwc(a, co), np.tanh(ops.convT2d_fwd(rnd(x, dtype), wt, None)), dtype, "relu on load")


@pytest.mark.parametrize("dtype", DTYPES, ids=DT_IDS)
@pytest.mark.parametrize("case", [("c1_dgrad", 2, 64, 13, 31, 3), ("ragged", 1, 64, 7, 33, 3), ("six", 2, 128, 6, 30, 6)],
                         ids=lambda c: c[0])
def test_tconv_scatter_fused_l1_tanh_epilogue(case, dtype, ws):
    """The generator phase's fused epilogue (train_gan.py:66-70): the launch that computes dL/dfake of the discriminator's
    first layer writes dL/dpre = (lambda sign(fake - real) / N + dL/dfake) (1 - fake^2) directly - bit for bit what the
    two-launch route (pfa_tconv_scatter, then pfa_l1_tanh_bwd on the stored dL/dfake) produces, and the oracle's arithmetic."""
    name, n, ci, h, w, co = case
    x = rnd(fill.uniform(name + ".lx", (n, ci, h, w)), dtype)
    wt = rnd(fill.uniform(name + ".lw", (ci, co, 4, 4), -0.25, 0.25), dtype)
    fake = rnd(np.tanh(fill.uniform(name + ".lf", (n, co, 2 * h, 2 * w), -2, 2)), dtype)
    real = rnd(fill.uniform(name + ".lr", (n, co, 2 * h, 2 * w)), dtype)
    real[0, 0, 0, :3] = fake[0, 0, 0, :3]                       # sign(0) = 0
    lam = 100.0
    src = to_nhwc(x, dtype)
    _, tr = shadows(wt, dtype, want=("tr",))
    cp = K.pad_ch(co, dtype)
    ft, rt = to_nhwc(fake, dtype), to_nhwc(real, dtype)
    dx = torch.full((n, 2 * h, 2 * w, cp), 7.0, dtype=dtype, device="cuda")
    K.tconv_scatter([src], tr, co, dx)
    two = torch.full_like(dx, 5.0)
    parts = torch.zeros(1024, device="cuda")
    K.l1_tanh_bwd_parts(ft, rt, co, dx, 0, lam, parts, two)
    one = torch.full_like(dx, 3.0)
    K.tconv_scatter([src], tr, co, one, l1=(ft, rt, co, lam))
    torch.cuda.synchronize()
    assert torch.equal(one, two), "fused epilogue and two-launch route differ"
    dgan = rnd(ops.convT2d_fwd(x, wt, None), dtype)
    d = fake.astype(np.float64) - real
    want = (lam * np.sign(d) / d.size + dgan) * (1.0 - fake.astype(np.float64) ** 2)
    close(from_nhwc(one, co), want, dtype, name)
    with pytest.raises(_lib.PfaError, match="no bias"):
        K.tconv_scatter([src], tr, co, one, act=K.ACT_TANH, l1=(ft, rt, co, lam))


def test_tconv_scatter_support_and_errors():
    bf = torch.bfloat16
    x16 = torch.zeros((1, 4, 4, 16), dtype=bf, device="cuda")
    x64 = torch.zeros((1, 4, 4, 64), dtype=bf, device="cuda")
    out8 = torch.zeros((1, 8, 8, 8), dtype=bf, device="cuda")
    out16 = torch.zeros((1, 8, 8, 16), dtype=bf, device="cuda")
    assert not K.tconv_scatter_ok([x16], 3, out8)          # channel count is not a whole 128-byte chunk
    assert not K.tconv_scatter_ok([x64], 12, out16)        # more than 8 output channels
    assert K.tconv_scatter_ok([x64], 3, out8)
    w = torch.zeros((8, 16, 16), dtype=bf, device="cuda")
    with pytest.raises(_lib.PfaError, match="128-byte chunk"):
        K.tconv_scatter([x16], w, 3, out8)


# ---------------------------------------------------------------------------- input gradients
@pytest.mark.parametrize("dtype", DTYPES, ids=DT_IDS)
@pytest.mark.parametrize("case", [("s2", 2, 16, 8, 8, 32, 2), ("s2_c1like", 2, 6, 16, 16, 64, 2),
                                  ("s1", 2, 32, 9, 9, 64, 1), ("s1_co1", 2, 64, 7, 6, 1, 1),
                                  ("s2_big", 2, 64, 32, 32, 128, 2)], ids=lambda c: c[0])
def test_conv_dgrad(case, dtype, ws):
    name, n, ci, h, w, co, stride = case
    x = fill.uniform(name + ".dx.x", (n, ci, h, w))
    wt = rnd(fill.uniform(name + ".dx.w", (co, ci, 4, 4), -0.25, 0.25), dtype)
    ho, wo = (h + 2 - 4) // stride + 1, (w + 2 - 4) // stride + 1
    dy = rnd(fill.uniform(name + ".dx.dy", (n, co, ho, wo)), dtype)
    want, _, _ = ops.conv2d_bwd(x, wt, dy, stride)
    _, tr = shadows(wt, dtype, want=("tr",))           # [Ci_p][16][Co_p]
    dyt = to_nhwc(dy, dtype)
    out = torch.full((n, h, w, K.pad_ch(ci, dtype)), 7.0, dtype=dtype, device="cuda")
    if stride == 2:
        K.gather_gemm(K.TCONV_S2, [dyt], tr, ci, [out], (n, ho, wo), ws)
    else:
        K.gather_gemm(K.CONV_S1_DGRAD, [dyt], tr, ci, [out], (n, h, w), ws)
    close(from_nhwc(out, ci), want, dtype, name)


@pytest.mark.parametrize("dtype", DTYPES, ids=DT_IDS)
@pytest.mark.parametrize("case", [("single", 2, [16], 4, 4, 8), ("cat_split", 2, [16, 32], 4, 4, 16),
                                  ("co3", 2, [16, 16], 8, 8, 3)], ids=lambda c: c[0])
def test_convT_dgrad(case, dtype, ws):
    name, n, cins, h, w, co = case
    ci = sum(cins)
    x = fill.uniform(name + ".tdx.x", (n, ci, h, w))
    wt = rnd(fill.uniform(name + ".tdx.w", (ci, co, 4, 4), -0.25, 0.25), dtype)
    dy = rnd(fill.uniform(name + ".tdx.dy", (n, co, 2 * h, 2 * w)), dtype)
    want, _, _ = ops.convT2d_bwd(x, wt, dy)
    same, _ = shadows(wt, dtype, want=("same",))        # [Ci][16][Co_p]
    dyt = to_nhwc(dy, dtype)
    outs = [torch.full((n, h, w, c), 7.0, dtype=dtype, device="cuda") for c in cins]
    K.gather_gemm(K.CONV_S2, [dyt], same, ci, outs, (n, h, w), ws)
    got = np.concatenate([from_nhwc(o, c) for o, c in zip(outs, cins)], axis=1)
    close(got, want, dtype, name)


# ---------------------------------------------------------------------------- weight gradients
@pytest.mark.parametrize("dtype", DTYPES, ids=DT_IDS)
@pytest.mark.parametrize("case", [("s2", 2, [16], 8, 8, 32, 2, None), ("s2_ci3", 2, [3], 16, 16, 16, 2, None),
                                  ("s2_cat33", 2, [3, 3], 16, 16, 64, 2, None), ("s1", 2, [32], 9, 9, 64, 1, None),
                                  ("s1_co1", 2, [64], 7, 6, 1, 1, None), ("s2_msplit", 4, [64], 32, 32, 128, 2, 5),
                                  ("s2_co136", 1, [16], 8, 8, 136, 2, 1),
                                  # K-group pairs (bf16, >= 2 slices): a single ragged 128-pixel step, and uneven slice pairs
                                  ("s2_pairs_ragged", 3, [32], 10, 14, 136, 2, 4), ("s1_pairs", 2, [64], 17, 19, 128, 1, 6)],
                         ids=lambda c: c[0])
def test_conv_wgrad(case, dtype, ws):
    name, n, cins, h, w, co, stride, msplit = case
    ci = sum(cins)
    x = rnd(fill.uniform(name + ".dw.x", (n, ci, h, w)), dtype)
    wt = fill.uniform(name + ".dw.w", (co, ci, 4, 4))
    ho, wo = (h + 2 - 4) // stride + 1, (w + 2 - 4) // stride + 1
    dy = rnd(fill.uniform(name + ".dw.dy", (n, co, ho, wo)), dtype)
    _, want, _ = ops.conv2d_bwd(x, wt, dy, stride, need_dx=False)
    srcs, o = [], 0
    for c in cins:
        srcs.append(to_nhwc(x[:, o:o + c], dtype))
        o += c
    dyt = to_nhwc(dy, dtype)
    for use_ws in (None, K.FORCE_SLAB):        # float-atomic combine, then the deterministic slab combine
        dw = torch.zeros((co, 16, ci), dtype=torch.float32, device="cuda")
        K.wgrad([dyt], srcs, dw, stride, [co], cins, msplit=msplit, ws=use_ws)
        torch.cuda.synchronize()
        got = dw.cpu().numpy().reshape(co, 4, 4, ci).transpose(0, 3, 1, 2)
        close(got, want, torch.float32, name)  # float32 accumulation of exact products in both modes
    dw2 = torch.zeros_like(dw)
    K.wgrad([dyt], srcs, dw2, stride, [co], cins, msplit=msplit, ws=K.FORCE_SLAB)
    K.wgrad([dyt], srcs, dw2, stride, [co], cins, msplit=msplit, ws=K.FORCE_SLAB)     # accumulates (zero_grad semantics)
    torch.cuda.synchronize()
    assert torch.equal(dw2, dw + dw), "slab combine must be bitwise reproducible and accumulate into dw"
    # fresh=True: the caller vouches for a zeroed dw with one writer; launches that can (one slice; the slab reduce)
    # store instead of adding - the same values
    for use_ws in (None, K.FORCE_SLAB):
        dw3 = torch.zeros_like(dw)
        K.wgrad([dyt], srcs, dw3, stride, [co], cins, msplit=msplit, ws=use_ws, fresh=True)
        torch.cuda.synchronize()
        close(dw3.cpu().numpy().reshape(co, 4, 4, ci).transpose(0, 3, 1, 2), want, torch.float32, name + " fresh")


@pytest.mark.parametrize("dtype", DTYPES, ids=DT_IDS)
@pytest.mark.parametrize("case", [("single", 2, [16], 4, 4, 8), ("cat", 2, [16, 32], 4, 4, 16),
                                  ("co3", 2, [16, 16], 8, 8, 3), ("big", 2, [64, 64], 16, 16, 64)],
                         ids=lambda c: c[0])
def test_convT_wgrad(case, dtype, ws):
    name, n, cins, h, w, co = case
    ci = sum(cins)
    x = rnd(fill.uniform(name + ".tdw.x", (n, ci, h, w)), dtype)
    wt = fill.uniform(name + ".tdw.w", (ci, co, 4, 4))
    dy = rnd(fill.uniform(name + ".tdw.dy", (n, co, 2 * h, 2 * w)), dtype)
    _, want, _ = ops.convT2d_bwd(x, wt, dy, need_dx=False)
    srcs, o = [], 0
    for c in cins:
        srcs.append(to_nhwc(x[:, o:o + c], dtype))
        o += c
    for use_ws in (None, K.FORCE_SLAB):
        dw = torch.zeros((ci, 16, co), dtype=torch.float32, device="cuda")
        K.wgrad(srcs, [to_nhwc(dy, dtype)], dw, 2, cins, [co], ws=use_ws)
        torch.cuda.synchronize()
        got = dw.cpu().numpy().reshape(ci, 4, 4, co).transpose(0, 3, 1, 2)
        close(got, want, torch.float32, name)


WGI_CASES = [  # name, kind, N, input channel parts, H, W of the layer INPUT, output channels, slices wanted (knob 12; 0 = default)
    ("e1_like", "conv", 4, [3], 32, 128, 64, 0),           # dy 16 x 64: 4 x 2 tiles per image, one gathered source
    ("c1_like", "conv", 2, [3, 3], 64, 128, 64, 0),        # two gathered sources (A | B images)
    ("d1_like", "convT", 4, [64, 64], 16, 64, 3, 0),       # transposed: direct = the two 64-channel inputs, gathered = dy (3 ch)
    ("e1_runs", "conv", 5, [3], 32, 128, 64, 16),          # 40 tiles in 14 slices: runs of 3 tiles through the ring, the last slice 1 tile
    ("e1_pairs", "conv", 4, [3], 32, 128, 64, 16),         # 32 tiles in 16 slices of 2: the ring never fills
    ("c1_wide", "conv", 4, [3, 3], 32, 128, 128, 0),       # 128 direct channels x two gathered sources: the two-stage instantiation
]


def _last_wgrad_kernel(fn):
    """Kernel name of the (first) weight-gradient launch fn() issues, from the library's timing store."""
    import ctypes
    lib = _lib.lib()
    lib.pfa_gather_timing(1)
    try:
        fn()
        torch.cuda.synchronize()
        ms, name, flop, byts, sk = ctypes.c_float(), ctypes.c_char_p(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
        assert lib.pfa_gather_timing_count() >= 1
        _lib.check(lib.pfa_gather_timing_get(0, ctypes.byref(ms), ctypes.byref(name), ctypes.byref(flop), ctypes.byref(byts),
                                             ctypes.byref(sk)), "pfa_gather_timing_get")
        return name.value.decode()
    finally:
        lib.pfa_gather_timing(0)


@pytest.mark.parametrize("case", WGI_CASES, ids=[c[0] for c in WGI_CASES])
def test_wgrad_image_kernel(case, ws):
    """Weight gradients of the image-side layers (models.py:183,184,223) on wgrad_image.hip: the oracle, and the general kernel
    (pfa_set_knob(4, 9)) - one and two gathered sources, the transposed layer's two direct sources."""
    name, kind, n, cins, h, w, co, want_slices = case
    dtype = torch.bfloat16
    ci = sum(cins)
    x = rnd(fill.uniform(name + ".wi.x", (n, ci, h, w)), dtype)
    srcs, o = [], 0
    for c in cins:
        srcs.append(to_nhwc(x[:, o:o + c], dtype))
        o += c
    if kind == "conv":
        wt = fill.uniform(name + ".wi.w", (co, ci, 4, 4))
        dy = rnd(fill.uniform(name + ".wi.dy", (n, co, h // 2, w // 2)), dtype)
        _, want, _ = ops.conv2d_bwd(x, wt, dy, 2, need_dx=False)
        shape, args = (co, 16, ci), lambda dw, dyt: ([dyt], srcs, dw, 2, [co], cins)
    else:
        wt = fill.uniform(name + ".wi.w", (ci, co, 4, 4))
        dy = rnd(fill.uniform(name + ".wi.dy", (n, co, 2 * h, 2 * w)), dtype)
        _, want, _ = ops.convT2d_bwd(x, wt, dy, need_dx=False)
        shape, args = (ci, 16, co), lambda dw, dyt: (srcs, [dyt], dw, 2, cins, [co])
    dyt = to_nhwc(dy, dtype)
    res = []
    for knob in (0, 9):
        _lib.lib().pfa_set_knob(4, knob)
        _lib.lib().pfa_set_knob(12, want_slices)
        try:
            for fresh in (True, False):
                dw = torch.zeros(shape, dtype=torch.float32, device="cuda") if fresh else torch.full(shape, 0.5, dtype=torch.float32, device="cuda")
                kern = _last_wgrad_kernel(lambda: K.wgrad(*args(dw, dyt), ws=ws, fresh=fresh))
                assert kern.startswith("k_wgrad_image") == (knob == 0), kern
                got = (dw if fresh else dw - 0.5).cpu().numpy().reshape(shape[0], 4, 4, shape[2]).transpose(0, 3, 1, 2)
                close(got, want, torch.float32, f"{name} knob {knob} fresh {fresh}")
                res.append(got)
        finally:
            _lib.lib().pfa_set_knob(4, 0)
            _lib.lib().pfa_set_knob(12, 0)
    close(res[0], res[2], torch.float32, name + " vs the general kernel")
    # slab + ordered reduce: the same bits on every run
    _lib.lib().pfa_set_knob(12, want_slices)
    try:
        runs = []
        for _ in range(3):
            dw = torch.zeros(shape, dtype=torch.float32, device="cuda")
            K.wgrad(*args(dw, dyt), ws=ws, fresh=True)
            runs.append(dw.clone())
    finally:
        _lib.lib().pfa_set_knob(12, 0)
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2]), name + ": not reproducible"
    assert np.array_equal(runs[0].cpu().numpy().reshape(shape[0], 4, 4, shape[2]).transpose(0, 3, 1, 2), res[0]), name


WGP_CASES = [  # name, kind, N, input channel parts, H, W of the layer INPUT, output channels, stride, slab
    ("conv_s2_32", "conv", 2, [64], 64, 64, 128, 2, False),          # e2-like: dy 32x32, 2x32 tiles, atomics
    ("conv_s2_ragged", "conv", 1, [64], 40, 72, 128, 2, True),       # dy 20x36: ragged tile rows and columns, slab
    ("conv_s2_16", "conv", 3, [128], 32, 32, 256, 2, False),         # dy 16x16: 4x16 tiles, two c tiles, two i tiles
    ("conv_s1", "conv", 2, [64], 32, 32, 128, 1, False),             # c4-like: dy 31x31 (ragged 2x32 tiles)
    ("conv_s1_16", "conv", 2, [64], 18, 17, 128, 1, True),           # dy 17x16: 4x16 tiles, ragged rows
    ("convT_cat", "convT", 2, [128, 128], 16, 16, 64, 2, False),     # decoder: direct = concat input, gathered = dy 32x32
    ("convT_wide", "convT", 1, [128], 16, 48, 64, 2, True),          # two tile columns of 4x16 ... (48 wide: 2x32 tiles)
]


@pytest.mark.parametrize("case", WGP_CASES, ids=[c[0] for c in WGP_CASES])
def test_wgrad_patch_kernel(case, ws):
    """Weight gradients on the patch kernel of wgrad_patch.hip (pfa_set_knob(4, 8) routes launches this small there): the
    oracle and the older kernel (knob 4 = 7) - both strides, both tile shapes, ragged maps, the decoder's virtual concat on
    the direct side, float atomics and the slab + ordered reduce; one and two tap rows per workgroup (knob 13)."""
    name, kind, n, cins, h, w, co, stride, slab = case
    dtype = torch.bfloat16
    ci = sum(cins)
    x = rnd(fill.uniform(name + ".pw.x", (n, ci, h, w)), dtype)
    if kind == "conv":
        ho, wo = (h + 2 - 4) // stride + 1, (w + 2 - 4) // stride + 1
        dy = rnd(fill.uniform(name + ".pw.dy", (n, co, ho, wo)), dtype)
        _, want, _ = ops.conv2d_bwd(x, fill.uniform(name + ".pw.w", (co, ci, 4, 4)), dy, stride, need_dx=False)
        rows, cols = co, ci
    else:
        dy = rnd(fill.uniform(name + ".pw.dy", (n, co, 2 * h, 2 * w)), dtype)
        _, want, _ = ops.convT2d_bwd(x, fill.uniform(name + ".pw.w", (ci, co, 4, 4)), dy, need_dx=False)
        rows, cols = ci, co
    srcs, o = [], 0
    for c in cins:
        srcs.append(to_nhwc(x[:, o:o + c], dtype))
        o += c
    dyt = to_nhwc(dy, dtype)
    lib = _lib.lib()
    res = []
    for knob, nky in ((8, 1), (7, 1), (8, 2)):                                           # (knob 13 = 2: two tap rows per workgroup)
        dw = torch.full((rows, 16, cols), 0.5, dtype=torch.float32, device="cuda")       # the result is ADDED to dw
        lib.pfa_set_knob(4, knob)
        lib.pfa_set_knob(13, nky)
        lib.pfa_gather_timing(1)
        try:
            before = _lib.launch_count()
            if kind == "conv":
                K.wgrad([dyt], srcs, dw, stride, [co], cins, ws=K.FORCE_SLAB if slab else None)
            else:
                K.wgrad(srcs, [dyt], dw, 2, cins, [co], ws=K.FORCE_SLAB if slab else None)
            launches = _lib.launch_count() - before
            torch.cuda.synchronize()
            import ctypes
            ms, nm, fl, by, sk = ctypes.c_float(), ctypes.c_char_p(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
            _lib.check(lib.pfa_gather_timing_get(0, ctypes.byref(ms), ctypes.byref(nm), ctypes.byref(fl), ctypes.byref(by),
                                                 ctypes.byref(sk)), "pfa_gather_timing_get")
        finally:
            lib.pfa_gather_timing(0)
            lib.pfa_set_knob(4, 0)
            lib.pfa_set_knob(13, 0)
        assert nm.value.decode().startswith("k_wgrad_patch") == (knob == 8), nm.value
        assert launches == (2 if slab else 1)
        res.append((dw.cpu().numpy() - 0.5).reshape(rows, 4, 4, cols).transpose(0, 3, 1, 2))
    close(res[0], want, torch.float32, name)
    close(res[0], res[1], torch.float32, name + " vs wgrad.hip", scale=np.abs(want).max())
    close(res[2], want, torch.float32, name + " two tap rows per workgroup")


WG_ROWS_CASES = [  # name, kind, N, input channel parts, H, W of the layer INPUT, output channels, stride, slices
    # (the patch kernel is kept off with knob 4 = 7 so that every case runs on wgrad.hip)
    ("row_piece", "conv", 2, [32], 8, 256, 128, 2, 1),          # dy 4 x 128: a 64-pixel step is half a row (left / right edge steps differ)
    ("row_piece_pairs", "conv", 2, [32], 8, 256, 128, 2, 4),    # the same on K-group pairs: a 128-pixel step is a whole row
    ("rows", "conv", 4, [64], 32, 32, 128, 2, 2),               # dy 16 x 16: four rows per step (eight with pairs), top / bottom edge steps
    ("image_per_step", "conv", 8, [64], 16, 16, 128, 2, 1),     # dy 8 x 8: a step is one image
    ("images_per_step", "conv", 16, [32], 8, 8, 256, 2, 2),     # dy 4 x 4: four (eight) images per step; two 128-row tiles
    ("tiny_maps", "conv", 64, [32], 4, 4, 128, 2, 1),           # dy 2 x 2: sixteen images per step
    ("stride1", "conv", 4, [32], 17, 17, 128, 1, 2),            # stride 1: dy 16 x 16 from a 17 x 17 input
    ("convT_cat", "convT", 4, [128, 128], 8, 8, 64, 2, 2),      # transposed: direct = two 128-channel inputs (a tile per source), gathered = dy 16 x 16
    ("convT_deep", "convT", 16, [128], 4, 4, 32, 2, 1),         # transposed on 4 x 4 maps, four images per step
]


@pytest.mark.parametrize("case", WG_ROWS_CASES, ids=[c[0] for c in WG_ROWS_CASES])
def test_wgrad_row_aligned_staging(case, ws):
    """k_wgrad<..., rows>: where a reduction step is a piece of a row, whole rows or whole images, both operands are staged
    with per-lane offsets fixed for the loop and one scalar per step (buffer-form LDS-DMA, hardware zero fill at the edges the
    step touches; opt-in: pfa_set_knob(1, 32768)).  Same arithmetic in the same order as the general staging: the oracle, and
    the SAME BITS as the default, through the slab combine and the fresh single-writer store."""
    name, kind, n, cins, h, w, co, stride, msplit = case
    dtype = torch.bfloat16
    ci = sum(cins)
    x = rnd(fill.uniform(name + ".wr.x", (n, ci, h, w)), dtype)
    if kind == "conv":
        ho, wo = (h + 2 - 4) // stride + 1, (w + 2 - 4) // stride + 1
        dy = rnd(fill.uniform(name + ".wr.dy", (n, co, ho, wo)), dtype)
        _, want, _ = ops.conv2d_bwd(x, fill.uniform(name + ".wr.w", (co, ci, 4, 4)), dy, stride, need_dx=False)
        rows, cols = co, ci
    else:
        dy = rnd(fill.uniform(name + ".wr.dy", (n, co, 2 * h, 2 * w)), dtype)
        _, want, _ = ops.convT2d_bwd(x, fill.uniform(name + ".wr.w", (ci, co, 4, 4)), dy, need_dx=False)
        rows, cols = ci, co
    srcs, o = [], 0
    for c in cins:
        srcs.append(to_nhwc(x[:, o:o + c], dtype))
        o += c
    dyt = to_nhwc(dy, dtype)
    lib = _lib.lib()

    def run(use_ws, fresh):
        dw = torch.zeros((rows, 16, cols), dtype=torch.float32, device="cuda")
        if kind == "conv":
            K.wgrad([dyt], srcs, dw, stride, [co], cins, msplit=msplit, ws=use_ws, fresh=fresh)
        else:
            K.wgrad(srcs, [dyt], dw, 2, cins, [co], msplit=msplit, ws=use_ws, fresh=fresh)
        torch.cuda.synchronize()
        return dw

    lib.pfa_set_knob(4, 7)
    try:
        got = {}
        for knob1 in (0, 32768):
            lib.pfa_set_knob(1, knob1)
            kernel = _last_wgrad_kernel(lambda: run(K.FORCE_SLAB, False))
            assert kernel.startswith("k_wgrad<bf16,128") and kernel.endswith(",rows>") == (knob1 == 32768), kernel
            got[knob1] = [run(K.FORCE_SLAB, False), run(K.FORCE_SLAB if msplit > 1 else None, True)]
    finally:
        lib.pfa_set_knob(4, 0)
        lib.pfa_set_knob(1, 0)
    for dw in got[32768]:
        close(dw.cpu().numpy().reshape(rows, 4, 4, cols).transpose(0, 3, 1, 2), want, torch.float32, name)
    for a, b in zip(got[0], got[32768]):
        assert torch.equal(a, b), name + ": the two stagings must give the same bits"


@pytest.mark.parametrize("case", [("conv_co320", "conv", 2, [32], 32, 32, 320, 2), ("convT_ci256_cat", "convT", 2, [128, 128], 16, 16, 24, 2),
                                  ("conv_s1_co256", "conv", 2, [16], 23, 23, 256, 1)], ids=lambda c: c[0])
def test_wgrad_ring_tile(case, ws):
    """pfa_set_knob(4, 3): the 256-row, 8-wave, 3-stage ring variant of the weight-gradient kernel (bf16, >= 256 direct
    channels; off by default) - a ragged 320-row direct operand, a virtual-concat direct operand and a stride-1 layer,
    both combine paths."""
    name, kind, n, cins, h, w, co, stride = case
    dtype = torch.bfloat16
    ci = sum(cins)
    x = rnd(fill.uniform(name + ".rw.x", (n, ci, h, w)), dtype)
    if kind == "conv":
        ho, wo = (h + 2 - 4) // stride + 1, (w + 2 - 4) // stride + 1
        dy = rnd(fill.uniform(name + ".rw.dy", (n, co, ho, wo)), dtype)
        _, want, _ = ops.conv2d_bwd(x, fill.uniform(name + ".rw.w", (co, ci, 4, 4)), dy, stride, need_dx=False)
        rows, cols = co, ci
    else:
        dy = rnd(fill.uniform(name + ".rw.dy", (n, co, 2 * h, 2 * w)), dtype)
        _, want, _ = ops.convT2d_bwd(x, fill.uniform(name + ".rw.w", (ci, co, 4, 4)), dy, need_dx=False)
        rows, cols = ci, co
    srcs, o = [], 0
    for c in cins:
        srcs.append(to_nhwc(x[:, o:o + c], dtype))
        o += c
    dyt = to_nhwc(dy, dtype)
    _lib.lib().pfa_set_knob(4, 3)
    try:
        assert n * dyt.shape[1] * dyt.shape[2] >= 2 * 3 * 64 or kind == "convT", "each of the two slices needs >= 3 steps"
        for use_ws in (None, K.FORCE_SLAB):
            dw = torch.zeros((rows, 16, cols), dtype=torch.float32, device="cuda")
            if kind == "conv":
                K.wgrad([dyt], srcs, dw, stride, [co], cins, msplit=2, ws=use_ws)
            else:
                K.wgrad(srcs, [dyt], dw, 2, cins, [co], msplit=2, ws=use_ws)
            torch.cuda.synchronize()
            got = dw.cpu().numpy().reshape(rows, 4, 4, cols).transpose(0, 3, 1, 2)
            close(got, want, torch.float32, name)
    finally:
        _lib.lib().pfa_set_knob(4, 0)


# ---------------------------------------------------------------------------- BatchNorm, losses, Adam
@pytest.mark.parametrize("dtype", DTYPES, ids=DT_IDS)
def test_batchnorm_roundtrip(dtype, ws):
    n, c, h, w = 3, 16, 9, 7
    x = rnd(fill.uniform("gbn.x", (n, c, h, w), -2, 3), dtype)
    g = fill.uniform("gbn.g", (c,), 0.5, 1.5)
    b = fill.uniform("gbn.b", (c,), -0.5, 0.5)
    rm0 = fill.uniform("gbn.rm", (c,))
    rv0 = fill.uniform("gbn.rv", (c,), 0.5, 2.0)
    y, cache, rm2, rv2, nbt2 = ops.bn_train_fwd(x, g, b, rm0, rv0, 5, repeats=2)
    xt = to_nhwc(x, dtype)
    rows = n * h * w
    # statistics partials as a GEMM epilogue would write them: here 3 arbitrary row tiles
    xf = torch.from_numpy(x.transpose(0, 2, 3, 1).reshape(rows, c)).cuda().double()
    cuts = [0, 50, 120, rows]
    psum = torch.stack([xf[a:e].sum(0) for a, e in zip(cuts[:-1], cuts[1:])]).float().contiguous()
    psq = torch.stack([(xf[a:e] ** 2).sum(0) for a, e in zip(cuts[:-1], cuts[1:])]).float().contiguous()
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    gt, bt, rm, rv = dev(g), dev(b), dev(rm0), dev(rv0)
    nbt = torch.tensor(5, dtype=torch.int64, device="cuda")
    scale, shift, mean, invstd = (torch.empty(c, device="cuda") for _ in range(4))
    K.bn_finalize(psum, psq, 3, c, rows, gt, bt, rm, rv, nbt, scale, shift, mean, invstd, repeats=2)
    torch.cuda.synchronize()
    close(rm.cpu().numpy(), rm2, torch.float32, "running_mean")
    close(rv.cpu().numpy(), rv2, torch.float32, "running_var")
    assert int(nbt) == nbt2 == 7
    out_l = torch.empty_like(xt)
    out_r = torch.empty_like(xt)
    K.bn_act(xt, scale, shift, out_l, out_r)
    close(from_nhwc(out_l, c), ops.leaky_relu(y), dtype, "bn+lrelu")
    close(from_nhwc(out_r, c), ops.relu(y), dtype, "bn+relu")
    # backward: dy = lrelu'(y) g1 + relu'(y) g2, then the BN backward
    g1 = rnd(fill.uniform("gbn.g1", (n, c, h, w)), dtype)
    g2 = rnd(fill.uniform("gbn.g2", (n, c, h, w)), dtype)
    dy_ref = ops.leaky_relu_bwd(y, g1) + ops.relu_bwd(y, g2)
    dx_ref, dg_ref, db_ref = ops.bn_train_bwd(rnd(dy_ref, dtype), g, cache)
    dy = torch.empty_like(xt)
    part, nt = K.bn_bwd_reduce(xt, scale, shift, mean, invstd, to_nhwc(g1, dtype), K.ACT_LRELU, to_nhwc(g2, dtype),
                               K.ACT_RELU, dy, ws)
    close(from_nhwc(dy, c), dy_ref, dtype, "dy")
    dgamma, dbeta = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
    coef = torch.empty(3 * c, device="cuda")
    K.bn_bwd_finalize(part, nt, c, rows, gt, mean, invstd, dgamma, dbeta, coef)
    dr = torch.empty_like(xt)
    K.bn_bwd_apply(dy, xt, coef, dr)
    torch.cuda.synchronize()
    lo = torch.float32 if dtype == torch.float32 else torch.bfloat16
    close(dgamma.cpu().numpy(), dg_ref, lo, "dgamma")
    close(dbeta.cpu().numpy(), db_ref, lo, "dbeta")
    close(from_nhwc(dr, c), dx_ref, dtype, "dx")
    # eval-mode coefficients
    K.bn_eval_coeffs(gt, bt, rm, rv, scale, shift)
    K.bn_act(xt, scale, shift, out_l, None)
    close(from_nhwc(out_l, c), ops.leaky_relu(ops.bn_eval_fwd(x, g, b, rm2, rv2)), dtype, "eval bn+lrelu")


STATS_FUSED_CASES = [  # name, N, C, H, W, partial rows
    ("c16_3tiles", 3, 16, 9, 7, 3),           # fewer channels than a group: one group of 16
    ("c64_17tiles", 2, 64, 12, 10, 17),       # one full group, more partial rows than virtual lanes
    ("c256_130tiles", 2, 256, 24, 23, 130),   # four groups, two load rounds per lane, ragged row blocks
    ("c128_512tiles", 1, 128, 40, 36, 512),   # the limit of the folded form
]


@pytest.mark.parametrize("dtype", DTYPES, ids=DT_IDS)
@pytest.mark.parametrize("case", STATS_FUSED_CASES, ids=[c[0] for c in STATS_FUSED_CASES])
def test_batchnorm_statistics_folded_into_their_consumers(case, dtype, ws):
    """pfa_bn_act_stats / pfa_bn_bwd_apply_stats (coefficient launch folded into the applying pass) against the oracle's
    BatchNorm (models.py:179,181 semantics) and against the two-launch route they replace."""
    name, n, c, h, w, nt = case
    x = rnd(fill.uniform(name + ".x", (n, c, h, w), -2, 3), dtype)
    g = fill.uniform(name + ".g", (c,), 0.5, 1.5)
    b = fill.uniform(name + ".b", (c,), -0.5, 0.5)
    rm0 = fill.uniform(name + ".rm", (c,))
    rv0 = fill.uniform(name + ".rv", (c,), 0.5, 2.0)
    y, cache, rm2, rv2, nbt2 = ops.bn_train_fwd(x, g, b, rm0, rv0, 5, repeats=2)
    xt = to_nhwc(x, dtype)
    rows = n * h * w
    assert K.bn_stats_fused_ok(dtype, c, nt) and not K.bn_stats_fused_ok(dtype, c, 513) and not K.bn_stats_fused_ok(dtype, 96, nt)
    xf = torch.from_numpy(x.transpose(0, 2, 3, 1).reshape(rows, c)).cuda().double()
    cuts = [round(i * rows / nt) for i in range(nt + 1)]          # nt row tiles of uneven length (some may be empty)
    zero = torch.zeros(c, dtype=torch.float64, device="cuda")
    psum = torch.stack([xf[a:e].sum(0) if e > a else zero for a, e in zip(cuts[:-1], cuts[1:])]).float().contiguous()
    psq = torch.stack([(xf[a:e] ** 2).sum(0) if e > a else zero for a, e in zip(cuts[:-1], cuts[1:])]).float().contiguous()
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    gt, bt = dev(g), dev(b)
    res = {}
    for route in ("folded", "two_launches"):
        rm, rv = dev(rm0), dev(rv0)
        nbt = torch.tensor(5, dtype=torch.int64, device="cuda")
        scale, shift, mean, invstd = (torch.full((c,), float("nan"), device="cuda") for _ in range(4))
        out_l, out_r = torch.empty_like(xt), torch.empty_like(xt)
        if route == "folded":
            K.bn_act_stats(xt, psum, psq, nt, rows, gt, bt, rm, rv, nbt, scale, shift, mean, invstd, out_l, out_r, repeats=2)
        else:
            K.bn_finalize(psum, psq, nt, c, rows, gt, bt, rm, rv, nbt, scale, shift, mean, invstd, repeats=2)
            K.bn_act(xt, scale, shift, out_l, out_r)
        torch.cuda.synchronize()
        assert int(nbt) == nbt2 == 7
        close(rm.cpu().numpy(), rm2, torch.float32, route + " running_mean")
        close(rv.cpu().numpy(), rv2, torch.float32, route + " running_var")
        close(from_nhwc(out_l, c), ops.leaky_relu(y), dtype, route + " bn+lrelu")
        close(from_nhwc(out_r, c), ops.relu(y), dtype, route + " bn+relu")
        res[route] = (scale, shift, mean, invstd, out_l, out_r)
    for a, e, what in zip(res["folded"], res["two_launches"], ("scale", "shift", "mean", "invstd", "out_l", "out_r")):
        # same partials, another summation order (double): the float32 coefficients agree to the last bit or two
        assert torch.allclose(a.float(), e.float(), rtol=2e-6 if what in ("scale", "shift", "mean", "invstd") else 1e-2, atol=1e-6), what
    # only one output
    only = torch.empty_like(xt)
    scale, shift, mean, invstd = res["folded"][:4]
    K.bn_act_stats(xt, psum, psq, nt, rows, gt, bt, None, None, None, scale, shift, mean, invstd, None, only, repeats=0)
    assert torch.equal(only, res["folded"][5])
    # backward
    g1 = rnd(fill.uniform(name + ".g1", (n, c, h, w)), dtype)
    g2 = rnd(fill.uniform(name + ".g2", (n, c, h, w)), dtype)
    dy_ref = ops.leaky_relu_bwd(y, g1) + ops.relu_bwd(y, g2)
    dx_ref, dg_ref, db_ref = ops.bn_train_bwd(rnd(dy_ref, dtype), g, cache)
    dy = torch.empty_like(xt)
    part, ntb = K.bn_bwd_reduce(xt, scale, shift, mean, invstd, to_nhwc(g1, dtype), K.ACT_LRELU, to_nhwc(g2, dtype),
                                K.ACT_RELU, dy, ws)
    if not K.bn_stats_fused_ok(dtype, c, ntb):          # more partial rows than the default limit: the kernel takes any number
        assert ntb > 512
        _lib.lib().pfa_set_knob(14, 4096)
    assert K.bn_stats_fused_ok(dtype, c, ntb)
    lo = torch.float32 if dtype == torch.float32 else torch.bfloat16
    outs = {}
    for route in ("folded", "two_launches"):
        dgamma, dbeta = torch.ones(c, device="cuda"), torch.full((c,), 2.0, device="cuda")     # gradients accumulate
        dr = torch.empty_like(xt)
        if route == "folded":
            K.bn_bwd_apply_stats(dy, xt, part, ntb, rows, gt, mean, invstd, dgamma, dbeta, dr)
        else:
            coef = torch.empty(3 * c, device="cuda")
            K.bn_bwd_finalize(part, ntb, c, rows, gt, mean, invstd, dgamma, dbeta, coef)
            K.bn_bwd_apply(dy, xt, coef, dr)
        torch.cuda.synchronize()
        close(dgamma.cpu().numpy() - 1.0, dg_ref, lo, route + " dgamma")
        close(dbeta.cpu().numpy() - 2.0, db_ref, lo, route + " dbeta")
        close(from_nhwc(dr, c), dx_ref, dtype, route + " dx")
        outs[route] = dr
    assert torch.allclose(outs["folded"].float(), outs["two_launches"].float(), rtol=1e-2, atol=1e-5)
    # in place on dy, without parameter gradients (the generator phase's discriminator pass)
    K.bn_bwd_apply_stats(dy, xt, part, ntb, rows, gt, mean, invstd, None, None, dy)
    assert torch.equal(dy, outs["folded"])
    _lib.lib().pfa_set_knob(14, 0)


FINISH_BN_CASES = [  # name, mode, N, cin parts, H, W (input), Co
    ("e5_like", "conv", 4, [64], 16, 16, 64),            # 8x8 maps: 256 rows, registers keep the slice sums
    ("e6_ragged", "conv", 3, [128], 6, 10, 32),          # 3x5 maps: 45 rows, two column groups
    ("d6_like", "tconv", 2, [64, 64], 4, 4, 64),         # transposed: 4 parity classes, two sources, 128 rows
    ("d5_rows", "tconv", 16, [256], 7, 7, 32),           # 16 x 14 x 14 = 3136 rows: the variant that sums the slices twice
]


@pytest.mark.parametrize("case", FINISH_BN_CASES, ids=[c[0] for c in FINISH_BN_CASES])
def test_whole_batch_finish_forward(case, ws):
    """pfa_gather_gemm(PFA_FUSE_DEFER) + pfa_finish_bn_fwd: K-slice combine, BatchNorm training forward (statistics, running
    buffers x2, coefficients for the backward) and both activations in one launch - against the oracle's conv + BatchNorm,
    and against the three-launch route (finish pass, bn_finalize, bn_act)."""
    _finish_forward(case, ws, coop=False)


FINISH_COOP_CASES = [  # name, mode, N, cin parts, H, W (input), Co (a multiple of 64)
    ("e5_like", "conv", 16, [128], 16, 16, 128),         # 8x8 maps at B = 16: 1024 rows, one row pass, 32 row blocks x 2 groups
    ("d5_like", "tconv", 4, [64, 64], 16, 16, 64),       # 32x32 outputs: 4096 rows, four row passes per thread
    ("ragged", "conv", 3, [64], 12, 20, 64),             # 6x10 maps: 180 rows, the last row block is partly empty
    ("two_passes", "tconv", 3, [128], 13, 13, 128),      # 26x26 outputs: 2028 rows, two passes, ragged
]


@pytest.mark.parametrize("case", FINISH_COOP_CASES, ids=[c[0] for c in FINISH_COOP_CASES])
def test_meeting_finish_forward(case, ws):
    """pfa_gather_gemm(PFA_FUSE_DEFER) + pfa_finish_coop_fwd (the workgroups of a 64-column group meet once inside the launch):
    the same checks as the whole-batch finish, the counters left zero, no launch gave up."""
    _finish_forward(case, ws, coop=True)


def _finish_forward(case, ws, coop):
    name, kind, n, cins, h, w, co = case
    dtype = torch.bfloat16
    ci = sum(cins)
    x = rnd(fill.uniform(name + ".bx", (n, ci, h, w)), dtype)
    if kind == "conv":
        wt = rnd(fill.uniform(name + ".bw", (co, ci, 4, 4), -0.25, 0.25), dtype)
        raw = ops.conv2d_fwd(x, wt, None, 2)
        mode, grid = K.CONV_S2, (n, h // 2, w // 2)
        wsh, _ = shadows(wt, dtype, cparts=[(c, c) for c in cins], want=("same",))
    else:
        wt = rnd(fill.uniform(name + ".bw", (ci, co, 4, 4), -0.25, 0.25), dtype)
        raw = ops.convT2d_fwd(x, wt, None)
        mode, grid = K.TCONV_S2, (n, h, w)
        _, wsh = shadows(wt, dtype, want=("tr",))
    ho, wo = raw.shape[2:]
    rows = n * ho * wo
    srcs, o = [], 0
    for c in cins:
        srcs.append(to_nhwc(x[:, o:o + c], dtype))
        o += c
    if coop:
        assert _lib.lib().pfa_finish_coop_supported(K.dt_code(dtype), rows, co)
    else:
        assert K.defer_ok(mode, srcs, co, grid, rows) == (rows <= K.FINISH_BN_ROWS)      # the engine's threshold; the kernels take more
        assert _lib.lib().pfa_finish_bn_supported(K.dt_code(dtype), rows, co)
    bn = torch.nn.BatchNorm2d(co).cuda()
    g, b = fill.uniform(name + ".bg", (co,), 0.5, 1.5), fill.uniform(name + ".bb", (co,), -0.5, 0.5)
    rm0, rv0 = fill.uniform(name + ".brm", (co,)), fill.uniform(name + ".brv", (co,), 0.5, 2.0)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    with torch.no_grad():
        bn.weight.copy_(dev(g)); bn.bias.copy_(dev(b)); bn.running_mean.copy_(dev(rm0)); bn.running_var.copy_(dev(rv0))
        bn.num_batches_tracked.fill_(5)
    raw_r = rnd(raw, dtype)                                             # the stored conv output is what is normalised
    y, cache, rm2, rv2, nbt2 = ops.bn_train_fwd(raw_r, g, b, rm0, rv0, 5, repeats=2)
    _, cache_x, rm_x, rv_x, _ = ops.bn_train_fwd(raw, g, b, rm0, rv0, 5, repeats=2)   # statistics come from the float32 sums
    shape = (n, ho, wo, co)
    t_raw, t_l, t_r = (torch.full(shape, 7.0, dtype=dtype, device="cuda") for _ in range(3))
    scale, shift, mean, invstd = (torch.empty(co, device="cuda") for _ in range(4))
    before = _lib.launch_count()
    slab, planes = K.gather_gemm(mode, srcs, wsh, co, [t_raw], grid, ws, defer=True)
    assert planes > 1
    if coop:
        K.finish_coop_fwd(slab, planes, co, bn, 2, t_raw, t_l, t_r, scale, shift, mean, invstd, ws)
    else:
        K.finish_bn_fwd(slab, planes, co, bn, 2, t_raw, t_l, t_r, scale, shift, mean, invstd)
    assert _lib.launch_count() - before == 2
    torch.cuda.synchronize()
    if coop:
        assert not K.coop_gave_up(ws) and int(ws._bufs["coop_bar"][2]) == 0
    amax = np.abs(raw).max()
    close(from_nhwc(t_raw, co), raw, dtype, name + ".raw", scale=amax)
    close(mean.cpu().numpy(), raw.astype(np.float64).mean(axis=(0, 2, 3)), torch.float32, name + ".mean", scale=amax)
    close(invstd.cpu().numpy(), cache_x[1], torch.float32, name + ".invstd")
    close(bn.running_mean.cpu().numpy(), rm_x, torch.float32, name + ".running_mean")
    close(bn.running_var.cpu().numpy(), rv_x, torch.float32, name + ".running_var")
    assert int(bn.num_batches_tracked) == nbt2 == 7
    # normalised outputs: the oracle on the ROUNDED conv output with the statistics of the unrounded one
    inv = cache_x[1].astype(np.float64)
    mu = raw.astype(np.float64).mean(axis=(0, 2, 3))
    yy = ((raw_r - mu[None, :, None, None]) * (inv * g)[None, :, None, None] + b[None, :, None, None]).astype(np.float32)
    close(from_nhwc(t_l, co), ops.leaky_relu(yy), dtype, name + ".lrelu", scale=np.abs(yy).max())
    close(from_nhwc(t_r, co), ops.relu(yy), dtype, name + ".relu", scale=np.abs(yy).max())
    # the three-launch route on the same inputs
    with torch.no_grad():
        bn.running_mean.copy_(dev(rm0)); bn.running_var.copy_(dev(rv0)); bn.num_batches_tracked.fill_(5)
    u_raw, u_l, u_r = (torch.empty(shape, dtype=dtype, device="cuda") for _ in range(3))
    st = K.gather_gemm(mode, srcs, wsh, co, [u_raw], grid, ws, stats=True)
    sc2, sh2, mu2, is2 = (torch.empty(co, device="cuda") for _ in range(4))
    K.bn_finalize(st[0], st[1], st[2], co, rows, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                  sc2, sh2, mu2, is2, repeats=2)
    K.bn_act(u_raw, sc2, sh2, u_l, u_r)
    torch.cuda.synchronize()
    if coop:        # (the un-deferred launch may run on another kernel with other slices)
        close(from_nhwc(t_raw, co), from_nhwc(u_raw, co), dtype, name + ".raw vs 3 launches", scale=amax)
    else:
        assert torch.equal(u_raw, t_raw)                                 # same slices, same ascending combine
    close(scale.cpu().numpy(), sc2.cpu().numpy(), torch.float32, name + ".scale vs 3 launches")
    close(shift.cpu().numpy(), sh2.cpu().numpy(), torch.float32, name + ".shift vs 3 launches", scale=np.abs(sh2.cpu().numpy()).max() + 1)
    close(from_nhwc(t_l, co), from_nhwc(u_l, co), dtype, name + ".lrelu vs 3 launches", scale=np.abs(yy).max())


FINISH_BWD_CASES = [  # name, N, cin, H, W, Co, BatchNorm, second gradient, two outputs
    ("bn", 2, [64], 8, 8, 64, True, False, False),
    ("bn_g2", 3, [128], 6, 10, 32, True, True, False),
    ("act_only", 2, [64], 8, 8, 32, False, False, False),            # the innermost level: ReLU' only
    ("two_outputs", 2, [64], 8, 8, 64, True, False, True),           # [plain 32 | fused 32]: the decoder's input gradient
    ("many_rows", 16, [64], 28, 28, 16, True, True, False),          # 3136 rows
]


@pytest.mark.parametrize("case", FINISH_BWD_CASES, ids=[c[0] for c in FINISH_BWD_CASES])
def test_whole_batch_finish_backward(case, ws):
    """pfa_gather_gemm(PFA_FUSE_DEFER) + pfa_finish_bn_bwd: K-slice combine, dy = LeakyReLU'(y) g (+ ReLU'(y) g2) and the
    BatchNorm backward of dy (dgamma, dbeta accumulated) in one launch, against the oracle."""
    _finish_backward(case, ws, coop=False)


FINISH_COOP_BWD_CASES = [  # name, N, cin, H, W, Co, BatchNorm, second gradient, two outputs
    ("bn", 16, [64], 16, 16, 64, True, False, False),                # 1024 rows
    ("bn_g2_ragged", 3, [128], 12, 20, 64, True, True, False),       # 180 rows
    ("two_outputs", 4, [64], 32, 32, 128, True, False, True),        # [plain 64 | fused 64], 1024 rows: the decoder's input gradient
    ("many_rows", 16, [64], 32, 32, 64, True, True, False),          # 4096 rows: four row passes
]


@pytest.mark.parametrize("case", FINISH_COOP_BWD_CASES, ids=[c[0] for c in FINISH_COOP_BWD_CASES])
def test_meeting_finish_backward(case, ws):
    """pfa_gather_gemm(PFA_FUSE_DEFER) + pfa_finish_coop_bwd, against the oracle; the counters left zero, no launch gave up."""
    _finish_backward(case, ws, coop=True)


def _finish_backward(case, ws, coop):
    name, n, cins, h, w, co, bn, with_g2, two = case
    dtype = torch.bfloat16
    ci = sum(cins)
    x = rnd(fill.uniform(name + ".wx", (n, ci, h, w)), dtype)
    wt = rnd(fill.uniform(name + ".ww", (co, ci, 4, 4), -0.25, 0.25), dtype)
    g = ops.conv2d_fwd(x, wt, None, 2)                                  # [n, co, ho, wo]: the incoming gradient
    ho, wo = g.shape[2:]
    rows = n * ho * wo
    wsh, _ = shadows(wt, dtype, want=("same",))
    srcs = [to_nhwc(x, dtype)]
    if not coop:
        assert K.defer_ok(K.CONV_S2, srcs, co, (n, ho, wo), rows) == (rows <= K.FINISH_BN_ROWS)
    c0 = co // 2 if two else 0
    cf = co - c0
    raw = rnd(fill.uniform(name + ".wraw", (n, cf, ho, wo), -1.0, 1.0), dtype)
    g2 = rnd(fill.uniform(name + ".wg2", (n, cf, ho, wo), -1.0, 1.0), dtype) if with_g2 else None
    gam = fill.uniform(name + ".wgam", (cf,), 0.5, 1.5)
    bet = fill.uniform(name + ".wbet", (cf,), -0.3, 0.3)
    if bn:
        y, cache, _, _, _ = ops.bn_train_fwd(raw, gam, bet, np.zeros(cf, np.float32), np.ones(cf, np.float32), 0)
        mean = raw.astype(np.float64).mean(axis=(0, 2, 3)).astype(np.float32)
        invstd = cache[1]
        scale, shift = (gam * invstd).astype(np.float32), (bet - mean * gam * invstd).astype(np.float32)
    else:
        y = raw
    gf = g[:, c0:]
    d = np.where(y > 0, 1.0, 0.2) * gf + ((y > 0) * g2 if with_g2 else 0.0)
    if bn:
        want, dg_ref, db_ref = ops.bn_train_bwd(d.astype(np.float32), gam, cache)
    else:
        want = d
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    spec = dict(raw=to_nhwc(raw, dtype), act1=K.ACT_LRELU, g2=to_nhwc(g2, dtype) if with_g2 else None,
                act2=K.ACT_RELU if with_g2 else K.ACT_NONE)
    if bn:
        spec.update(scale=dev(scale), shift=dev(shift), mean=dev(mean), invstd=dev(invstd))
    plain = torch.full((n, ho, wo, c0), 7.0, dtype=dtype, device="cuda") if two else None
    dr = torch.full((n, ho, wo, cf), 7.0, dtype=dtype, device="cuda")
    dgamma, dbeta = torch.full((cf,), 0.5, device="cuda"), torch.full((cf,), 0.25, device="cuda")
    before = _lib.launch_count()
    slab, planes = K.gather_gemm(K.CONV_S2, srcs, wsh, co, [plain, dr] if two else [dr], (n, ho, wo), ws, defer=True)
    if coop:
        K.finish_coop_bwd(slab, planes, co, plain, spec, dev(gam), dgamma, dbeta, dr, ws)
    else:
        K.finish_bn_bwd(slab, planes, co, plain, spec, dev(gam) if bn else None, dgamma if bn else None, dbeta if bn else None, dr)
    assert _lib.launch_count() - before == 2 and planes > 1
    torch.cuda.synchronize()
    if coop:
        assert not K.coop_gave_up(ws) and int(ws._bufs["coop_bar"][2]) == 0
    close(from_nhwc(dr, cf), want, dtype, name + ".dr", scale=np.abs(want).max())
    if two:
        close(from_nhwc(plain, c0), g[:, :c0], dtype, name + ".plain", scale=np.abs(g).max())
    if bn:
        close((dgamma - 0.5).cpu().numpy(), dg_ref, dtype, name + ".dgamma", scale=np.abs(dg_ref).max())
        close((dbeta - 0.25).cpu().numpy(), db_ref, dtype, name + ".dbeta", scale=np.abs(db_ref).max())


def test_meeting_finish_under_uneven_load(ws):
    """The in-launch meeting of pfa_finish_coop_fwd must not depend on placement, timing or what the caches hold: the same
    launch repeated while another stream keeps the chip unevenly busy (GEMMs of changing size), its inputs and partial sums
    pre-read (L1 / L2 warm), gives the same bits every time; no launch gives up, the counters return to zero."""
    dtype = torch.bfloat16
    rows, co, planes = 4096, 512, 4
    g = torch.Generator(device="cuda").manual_seed(5)
    slab = torch.randn((planes, rows, co), device="cuda", generator=g)
    bn = torch.nn.BatchNorm2d(co).cuda()
    shape = (16, 16, 16, co)
    outs = [tuple(torch.empty(shape, dtype=dtype, device="cuda") for _ in range(3)) for _ in range(2)]
    coef = [tuple(torch.empty(co, device="cuda") for _ in range(4)) for _ in range(2)]
    side = torch.cuda.Stream()
    a = torch.randn((4096, 4096), device="cuda", dtype=dtype)
    K.finish_coop_fwd(slab, planes, co, bn, 0, *outs[0], *coef[0], ws)
    torch.cuda.synchronize()
    ref = [t.clone() for t in outs[0]] + [t.clone() for t in coef[0]]
    for it in range(40):
        with torch.cuda.stream(side):
            for k in range(1 + it % 4):
                m = 512 << (it + k) % 4
                torch.mm(a[:m], a)
        part = ws._bufs["coop_part"]
        part.sum()                                   # the granules of the previous launch, read by plain loads
        o, c = outs[it & 1], coef[it & 1]
        K.finish_coop_fwd(slab, planes, co, bn, 0, *o, *c, ws)
        if it % 8 == 7:
            torch.cuda.synchronize()
            for t, r in zip(list(o) + list(c), ref):
                assert torch.equal(t, r), it
    torch.cuda.synchronize()
    assert not K.coop_gave_up(ws) and int(ws._bufs["coop_bar"][2]) == 0


@pytest.mark.parametrize("dtype", DTYPES, ids=DT_IDS)
def test_losses(dtype, ws):
    x = rnd(fill.uniform("gl.x", (3, 1, 6, 5), -4, 4), dtype)
    xt = to_nhwc(x, dtype)
    acc = torch.zeros(1, device="cuda")
    dl = torch.full_like(xt, 7.0)
    K.bce_logits(xt, 1.0, 0.5, 0.5, acc, dl)
    K.bce_logits(xt, 0.0, 0.5, 0.5, acc, None)
    l1, g1 = ops.bce_logits(x, 1.0)
    l0, _ = ops.bce_logits(x, 0.0)
    torch.cuda.synchronize()
    close(acc.cpu().numpy()[0], 0.5 * (l1 + l0), torch.float32, "bce loss")
    close(from_nhwc(dl, 1), 0.5 * g1, dtype, "bce grad")
    assert float(dl[..., 1:].float().abs().max()) == 0.0
    f = rnd(np.tanh(fill.uniform("gl.f", (2, 3, 8, 8), -2, 2)), dtype)
    r = rnd(fill.uniform("gl.r", (2, 3, 8, 8)), dtype)
    cp = K.pad_ch(3, dtype)
    dgan_full = rnd(fill.uniform("gl.dg", (2, 2 * cp, 8, 8), -1e-3, 1e-3), dtype)
    acc.zero_()
    dpre = torch.full((2, 8, 8, cp), 7.0, dtype=dtype, device="cuda")
    K.l1_tanh_bwd(to_nhwc(f, dtype), to_nhwc(r, dtype), 3, to_nhwc(dgan_full, dtype), cp, 100.0, acc, dpre)
    ll, gl = ops.l1_loss(f, r)
    want = (dgan_full[:, cp:cp + 3] + 100.0 * gl) * (1 - f.astype(np.float64) ** 2)
    torch.cuda.synchronize()
    close(acc.cpu().numpy()[0], 100.0 * ll, torch.float32, "l1 loss")
    close(from_nhwc(dpre, 3), want, dtype, "dpre")
    cs = torch.zeros(cp, device="cuda")
    K.colsum(dpre, cs, ws)
    torch.cuda.synchronize()
    close(cs.cpu().numpy()[:3], from_nhwc(dpre, 3).astype(np.float64).sum(axis=(0, 2, 3)), torch.float32, "colsum")


@pytest.mark.parametrize("dtype", DTYPES, ids=DT_IDS)
def test_losses_from_partials_are_reproducible(dtype):
    """pfa_bce_logits_parts / pfa_l1_tanh_bwd_parts + pfa_loss_finish: one partial per workgroup added in index order - the
    oracle's losses and gradients (train_gan.py:42-43,58-69), the accumulator form's values, and the same bits on every run
    at a size with many workgroups."""
    x = rnd(fill.uniform("glp.x", (16, 1, 30, 30), -4, 4), dtype)
    xt = to_nhwc(x, dtype)
    parts = torch.full((64 + 64 + 1024,), 3.0, device="cuda")               # (stale values: every used entry is overwritten)
    dl = torch.full_like(xt, 7.0)
    n1 = K.bce_logits_parts(xt, 1.0, 0.5, 0.5, parts[0:64], dl)
    n0 = K.bce_logits_parts(xt, 0.0, 0.5, 0.5, parts[64:128], None)
    assert n1 == n0 == K.bce_logits_nparts(xt) and 1 < n1 <= 64
    f = rnd(np.tanh(fill.uniform("glp.f", (4, 3, 128, 128), -2, 2)), dtype)
    r = rnd(fill.uniform("glp.r", (4, 3, 128, 128)), dtype)
    cp = K.pad_ch(3, dtype)
    ft, rt = to_nhwc(f, dtype), to_nhwc(r, dtype)
    dpre = torch.full((4, 128, 128, cp), 7.0, dtype=dtype, device="cuda")
    nl = K.l1_tanh_bwd_parts(ft, rt, 3, None, 0, 100.0, parts[128:], dpre)
    assert 64 < nl <= 1024
    out = torch.full((2,), -1.0, device="cuda")
    K.loss_finish(parts, [(0, n1), (64, 64 + n0), (128, 128 + nl), (128, 128)], out)
    l1, g1 = ops.bce_logits(x, 1.0)
    l0, _ = ops.bce_logits(x, 0.0)
    ll, gl = ops.l1_loss(f, r)
    torch.cuda.synchronize()
    close(out.cpu().numpy()[0], 0.5 * (l1 + l0), torch.float32, "bce loss from partials")
    close(out.cpu().numpy()[1], 100.0 * ll, torch.float32, "l1 loss from partials")
    close(from_nhwc(dl, 1), 0.5 * g1, dtype, "bce grad")
    close(from_nhwc(dpre, 3), 100.0 * gl * (1 - f.astype(np.float64) ** 2), dtype, "dpre")
    acc = torch.zeros(1, device="cuda")                                      # the accumulator form agrees to rounding
    K.l1_tanh_bwd(ft, rt, 3, None, 0, 100.0, acc, None)
    assert abs(float(acc[0]) - float(out[1])) <= 1e-5 * abs(float(out[1]))
    one = torch.zeros(1, device="cuda")                                      # two ranges -> out[0] only
    K.loss_finish(parts, [(128, 128 + nl), (0, 0)], one)
    assert float(one[0]) == float(out[1])
    for _ in range(5):                                                       # bitwise repeatable
        p2 = torch.zeros_like(parts)
        K.bce_logits_parts(xt, 1.0, 0.5, 0.5, p2[0:64], None)
        K.bce_logits_parts(xt, 0.0, 0.5, 0.5, p2[64:128], None)
        K.l1_tanh_bwd_parts(ft, rt, 3, None, 0, 100.0, p2[128:], None)
        o2 = torch.zeros(2, device="cuda")
        K.loss_finish(p2, [(0, n1), (64, 64 + n0), (128, 128 + nl), (128, 128)], o2)
        assert torch.equal(o2, out)
    with pytest.raises(_lib.PfaError):
        K.loss_finish(parts, [(0, 4), (2, 1)], one)


def test_adam():
    n = 1003
    p = fill.uniform("ga.p", (n,))
    m = np.zeros_like(p)
    v = np.zeros_like(p)
    pt, mt, vt = (torch.from_numpy(a.copy()).cuda() for a in (np.pad(p, (0, 1)), np.pad(m, (0, 1)), np.pad(v, (0, 1))))
    step_dev = torch.zeros(1, dtype=torch.int32, device="cuda")
    for i in range(3):
        g = fill.uniform(f"ga.g{i}", (n,), -1e-2, 1e-2)
        p, m, v = ops.adam_step(p, g, m, v, i + 1)
        gt = torch.from_numpy(np.pad(g, (0, 1))).cuda()
        if i < 2:
            K.adam(pt[:n], gt[:n], mt[:n], vt[:n], 1e-4, 0.5, 0.999, 1e-8, i + 1)
        else:   # device-side step counter (hipGraph replay form)
            step_dev.fill_(2)
            K.adam_tick(step_dev)
            K.adam(pt[:n], gt[:n], mt[:n], vt[:n], 1e-4, 0.5, 0.999, 1e-8, 0, step_dev=step_dev)
        torch.cuda.synchronize()
        np.testing.assert_allclose(pt[:n].cpu().numpy(), p, rtol=2e-6, atol=2e-7)
    assert float(pt[n]) == 0.0


def test_argument_errors_are_reported():
    a = _lib.GatherArgs()
    with pytest.raises(_lib.PfaError, match="struct_size"):
        _lib.check(_lib.lib().pfa_gather_gemm(a, None), "pfa_gather_gemm")
    with pytest.raises(_lib.PfaError, match="GPU tensors"):
        K.pack_nhwc(torch.zeros(1, 3, 4, 4), torch.float32)



@pytest.mark.parametrize("dtype", DTYPES, ids=DT_IDS)
def test_unpack_u8_matches_to_pil_image_arithmetic(dtype):
    """pfa_unpack_u8 against `(t*0.5+0.5).mul(255).byte()` (generate_synthetic_data.py:66-71 + TF.to_pil_image), bit
    for bit, including the ends of the Tanh range."""
    x = fill.uniform("u8.x", (2, 3, 9, 7))
    x[0, :, 0, 0] = [-1.0, 1.0, 0.0]
    x[0, :, 0, 1] = [0.99999994, -0.99999994, 0.5]
    x = rnd(x, dtype)
    t = to_nhwc(x, dtype)
    got = K.unpack_u8(t, 3).cpu().numpy()
    want = (torch.from_numpy(x) * 0.5 + 0.5).mul(255).byte().permute(0, 2, 3, 1).numpy()
    assert got.shape == want.shape and np.array_equal(got, want)
