"""Tensor-level wrappers over the C ABI (one Python function per libpfa_hip entry point).

torch is used for device memory and the current HIP stream only; every arithmetic kernel is in
libpfa_hip.so.  Activations are NHWC tensors [B, H, W, Cp] (bf16 or f32) whose channel count is
padded to 16 bytes; pad lanes are zero.
"""
import ctypes
import math

import torch

from . import _lib
from ._lib import (ACT_LRELU, ACT_NONE, ACT_RELU, ACT_TANH, BF16, CONV_S1, CONV_S1_DGRAD, CONV_S2, F32, TCONV_S2,
                   GatherArgs, ScatterArgs, WgradArgs, check, lib)

SLOPE = 0.2          # nn.LeakyReLU(0.2): models.py:178,223,232,240
BN_EPS = 1e-5        # nn.BatchNorm2d defaults: models.py:179,181,231,239
BN_MOMENTUM = 0.1


def dt_code(dtype):
    if dtype == torch.bfloat16:
        return BF16
    if dtype == torch.float32:
        return F32
    raise TypeError(f"unsupported activation dtype {dtype}")


def ch_of(dtype):
    return 8 if dtype == torch.bfloat16 else 4


def pad_ch(c, dtype):
    ch = ch_of(dtype)
    return (c + ch - 1) // ch * ch


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """hipStream_t of torch's current stream on the current device (what every launch is ordered on).
    torch.cuda.current_stream() builds a Stream object through several Python layers - a third of the host time of a
    step at ~230 launches - so the raw accessor is used when this torch build has it."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


# Event records / stream waits between the step's streams: the torch call, noted when a launch list is being recorded.
def record_event(event, st):
    event.record(st)
    rec = _lib.recorder()
    if rec is not None:
        rec.note_record(event, st)


def wait_event(st, event):
    st.wait_event(event)
    rec = _lib.recorder()
    if rec is not None:
        rec.note_wait(st, event)


def wait_stream(dst, src):
    """dst.wait_stream(src)"""
    rec = _lib.recorder()
    if rec is None:
        dst.wait_stream(src)
        return
    ev = torch.cuda.Event()
    record_event(ev, src)
    wait_event(dst, ev)


SOLO_HINT = __import__("os").environ.get("PFA_SOLO_HINT", "1") != "0"      # A/B switch


def set_solo(flag):
    """Tell the library that the launches that follow run alone on the GPU (no other stream of the step has work): launch
    plans then fill the chip instead of leaving CUs to the other streams.  A library call like any other: launch lists
    record and replay it in order."""
    if SOLO_HINT:
        check(lib().pfa_set_knob(15, 1 if flag else 0), "pfa_set_knob")


def axpy(y, x, alpha=1.0):
    """y += alpha * x (float32, same length) on the current stream."""
    check(lib().pfa_axpy(ptr(y), ptr(x), y.numel(), float(alpha), stream()), "pfa_axpy")


def ptr(t):
    return None if t is None else t.data_ptr()


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.PfaError("libpfa_hip kernels need GPU tensors; got a CPU tensor")


class Workspace:
    """Caller-owned scratch the library never allocates itself (SURVEY 8b 'Ownership')."""

    def __init__(self, device):
        self.device = device
        self._bufs = {}
        self._retired = []      # outgrown buffers stay alive: a captured hipGraph may still launch kernels on them

    def get(self, name, numel, dtype=torch.float32):
        buf = self._bufs.get(name)
        if buf is None or buf.numel() < numel or buf.dtype != dtype:
            if buf is not None:
                self._retired.append(buf)
            buf = torch.empty(max(int(numel), 1), dtype=dtype, device=self.device)
            self._bufs[name] = buf
        return buf

    def get_zeroed(self, name, numel, dtype=torch.int32):
        """A buffer that holds zeros when it is created (the library's in-launch meeting counters: every launch leaves them
        zero again).  Sized once: it is never re-created behind a launch that may still use it."""
        buf = self._bufs.get(name)
        if buf is None or buf.numel() < numel or buf.dtype != dtype:
            if buf is not None:
                self._retired.append(buf)
            buf = torch.zeros(max(int(numel), 1), dtype=dtype, device=self.device)
            self._bufs[name] = buf
        return buf


# ------------------------------------------------------------------------------- gather GEMM
# Workgroups a K-sliced launch aims for.  In-situ sweeps (bench.py, same box): the training step is fastest at 256 (512 ->
# 3.41 ms, 256 -> 3.37, 128 -> 3.41, 64 -> 3.52: fewer float32 partial planes, and its other streams fill the CUs a launch
# leaves idle), the single-stream eval forward at 512 (1.094 against 1.146 ms at B = 64).  The engines set the phase.
SPLITK_TARGET = {True: int(__import__("os").environ.get("PFA_SPLITK_TARGET", "256")),
                 False: int(__import__("os").environ.get("PFA_SPLITK_TARGET_EVAL", "512"))}
_phase_train = True


def set_phase(train):
    """Called by the engines at the start of a forward / backward pass: training passes and eval forwards slice K differently."""
    global _phase_train
    _phase_train = bool(train)


def choose_splitk(mode, M, n, ctot, dtype):
    """Slice K over extra workgroups when the output grid alone cannot fill 256 CUs
    (the bottleneck layers at 8x8 and below are weight-bandwidth bound, SURVEY 7 'Hard parts')."""
    bn = 16 if n <= 16 else (64 if n <= 64 else 128)
    bm = 256 if bn == 16 else 128
    classes = 4 if mode == TCONV_S2 else 1
    blocks = math.ceil(M / bm) * math.ceil(n / bn) * classes
    ksteps = math.ceil((4 if mode == TCONV_S2 else 16) * (ctot // ch_of(dtype)) / 8)
    if ksteps < 8:
        return 1
    if blocks >= 192:
        # fewer than 384 tiles of 128 rows = fewer than 192 of the 256-row ring tile, one per CU at most: two K slices
        # fill the chip (measured with the ring tile: e3/c3 forward 48 -> 41 us, c4 dgrad 135 -> 87 us, d4 72 -> 53 us)
        return 2 if blocks < 384 and ksteps >= 32 and n % 4 == 0 else 1
    # enough slices for ~512 workgroups, at most 16 and a power of two: even slices (c5: 9 uneven slices 31 us, 8 slices
    # 24 us) and a bounded slab (the 2x2 / 4x4 layers: 32 slices 32 us, 8-16 slices 26 us; tools/tune_gather.py)
    want = max(1, min(ksteps // 4, math.ceil(SPLITK_TARGET[_phase_train] / blocks), 16))
    return 1 << (want.bit_length() - 1)


import os as _os
BN_STATS_FUSED = _os.environ.get("PFA_BN_STATS_FUSED", "1") != "0"   # A/B switch: coefficient launches folded into their consumers
FUSE_BWD = _os.environ.get("PFA_FUSE_BWD", "1") != "0"     # A/B switch for the fused dgrad epilogue
IMAGE_DGRAD_FUSE = _os.environ.get("PFA_IMAGE_DGRAD_FUSE", "1") != "0"   # A/B switch: fused backward on the image-side input gradient
FUSE_SPLIT = _os.environ.get("PFA_FUSE_SPLIT", "1") != "0"  # ... and for fusing it into the finish pass of K-sliced launches


def auto_splitk(mode, srcs, n, grid):
    if not FUSE_BWD:
        return 0            # "not 1": callers fall back to the separate backward-reduce pass
    if sum(s.shape[3] for s in srcs) < 8 * ch_of(srcs[0].dtype) and not (
            IMAGE_DGRAD_FUSE and mode == CONV_S2 and srcs[0].dtype == torch.bfloat16 and len(srcs) == 1 and n in (64, 128)):
        # narrow source (the 3-channel image gradient) on the implicit GEMM: the fused epilogue doubles that short launch
        # (104 vs 49 us in the step, tools/insitu_launches.py) while the separate streaming pass costs ~25 us.  In bf16 the
        # launch runs on the image-side kernel (conv_image.hip), whose whole-row epilogue carries the fused backward cheaply.
        return 0
    return choose_splitk(mode, grid[0] * grid[1] * grid[2], n, sum(s.shape[3] for s in srcs), srcs[0].dtype)


def fuse_ok(mode, srcs, n, grid, single_ld=None):
    """Can the activation / BatchNorm backward ride on this input-gradient launch (gather_gemm(fuse=...))?  Yes for an
    un-split launch (GEMM epilogue); for a K-sliced one when its finish pass takes the vector path: whole 16-byte column
    chunks that tile 256 lanes, and unpadded rows if there is a single output (single_ld = its row length)."""
    sk = auto_splitk(mode, srcs, n, grid)
    if sk <= 1:
        return sk == 1
    if not FUSE_SPLIT:
        return False
    return n % 4 == 0 and n // 4 <= 256 and 256 % (n // 4) == 0 and (single_ld is None or single_ld == n)


# Whole-batch finish of the small K-sliced layers: rows up to which it is taken (0 = never).  One workgroup per 16 columns
# means n / 16 = 32 workgroups read the whole slab: measured against finish pass + bn_finalize + bn_act it wins at 256 rows
# (4x4 maps at B = 16: 8.9 / 14.5 us against 15.7 / 18.9) and loses from 1024 rows on (43 us against 21: the 33 MB slab of a
# 16-slice launch needs the whole chip's bandwidth).
FINISH_BN_ROWS = int(_os.environ.get("PFA_FINISH_BN_ROWS", "256"))


def defer_ok(mode, srcs, n, grid, rows):
    """Will this launch be K-sliced on the implicit GEMM (a map too small for the patch kernels) with few enough rows that
    one workgroup can finish a column group over the whole batch (finish_bn_fwd / finish_bn_bwd)?"""
    if rows > FINISH_BN_ROWS or srcs[0].dtype != torch.bfloat16 or (grid[1] >= 8 and grid[2] >= 16):
        return False
    if not lib().pfa_finish_bn_supported(dt_code(srcs[0].dtype), rows, n):
        return False
    return choose_splitk(mode, grid[0] * grid[1] * grid[2], n, sum(t.shape[3] for t in srcs), srcs[0].dtype) > 1


def gather_gemm(mode, srcs, w, n, outs, grid, ws, bias=None, act=ACT_NONE, stats=False, splitk=None, fuse=None,
                affine=None, c_real=0, defer=False):
    """out = act(bias + implicit_gemm(srcs, w)).  srcs/outs: lists of 1 or 2 NHWC tensors.

    grid = (nb, hq, wq) position grid.  Returns (psum, psq, ntiles) views when stats=True.
    fuse: dict(out=1|2, raw=, act1=, g2=None, act2=, scale=, shift=, mean=, invstd=) - the named output receives
    dy = act1'(y)*g + act2'(y)*g2 (activation / BatchNorm backward fused into this input-gradient launch); with
    mean/invstd the call returns (part, None, ntiles) for bn_bwd_finalize.
    affine: (scale, shift, act2) - eval-mode BatchNorm in the epilogue: outs[0] = act(scale*acc + shift) and, with two
    outs, outs[1] = act2(scale*acc + shift) over the same columns (needs splitk == 1: check auto_splitk first).
    c_real: real (unpadded) contraction channels, for the FLOP accounting of pfa_gather_timing only (0 = as padded).
    defer: run the K slices only (defer_ok must hold) and return (slab, planes): the caller finishes them with
    finish_bn_fwd / finish_bn_bwd; `outs` then only give the output geometry.  defer="try": the same if the launch will be
    K-sliced (whichever kernel runs it), else nothing is launched and None is returned (finish_coop_*).
    """
    _require_cuda(*srcs, w, *outs)
    dtype = srcs[0].dtype
    a = GatherArgs()
    a.struct_size = ctypes.sizeof(GatherArgs)
    a.mode, a.dtype = mode, dt_code(dtype)
    a.nb, a.hq, a.wq = grid
    a.hs, a.ws = srcs[0].shape[1], srcs[0].shape[2]
    a.c1 = srcs[0].shape[3]
    a.c2 = srcs[1].shape[3] if len(srcs) > 1 else 0
    a.n = n
    a.ld1 = outs[0].shape[3]
    a.ld2 = outs[1].shape[3] if len(outs) > 1 else 0
    a.n1 = outs[0].shape[3]
    a.act, a.slope = act, SLOPE
    a.c_real = c_real
    M = grid[0] * grid[1] * grid[2]
    a.splitk = choose_splitk(mode, M, n, a.c1 + a.c2, dtype) if splitk is None else splitk
    a.src1, a.src2 = ptr(srcs[0]), ptr(srcs[1]) if len(srcs) > 1 else None
    a.w, a.bias = ptr(w), ptr(bias)
    a.out1, a.out2 = ptr(outs[0]), ptr(outs[1]) if len(outs) > 1 else None
    npix = outs[0].shape[0] * outs[0].shape[1] * outs[0].shape[2]
    slab = None
    if a.splitk > 1:
        slab = ws.get("slab", lib().pfa_gather_slab_floats(ctypes.byref(a)))
        a.slab = ptr(slab)
    if defer:
        a.fuse_out = _lib.FUSE_DEFER
        if defer == "try":          # only if the library will really slice K (its planners may keep one slice): else None
            if a.splitk <= 1 or lib().pfa_gather_plan_splitk(ctypes.byref(a)) <= 1:
                return None
        check(lib().pfa_gather_gemm(ctypes.byref(a), stream()), "pfa_gather_gemm")
        return slab, a.splitk
    psum = psq = None
    if affine is not None:
        a.fuse_out, a.fuse_act2 = 3, affine[2]
        a.fuse_scale, a.fuse_shift = ptr(affine[0]), ptr(affine[1])
    if fuse is not None:
        a.fuse_out, a.fuse_act1, a.fuse_act2 = fuse["out"], fuse["act1"], fuse.get("act2", ACT_NONE)
        a.fuse_raw, a.fuse_g2 = ptr(fuse["raw"]), ptr(fuse.get("g2"))
        a.fuse_scale, a.fuse_shift = ptr(fuse.get("scale")), ptr(fuse.get("shift"))
        a.fuse_mean, a.fuse_invstd = ptr(fuse.get("mean")), ptr(fuse.get("invstd"))
        if fuse.get("mean") is not None:
            tiles = lib().pfa_gather_stat_tiles(ctypes.byref(a))
            if tiles < 0:
                check(tiles, "pfa_gather_stat_tiles")
            c = outs[fuse["out"] - 1].shape[3]
            part = ws.get("bwdpart", 2 * tiles * c)
            a.psum, a.psq = ptr(part), part.data_ptr() + 4 * tiles * c
            check(lib().pfa_gather_gemm(ctypes.byref(a), stream()), "pfa_gather_gemm")
            return part, None, a.ntiles_stat
    if stats:
        tiles = lib().pfa_gather_stat_tiles(ctypes.byref(a))
        if tiles < 0:
            check(tiles, "pfa_gather_stat_tiles")
        psum = ws.get("psum", tiles * n)
        psq = ws.get("psq", tiles * n)
        a.psum, a.psq = ptr(psum), ptr(psq)
    check(lib().pfa_gather_gemm(ctypes.byref(a), stream()), "pfa_gather_gemm")
    return (psum, psq, a.ntiles_stat) if stats else None


SCATTER = _os.environ.get("PFA_SCATTER", "1") != "0"     # PFA_SCATTER=0: always take the implicit-GEMM path (A/B runs)


def tconv_scatter_ok(srcs, n, out):
    """True when tconv_scatter() can replace gather_gemm(TCONV_S2, ...) for these operands."""
    if not SCATTER or len(srcs) > 2:
        return False
    c2 = srcs[1].shape[3] if len(srcs) > 1 else 0
    return bool(lib().pfa_tconv_scatter_supported(dt_code(srcs[0].dtype), srcs[0].shape[3], c2, n, out.shape[3]))


def tconv_scatter_ok_ch(dtype, c1, c2, n, ld):
    """tconv_scatter_ok from channel counts alone (the engine decides before the second source exists)."""
    return SCATTER and bool(lib().pfa_tconv_scatter_supported(dt_code(dtype), c1, c2, n, ld))


def tconv_scatter(srcs, w, n, out, bias=None, act=ACT_NONE, l1=None, relu_src1=False):
    """out = act(bias + ConvTranspose2d_{k4,s2,p1}([srcs]))) for n <= 8 output channels: pointwise MFMA GEMM per input
    patch + col2im in LDS (csrc/tconv_scatter.hip).  Same operands and result as gather_gemm(TCONV_S2, ...).
    l1 = (fake, real, c_real, lambda): the generator phase's fused epilogue - `out` receives dL/dpre of the generator's
    output layer, what l1_tanh_bwd computes from the stored dL/dfake (see pfa_hip.h)."""
    _require_cuda(*srcs, w, out)
    a = ScatterArgs()
    a.struct_size = ctypes.sizeof(ScatterArgs)
    a.dtype = dt_code(srcs[0].dtype)
    a.nb, a.hs, a.ws = srcs[0].shape[:3]
    a.c1 = srcs[0].shape[3]
    a.c2 = srcs[1].shape[3] if len(srcs) > 1 else 0
    a.n, a.ld = n, out.shape[3]
    a.act, a.slope = act, SLOPE
    a.relu_src1 = 1 if relu_src1 else 0          # source 1 is the LeakyReLU copy of a tensor whose ReLU the layer takes
    a.src1, a.src2 = ptr(srcs[0]), ptr(srcs[1]) if len(srcs) > 1 else None
    a.w, a.bias, a.out = ptr(w), ptr(bias), ptr(out)
    if l1 is not None:
        fake, real, c_real, lam = l1
        if fake.shape != out.shape or real.shape != out.shape:
            raise _lib.PfaError("tconv_scatter: the fused L1 epilogue needs fake / real of the output's shape")
        a.l1_fake, a.l1_real, a.l1_creal = ptr(fake), ptr(real), c_real
        a.l1_lambda = float(lam)
    check(lib().pfa_tconv_scatter(ctypes.byref(a), stream()), "pfa_tconv_scatter")


WGRAD_MINSTEPS = int(_os.environ.get("PFA_WGRAD_MINSTEPS", "8"))   # reduction steps a slice keeps at least (A/B runs)
WGRAD_BLOCKS = int(_os.environ.get("PFA_WGRAD_BLOCKS", "512"))     # workgroups a weight-gradient launch aims for (A/B runs)
FORCE_SLAB = object()     # pass as ws= to force the slab combine whatever the slice count (tests)
_FORCE_WS = None


WGRAD_FRESH = _os.environ.get("PFA_WGRAD_FRESH", "1") != "0"       # A/B switch
# Reduction slices from which a slab + ordered reduce is used instead of float atomics.  3: measured faster than atomics from
# 4 slices on (e4 47 -> 36 us, c4 90 -> 81, d4 62 -> 54) and slower at 2 (e5 28 -> 37); and two atomic adds into a zeroed
# buffer commute, so from 3 on every weight gradient is bitwise reproducible run to run.
WGRAD_SLAB_MIN = int(_os.environ.get("PFA_WGRAD_SLAB_MIN", "3"))


def wgrad(direct, gathered, dw, stride, dreal, greal, msplit=None, ws=None, fresh=False):
    """dw[i][tap][c] += sum_m direct[m][i] * gathered[pix(m)+tap][c] (float32, atomically accumulated).
    fresh: dw holds zeros and nothing else is added to it (a zeroed gradient arena, one weight gradient per layer): a
    single-slice launch may then store its tiles (the deep layers' 16-33 MB of dw leave ~1.5x faster than through atomics).

    direct/gathered: lists of 1-2 NHWC tensors; dreal/greal: real channel counts per part.
    """
    _require_cuda(*direct, *gathered, dw)
    global _FORCE_WS
    if ws is FORCE_SLAB and _FORCE_WS is None:
        _FORCE_WS = Workspace(dw.device)
    a = WgradArgs()
    a.struct_size = ctypes.sizeof(WgradArgs)
    a.dtype, a.stride = dt_code(direct[0].dtype), stride
    a.nb, a.hq, a.wq = direct[0].shape[:3]
    a.hg, a.wg = gathered[0].shape[1:3]
    a.d1p, a.d1r = direct[0].shape[3], dreal[0]
    a.d2p, a.d2r = (direct[1].shape[3], dreal[1]) if len(direct) > 1 else (0, 0)
    a.g1p, a.g1r = gathered[0].shape[3], greal[0]
    a.g2p, a.g2r = (gathered[1].shape[3], greal[1]) if len(gathered) > 1 else (0, 0)
    if msplit is None:
        km = 64 if direct[0].dtype == torch.bfloat16 else 32
        M = a.nb * a.hq * a.wq
        tiles = math.ceil(16 * (a.g1p + a.g2p) / 128) * math.ceil((a.d1p + a.d2p) / 128)
        # ~2 workgroups per CU; every extra slice costs one more float32-atomic pass over dw (chip-wide ~1.3 TB/s)
        msplit = max(1, min(math.ceil(M / km / WGRAD_MINSTEPS), math.ceil(WGRAD_BLOCKS / tiles)))
    a.msplit = msplit
    a.fresh = 1 if fresh and WGRAD_FRESH else 0
    a.dir1, a.dir2 = ptr(direct[0]), ptr(direct[1]) if len(direct) > 1 else None
    a.gat1, a.gat2 = ptr(gathered[0]), ptr(gathered[1]) if len(gathered) > 1 else None
    a.dw = ptr(dw)
    slices = lib().pfa_wgrad_slices(ctypes.byref(a))       # (the library may route the call to a kernel with its own slicing)
    if slices < 0:
        check(slices, "pfa_wgrad_slices")
    if ws is not None and (slices >= WGRAD_SLAB_MIN or ws is FORCE_SLAB):
        # per-slice slabs + an ordered reduce beat ~1.3 TB/s float atomics (and are reproducible)
        a.slab = ptr((ws if ws is not FORCE_SLAB else _FORCE_WS).get("wgslab", lib().pfa_wgrad_slab_floats(ctypes.byref(a))))
    check(lib().pfa_wgrad(ctypes.byref(a), stream()), "pfa_wgrad")


# ------------------------------------------------------------------------------- whole-batch finish of small layers
def finish_bn_fwd(slab, planes, n, bn, repeats, raw, out_l, out_r, scale, shift, mean, invstd):
    """K-slice combine + BatchNorm2d training forward + LeakyReLU / ReLU outputs in one launch (see pfa_hip.h)."""
    rows = rows_of(raw)
    track = bn.track_running_stats and bn.running_mean is not None
    check(lib().pfa_finish_bn_fwd(ptr(slab), planes, rows, n, ptr(bn.weight), ptr(bn.bias), BN_EPS, BN_MOMENTUM,
                                  repeats if track else 0,
                                  ptr(bn.running_mean) if track else None, ptr(bn.running_var) if track else None,
                                  ptr(bn.num_batches_tracked) if track else None, dt_code(raw.dtype), ptr(raw), raw.shape[3],
                                  ptr(out_l), ptr(out_r), SLOPE, ptr(scale), ptr(shift), ptr(mean), ptr(invstd), stream()),
          "pfa_finish_bn_fwd")


def finish_bn_bwd(slab, planes, n, plain, fuse, gamma, dgamma, dbeta, dr):
    """K-slice combine + the fused backward of gather_gemm(fuse=...) + bn_bwd_finalize + bn_bwd_apply in one launch.
    plain: the first output of a two-output launch (columns [0, n1)) or None; dr: the fused output."""
    rows = rows_of(dr)
    n1 = plain.shape[3] if plain is not None else 0
    check(lib().pfa_finish_bn_bwd(ptr(slab), planes, rows, n, n1, ptr(plain), n1, ptr(fuse["raw"]), ptr(fuse.get("g2")),
                                  fuse["act1"], fuse.get("act2", ACT_NONE), SLOPE, ptr(fuse.get("scale")),
                                  ptr(fuse.get("shift")), ptr(fuse.get("mean")), ptr(fuse.get("invstd")), ptr(gamma),
                                  ptr(dgamma), ptr(dbeta), dt_code(dr.dtype), ptr(dr), dr.shape[3], stream()),
          "pfa_finish_bn_bwd")


# Mid-size maps (up to 4096 rows): the same two finishes on up to 256 workgroups that meet once inside the launch
# (csrc/finish_coop.hip).  OPT-IN (PFA_COOP=1): same-box A/B at B = 16 / 256x256 - the eight launch pairs it replaces per step
# are 39 us shorter on one stream (+1.1 %), the three-stream step gains +0.2 % (noise): there the GPU is busy with the other
# streams' workgroups whenever the chain waits, so a shorter chain alone buys nothing, and an in-launch hand-off is not worth
# carrying in the default path for that.
COOP = _os.environ.get("PFA_COOP", "0") == "1"
COOP_BAR_WORDS = 32                    # error flag, epoch, leave counter


def coop_ok(t, n):
    """Can the K slices of a launch whose (fused) output is the bf16 tensor t [B,h,w,n'] be finished by finish_coop_*?"""
    return (COOP and t.dtype == torch.bfloat16 and rows_of(t) > FINISH_BN_ROWS and
            bool(lib().pfa_finish_coop_supported(dt_code(t.dtype), rows_of(t), n)))


def _coop_ws(ws, rows, n):
    part = ws.get_zeroed("coop_part", max(lib().pfa_finish_coop_part_floats(rows, n), 1 << 18), torch.float32)   # (tag 0 is never an epoch)
    bar = ws.get_zeroed("coop_bar", COOP_BAR_WORDS)
    if lib().pfa_finish_coop_bar_words(n) > bar.numel():
        raise _lib.PfaError("finish_coop: more columns than the counter block was sized for")
    return part, bar


def coop_gave_up(ws):
    """Did a finish_coop launch on this workspace give up waiting for its partners (synchronises)?  Its results are garbage."""
    bar = ws._bufs.get("coop_bar")
    return bar is not None and int(bar[0].item()) != 0


def finish_coop_fwd(slab, planes, n, bn, repeats, raw, out_l, out_r, scale, shift, mean, invstd, ws):
    """finish_bn_fwd for up to 4096 rows, n % 64 == 0 (see pfa_hip.h); ws: the Workspace of the calling stream."""
    rows = rows_of(raw)
    track = bn.track_running_stats and bn.running_mean is not None
    part, bar = _coop_ws(ws, rows, n)
    check(lib().pfa_finish_coop_fwd(ptr(slab), planes, rows, n, ptr(bn.weight), ptr(bn.bias), BN_EPS, BN_MOMENTUM,
                                    repeats if track else 0,
                                    ptr(bn.running_mean) if track else None, ptr(bn.running_var) if track else None,
                                    ptr(bn.num_batches_tracked) if track else None, dt_code(raw.dtype), ptr(raw), raw.shape[3],
                                    ptr(out_l), ptr(out_r), SLOPE, ptr(scale), ptr(shift), ptr(mean), ptr(invstd), ptr(part),
                                    ptr(bar), stream()), "pfa_finish_coop_fwd")


def finish_coop_bwd(slab, planes, n, plain, fuse, gamma, dgamma, dbeta, dr, ws):
    """finish_bn_bwd for up to 4096 rows (the fused node carries a BatchNorm; n1 % 64 == 0)."""
    rows = rows_of(dr)
    n1 = plain.shape[3] if plain is not None else 0
    part, bar = _coop_ws(ws, rows, n)
    check(lib().pfa_finish_coop_bwd(ptr(slab), planes, rows, n, n1, ptr(plain), n1, ptr(fuse["raw"]), ptr(fuse.get("g2")),
                                    fuse["act1"], fuse.get("act2", ACT_NONE), SLOPE, ptr(fuse.get("scale")),
                                    ptr(fuse.get("shift")), ptr(fuse.get("mean")), ptr(fuse.get("invstd")), ptr(gamma),
                                    ptr(dgamma), ptr(dbeta), dt_code(dr.dtype), ptr(dr), dr.shape[3], ptr(part), ptr(bar),
                                    stream()), "pfa_finish_coop_bwd")


# ------------------------------------------------------------------------------- the PatchGAN logit layer
def logit_ok(x, ld):
    """Are the single-output-channel kernels built for this input [B,h,w,c]?"""
    return bool(lib().pfa_logit_supported(dt_code(x.dtype), x.shape[3], ld, x.shape[1], x.shape[2]))


def logit_fwd(x, w, bias, logits):
    B, h, wd, c = x.shape
    check(lib().pfa_logit_fwd(ptr(x), ptr(w), ptr(bias), ptr(logits), B, h, wd, c, logits.shape[3], stream()),
          "pfa_logit_fwd")


def logit_fwd_nparts(x):
    """Loss partials logit_fwd_bce writes for the input x [B,h,w,c] (one per workgroup)."""
    return int(lib().pfa_logit_fwd_nparts(x.shape[0], x.shape[1], x.shape[2]))


def logit_fwd_bce(x, w, bias, logits, target, lscale, gscale, parts, dlogits):
    """logit_fwd + bce_logits_parts in one launch.  -> number of partials written (parts[0 .. n))"""
    B, h, wd, c = x.shape
    n = int(lib().pfa_logit_fwd_nparts(B, h, wd))
    if parts.numel() < n:
        raise _lib.PfaError("logit_fwd_bce: partials buffer too small")
    check(lib().pfa_logit_fwd_bce(ptr(x), ptr(w), ptr(bias), ptr(logits), B, h, wd, c, logits.shape[3], float(target),
                                  float(lscale), float(gscale), ptr(parts), ptr(dlogits), stream()), "pfa_logit_fwd_bce")
    return n


def logit_dgrad(dlogits, w, raw, scale, shift, mean, invstd, act, dy, ws):
    """dy = act'(scale*raw+shift) * (dlogits (*) w); returns (part, ntiles) for bn_bwd_finalize, or (None, 0)."""
    B, h, wd, c = raw.shape
    part, nt = None, 0
    if mean is not None:
        nt = lib().pfa_logit_dgrad_tiles(B, h, wd)
        part = ws.get("bnpart", 2 * nt * c)
    check(lib().pfa_logit_dgrad(ptr(dlogits), dlogits.shape[3], ptr(w), ptr(raw), ptr(scale), ptr(shift), ptr(mean),
                                ptr(invstd), act, SLOPE, ptr(dy), ptr(part), B, h, wd, c, stream()), "pfa_logit_dgrad")
    return part, nt


def logit_wgrad(dlogits, x, dw, db, ws=None):
    """ws: Workspace for the partial rows of the reproducible form (ordered reduce); None: float atomics."""
    B, h, wd, c = x.shape
    slab = ws.get("logit_wgrad", lib().pfa_logit_wgrad_slab_floats(B, h)) if ws is not None else None
    check(lib().pfa_logit_wgrad(ptr(dlogits), dlogits.shape[3], ptr(x), ptr(dw), ptr(db), B, h, wd, c, ptr(slab), stream()),
          "pfa_logit_wgrad")


# ------------------------------------------------------------------------------- BatchNorm
def bn_finalize(psum, psq, ntiles, c, count, gamma, beta, running_mean, running_var, nbt, scale, shift, mean, invstd,
                repeats=1):
    check(lib().pfa_bn_finalize(ptr(psum), ptr(psq), ntiles, c, float(count), ptr(gamma), ptr(beta), BN_EPS,
                                BN_MOMENTUM, repeats, ptr(running_mean), ptr(running_var), ptr(nbt), ptr(scale),
                                ptr(shift), ptr(mean), ptr(invstd), stream()), "pfa_bn_finalize")


def bn_eval_coeffs(gamma, beta, running_mean, running_var, scale, shift):
    check(lib().pfa_bn_eval_coeffs(ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), BN_EPS,
                                   running_mean.numel(), ptr(scale), ptr(shift), stream()), "pfa_bn_eval_coeffs")


def rows_of(t):
    return t.shape[0] * t.shape[1] * t.shape[2]


def bn_act(raw, scale, shift, out_l=None, out_r=None):
    _require_cuda(raw)
    check(lib().pfa_bn_act(ptr(raw), rows_of(raw), raw.shape[3], dt_code(raw.dtype), ptr(scale), ptr(shift), SLOPE,
                           ptr(out_l), ptr(out_r), stream()), "pfa_bn_act")


# Bytes a folded pass may move (read + written).  Every workgroup of the folded form re-reads its channels' partial sums, so its
# grid stays small and the bandwidth comes from each lane's queue depth: up to ~50 MB (the 16.8 MB tensors of B = 16 at 256 x
# 256 with two outputs) it is as fast as or faster than the two launches, on the 100 MB passes of 512 x 512 the two launches
# with their 2048-workgroup apply win (same-box A/B: 512 x 512 step +0.4 % without the folded form, 256 x 256 -0.5 %).
BN_STATS_MAX_BYTES = int(_os.environ.get("PFA_BN_STATS_MAX_MB", "64")) << 20


def bn_stats_fused_ok(dtype, c, ntiles, traffic=0):
    """Should the coefficient launch be folded into the pass that applies it (bn_act_stats / bn_bwd_apply_stats)?
    traffic: bytes the pass reads + writes (0 = unknown: shape support only)."""
    return (BN_STATS_FUSED and traffic <= BN_STATS_MAX_BYTES
            and bool(lib().pfa_bn_stats_fused_supported(dt_code(dtype), c, ntiles)))


def bn_act_stats(raw, psum, psq, ntiles, count, gamma, beta, running_mean, running_var, nbt, scale, shift, mean, invstd,
                 out_l=None, out_r=None, repeats=1):
    """bn_finalize + bn_act in one launch (the statistics are finished in the prologue of every workgroup)."""
    _require_cuda(raw)
    check(lib().pfa_bn_act_stats(ptr(raw), rows_of(raw), raw.shape[3], dt_code(raw.dtype), ptr(psum), ptr(psq), ntiles,
                                 float(count), ptr(gamma), ptr(beta), BN_EPS, BN_MOMENTUM, repeats, ptr(running_mean),
                                 ptr(running_var), ptr(nbt), ptr(scale), ptr(shift), ptr(mean), ptr(invstd), SLOPE,
                                 ptr(out_l), ptr(out_r), stream()), "pfa_bn_act_stats")


def bn_bwd_apply_stats(dy, raw, part, ntiles, count, gamma, mean, invstd, dgamma, dbeta, dr):
    """bn_bwd_finalize + bn_bwd_apply in one launch."""
    check(lib().pfa_bn_bwd_apply_stats(ptr(dy), ptr(raw), rows_of(raw), raw.shape[3], dt_code(raw.dtype), ptr(part), ntiles,
                                       float(count), ptr(gamma), ptr(mean), ptr(invstd), ptr(dgamma), ptr(dbeta), ptr(dr),
                                       stream()), "pfa_bn_bwd_apply_stats")


def bn_bwd_reduce(raw, scale, shift, mean, invstd, g1, act1, g2, act2, dy, ws):
    """dy = act1'(y)*g1 + act2'(y)*g2; returns (part, ntiles) or (None, 0) when mean is None."""
    rows, c = rows_of(raw), raw.shape[3]
    part = None
    if mean is not None:
        tiles = lib().pfa_bn_bwd_tiles(rows, c, dt_code(raw.dtype))
        if tiles < 0:
            raise _lib.PfaError(f"BatchNorm backward: channel count {c} unsupported (c/chunk must divide 256)")
        part = ws.get("bnpart", 2 * tiles * c)
    nt = ctypes.c_int(0)
    check(lib().pfa_bn_bwd_reduce(ptr(raw), rows, c, dt_code(raw.dtype), ptr(scale), ptr(shift), ptr(mean),
                                  ptr(invstd), ptr(g1), act1, ptr(g2), act2, SLOPE, ptr(dy), ptr(part),
                                  ctypes.byref(nt), stream()), "pfa_bn_bwd_reduce")
    return part, nt.value


def bn_bwd_finalize(part, ntiles, c, count, gamma, mean, invstd, dgamma, dbeta, coef):
    check(lib().pfa_bn_bwd_finalize(ptr(part), ntiles, c, float(count), ptr(gamma), ptr(mean), ptr(invstd),
                                    ptr(dgamma), ptr(dbeta), ptr(coef), stream()), "pfa_bn_bwd_finalize")


def bn_bwd_apply(dy, raw, coef, dr):
    check(lib().pfa_bn_bwd_apply(ptr(dy), ptr(raw), rows_of(raw), raw.shape[3], dt_code(raw.dtype), ptr(coef),
                                 ptr(dr), stream()), "pfa_bn_bwd_apply")


# ------------------------------------------------------------------------------- losses
def bce_logits(logits, target, lscale, gscale, loss_acc, dlogits):
    check(lib().pfa_bce_logits(ptr(logits), rows_of(logits), logits.shape[3], dt_code(logits.dtype), float(target),
                               float(lscale), float(gscale), ptr(loss_acc), ptr(dlogits), stream()), "pfa_bce_logits")


def l1_tanh_bwd(fake, real, c_real, dgan, dgan_off, lam, loss_acc, dpre):
    check(lib().pfa_l1_tanh_bwd(ptr(fake), ptr(real), rows_of(fake), fake.shape[3], c_real, dt_code(fake.dtype),
                                ptr(dgan), dgan.shape[3] if dgan is not None else 0, dgan_off, float(lam),
                                ptr(loss_acc), ptr(dpre), stream()), "pfa_l1_tanh_bwd")


# Reproducible loss sums: one partial per workgroup into `parts` (a float32 tensor of at least *_nparts elements), added in
# index order by loss_finish - no float atomics, the same bits on every run.
def bce_logits_nparts(logits):
    return int(lib().pfa_bce_logits_nparts(rows_of(logits)))


def bce_logits_parts(logits, target, lscale, gscale, parts, dlogits):
    """-> number of partials written (parts[0 .. n))"""
    n = bce_logits_nparts(logits)
    if parts.numel() < n:
        raise _lib.PfaError("bce_logits_parts: partials buffer too small")
    check(lib().pfa_bce_logits_parts(ptr(logits), rows_of(logits), logits.shape[3], dt_code(logits.dtype), float(target),
                                     float(lscale), float(gscale), ptr(parts), ptr(dlogits), stream()), "pfa_bce_logits_parts")
    return n


def l1_tanh_bwd_parts(fake, real, c_real, dgan, dgan_off, lam, parts, dpre):
    n = int(lib().pfa_l1_tanh_bwd_nparts(rows_of(fake), fake.shape[3], dt_code(fake.dtype)))
    if parts.numel() < n:
        raise _lib.PfaError("l1_tanh_bwd_parts: partials buffer too small")
    check(lib().pfa_l1_tanh_bwd_parts(ptr(fake), ptr(real), rows_of(fake), fake.shape[3], c_real, dt_code(fake.dtype),
                                      ptr(dgan), dgan.shape[3] if dgan is not None else 0, dgan_off, float(lam),
                                      ptr(parts), ptr(dpre), stream()), "pfa_l1_tanh_bwd_parts")
    return n


def loss_finish(parts, ranges, out):
    """out[0] = sum of parts over ranges[0] and ranges[1]; with four ranges also out[1] = ... over ranges[2] and ranges[3].
    A range is (begin, end)."""
    if len(ranges) not in (2, 4) or any(e < b or b < 0 or e > parts.numel() for b, e in ranges):
        raise _lib.PfaError("loss_finish: two or four (begin, end) ranges inside the partials buffer")
    flat = [int(v) for r in ranges for v in r] + ([-1, -1, -1, -1] if len(ranges) == 2 else [])
    check(lib().pfa_loss_finish(ptr(parts), *flat, ptr(out), stream()), "pfa_loss_finish")


def colsum(x, out, ws, c_out=None):
    """out[j] += sum over pixels of x[..., j], j < c_out (default: len(out))."""
    rows, c = rows_of(x), x.shape[3]
    scratch = ws.get("colsum", lib().pfa_colsum_ws(rows, c))
    check(lib().pfa_colsum(ptr(x), rows, c, out.numel() if c_out is None else c_out, dt_code(x.dtype), ptr(out),
                           ptr(scratch), stream()), "pfa_colsum")


# ------------------------------------------------------------------------------- Adam, shadows, layout
def adam(p, g, m, v, lr, b1, b2, eps, step, step_dev=None, grad_scale=1.0, g2=None, grid=0):
    """grid: workgroups of the launch (0 = the library's default, 2048).  Every call states it: the setting is process-wide."""
    lib().pfa_set_knob(11, int(grid))
    check(lib().pfa_adam(ptr(p), ptr(g), ptr(g2), ptr(m), ptr(v), p.numel(), lr, b1, b2, eps, int(step),
                         ptr(step_dev), float(grad_scale), stream()), "pfa_adam")


def adam_tick(step_dev):
    check(lib().pfa_adam_tick(ptr(step_dev), stream()), "pfa_adam_tick")


def weight_shadow(master, rows, c1r, c1p, c2r, c2p, rows_p, same, tr):
    dtype = same.dtype if same is not None else tr.dtype
    check(lib().pfa_weight_shadow(ptr(master), rows, c1r, c1p, c2r, c2p, rows_p, dt_code(dtype), ptr(same), ptr(tr),
                                  stream()), "pfa_weight_shadow")


def pack_nhwc(x, dtype, out=None):
    """NCHW float32 -> NHWC `dtype` with the channel count padded to 16 bytes."""
    _require_cuda(x)
    n, c, h, w = x.shape
    x = x.contiguous().float()
    cp = pad_ch(c, dtype)
    if out is None:
        out = torch.empty((n, h, w, cp), dtype=dtype, device=x.device)
    check(lib().pfa_pack_nhwc(ptr(x), n, c, h, w, cp, dt_code(dtype), ptr(out), stream()), "pfa_pack_nhwc")
    return out


def unpack_nchw(x, c, out=None):
    """NHWC (padded) -> NCHW float32, real channels only."""
    n, h, w, cp = x.shape
    if out is None:
        out = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
    check(lib().pfa_unpack_nchw(ptr(x), n, c, h, w, cp, dt_code(x.dtype), ptr(out), stream()), "pfa_unpack_nchw")
    return out


def unpack_u8(x, c, mul=0.5, add=0.5, out=None):
    """NHWC (padded) -> uint8 [N,H,W,c] = byte((x*mul + add) * 255): the PNG bytes of generate_synthetic_data.py:66-71
    (`t*0.5+0.5` then TF.to_pil_image), produced on the device so one byte per sample crosses PCIe."""
    _require_cuda(x)
    n, h, w, cp = x.shape
    if out is None:
        out = torch.empty((n, h, w, c), dtype=torch.uint8, device=x.device)
    check(lib().pfa_unpack_u8(ptr(x), n * h * w, c, cp, dt_code(x.dtype), float(mul), float(add), ptr(out), stream()),
          "pfa_unpack_u8")
    return out


def fill_zero(t):
    check(lib().pfa_fill_zero(ptr(t), t.numel() * t.element_size(), stream()), "pfa_fill_zero")


# ------------------------------------------------------------------------------- data-parallel exchange (pfa_comm_*)
class Comm:
    """An RCCL communicator behind the C boundary (include/pfa_hip.h, "data-parallel exchange"): the transport a host
    without torch.distributed would use, and the one launch lists can hold.  allreduce() is ordered on the current stream
    like every other launch: no handle, no separate communicator stream."""
    ID_BYTES = 128

    def __init__(self, handle, nranks, rank):
        self.handle, self.nranks, self.rank = handle, nranks, rank

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(Comm.ID_BYTES)
        check(lib().pfa_comm_unique_id(buf, Comm.ID_BYTES), "pfa_comm_unique_id")
        return bytes(buf.raw)

    @classmethod
    def create(cls, ident, nranks, rank):
        """Collective over the ranks; the calling thread's current device is the rank's GPU."""
        out = ctypes.c_void_p()
        check(lib().pfa_comm_create(ident, len(ident), nranks, rank, ctypes.byref(out)), "pfa_comm_create")
        return cls(out.value, nranks, rank)

    @classmethod
    def from_process_group(cls, pg=None):
        """One communicator over the ranks of a torch.distributed group; the group only carries the identifier."""
        import torch.distributed as dist
        rank, world = dist.get_rank(pg), dist.get_world_size(pg)
        ident = [cls.unique_id() if rank == 0 else None]
        src = dist.get_global_rank(pg, 0) if pg is not None and pg is not dist.group.WORLD else 0
        dist.broadcast_object_list(ident, src=src, group=pg)
        return cls.create(ident[0], world, rank)

    def allreduce(self, t):
        """t (contiguous float32 / bfloat16 device tensor) := sum over the ranks, in place, on the current stream."""
        _require_cuda(t)
        if not t.is_contiguous():
            raise _lib.PfaError("Comm.allreduce: the range must be contiguous")
        check(lib().pfa_allreduce(self.handle, ptr(t), t.numel(), dt_code(t.dtype), stream()), "pfa_allreduce")

    def close(self):
        if self.handle is not None:
            h, self.handle = self.handle, None
            check(_lib.lib().pfa_comm_destroy(h), "pfa_comm_destroy")

