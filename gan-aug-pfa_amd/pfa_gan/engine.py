"""Whole-network executors for UNetGenerator / NLayerDiscriminator on libpfa_hip.

What the reference leaves to PyTorch's autograd and operator library (models.py:163-208,
:246-247; train_gan.py:56-70) is scheduled here explicitly, layer by layer:

  conv / convT  -> pfa_gather_gemm (MFMA implicit GEMM, BN statistics in its epilogue)
  BatchNorm     -> pfa_bn_finalize + pfa_bn_act: the raw conv output is normalised once and
                   written as the activated operand(s) its consumers read - LeakyReLU(y) for the next
                   down-conv and ReLU(y) for the decoder (ReLU(LeakyReLU(y)) == ReLU(y));
                   torch.cat of models.py:208 is never materialised (two source pointers)
  backward      -> pfa_bn_bwd_* , pfa_wgrad, pfa_gather_gemm in dgrad geometry

Activations are NHWC, bf16 or f32, channel-padded to 16 bytes.  Parameters live in one flat
float32 arena per network (every nn.Parameter is a view into it, conv weights in channels-last
memory = the kernels' [rows][16][c] layout); gradients accumulate into a matching arena.
"""
import os

import torch
import torch.nn as nn

from . import kernels as K
from ._lib import PfaError


# ------------------------------------------------------------------------------------ flat arenas
# Outermost levels whose parameter gradients - the last work of a backward pass - run on the main stream instead of behind
# the side stream's backlog (the main stream would only wait for that stream).  Same-box A/B of the step: discriminator 0 ->
# 3.22 ms, 1 -> 3.18; generator 0 / 1: within noise in round 2; with the last bucket's update on the main stream (round 3:
# the main stream then waits for the side stream's LAST weight gradient anyway) 1 is worth +0.67 % and 2 another +0.67 %
# (3: +0.2 % only; discriminator 2: -1.3 %; three same-box runs per arm, tools/ab.py).
BCE_FUSED = os.environ.get("PFA_BCE_FUSED", "1") != "0"      # the loss in the logit layer's launch (A/B switch)
D_WGRAD_MAIN_LEVELS = int(os.environ.get("PFA_D_WGRAD_MAIN_LEVELS", "1"))
G_WGRAD_MAIN_LEVELS = int(os.environ.get("PFA_G_WGRAD_MAIN_LEVELS", "2"))

class FlatParams:
    """One float32 arena for all parameters of a module (+ a gradient arena of the same shape)."""

    ALIGN = 4          # floats: every parameter starts on a 16-byte boundary (vector Adam)
    registry = {}      # id(parameter) -> FlatParams, so an optimizer built from .parameters() finds its arena

    def __init__(self, module):
        params = [p for p in module.parameters()]
        if not params:
            raise PfaError("module has no parameters")
        self.device = params[0].device
        if self.device.type != "cuda":
            raise PfaError("the HIP engine needs the module on a GPU (module.to('cuda'))")
        self.slots = {}
        off = 0
        for p in params:
            if p.dtype != torch.float32:
                raise PfaError("master parameters must be float32")
            self.slots[id(p)] = (off, p.numel())
            off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.total = off
        self.epoch = 0     # bumped by every library kernel that writes the arena behind torch's back (FusedAdam.step)
        self.data = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.params = params
        with torch.no_grad():
            for p in params:
                o, n = self.slots[id(p)]
                view = self._view(self.data, p, o, n)
                view.copy_(p.data)
                p.data = view
        self.attach_grads()
        for p in params:
            FlatParams.registry[id(p)] = self

    @staticmethod
    def _view(arena, p, o, n):
        seg = arena[o:o + n]
        if p.dim() == 4:      # [R, C, 4, 4] logical, [R][4][4][C] in memory (channels last)
            r, c = p.shape[0], p.shape[1]
            return seg.view(r, 4, 4, c).permute(0, 3, 1, 2)
        return seg.view(p.shape)

    def attach_grads(self):
        """Point every p.grad at its slice of the gradient arena (after zero_grad(set_to_none=True))."""
        for p in self.params:
            o, n = self.slots[id(p)]
            p.grad = self._view(self.grad, p, o, n)

    def seg(self, arena, p):
        o, n = self.slots[id(p)]
        return arena[o:o + n]

    def bound(self):
        p = self.params[0]
        o, _ = self.slots[id(p)]
        return p.data.data_ptr() == self.data.data_ptr() + 4 * o


def flat_of(module):
    flat = getattr(module, "_pfa_flat", None)
    if flat is None or not flat.bound():
        flat = FlatParams(module)
        object.__setattr__(module, "_pfa_flat", flat)
    return flat


class Ctx:
    """Named, shape-stable device buffers of one forward/backward pass (allocated once, reused every step)."""

    def __init__(self, device):
        self.device = device
        self._t = {}
        self.pinned = False     # set by whoever captures a hipGraph / records a launch list on these buffers (pin())
        self._shelf = {}        # pinned contexts: (name, shape, dtype) -> buffer of a shape that is not the current one
        self.v = {}

    def pin(self):
        """A hipGraph or launch list now holds these buffers' addresses - or other streams use them without telling the
        allocator: from here on a buffer that is replaced (a batch of another shape between two replays) is shelved under
        its shape and taken up again when that shape returns, never handed back to the allocator.  Memory is bounded by the
        number of distinct shapes, not by the number of shape changes."""
        self.pinned = True

    def buf(self, name, shape, dtype):
        t = self._t.get(name)
        shape = tuple(shape)
        if t is None or tuple(t.shape) != shape or t.dtype != dtype:
            if t is not None and self.pinned:
                # a batch of another shape (a ragged last batch between two hipGraph replays): the graph keeps launching
                # on the old buffer.  An eager context (eval forwards, inference on tiles of changing size: everything on
                # the current stream) just drops it.
                self._shelf[(name, tuple(t.shape), t.dtype)] = t
            t = self._shelf.pop((name, shape, dtype), None) if self.pinned else None
            if t is None:
                t = torch.zeros(shape, dtype=dtype, device=self.device)
            self._t[name] = t
        return t


# ------------------------------------------------------------------------------------ one conv layer
class Layer:
    """A Conv2d or ConvTranspose2d with its optional BatchNorm2d, plus the kernel-side weight shadows."""

    def __init__(self, flat, dtype, conv, norm, in_parts, name):
        self.name = name
        self.flat, self.dtype = flat, dtype
        self.conv, self.w, self.b = conv, conv.weight, conv.bias
        self.transposed = isinstance(conv, nn.ConvTranspose2d)
        self.stride = conv.stride[0]
        if conv.kernel_size != (4, 4) or conv.padding != (1, 1) or self.stride not in (1, 2):
            raise PfaError(f"{name}: only 4x4 / padding 1 / stride 1|2 convolutions are built")
        if norm is not None and not isinstance(norm, nn.BatchNorm2d):
            raise NotImplementedError(f"{name}: the fused engine (GanTrainer / FusedAdam) implements nn.BatchNorm2d, the "
                                      f"reference's default (train_gan.py:138-139); with {type(norm).__name__} call the "
                                      f"modules themselves - their GPU forward/backward run layer by layer on the HIP "
                                      f"convolution kernels (pfa_gan.layerwise) under torch.optim")
        self.bn = norm
        self.cout = conv.out_channels
        self.cout_p = K.pad_ch(self.cout, dtype)
        self.in_real = list(in_parts)
        self.in_pad = [K.pad_ch(c, dtype) for c in in_parts]
        if sum(in_parts) != conv.in_channels:
            raise PfaError(f"{name}: input split {in_parts} != in_channels {conv.in_channels}")
        dev = flat.device
        if self.transposed:
            # master [Ci][16][Co]: rows are the (unpadded) concat of the inputs
            if self.in_real != self.in_pad:
                raise PfaError(f"{name}: transposed-conv input channels {in_parts} must be multiples of "
                               f"{K.ch_of(dtype)} in {dtype} mode")
            self.rows, self.cparts = conv.in_channels, [(self.cout, self.cout_p)]
        else:
            self.rows, self.cparts = self.cout, list(zip(self.in_real, self.in_pad))
        if self.bn is not None and self.cout != self.cout_p:
            raise PfaError(f"{name}: BatchNorm layers need out_channels % {K.ch_of(dtype)} == 0 in {dtype} mode")
        cp = sum(p for _, p in self.cparts)
        self.rows_p = K.pad_ch(self.rows, dtype)
        self.same = torch.zeros((self.rows, 16, cp), dtype=dtype, device=dev)
        self.tr = torch.zeros((cp, 16, self.rows_p), dtype=dtype, device=dev)

    # kernel operands -------------------------------------------------------------
    @property
    def w_fwd(self):
        return self.tr if self.transposed else self.same

    @property
    def w_bwd(self):
        return self.same if self.transposed else self.tr

    def refresh(self):
        (c1r, c1p), (c2r, c2p) = self.cparts[0], (self.cparts[1] if len(self.cparts) > 1 else (0, 0))
        K.weight_shadow(self.flat.seg(self.flat.data, self.w), self.rows, c1r, c1p, c2r, c2p, self.rows_p,
                        self.same, self.tr)

    def dw(self, garena):
        return self.flat.seg(garena, self.w)

    def db(self, garena):
        return self.flat.seg(garena, self.b)


class _Engine:
    def __init__(self, module, dtype):
        self.module, self.dtype = module, dtype
        self.flat = flat_of(module)
        self.device = self.flat.device
        self.ws = K.Workspace(self.device)           # main-stream scratch
        self.ws_side = K.Workspace(self.device)      # scratch of the parameter-gradient stream
        self.ws_side_b = K.Workspace(self.device)    # ... of the second one when they alternate
        self.ws_alt = K.Workspace(self.device)       # scratch of a whole pass that runs on another stream
        self.ws_cur = self.ws
        self.layers = []
        self._versions = None
        self.bn_epoch = 0            # bumped whenever a kernel may have written BatchNorm parameters or buffers
        self._scratch = None
        self._table = None

    def still_bound(self):
        return self.flat.bound() and getattr(self.module, "_pfa_flat", None) is self.flat

    def _param_grads(self, side, fn):
        """Run a layer's parameter-gradient launches (wgrad, bias column sums).  They depend only on the layer's
        output gradient, so with a side stream they run concurrently with the dgrad / BatchNorm chain on the main
        stream (fork here; the caller joins before the optimizer).  Their workspace is not shared with the main
        stream's kernels.  side may be a pair of streams: consecutive layers then alternate between them (each with a
        workspace of its own) and _join_sides() folds the second into the first before anything waits for `side`."""
        if side is None:
            fn(self.ws_cur)
            return
        ws = self.ws_side
        if isinstance(side, (tuple, list)):
            self._side_turn = getattr(self, "_side_turn", 0) ^ 1
            if self._side_turn and side[1] is not None:
                side, ws = side[1], self.ws_side_b
            else:
                side = side[0]
        K.wait_stream(side, torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn(ws)

    @staticmethod
    def _join_sides(side):
        if isinstance(side, (tuple, list)) and side[1] is not None:
            K.wait_stream(side[0], side[1])

    def new_ctx(self):
        return Ctx(self.device)

    def scratch_ctx(self):
        if self._scratch is None:
            self._scratch = self.new_ctx()
        return self._scratch

    def _shadow_table(self, layers=None):
        """Device-resident descriptor table so that ALL layers (or the given ones) re-emit their shadows in one launch."""
        import ctypes
        from ._lib import ShadowDesc, lib
        layers = self.layers if layers is None else layers
        n = len(layers)
        arr = (ShadowDesc * n)()
        sb = tb = 0
        for i, L in enumerate(layers):
            (c1r, c1p), (c2r, c2p) = L.cparts[0], (L.cparts[1] if len(L.cparts) > 1 else (0, 0))
            a, b = ctypes.c_int(0), ctypes.c_int(0)
            K.check(lib().pfa_shadow_blocks(L.rows, c1p, c2p, L.rows_p, ctypes.byref(a), ctypes.byref(b)), "pfa_shadow_blocks")
            arr[i] = ShadowDesc(self.flat.seg(self.flat.data, L.w).data_ptr(), L.same.data_ptr(), L.tr.data_ptr(),
                                L.rows, c1r, c1p, c2r, c2p, L.rows_p, sb, tb)
            sb += a.value
            tb += b.value
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        return host.to(self.device), n, sb, tb

    def refresh_shadows(self, force=False):
        """Re-emit the kernel-side weight copies when a master parameter changed: torch bumps Tensor._version on every
        in-place write, the library's own optimizer bumps FlatParams.epoch; the fused trainer forces it after its Adam.
        (Writes through `param.data` move neither: the train-mode module forward therefore always forces; eval-mode
        forwards do not - code that edits `.data` of an eval-mode module calls refresh_shadows(force=True) itself.)"""
        versions = (self.flat.epoch,) + tuple(L.w._version for L in self.layers)
        if force or versions != self._versions:
            self.bn_epoch += 1          # parameters were written (the fused Adam updates the norm layers' too)
            if self._table is None:
                self._table = self._shadow_table()
            tab, n, sb, tb = self._table
            K.check(K.lib().pfa_weight_shadow_batch(tab.data_ptr(), n, sb, tb, K.dt_code(self.dtype), K.stream()),
                    "pfa_weight_shadow_batch")
            self._versions = versions

    def refresh_shadows_range(self, lo, hi):
        """Re-emit the kernel-side copies of the layers whose weights lie in [lo, hi) of the parameter arena (one bucket of
        grad_buckets(): the fused trainer updates and refreshes bucket by bucket while the backward is still running)."""
        tabs = self.__dict__.setdefault("_range_tables", {})
        if (lo, hi) not in tabs:
            sub = [L for L in self.layers if lo <= self.flat.slots[id(L.w)][0] < hi]
            tabs[(lo, hi)] = self._shadow_table(sub) if sub else None
        if tabs[(lo, hi)] is not None:
            tab, n, sb, tb = tabs[(lo, hi)]
            K.check(K.lib().pfa_weight_shadow_batch(tab.data_ptr(), n, sb, tb, K.dt_code(self.dtype), K.stream()),
                    "pfa_weight_shadow_batch")

    def _gap_table(self, lo, hi):
        """Device table of the arena ranges in [lo, hi) that are not convolution weights (BatchNorm gamma / beta, biases),
        adjacent ones merged: int64 pairs {first element, length}.  -> (tensor or None, number of ranges)"""
        conv = {id(L.w) for L in self.layers}
        gaps = []
        for p in self.flat.params:
            o, n = self.flat.slots[id(p)]
            if id(p) in conv or not (lo <= o < hi):
                continue
            if gaps and gaps[-1][0] + gaps[-1][1] == o:
                gaps[-1][1] += n
            else:
                gaps.append([o, n])
        if not gaps:
            return None, 0
        return torch.tensor(gaps, dtype=torch.int64).to(self.device), len(gaps)

    def adam_refresh_range(self, opt, lo, hi, grad_scale=1.0, use_grad2=False):
        """FusedAdam.step_range(lo, hi) + refresh_shadows_range(lo, hi) in one pass over the parameters
        (pfa_adam_shadow_batch): the launch that emits the kernel-side copies of the range's layers applies Adam to their
        master elements first; the range's other parameters (BatchNorm, biases) take a small launch of their own.  The step
        count is opt's (step_begin() / the device counter), as in step_range."""
        tabs = self.__dict__.setdefault("_adam_tables", {})
        if (lo, hi) not in tabs:
            sub = [L for L in self.layers if lo <= self.flat.slots[id(L.w)][0] < hi]
            tabs[(lo, hi)] = (self._shadow_table(sub) if sub else None, self._gap_table(lo, hi))
        shadow, (gaps, ngaps) = tabs[(lo, hi)]
        tab, n, _, tb = shadow if shadow is not None else (None, 0, 0, 0)
        flat = opt.flat()
        if flat is not self.flat:
            raise PfaError("adam_refresh_range: the optimizer belongs to another module")
        g = opt.param_groups[0]
        dev_step = getattr(opt, "_range_dev", False)
        K.check(K.lib().pfa_adam_shadow_batch(K.ptr(tab), n, tb, K.dt_code(self.dtype), K.ptr(gaps), ngaps,
                                              flat.data.data_ptr(), flat.grad.data_ptr(),
                                              opt.grad2().data_ptr() if use_grad2 else None, opt.m.data_ptr(), opt.v.data_ptr(),
                                              g["lr"], g["betas"][0], g["betas"][1], g["eps"], 0 if dev_step else opt.steps,
                                              opt.step_dev.data_ptr() if dev_step else None, float(grad_scale), K.stream()),
                "pfa_adam_shadow_batch")

    def shadows_refreshed(self):
        """All ranges have been re-emitted for the current parameters (bookkeeping of refresh_shadows)."""
        self.bn_epoch += 1
        self._versions = (self.flat.epoch,) + tuple(L.w._version for L in self.layers)

    # BatchNorm plumbing ------------------------------------------------------------
    @staticmethod
    def _affine_ok(mode, srcs, n, grid, ld):
        """The affine epilogue runs in the GEMM epilogue of an un-split launch, or in the finish pass of a K-sliced one
        when that takes its vector path (whole 16-byte column chunks that tile 256 lanes, unpadded rows)."""
        if os.environ.get("PFA_AFFINE", "1") == "0":       # A/B switch
            return False
        if K.choose_splitk(mode, grid[0] * grid[1] * grid[2], n, sum(t.shape[3] for t in srcs), srcs[0].dtype) == 1:
            return True
        return n % 4 == 0 and n // 4 <= 256 and 256 % (n // 4) == 0 and ld == n

    def _eval_affine(self, L, ctx, tag):
        """(scale, shift) of the layer's eval-mode normalisation; the identity for a layer without a norm."""
        if L.bn is None:
            one = ctx.buf(tag + ".one", (L.cout,), torch.float32)
            zero = ctx.buf(tag + ".zero", (L.cout,), torch.float32)
            if ctx.v.get(tag + ".ident") is not one:        # constants: written when the buffers are (re)created only
                one.fill_(1.0)
                zero.zero_()
                ctx.v[tag + ".ident"] = one
            return one, zero
        return self._bn_coeffs(L, ctx, tag, None, 1, False, 1)

    def _bn_coeffs(self, L, ctx, tag, st, count, train, repeats):
        c = L.cout
        f32 = torch.float32
        scale, shift = ctx.buf(tag + ".scale", (c,), f32), ctx.buf(tag + ".shift", (c,), f32)
        bn = L.bn
        if train:
            psum, psq, nt = st
            mean, invstd = ctx.buf(tag + ".mean", (c,), f32), ctx.buf(tag + ".invstd", (c,), f32)
            track = bn.track_running_stats and bn.running_mean is not None
            self.bn_epoch += 1          # scale/shift now hold batch statistics; the running statistics move
            K.bn_finalize(psum, psq, nt, c, count, bn.weight, bn.bias, bn.running_mean if track else None,
                          bn.running_var if track else None, bn.num_batches_tracked if track else None,
                          scale, shift, mean, invstd, repeats=repeats if track else 0)
        else:
            # eval coefficients depend on four tensors that change rarely (load_state_dict, a training step): recompute
            # them only when one of those was written since - by torch (a tensor's version counter moves on every
            # in-place write) or by this library's kernels, which write behind torch's back (running statistics in a
            # train-mode forward, parameters in the fused Adam: both bump self.bn_epoch) - instead of one tiny launch
            # per layer and forward, 10 % of a B=16 forward
            key = (self.bn_epoch, scale.data_ptr()) + tuple((t.data_ptr(), t._version) if t is not None else None
                                              for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var))
            if ctx.v.get(tag + ".evalkey") != key:
                K.bn_eval_coeffs(bn.weight, bn.bias, bn.running_mean, bn.running_var, scale, shift)
                ctx.v[tag + ".evalkey"] = key
        return scale, shift

    def _bn_act_train(self, L, ctx, tag, raw, st, count, repeats, out_l, out_r):
        """Training-mode BatchNorm of a conv output whose per-tile partial sums `st` the convolution's epilogue wrote:
        coefficients + running buffers + normalise / activate.  One launch (every workgroup finishes the statistics of
        its own channel group in its prologue) where the shape allows it, else bn_finalize + bn_act."""
        psum, psq, nt = st
        traffic = raw.numel() * raw.element_size() * (1 + (out_l is not None) + (out_r is not None))
        if not K.bn_stats_fused_ok(raw.dtype, L.cout, nt, traffic) or raw.shape[3] != L.cout:
            scale, shift = self._bn_coeffs(L, ctx, tag, st, count, True, repeats)
            K.bn_act(raw, scale, shift, out_l, out_r)
            return
        c, f32, bn = L.cout, torch.float32, L.bn
        track = bn.track_running_stats and bn.running_mean is not None
        self.bn_epoch += 1          # scale/shift now hold batch statistics; the running statistics move
        K.bn_act_stats(raw, psum, psq, nt, count, bn.weight, bn.bias, bn.running_mean if track else None,
                       bn.running_var if track else None, bn.num_batches_tracked if track else None,
                       ctx.buf(tag + ".scale", (c,), f32), ctx.buf(tag + ".shift", (c,), f32),
                       ctx.buf(tag + ".mean", (c,), f32), ctx.buf(tag + ".invstd", (c,), f32), out_l, out_r,
                       repeats=repeats if track else 0)

    def _conv_bn_act(self, mode, srcs, L, ctx, tag, raw, grid, train, repeats, out_l, out_r):
        """raw = conv(srcs); BatchNorm (batch statistics when train); out_l = LeakyReLU(y), out_r = ReLU(y) (either may be
        None).  A K-sliced launch on a small map is finished over the whole batch in one launch (K.finish_bn_fwd) instead
        of finish pass + bn_finalize + bn_act."""
        count = raw.shape[0] * raw.shape[1] * raw.shape[2]
        if train and K.defer_ok(mode, srcs, L.cout, grid, count):
            f32, c = torch.float32, L.cout
            slab, planes = K.gather_gemm(mode, srcs, L.w_fwd, L.cout, [raw], grid, self.ws, defer=True)
            self.bn_epoch += 1
            K.finish_bn_fwd(slab, planes, c, L.bn, repeats, raw, out_l, out_r,
                            ctx.buf(tag + ".scale", (c,), f32), ctx.buf(tag + ".shift", (c,), f32),
                            ctx.buf(tag + ".mean", (c,), f32), ctx.buf(tag + ".invstd", (c,), f32))
            return
        if train and raw.shape[3] == L.cout and K.coop_ok(raw, L.cout):
            # mid-size map: if the launch is K-sliced, its slices meet in ONE launch on the whole chip (finish_coop.hip) instead
            # of finish pass + folded BatchNorm pass
            res = K.gather_gemm(mode, srcs, L.w_fwd, L.cout, [raw], grid, self.ws, defer="try")
            if res is not None:
                f32, c = torch.float32, L.cout
                self.bn_epoch += 1
                K.finish_coop_fwd(res[0], res[1], c, L.bn, repeats, raw, out_l, out_r,
                                  ctx.buf(tag + ".scale", (c,), f32), ctx.buf(tag + ".shift", (c,), f32),
                                  ctx.buf(tag + ".mean", (c,), f32), ctx.buf(tag + ".invstd", (c,), f32), self.ws)
                return
        st = K.gather_gemm(mode, srcs, L.w_fwd, L.cout, [raw], grid, self.ws, stats=train)
        if train:
            self._bn_act_train(L, ctx, tag, raw, st, count, repeats, out_l, out_r)
            return
        scale, shift = self._bn_coeffs(L, ctx, tag, st, count, train, repeats)
        K.bn_act(raw, scale, shift, out_l, out_r)

    def _dgrad_fused(self, mode, d, w, n, outs, grid, L, ctx, tag, which, raw, act1, garena, g2=None, act2=K.ACT_NONE,
                     param_grads=True):
        """Input-gradient launch whose output `which` (1 | 2 of outs) is the gradient of node `tag` (layer L): fused with
        the node's activation / BatchNorm backward.  Returns the node's conv-output gradient (= outs[which - 1])."""
        dy = outs[which - 1]
        count = dy.shape[0] * dy.shape[1] * dy.shape[2]
        spec = self._fuse_spec(L, ctx, tag, which, raw, act1, g2, act2)
        if K.defer_ok(mode, [d], n, grid, count) and (which == 1 or outs[0].shape[3] % 16 == 0) and dy.shape[3] == n - (
                outs[0].shape[3] if which == 2 else 0):
            slab, planes = K.gather_gemm(mode, [d], w, n, outs, grid, self.ws_cur, defer=True)
            bn = L.bn
            K.finish_bn_bwd(slab, planes, n, outs[0] if which == 2 else None, spec, bn.weight if bn is not None else None,
                            self.flat.seg(garena, bn.weight) if bn is not None and param_grads else None,
                            self.flat.seg(garena, bn.bias) if bn is not None and param_grads else None, dy)
            return dy
        n1 = outs[0].shape[3] if which == 2 else 0
        if L.bn is not None and n1 % 64 == 0 and dy.shape[3] == n - n1 and K.coop_ok(dy, n):
            res = K.gather_gemm(mode, [d], w, n, outs, grid, self.ws_cur, defer="try")
            if res is not None:
                bn = L.bn
                K.finish_coop_bwd(res[0], res[1], n, outs[0] if which == 2 else None, spec, bn.weight,
                                  self.flat.seg(garena, bn.weight) if param_grads else None,
                                  self.flat.seg(garena, bn.bias) if param_grads else None, dy, self.ws_cur)
                return dy
        st = K.gather_gemm(mode, [d], w, n, outs, grid, self.ws_cur, fuse=spec)
        return self._finish_fused(L, ctx, tag, raw, count, garena, dy, st, param_grads=param_grads)

    def _bn_backward(self, L, ctx, tag, raw, count, garena, g1, act1, g2=None, act2=K.ACT_NONE, param_grads=True):
        f32 = torch.float32
        c = L.cout
        dy = ctx.buf(tag + ".dy", raw.shape, raw.dtype)
        scale, shift = ctx.buf(tag + ".scale", (c,), f32), ctx.buf(tag + ".shift", (c,), f32)
        mean, invstd = ctx.buf(tag + ".mean", (c,), f32), ctx.buf(tag + ".invstd", (c,), f32)
        part, nt = K.bn_bwd_reduce(raw, scale, shift, mean, invstd, g1, act1, g2, act2, dy, self.ws_cur)
        return self._bn_bwd_finish(L, ctx, tag, raw, count, garena, dy, part, nt, param_grads)

    def _bn_bwd_finish(self, L, ctx, tag, raw, count, garena, dy, part, nt, param_grads):
        """dy (in place) -> gradient of the conv output, from the partial sums of dy and dy * xhat; dgamma / dbeta.  One
        launch where the shape allows it (see _bn_act_train), else bn_bwd_finalize + bn_bwd_apply."""
        f32, c = torch.float32, L.cout
        mean, invstd = ctx.buf(tag + ".mean", (c,), f32), ctx.buf(tag + ".invstd", (c,), f32)
        dgamma = self.flat.seg(garena, L.bn.weight) if param_grads else None
        dbeta = self.flat.seg(garena, L.bn.bias) if param_grads else None
        if K.bn_stats_fused_ok(raw.dtype, c, nt, 3 * raw.numel() * raw.element_size()) and raw.shape[3] == c:
            K.bn_bwd_apply_stats(dy, raw, part, nt, count, L.bn.weight, mean, invstd, dgamma, dbeta, dy)
            return dy
        coef = ctx.buf(tag + ".coef", (3 * c,), f32)
        K.bn_bwd_finalize(part, nt, c, count, L.bn.weight, mean, invstd, dgamma, dbeta, coef)
        K.bn_bwd_apply(dy, raw, coef, dy)
        return dy

    def _fuse_spec(self, L, ctx, tag, out, raw, act1, g2=None, act2=K.ACT_NONE):
        """gather_gemm(fuse=...) for the node 'conv output of layer L' (raw = its stored output; without a norm the
        activated output stands in: sign(act(x)) == sign(x))."""
        spec = dict(out=out, raw=raw, act1=act1, g2=g2, act2=act2)
        if L.bn is not None:
            f32, c = torch.float32, L.cout
            spec.update(scale=ctx.buf(tag + ".scale", (c,), f32), shift=ctx.buf(tag + ".shift", (c,), f32),
                        mean=ctx.buf(tag + ".mean", (c,), f32), invstd=ctx.buf(tag + ".invstd", (c,), f32))
        return spec

    def _finish_fused(self, L, ctx, tag, raw, count, garena, dy, st, param_grads=True):
        """Second half of the BatchNorm backward after a fused dgrad epilogue wrote dy and the partial sums."""
        if L.bn is None:
            return dy
        part, _, nt = st
        return self._bn_bwd_finish(L, ctx, tag, raw, count, garena, dy, part, nt, param_grads)

    def _act_backward(self, ctx, tag, act_out, g1, act1, g2=None, act2=K.ACT_NONE):
        """Conv output without a norm: its gradient is act'(.)*g, and sign(act_out) == sign(pre-activation)."""
        dr = ctx.buf(tag + ".dy", act_out.shape, act_out.dtype)
        K.bn_bwd_reduce(act_out, None, None, None, None, g1, act1, g2, act2, dr, self.ws_cur)
        return dr


ONE_COPY_E1 = os.environ.get("PFA_ONE_COPY_E1", "1") != "0"      # A/B switch: eval forward without the ReLU copy of e1's activation


# ------------------------------------------------------------------------------------ generator
class GeneratorEngine(_Engine):
    """UNetGenerator.forward / backward (models.py:163-208) on the HIP kernels."""

    def __init__(self, gen, dtype):
        super().__init__(gen, dtype)
        if getattr(gen, "_use_dropout", False):
            raise NotImplementedError("use_dropout=True is outside the fused engine (the reference's train_gan.py "
                                      "constructs the generator without it): call the module itself - its GPU "
                                      "forward/backward run layer by layer on the HIP convolution kernels "
                                      "(pfa_gan.layerwise)")
        blocks = gen.blocks()
        self.D = len(blocks)
        self.down, self.up = [], []
        enc = [b._parts["down"].out_channels for b in blocks]
        for k, b in enumerate(blocks):
            pr = b._parts
            cin = [pr["down"].in_channels]
            self.down.append(Layer(self.flat, dtype, pr["down"], pr["down_norm"], cin, f"e{k + 1}"))
            up_in = [enc[k]] if k == self.D - 1 else [enc[k], blocks[k + 1]._parts["up"].out_channels]
            self.up.append(Layer(self.flat, dtype, pr["up"], pr["up_norm"], up_in, f"d{k + 1}"))
        for k in range(1, self.D - 1):
            if self.down[k].bn is None:
                raise PfaError("unexpected generator structure: middle level without a norm layer")
        self.layers = self.down + self.up
        self.out_nc = self.up[0].cout

    def forward(self, x, ctx, train=True, bn_repeats=1, differentiable=None):
        """x: NHWC [B,H,W,cp_in].  Returns tanh output NHWC [B,H,W,cp_out] (pad channels zero).
        differentiable (default: train): a backward over this context may follow - the forward keeps every tensor it reads."""
        if differentiable is None:
            differentiable = train
        D, dt = self.D, self.dtype
        B, H, W, _ = x.shape
        if H % (1 << D) or W % (1 << D):
            raise PfaError(f"input {H}x{W} must be a multiple of 2^num_downs = {1 << D}")
        ctx.v.update(x=x, aL={}, aR={}, uR={}, train=train)
        K.set_phase(train)
        a = x
        for k in range(D):
            L = self.down[k]
            h, w = H >> (k + 1), W >> (k + 1)
            shape = (B, h, w, L.cout_p)
            if k == D - 1:          # bottom of the U: no norm, consumed through ReLU only
                aR = ctx.buf(f"e{k}.aR", shape, dt)
                K.gather_gemm(K.CONV_S2, [a], L.w_fwd, L.cout, [aR], (B, h, w), self.ws, act=K.ACT_RELU)
                ctx.v["aR"][k] = aR
                break
            if k == 0 and not train and not differentiable and ONE_COPY_E1 and D > 1 and self._affine_ok(K.CONV_S2, [a], L.cout, (B, h, w), L.cout_p) \
                    and K.tconv_scatter_ok_ch(dt, L.cout_p, self.up[1].cout_p, self.up[0].cout, self.up[0].cout_p):
                # eval: ONE copy of the outermost encoder activation (the largest tensor of the forward: 134 MB at batch 64).
                # The last layer takes ReLU of it, and relu(lrelu(y)) == relu(y): its kernel applies max(x, 0) to the staged
                # operand (pfa_scatter_args.relu_src1) instead of this launch writing a second, ReLU copy
                aL = ctx.buf(f"e{k}.aL", shape, dt)
                K.gather_gemm(K.CONV_S2, [a], L.w_fwd, L.cout, [aL], (B, h, w), self.ws, act=K.ACT_LRELU,
                              affine=self._eval_affine(L, ctx, f"e{k}") + (K.ACT_NONE,), c_real=sum(L.in_real))
                ctx.v["aL"][k], ctx.v["aR"][k] = aL, None
                a = aL
                continue
            aL, aR = ctx.buf(f"e{k}.aL", shape, dt), ctx.buf(f"e{k}.aR", shape, dt)
            if (not train or L.bn is None) and self._affine_ok(K.CONV_S2, [a], L.cout, (B, h, w), L.cout_p):
                # eval: the normalisation is a per-channel affine map known up front - apply it in the GEMM epilogue
                # and emit both readers' operands from there (no raw tensor, no separate pass).  The outermost level
                # has no norm at all (identity map), so it takes this route in training too: its backward reads the
                # activation sign from aL, never the raw output.
                K.gather_gemm(K.CONV_S2, [a], L.w_fwd, L.cout, [aL, aR], (B, h, w), self.ws, act=K.ACT_LRELU,
                              affine=self._eval_affine(L, ctx, f"e{k}") + (K.ACT_RELU,), c_real=sum(L.in_real))
                ctx.v["aL"][k], ctx.v["aR"][k] = aL, aR
                a = aL
                continue
            raw = ctx.buf(f"e{k}.raw", shape, dt)
            if L.bn is None:        # outermost: no norm
                K.gather_gemm(K.CONV_S2, [a], L.w_fwd, L.cout, [raw], (B, h, w), self.ws, c_real=sum(L.in_real))
                K.bn_act(raw, None, None, aL, aR)
            else:
                self._conv_bn_act(K.CONV_S2, [a], L, ctx, f"e{k}", raw, (B, h, w), train, bn_repeats, aL, aR)
            ctx.v["aL"][k], ctx.v["aR"][k] = aL, aR
            a = aL
        for k in range(D - 1, 0, -1):
            L = self.up[k]
            h, w = H >> (k + 1), W >> (k + 1)
            srcs = [ctx.v["aR"][k]] if k == D - 1 else [ctx.v["aR"][k], ctx.v["uR"][k + 1]]
            if not train and self._affine_ok(K.TCONV_S2, srcs, L.cout, (B, h, w), L.cout_p):
                uR = ctx.buf(f"u{k}.uR", (B, 2 * h, 2 * w, L.cout_p), dt)
                K.gather_gemm(K.TCONV_S2, srcs, L.w_fwd, L.cout, [uR], (B, h, w), self.ws, act=K.ACT_RELU,
                              affine=self._eval_affine(L, ctx, f"u{k}") + (K.ACT_NONE,))
                ctx.v["uR"][k] = uR
                continue
            raw = ctx.buf(f"u{k}.raw", (B, 2 * h, 2 * w, L.cout_p), dt)
            uR = ctx.buf(f"u{k}.uR", raw.shape, dt)
            self._conv_bn_act(K.TCONV_S2, srcs, L, ctx, f"u{k}", raw, (B, h, w), train, bn_repeats, None, uR)
            ctx.v["uR"][k] = uR
        L = self.up[0]
        fake = ctx.buf("fake", (B, H, W, L.cout_p), dt)
        if ctx.v["aR"][0] is None:                         # eval, one copy of e1's activation: ReLU on the operand's way in
            K.tconv_scatter([ctx.v["aL"][0], ctx.v["uR"][1]], L.w_fwd, L.cout, fake, bias=L.b, act=K.ACT_TANH, relu_src1=True)
            ctx.v["fake"] = fake
            return fake
        srcs = [ctx.v["aR"][0], ctx.v["uR"][1]]
        if K.tconv_scatter_ok(srcs, L.cout, fake):        # few output channels: pointwise GEMM + col2im
            K.tconv_scatter(srcs, L.w_fwd, L.cout, fake, bias=L.b, act=K.ACT_TANH)
        else:
            K.gather_gemm(K.TCONV_S2, srcs, L.w_fwd, L.cout, [fake], (B, H // 2, W // 2), self.ws, bias=L.b,
                          act=K.ACT_TANH)
        ctx.v["fake"] = fake
        return fake

    def grad_buckets(self):
        """Contiguous ranges of the gradient arena in the order the backward completes them, with the point of the backward
        at which each is final: -> (buckets, whens); whens[i] = ("dec", k): after decoder level k (0-based, outermost = 0),
        ("enc", k): after encoder level k, ("end",): when the backward ends.
          [d4..d1]            final after decoder level 3: the outer decoder levels run first (22 MB that can cross the ranks -
                              and be updated - under the whole inner decoder and encoder backward)
          [dD..d5]            final when the decoder loop ends (84 MB)
          [e_{D-2}..e_D]      after the three innermost encoder levels (50 MB)
          [e3..e_{D-3}]       after level e3
          [e1, e2]            end of backward: a small last bucket keeps the update that nothing can hide short
        Arena order is module.parameters() order: e1, e2, bn2, ..., eD, dD, bnD, ..., d2, bn2, d1 - so each bucket is one
        slice.  Shallow generators merge what they do not have."""
        off = lambda L: self.flat.slots[id(L.w)][0]
        D = self.D
        x_dec = off(self.up[D - 1])
        buckets, whens = [], []
        split = 3 if D - 1 > 3 else None          # d4's weight is the first parameter of the outer-decoder range
        if split is not None:
            x_out = off(self.up[split])
            buckets += [(x_out, self.flat.total), (x_dec, x_out)]
            whens += [("dec", split), ("dec", D - 1)]
        else:
            buckets.append((x_dec, self.flat.total))
            whens.append(("dec", D - 1))
        mid = max(D - 3, 1)
        x_mid = off(self.down[mid])
        buckets.append((x_mid, x_dec))
        whens.append(("enc", mid))
        if mid > 2:
            x_low = off(self.down[2])
            buckets += [(x_low, x_mid), (0, x_low)]
            whens += [("enc", 2), ("end",)]
        else:
            buckets.append((0, x_mid))
            whens.append(("end",))
        return buckets, whens

    def backward(self, ctx, dpre, garena, side=None, bucket_ready=None):
        """dpre: gradient w.r.t. the pre-Tanh output, NHWC.  Accumulates every parameter gradient into garena.
        side: optional second stream for the parameter-gradient kernels (join it before using garena).
        bucket_ready(i): called when every kernel that writes grad_buckets()[0][i] or reads its parameters has been issued
        (the caller may all-reduce that range and update its parameters under the rest of the backward)."""
        if not ctx.v.get("train", False):
            raise PfaError("generator backward needs a training-mode forward")
        K.set_phase(True)
        D, dt = self.D, self.dtype
        v = ctx.v
        B, H, W, _ = v["x"].shape
        g_skip, g_up = {}, {}
        ready = {}          # node tag -> conv-output gradient already produced by a fused dgrad epilogue
        whens = self.grad_buckets()[1]

        def dec_hook(k):
            # every kernel that writes the gradients of decoder levels 0..k or reads their weights has been issued
            if bucket_ready is not None and ("dec", k) in whens and k < D - 1:
                self._join_sides(side)
                bucket_ready(whens.index(("dec", k)))
        # ---- decoder, outermost first
        d = dpre
        for k in range(D):
            L = self.up[k]
            h, w = H >> (k + 1), W >> (k + 1)            # input grid of this transposed conv
            if k > 0:
                if f"u{k}" in ready:
                    d = ready.pop(f"u{k}")
                else:
                    raw = ctx.buf(f"u{k}.raw", (B, 2 * h, 2 * w, L.cout_p), dt)
                    d = self._bn_backward(L, ctx, f"u{k}", raw, B * 4 * h * w, garena, g_up[k], K.ACT_RELU)
            srcs = [v["aR"][k]] if k == D - 1 else [v["aR"][k], v["uR"][k + 1]]

            def pg(ws, L=L, srcs=srcs, d=d):
                K.wgrad(srcs, [d], L.dw(garena), 2, L.in_real, [L.cout], ws=ws, fresh=True)
                if L.b is not None:
                    K.colsum(d, L.db(garena), ws, c_out=L.cout)
            self._param_grads(side, pg)
            fusable = K.fuse_ok(K.CONV_S2, [d], L.conv.in_channels, (B, h, w),
                                single_ld=srcs[0].shape[3] if k == D - 1 else None)
            if k == D - 1:
                # the only consumer of the innermost conv output is this ReLU: its gradient is final here
                Ln = self.down[k]
                if fusable:
                    dr = ctx.buf(f"e{k}.dy", srcs[0].shape, dt)
                    ready[f"e{k}"] = self._dgrad_fused(K.CONV_S2, d, L.w_bwd, L.conv.in_channels, [dr], (B, h, w), Ln, ctx,
                                                       f"e{k}", 1, v["aR"][k], K.ACT_RELU, garena)
                else:
                    g_skip[k] = ctx.buf(f"g.skip{k}", srcs[0].shape, dt)
                    K.gather_gemm(K.CONV_S2, [d], L.w_bwd, L.conv.in_channels, [g_skip[k]], (B, h, w), self.ws)
                continue
            g_skip[k] = ctx.buf(f"g.skip{k}", srcs[0].shape, dt)
            Ln = self.up[k + 1]
            if fusable:
                dyn = ctx.buf(f"u{k + 1}.dy", srcs[1].shape, dt)
                rawn = ctx.buf(f"u{k + 1}.raw", srcs[1].shape, dt)
                ready[f"u{k + 1}"] = self._dgrad_fused(K.CONV_S2, d, L.w_bwd, L.conv.in_channels, [g_skip[k], dyn], (B, h, w),
                                                       Ln, ctx, f"u{k + 1}", 2, rawn, K.ACT_RELU, garena)
            else:
                g_up[k + 1] = ctx.buf(f"g.up{k + 1}", srcs[1].shape, dt)
                K.gather_gemm(K.CONV_S2, [d], L.w_bwd, L.conv.in_channels, [g_skip[k], g_up[k + 1]], (B, h, w), self.ws)
            dec_hook(k)
        # (the last decoder bucket: final here)
        if bucket_ready is not None:
            self._join_sides(side)
            bucket_ready(whens.index(("dec", D - 1)))
        else:
            self._join_sides(side)
        # ---- encoder, innermost first
        g_down = None
        for k in range(D - 1, -1, -1):
            L = self.down[k]
            h, w = H >> (k + 1), W >> (k + 1)
            if f"e{k}" in ready:
                dr = ready.pop(f"e{k}")
            elif k == D - 1:
                dr = self._act_backward(ctx, f"e{k}", v["aR"][k], g_skip[k], K.ACT_RELU)
            elif L.bn is None:
                dr = self._act_backward(ctx, f"e{k}", v["aL"][k], g_down, K.ACT_LRELU, g_skip[k], K.ACT_RELU)
            else:
                raw = ctx.buf(f"e{k}.raw", (B, h, w, L.cout_p), dt)
                dr = self._bn_backward(L, ctx, f"e{k}", raw, B * h * w, garena, g_down, K.ACT_LRELU, g_skip[k],
                                       K.ACT_RELU)
            src = v["x"] if k == 0 else v["aL"][k - 1]

            def pg(ws, L=L, dr=dr, src=src):
                K.wgrad([dr], [src], L.dw(garena), 2, [L.cout], L.in_real, ws=ws, fresh=True)
                if L.b is not None:
                    K.colsum(dr, L.db(garena), ws, c_out=L.cout)
            self._param_grads(None if k < G_WGRAD_MAIN_LEVELS else side, pg)
            if k > 0:
                Lp = self.down[k - 1]             # the node this input gradient belongs to
                hp, wp = H >> k, W >> k
                if K.fuse_ok(K.TCONV_S2, [dr], src.shape[3], (B, h, w), single_ld=src.shape[3]):
                    rawp = ctx.buf(f"e{k - 1}.raw", src.shape, dt) if Lp.bn is not None else v["aL"][k - 1]
                    dyp = ctx.buf(f"e{k - 1}.dy", src.shape, dt)
                    ready[f"e{k - 1}"] = self._dgrad_fused(K.TCONV_S2, dr, L.w_bwd, src.shape[3], [dyp], (B, h, w), Lp, ctx,
                                                           f"e{k - 1}", 1, rawp, K.ACT_LRELU, garena, g_skip[k - 1], K.ACT_RELU)
                else:
                    g_down = ctx.buf(f"g.down{k - 1}", src.shape, dt)
                    K.gather_gemm(K.TCONV_S2, [dr], L.w_bwd, src.shape[3], [g_down], (B, h, w), self.ws)
            # (after this level's input gradient, the last reader of its weights: the caller may update the bucket now)
            if bucket_ready is not None and ("enc", k) in whens and k > 0:
                self._join_sides(side)
                bucket_ready(whens.index(("enc", k)))
        self._join_sides(side)
        if bucket_ready is not None:
            bucket_ready(whens.index(("end",)))


# ------------------------------------------------------------------------------------ discriminator
class DiscriminatorEngine(_Engine):
    """NLayerDiscriminator.forward / backward (models.py:246-247).

    in_split: how the caller hands over the input channels, e.g. (3, 3) for (real_A, B) as two tensors -
    torch.cat((real_A, x), 1) of train_gan.py:57,59,66 then never happens; None = one tensor.
    """

    def __init__(self, disc, dtype, in_split=None):
        super().__init__(disc, dtype)
        convs, norms = [], []
        mods = list(disc.model)
        for i, m in enumerate(mods):
            if isinstance(m, nn.Conv2d):
                convs.append(m)
                nxt = mods[i + 1] if i + 1 < len(mods) else None
                norms.append(nxt if nxt is not None and not isinstance(nxt, (nn.LeakyReLU, nn.Conv2d)) else None)
        in_parts = list(in_split) if in_split else [convs[0].in_channels]
        self.in_parts = in_parts
        self.layers = []
        for i, (c, nm) in enumerate(zip(convs, norms)):
            parts = in_parts if i == 0 else [c.in_channels]
            self.layers.append(Layer(self.flat, dtype, c, nm, parts, f"c{i + 1}"))
        if self.layers[0].bn is not None or self.layers[-1].bn is not None:
            raise PfaError("unexpected discriminator structure")

    @staticmethod
    def _out_hw(L, h, w):
        return (h // 2, w // 2) if L.stride == 2 else (h - 1, w - 1)

    def forward(self, srcs, ctx, train=True, bce=None):
        """srcs: list of NHWC tensors matching in_split.  Returns logits NHWC [B,ho,wo,cp(1)].
        bce = (target, lscale, gscale, parts): BCE-with-logits of the pass (train_gan.py:58,60,67) - where the logit layer
        runs on its vector kernel the loss rides in that launch's epilogue: ctx.v["bce_nparts"] then holds the number of loss
        partials written to `parts` and ctx.buf("dlogits") the gradient; otherwise ctx.v["bce_nparts"] is None and the caller
        runs kernels.bce_logits_parts on the returned logits."""
        dt = self.dtype
        B, H, W, _ = srcs[0].shape
        ctx.v.update(srcs=list(srcs), a={}, train=train, bce_nparts=None)
        K.set_phase(train)
        a, h, w = list(srcs), H, W
        n = len(self.layers)
        for i, L in enumerate(self.layers):
            if L.stride == 2 and (h % 2 or w % 2):
                raise PfaError(f"{L.name}: stride-2 conv needs an even input, got {h}x{w}")
            ho, wo = self._out_hw(L, h, w)
            if ho < 1 or wo < 1:
                raise PfaError("input too small for this discriminator")
            mode = K.CONV_S2 if L.stride == 2 else K.CONV_S1
            shape = (B, ho, wo, L.cout_p)
            if L.bn is None:
                out = ctx.buf(f"c{i}.a", shape, dt)
                if self._logit_layer(i, a):
                    if bce is not None and BCE_FUSED and K.logit_fwd_nparts(a[0]) <= bce[3].numel():
                        target, lscale, gscale, parts = bce
                        ctx.v["bce_nparts"] = K.logit_fwd_bce(a[0], L.same, L.b, out, target, lscale, gscale, parts,
                                                              ctx.buf("dlogits", shape, dt))
                    else:
                        K.logit_fwd(a[0], L.same, L.b, out)
                else:
                    K.gather_gemm(mode, a, L.w_fwd, L.cout, [out], (B, ho, wo), self.ws, bias=L.b,
                                  act=K.ACT_LRELU if i < n - 1 else K.ACT_NONE, c_real=sum(L.in_real))
            else:
                raw = ctx.buf(f"c{i}.raw", shape, dt)
                st = K.gather_gemm(mode, a, L.w_fwd, L.cout, [raw], (B, ho, wo), self.ws, bias=L.b, stats=train)
                out = ctx.buf(f"c{i}.a", shape, dt)
                if train:
                    self._bn_act_train(L, ctx, f"c{i}", raw, st, B * ho * wo, 1, out, None)
                else:
                    scale, shift = self._bn_coeffs(L, ctx, f"c{i}", st, B * ho * wo, train, 1)
                    K.bn_act(raw, scale, shift, out, None)
            ctx.v["a"][i] = out
            a, h, w = [out], ho, wo
        return a[0]

    def _logit_layer(self, i, inputs):
        """Is layer i the one-channel logit convolution (models.py:243) on a shape its vector kernels are built for?"""
        L = self.layers[i]
        return (i == len(self.layers) - 1 and i > 0 and L.cout == 1 and L.stride == 1 and len(inputs) == 1
                and K.logit_ok(inputs[0], L.cout_p))

    def grad_bucket(self):
        """-> (lo, level): [lo, total) of the gradient arena - the two innermost convolutions (76 % of the parameters) and the
        BatchNorm between them - is final once the parameter gradients of layer `level` have been issued (arena order =
        module.parameters() order: c1, c2, bn2, c3, bn3, c4, bn4, c5)."""
        n = len(self.layers)
        L = self.layers[n - 2]
        return self.flat.slots[id(L.w)][0], n - 2

    def backward(self, ctx, dlogits, garena, param_grads=True, need_dx=False, side=None, alt_ws=False, dx_part=None,
                 fresh=True, bucket_ready=None, bucket_done=None, dx_l1=None):
        """dlogits NHWC.  param_grads=False is the generator phase: input gradient only (the reference
        computes and discards D's weight gradients there, train_gan.py:70 then :55).
        need_dx: return dL/d(input) [B,H,W,sum(in_pad)]; with dx_part=i only the gradient of input source i
        ([B,H,W,in_pad[i]]) is computed - the G phase needs dL/dfake, not dL/dreal_A.
        fresh: garena holds zeros where this pass writes weight gradients (False for a second pass that accumulates into the
        arena of a first one).
        bucket_ready(): called when every kernel that writes grad_bucket()'s range has been issued (the caller may all-reduce
        that range under the rest of the backward).
        bucket_done(): called when, in addition, every kernel that READS the parameters of that range has been issued - behind
        the input gradient of the range's outermost layer (the caller may update them under the rest of the backward).
        dx_l1 = (fake, real, c_real, lambda, out): the generator phase - the launch that computes dL/dfake writes, into `out`,
        the gradient w.r.t. the generator's pre-Tanh output instead (kernels.tconv_scatter(l1=...)) and backward returns `out`;
        ignored (None returned through dx as usual) when that launch is not the scatter kernel: check l1_fused()."""
        if not ctx.v.get("train", False):
            raise PfaError("discriminator backward needs a training-mode forward")
        K.set_phase(True)
        # alt_ws: this whole pass runs on a stream of its own, concurrently with another pass of the same engine
        self.ws_cur = self.ws_alt if alt_ws else self.ws
        try:
            return self._backward(ctx, dlogits, garena, param_grads, need_dx, side, dx_part, fresh, bucket_ready, bucket_done,
                                  dx_l1)
        finally:
            self.ws_cur = self.ws

    def l1_fused(self, ctx, dx_part):
        """Will backward(..., dx_part=, dx_l1=...) of this context's pass take the fused L1 / Tanh' epilogue?  (the first
        layer's input gradient onto one source of <= 8 channels through the scatter kernel)"""
        L = self.layers[0]
        out = ctx.v["a"][0]
        if dx_part is None or L.stride != 2:
            return False
        cp = L.in_pad[dx_part]
        return K.SCATTER and bool(K.lib().pfa_tconv_scatter_supported(K.dt_code(self.dtype), out.shape[3], 0, cp, cp))

    def _backward(self, ctx, dlogits, garena, param_grads, need_dx, side, dx_part=None, fresh=True, bucket_ready=None,
                  bucket_done=None, dx_l1=None):
        dt = self.dtype
        v = ctx.v
        n = len(self.layers)
        g = dlogits
        dx = None
        ready = None        # conv-output gradient of layer i already produced by the fused epilogue of layer i+1's dgrad
        for i in range(n - 1, -1, -1):
            L = self.layers[i]
            out = v["a"][i]
            B, ho, wo, _ = out.shape
            if i == n - 1:
                dr = g
            elif ready is not None:
                dr, ready = ready, None
            elif L.bn is None:
                dr = self._act_backward(ctx, f"c{i}", out, g, K.ACT_LRELU)
            else:
                raw = ctx.buf(f"c{i}.raw", out.shape, dt)
                dr = self._bn_backward(L, ctx, f"c{i}", raw, B * ho * wo, garena, g, K.ACT_LRELU,
                                       param_grads=param_grads)
            inputs = v["srcs"] if i == 0 else [v["a"][i - 1]]
            logit = self._logit_layer(i, inputs)
            if param_grads:
                def pg(ws, L=L, dr=dr, inputs=inputs, logit=logit):
                    if logit:
                        K.logit_wgrad(dr, inputs[0], L.dw(garena), L.db(garena) if L.b is not None else None, ws=ws)
                        return
                    K.wgrad([dr], inputs, L.dw(garena), L.stride, [L.cout], L.in_real, ws=ws, fresh=fresh)
                    if L.b is not None:
                        K.colsum(dr, L.db(garena), ws, c_out=L.cout)
                self._param_grads(None if i < D_WGRAD_MAIN_LEVELS else side, pg)
                if bucket_ready is not None and i == n - 2:
                    # (this layer's BatchNorm gradients were written with the fused finish of the layer above, on this stream)
                    bucket_ready()
            if logit:
                # input gradient fused with the LeakyReLU / BatchNorm backward of the layer below
                Lp, prev = self.layers[i - 1], v["a"][i - 1]
                tag = f"c{i - 1}"
                rawp = ctx.buf(tag + ".raw", prev.shape, dt) if Lp.bn is not None else prev
                dyp = ctx.buf(tag + ".dy", prev.shape, dt)
                f = self._fuse_spec(Lp, ctx, tag, 1, rawp, K.ACT_LRELU)
                part, nt = K.logit_dgrad(dr, L.same, rawp, f.get("scale"), f.get("shift"), f.get("mean"), f.get("invstd"),
                                         K.ACT_LRELU, dyp, self.ws_cur)
                hp, wp = prev.shape[1:3]
                ready = self._finish_fused(Lp, ctx, tag, rawp, B * hp * wp, garena, dyp, (part, None, nt),
                                           param_grads=param_grads)
                continue
            if i > 0 or need_dx:
                hin, win = inputs[0].shape[1:3]
                cp = sum(L.in_pad)
                mode = K.TCONV_S2 if L.stride == 2 else K.CONV_S1_DGRAD
                grid = (B, ho, wo) if L.stride == 2 else (B, hin, win)
                if i > 0 and K.fuse_ok(mode, [dr], cp, grid, single_ld=v["a"][i - 1].shape[3]):
                    Lp = self.layers[i - 1]
                    prev = v["a"][i - 1]
                    rawp = ctx.buf(f"c{i - 1}.raw", prev.shape, dt) if Lp.bn is not None else prev
                    dyp = ctx.buf(f"c{i - 1}.dy", prev.shape, dt)
                    st = K.gather_gemm(mode, [dr], L.w_bwd, cp, [dyp], grid, self.ws_cur,
                                       fuse=self._fuse_spec(Lp, ctx, f"c{i - 1}", 1, rawp, K.ACT_LRELU))
                    hp, wp = prev.shape[1:3]
                    ready = self._finish_fused(Lp, ctx, f"c{i - 1}", rawp, B * hp * wp, garena, dyp, st,
                                               param_grads=param_grads)
                else:
                    wb = L.w_bwd
                    if i == 0 and dx_part is not None:       # rows of w_bwd = input channels: one source's slice
                        r0 = sum(L.in_pad[:dx_part])
                        cp = L.in_pad[dx_part]
                        wb = wb[r0:r0 + cp]
                    use_l1 = i == 0 and dx_l1 is not None and dx_part is not None
                    g = dx_l1[4] if use_l1 else ctx.buf(f"g.c{i}" if dx_part is None or i else f"g.c0.{dx_part}",
                                                        (B, hin, win, cp), dt)
                    if L.stride == 2 and K.tconv_scatter_ok([dr], cp, g):
                        K.tconv_scatter([dr], wb, cp, g, l1=dx_l1[:4] if use_l1 else None)
                    elif use_l1:
                        raise PfaError("discriminator backward: dx_l1 needs the scatter kernel on the first layer (l1_fused())")
                    else:
                        K.gather_gemm(mode, [dr], wb, cp, [g], grid, self.ws_cur)
                    if i == 0:
                        dx = g
            if bucket_done is not None and i == n - 2:
                bucket_done()
        return dx
