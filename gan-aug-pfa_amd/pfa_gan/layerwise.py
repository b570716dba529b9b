"""Layer-wise execution of the 4x4 convolutions on the HIP kernels, for the configurations the fused engine does not cover.

The reference's constructors accept `norm_layer=nn.InstanceNorm2d` (convolutions then carry a bias, models.py:171-174,
216-219) and `use_dropout=True` (nn.Dropout(0.5) in the middle U-Net levels, models.py:197-198); its train_gan.py never
selects either.  The fused engine (pfa_gan.engine) is built around the default BatchNorm2d graph.  For every other
configuration the drop-in modules run their own nn.Sequential graph, in which each Conv2d / ConvTranspose2d is one of the
subclasses below: on a CUDA tensor its forward and backward are the implicit-GEMM and weight-gradient kernels of
libpfa_hip (one autograd node per convolution), and whatever sits between the convolutions - InstanceNorm, Dropout,
activations, torch.cat - executes as the module graph says.  On a CPU tensor they are plain nn.Conv2d / nn.ConvTranspose2d.

This mode packs / unpacks NCHW <-> NHWC and re-emits the weight shadows around every convolution: it is the
compatibility route, not the measured one.
"""
import torch
import torch.nn as nn

from . import kernels as K
from ._lib import PfaError

_WS = {}


def _ws(device):
    ws = _WS.get(device)
    if ws is None:
        ws = _WS[device] = K.Workspace(device)
    return ws


_PRECISION = []          # set by the owning network's forward (UNetGenerator / NLayerDiscriminator .precision)


class precision:
    """with precision("bf16"): ... - the arithmetic type the convolutions inside run in."""

    def __init__(self, p):
        self.p = p

    def __enter__(self):
        _PRECISION.append(self.p)

    def __exit__(self, *exc):
        _PRECISION.pop()


def _dtype_of(module):
    from .models import default_precision, resolve_dtype
    return resolve_dtype(_PRECISION[-1] if _PRECISION else default_precision())


def _master(weight):
    """[rows][Cin-or-Cout][4][4] parameter -> float32 master [rows][16][c] (the layout pfa_weight_shadow reads)."""
    r, c = weight.shape[0], weight.shape[1]
    return weight.detach().float().permute(0, 2, 3, 1).reshape(r, 16, c).contiguous()


class _Conv4x4Fn(torch.autograd.Function):
    """y = conv(x, w) + b for Conv2d(k4, s1|s2, p1) and ConvTranspose2d(k4, s2, p1); NCHW float32 at the boundary."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, transposed, dtype):
        if weight.shape[2:] != (4, 4):
            raise PfaError("layer-wise HIP convolution: only 4x4 kernels are built")
        n, cin, h, w = x.shape
        dev = x.device
        ws = _ws(dev)
        xh = K.pack_nhwc(x, dtype)
        rows, c = weight.shape[0], weight.shape[1]
        cp, rows_p = K.pad_ch(c, dtype), K.pad_ch(rows, dtype)
        same = torch.empty((rows, 16, cp), dtype=dtype, device=dev)
        tr = torch.empty((cp, 16, rows_p), dtype=dtype, device=dev)
        K.weight_shadow(_master(weight), rows, c, cp, 0, 0, rows_p, same, tr)
        if transposed:
            if cin != rows or cin % K.ch_of(dtype):
                raise PfaError(f"layer-wise transposed conv: in_channels {cin} must be a multiple of {K.ch_of(dtype)}")
            cout, ho, wo = c, 2 * h, 2 * w
            mode, w_fwd, grid = K.TCONV_S2, tr, (n, h, w)
        else:
            cout = rows
            ho, wo = (h // 2, w // 2) if stride == 2 else (h - 1, w - 1)
            if stride == 2 and (h % 2 or w % 2):
                raise PfaError(f"layer-wise stride-2 conv needs an even input, got {h}x{w}")
            mode, w_fwd, grid = (K.CONV_S2 if stride == 2 else K.CONV_S1), same, (n, ho, wo)
        out = torch.empty((n, ho, wo, K.pad_ch(cout, dtype)), dtype=dtype, device=dev)
        b32 = bias.detach().float().contiguous() if bias is not None else None
        K.gather_gemm(mode, [xh], w_fwd, cout, [out], grid, ws, bias=b32)
        ctx.save_for_backward(xh)
        ctx.meta = (stride, transposed, dtype, same, tr, tuple(weight.shape), tuple(x.shape), bias is not None)
        return K.unpack_nchw(out, cout)

    @staticmethod
    def backward(ctx, gy):
        (xh,) = ctx.saved_tensors
        stride, transposed, dtype, same, tr, wshape, xshape, has_bias = ctx.meta
        n, cin, h, w = xshape
        dev = gy.device
        ws = _ws(dev)
        gyh = K.pack_nhwc(gy.contiguous(), dtype)
        ho, wo = gyh.shape[1:3]
        rows, c = wshape[0], wshape[1]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gxh = torch.empty_like(xh)
            if transposed:          # dgrad of the transposed conv = stride-2 correlation of gy with the master layout
                K.gather_gemm(K.CONV_S2, [gyh], same, cin, [gxh], (n, h, w), ws)
            elif stride == 2:
                K.gather_gemm(K.TCONV_S2, [gyh], tr, cin, [gxh], (n, ho, wo), ws)
            else:
                K.gather_gemm(K.CONV_S1_DGRAD, [gyh], tr, cin, [gxh], (n, h, w), ws)
            gx = K.unpack_nchw(gxh, cin)
        if ctx.needs_input_grad[1]:
            dw = torch.zeros((rows, 16, c), dtype=torch.float32, device=dev)
            if transposed:          # master rows = input channels: the input is the direct operand, gy the gathered one
                K.wgrad([xh], [gyh], dw, 2, [rows], [c], ws=ws, fresh=True)
            else:
                K.wgrad([gyh], [xh], dw, stride, [rows], [c], ws=ws, fresh=True)
            gw = dw.reshape(rows, 4, 4, c).permute(0, 3, 1, 2).contiguous()
        if has_bias and ctx.needs_input_grad[2]:
            cout = c if transposed else rows
            gb = torch.zeros(cout, dtype=torch.float32, device=dev)
            K.colsum(gyh, gb, ws, c_out=cout)
        return gx, gw, gb, None, None, None


class Conv2d(nn.Conv2d):
    """nn.Conv2d whose CUDA forward/backward run on libpfa_hip (4x4, padding 1, stride 1 or 2)."""

    def forward(self, x):
        if not x.is_cuda:
            return super().forward(x)
        if self.kernel_size != (4, 4) or self.padding != (1, 1) or self.stride not in ((1, 1), (2, 2)) or self.groups != 1:
            raise PfaError("layer-wise HIP convolution: only 4x4 / padding 1 / stride 1|2 / groups 1 is built")
        return _Conv4x4Fn.apply(x, self.weight, self.bias, self.stride[0], False, _dtype_of(self))


class ConvTranspose2d(nn.ConvTranspose2d):
    """nn.ConvTranspose2d whose CUDA forward/backward run on libpfa_hip (4x4, stride 2, padding 1)."""

    def forward(self, x, output_size=None):
        if not x.is_cuda:
            return super().forward(x, output_size)
        if self.kernel_size != (4, 4) or self.padding != (1, 1) or self.stride != (2, 2) or self.output_padding != (0, 0):
            raise PfaError("layer-wise HIP transposed convolution: only 4x4 / stride 2 / padding 1 is built")
        return _Conv4x4Fn.apply(x, self.weight, self.bias, 2, True, _dtype_of(self))
