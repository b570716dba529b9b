"""Drop-in nn.Module definitions for the Pix2Pix path of GAN-AUG-PFA.

Same class names, constructor signatures and state_dict keys as the reference's models.py
(UNetGenerator :149-164, UnetSkipConnectionBlock :167-208, NLayerDiscriminator :212-247), so
`generator_epoch_*.pth` / `discriminator_epoch_*.pth` load strictly in either direction.

The sub-modules (nn.Conv2d, nn.ConvTranspose2d, nn.BatchNorm2d ...) sit at the reference's
nn.Sequential indices and own the parameters and buffers; on a GPU tensor `forward` does NOT call
them: it runs the hand-written HIP engine (pfa_gan.engine) through libpfa_hip.so and raises if the
library is missing.  On a CPU tensor the same sub-modules execute as plain PyTorch - that is the
reference's own CPU semantics (BASELINE config 1, "plumbing"), selected only by the caller putting
the model on the CPU, never as a fallback for a GPU input.
"""
import functools
import os

import torch
import torch.nn as nn

from . import layerwise


def _norm_is_instance(norm_layer):
    """models.py:171-174 / :216-219: convs carry a bias only under InstanceNorm2d."""
    base = norm_layer.func if isinstance(norm_layer, functools.partial) else norm_layer
    return base == nn.InstanceNorm2d


def _norm_is_batch(norm_layer):
    base = norm_layer.func if isinstance(norm_layer, functools.partial) else norm_layer
    return base == nn.BatchNorm2d


def default_precision():
    """'fp32' (parity mode, exact-f32 MFMA) unless PFA_PRECISION=bf16."""
    return os.environ.get("PFA_PRECISION", "fp32").lower()


_DTYPES = {"fp32": torch.float32, "f32": torch.float32, "float32": torch.float32,
           "bf16": torch.bfloat16, "bfloat16": torch.bfloat16}


def _contiguous_state_dict(module, state_dict, prefix, local_metadata):
    """Checkpoints keep the reference layout (train_gan.py:149-153): plain NCHW-contiguous float32 tensors with
    their own storage, whatever memory layout the engine keeps the live parameters in."""
    for k, v in list(state_dict.items()):
        if k.startswith(prefix) and isinstance(v, torch.Tensor) and v.dim() > 0:
            if not v.is_contiguous() or v.untyped_storage().nbytes() != v.numel() * v.element_size():
                state_dict[k] = v.detach().clone(memory_format=torch.contiguous_format)
    return state_dict


def resolve_dtype(precision):
    if isinstance(precision, torch.dtype):
        return precision
    try:
        return _DTYPES[str(precision).lower()]
    except KeyError:
        raise ValueError(f"unknown precision {precision!r}; use 'fp32' or 'bf16'") from None


class UnetSkipConnectionBlock(nn.Module):
    """One U-Net level: down-conv, the nested sub-block, up-conv, concatenated with its input."""

    def __init__(self, outer_nc, inner_nc, input_nc=None, submodule=None, outermost=False, innermost=False,
                 norm_layer=nn.BatchNorm2d, use_dropout=False):
        super().__init__()
        self.outermost = outermost
        self.innermost = innermost
        bias = _norm_is_instance(norm_layer)
        if input_nc is None:
            input_nc = outer_nc
        # (layerwise.Conv2d / ConvTranspose2d are nn.Conv2d / nn.ConvTranspose2d whose CUDA forward runs on libpfa_hip)
        down = layerwise.Conv2d(input_nc, inner_nc, kernel_size=4, stride=2, padding=1, bias=bias)
        # the transposed conv sees [skip | up] channels except at the bottom of the U
        up_in = inner_nc if innermost else inner_nc * 2
        up = layerwise.ConvTranspose2d(up_in, outer_nc, kernel_size=4, stride=2, padding=1,
                                       bias=True if outermost else bias)
        seq = []
        if not outermost:
            seq.append(nn.LeakyReLU(0.2, True))
        seq.append(down)
        if not (outermost or innermost):
            seq.append(norm_layer(inner_nc))
        if submodule is not None:
            seq.append(submodule)
        seq += [nn.ReLU(True), up]
        seq.append(nn.Tanh() if outermost else norm_layer(outer_nc))
        if use_dropout and not (outermost or innermost):
            seq.append(nn.Dropout(0.5))
        self.model = nn.Sequential(*seq)
        # typed handles for the HIP engine (plain attributes of already-registered modules: no new keys)
        object.__setattr__(self, "_parts", dict(
            down=down, up=up, sub=submodule,
            down_norm=None if (outermost or innermost) else self.model[2],
            up_norm=None if outermost else self.model[len(seq) - 1 - (1 if (use_dropout and not (outermost or innermost)) else 0)],
            dropout=use_dropout and not (outermost or innermost)))

    def forward(self, x):
        # As a module of its own (and inside a generator with a non-default norm / dropout) a level runs its Sequential:
        # on the GPU the convolutions are libpfa_hip kernels (pfa_gan.layerwise), the rest is the module graph.
        return self.model(x) if self.outermost else torch.cat([x, self.model(x)], 1)


class UNetGenerator(nn.Module):
    """U-Net generator (Pix2Pix).  UNetGenerator(input_nc, output_nc, num_downs=7, ngf=64, ...)."""

    def __init__(self, input_nc, output_nc, num_downs=7, ngf=64, norm_layer=nn.BatchNorm2d, use_dropout=False):
        super().__init__()
        if num_downs < 5:
            raise ValueError("num_downs must be >= 5 (five levels are structural, models.py:155-161)")
        widths = [ngf, ngf * 2, ngf * 4, ngf * 8] + [ngf * 8] * (num_downs - 4)   # encoder output channels per level
        block = None
        for level in range(num_downs - 1, -1, -1):          # build from the bottom of the U outwards
            inner = widths[level]
            outer = output_nc if level == 0 else widths[level - 1]
            block = UnetSkipConnectionBlock(
                outer, inner, input_nc=input_nc if level == 0 else None, submodule=block,
                outermost=level == 0, innermost=level == num_downs - 1, norm_layer=norm_layer,
                use_dropout=use_dropout and 4 <= level < num_downs - 1)
        self.model = block
        self.input_nc, self.output_nc, self.num_downs, self.ngf = input_nc, output_nc, num_downs, ngf
        self._norm_layer = norm_layer
        self._use_dropout = use_dropout
        self.precision = default_precision()
        object.__setattr__(self, "_engines", {})
        self._register_state_dict_hook(_contiguous_state_dict)

    def blocks(self):
        """Levels outermost -> innermost."""
        out, b = [], self.model
        while b is not None:
            out.append(b)
            b = b._parts["sub"]
        return out

    def set_precision(self, precision):
        self.precision = precision
        return self

    def engine(self, precision=None):
        from .engine import GeneratorEngine
        dtype = resolve_dtype(self.precision if precision is None else precision)
        eng = self._engines.get(dtype)
        if eng is None or not eng.still_bound():
            eng = GeneratorEngine(self, dtype)
            self._engines[dtype] = eng
        return eng

    def fused(self):
        """True for the configuration the fused HIP engine is built for: BatchNorm2d, no dropout (train_gan.py:138)."""
        return _norm_is_batch(self._norm_layer) and not self._use_dropout

    def forward(self, input):
        if not input.is_cuda:
            return self.model(input)
        if not self.fused():                      # InstanceNorm / dropout variants: layer by layer, HIP convolutions
            with layerwise.precision(self.precision):
                return self.model(input)
        from .autograd import generator_apply
        return generator_apply(self, input)

    @torch.no_grad()
    def generate_uint8(self, input):
        """Inference fast path of generate_synthetic_data.py:61-71: G(input) de-normalised (x*0.5+0.5) and converted to
        PNG bytes the way TF.to_pil_image does (mul(255).byte()).  NCHW float batch in [-1,1] -> uint8 [B,H,W,C]; on a
        GPU input the conversion runs on the device (libpfa_hip), on a CPU input it is the same arithmetic in torch."""
        if not input.is_cuda or not self.fused():
            return (self(input) * 0.5 + 0.5).mul(255).byte().permute(0, 2, 3, 1).contiguous()
        from . import kernels as K
        eng = self.engine()
        eng.refresh_shadows(force=self.training)
        fake = eng.forward(K.pack_nhwc(input, eng.dtype), eng.scratch_ctx(), train=self.training, bn_repeats=1)
        return K.unpack_u8(fake, eng.out_nc, 0.5, 0.5)


class NLayerDiscriminator(nn.Module):
    """PatchGAN discriminator.  NLayerDiscriminator(input_nc, ndf=64, n_layers=3, norm_layer=nn.BatchNorm2d)."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=nn.BatchNorm2d):
        super().__init__()
        bias = _norm_is_instance(norm_layer)
        mults = [min(2 ** n, 8) for n in range(n_layers + 1)]
        seq = [layerwise.Conv2d(input_nc, ndf, kernel_size=4, stride=2, padding=1), nn.LeakyReLU(0.2, True)]
        for n in range(1, n_layers + 1):
            stride = 2 if n < n_layers else 1          # the last widening conv keeps the resolution (stride 1)
            seq += [layerwise.Conv2d(ndf * mults[n - 1], ndf * mults[n], kernel_size=4, stride=stride, padding=1, bias=bias),
                    norm_layer(ndf * mults[n]), nn.LeakyReLU(0.2, True)]
        seq.append(layerwise.Conv2d(ndf * mults[n_layers], 1, kernel_size=4, stride=1, padding=1))
        self.model = nn.Sequential(*seq)
        self.input_nc, self.ndf, self.n_layers = input_nc, ndf, n_layers
        self._norm_layer = norm_layer
        self.precision = default_precision()
        object.__setattr__(self, "_engines", {})
        self._register_state_dict_hook(_contiguous_state_dict)

    def set_precision(self, precision):
        self.precision = precision
        return self

    def engine(self, precision=None, in_split=None):
        from .engine import DiscriminatorEngine
        dtype = resolve_dtype(self.precision if precision is None else precision)
        key = (dtype, tuple(in_split) if in_split else None)
        eng = self._engines.get(key)
        if eng is None or not eng.still_bound():
            eng = DiscriminatorEngine(self, dtype, in_split)
            self._engines[key] = eng
        return eng

    def fused(self):
        return _norm_is_batch(self._norm_layer)

    def forward(self, input):
        if not input.is_cuda:
            return self.model(input)
        if not self.fused():
            with layerwise.precision(self.precision):
                return self.model(input)
        from .autograd import discriminator_apply
        return discriminator_apply(self, input)
