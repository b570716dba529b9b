"""torch.autograd glue: lets the drop-in modules be used exactly like the reference's
(`loss.backward()`, `.detach()`, `optim.Adam(model.parameters())` - train_gan.py:55-71) while every
forward and backward kernel is the HIP engine's.  One autograd node per network, not per layer.
"""
import torch

from . import kernels as K
from ._lib import PfaError


def _params(module):
    return [p for p in module.parameters()]


class _GeneratorFn(torch.autograd.Function):
    @staticmethod
    def forward(fctx, x, module, needs_grad, *params):
        eng = module.engine()
        # training: any optimizer may have rewritten the masters since the last call (also through `.data`, which no
        # counter sees): always re-emit the kernel-side copies, one pass over the weights.  eval (the inference loop of
        # generate_synthetic_data.py:61-71): only when a version counter moved.
        eng.refresh_shadows(force=module.training)
        ctx = eng.new_ctx() if needs_grad else eng.scratch_ctx()
        xh = K.pack_nhwc(x, eng.dtype)
        fake = eng.forward(xh, ctx, train=module.training, bn_repeats=1, differentiable=needs_grad or module.training)
        fctx.eng, fctx.ctx = eng, ctx
        return K.unpack_nchw(fake, eng.out_nc)

    @staticmethod
    def backward(fctx, gout):
        eng, ctx = fctx.eng, fctx.ctx
        fake = ctx.v["fake"]
        g = K.pack_nhwc(gout, eng.dtype)
        dpre = ctx.buf("dpre", fake.shape, fake.dtype)
        # Tanh backward: dpre = gout * (1 - fake^2)   (the L1 term is switched off: lambda = 0)
        K.l1_tanh_bwd(fake, fake, eng.out_nc, g, 0, 0.0, None, dpre)
        garena = torch.zeros_like(eng.flat.grad)
        eng.backward(ctx, dpre, garena)
        grads = tuple(eng.flat._view(garena, p, *eng.flat.slots[id(p)]) for p in eng.flat.params)
        return (None, None, None) + grads


class _DiscriminatorFn(torch.autograd.Function):
    @staticmethod
    def forward(fctx, x, module, needs_grad, *params):
        eng = module.engine()
        eng.refresh_shadows(force=module.training)
        ctx = eng.new_ctx() if needs_grad else eng.scratch_ctx()
        xh = K.pack_nhwc(x, eng.dtype)
        logits = eng.forward([xh], ctx, train=module.training)
        fctx.eng, fctx.ctx = eng, ctx
        fctx.in_nc = x.shape[1]
        return K.unpack_nchw(logits, 1)

    @staticmethod
    def backward(fctx, gout):
        eng, ctx = fctx.eng, fctx.ctx
        dlogits = K.pack_nhwc(gout, eng.dtype)
        if ctx is eng.scratch_ctx():
            raise PfaError("backward through a forward that ran without gradient tracking")
        need_dx = fctx.needs_input_grad[0]
        garena = torch.zeros_like(eng.flat.grad)
        dxh = eng.backward(ctx, dlogits, garena, param_grads=True, need_dx=need_dx)
        dx = K.unpack_nchw(dxh, fctx.in_nc) if need_dx else None
        grads = tuple(eng.flat._view(garena, p, *eng.flat.slots[id(p)]) for p in eng.flat.params)
        return (dx, None, None) + grads


def _check(module, x):
    p = next(module.parameters())
    if not p.is_cuda or p.device != x.device:
        raise PfaError(f"input on {x.device} but parameters on {p.device}: move the model with .to(device) first")
    if x.dim() != 4:
        raise PfaError("expected an NCHW batch")


def generator_apply(module, x):
    _check(module, x)
    if x.requires_grad:
        raise PfaError("the generator engine does not produce an input gradient (the reference never needs one)")
    eng = module.engine()        # binds the flat arenas before autograd sees the parameters
    # (grad mode is always off INSIDE Function.forward, so whether a backward can follow is decided here:
    #  a pass that may be differentiated gets its own activation context, inference reuses one scratch context)
    needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in eng.flat.params)
    return _GeneratorFn.apply(x, module, needs_grad, *eng.flat.params)


def discriminator_apply(module, x):
    _check(module, x)
    eng = module.engine()
    needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in eng.flat.params))
    return _DiscriminatorFn.apply(x, module, needs_grad, *eng.flat.params)
