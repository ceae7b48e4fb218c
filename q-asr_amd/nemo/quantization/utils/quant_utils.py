"""API-compatible surface of the reference's nemo/quantization/utils/quant_utils.py, built
on qasr.quant_math (host PyTorch).  Same names, argument meaning and results:
  linear_quantize                      quant_utils.py:12-26
  symmetric_linear_quantization_params quant_utils.py:28-54
  SymmetricQuantFunction               quant_utils.py:57-95
  batch_frexp                          quant_utils.py:121-147
  fixedpoint_mul                       quant_utils.py:149-223
Unlike the reference nothing here forces `.cuda()`: tensors stay on the device they come from.
"""
import torch
from torch.autograd import Function

from qasr import quant_math as Q


def linear_quantize(input, scale, zero_point, inplace=False):
    if inplace:
        return input.mul_(1.0 / scale).add_(zero_point).round_()
    return torch.round(1.0 / scale * input + zero_point)


def symmetric_linear_quantization_params(num_bits, saturation_min, saturation_max, per_channel=False):
    with torch.no_grad():
        return Q.sym_scale(num_bits, saturation_min, saturation_max)


class SymmetricQuantFunction(Function):
    """Integers in [-n, n-1], n = 2^(k-1)-1; straight-through gradient scaled by 1/scale."""

    @staticmethod
    def forward(ctx, x, k, specified_scale=None):
        if specified_scale is None:
            raise ValueError('SymmetricQuantFunction needs the pre-computed scale')
        ctx.scale = specified_scale
        return Q.quantize(x, k, specified_scale)

    @staticmethod
    def backward(ctx, grad_output):
        s = ctx.scale
        shape = {4: (-1, 1, 1, 1), 2: (-1, 1)}.get(grad_output.dim(), (-1,))
        return grad_output.clone() / s.view(*shape), None, None


class round_ste(Function):
    @staticmethod
    def forward(ctx, x):
        return torch.round(x)

    @staticmethod
    def backward(ctx, g):
        return g.clone()


class floor_ste(Function):
    @staticmethod
    def forward(ctx, x):
        return torch.floor(x)

    @staticmethod
    def backward(ctx, g):
        return g.clone()


def batch_frexp(inputs, max_bit=31):
    """(mantissa as integer m, exponent e) with inputs = m * 2^-e, m rounded half-up to max_bit bits.
    Computed with torch.frexp on the tensor's own device (the reference round-trips through the
    host and a Python Decimal loop on every forward)."""
    mant, ex = torch.frexp(inputs.double())
    m = torch.floor(mant * float(2 ** max_bit) + 0.5)
    return m, float(max_bit) - ex.double()


class fixedpoint_mul(Function):
    """Integer requantisation out = clamp(round(z*m/2^e) [+ same for identity], -n-1, n)."""

    @staticmethod
    def forward(ctx, pre_act, pre_act_scaling_factor, bit_num, quant_mode, z_scaling_factor, identity=None,
                identity_scaling_factor=None):
        ctx.identity = identity
        ctx.z_scaling_factor = z_scaling_factor
        if quant_mode != 'symmetric':
            raise NotImplementedError('only symmetric mode is on the hot path')
        lo, hi = Q.qrange(bit_num)
        with torch.no_grad():
            def one(x, sf):
                sf = sf if sf.dim() == 3 else sf.view(1, -1, 1)
                z = torch.round(x / sf)
                return torch.round(z.double() * Q.requant_multiplier(sf, z_scaling_factor))

            out = one(pre_act, pre_act_scaling_factor)
            if identity is not None:
                out = one(identity, identity_scaling_factor) + out
            return torch.clamp(out.float(), lo, hi)

    @staticmethod
    def backward(ctx, grad_output):
        g = grad_output.clone() / ctx.z_scaling_factor
        return g, None, None, None, None, (g.clone() if ctx.identity is not None else None), None
