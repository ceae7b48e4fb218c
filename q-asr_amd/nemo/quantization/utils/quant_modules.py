"""QuantAct / QuantConv1d with the reference's constructor arguments, buffers, state-dict keys and
forward contract (nemo/quantization/utils/quant_modules.py:18-367), written from scratch.

These modules are the HOST side of the drop-in: they carry calibration (running ranges, percentile,
EMA), dynamic mode and the un-quantised path in PyTorch - what the north-star keeps in host PyTorch.
Once a model is calibrated and frozen (`qm.evaluate`) EncDecCTCModel packs it and the forward runs in
the HIP engine instead (nemo/collections/asr/models/ctc_models.py in this repo).

Differences from the reference that do not change integer results: activations are rounded to exact
integers before the conv (the reference feeds the un-rounded x/scale, which is off-integer by <=1.5e-5;
`rint(reference conv_int)` equals our accumulator, SURVEY hard-part 1), and no `.cuda()` is forced.
"""
import torch
import torch.nn.functional as F
from torch.nn import Module, Parameter

from qasr import quant_math as Q
from .quant_utils import *  # noqa: F401,F403  (re-exported like the reference does)
from .quant_utils import fixedpoint_mul


class QuantAct(Module):
    def __init__(self, activation_bit, act_range_momentum=0.95, running_stat=True, per_channel=False,
                 channel_len=None, quant_mode="none", dynamic=False, percentile=None):
        super().__init__()
        if quant_mode not in ("none", "symmetric"):
            if quant_mode == "asymmetric":
                raise NotImplementedError(f"unsupported quant mode: {quant_mode}")
            raise ValueError(f"unknown quant mode: {quant_mode}")
        self.activation_bit = activation_bit
        self.act_range_momentum = act_range_momentum
        self.running_stat = running_stat
        self.quant_mode = quant_mode
        self.per_channel = per_channel
        self.dynamic = dynamic
        self.percentile = percentile
        n = 1 if not per_channel else channel_len
        if n is None:
            raise AssertionError('channel_len is required with per_channel=True')
        for name in ('x_min', 'x_max', 'act_scaling_factor'):
            self.register_buffer(name, torch.zeros(n))

    def __repr__(self):
        return (f"{self.__class__.__name__}(activation_bit={self.activation_bit}, quant_mode: {self.quant_mode}, "
                f"Act_min: {self.x_min.min().item():.2f}, Act_max: {self.x_max.max().item():.2f})")

    def fix(self):
        self.running_stat = False

    def unfix(self):
        self.running_stat = True

    def set_percentile(self, percentile):
        assert not self.per_channel, 'percentile mode is only available for the global quantization mode'
        self.percentile = percentile

    # -- range tracking (quant_modules.py:112-141) ------------------------------------------------
    def _range_of(self, x, use_percentile):
        x = x.detach()
        if not use_percentile:
            if self.per_channel:
                return x.amin(dim=(0, 2)), x.amax(dim=(0, 2))
            return x.min(), x.max()
        assert not self.per_channel, 'percentile mode is only available for the global quantization mode'
        if x.is_cuda and x.numel() > 1:
            # both quantiles in one radix select on device (csrc/qasr_calib.hip), bit-identical to torch.quantile on CPU
            from qasr import engine
            r = engine.quantile2(x.float(), 1 - self.percentile / 100, self.percentile / 100)
            return r[0], r[1]
        return (_quantile(x, torch.tensor(1 - self.percentile / 100, device=x.device)),
                _quantile(x, torch.tensor(self.percentile / 100, device=x.device)))

    def _observe(self, lo, hi):
        if torch.eq(self.x_min, self.x_max).all():          # first batch initialises the range
            self.x_min = self.x_min + lo
            self.x_max = self.x_max + hi
        elif self.act_range_momentum == -1:
            self.x_min = torch.min(self.x_min, lo)
            self.x_max = torch.max(self.x_max, hi)
        else:
            mom = self.act_range_momentum
            self.x_min = self.x_min * mom + lo * (1 - mom)
            self.x_max = self.x_max * mom + hi * (1 - mom)

    def forward(self, x, pre_act_scaling_factor=None, identity=None, identity_scaling_factor=None):
        x_act = x if identity is None else identity + x
        batch = None
        if self.running_stat:
            batch = self._range_of(x_act, self.percentile is not None)
            self._observe(*batch)
        if self.quant_mode == 'none':
            return x_act, None
        if self.dynamic:                                     # per-batch range (quant_modules.py:149-167)
            reuse = batch is not None and bool(self.percentile)
            lo, hi = batch if reuse else self._range_of(x_act, bool(self.percentile))
        else:
            lo, hi = self.x_min, self.x_max
        sf = Q.sym_scale(self.activation_bit, lo.reshape(1, -1, 1), hi.reshape(1, -1, 1))
        self.act_scaling_factor = sf.reshape(-1)
        if pre_act_scaling_factor is None:                   # first layer: quantise the float input
            x = Q.quantize(x, self.activation_bit, sf) * sf
            pre_act_scaling_factor = sf
        q = fixedpoint_mul.apply(x, pre_act_scaling_factor, self.activation_bit, self.quant_mode, sf, identity,
                                 identity_scaling_factor)
        return q * sf, sf


def _quantile(x, q):
    """torch.quantile over the whole tensor; falls back to kthvalue interpolation above its size limit."""
    flat = x.reshape(-1).float()
    if flat.numel() <= 16_000_000:
        return torch.quantile(flat, q.to(flat.dtype))
    pos = q.double() * (flat.numel() - 1)
    lo = int(torch.floor(pos))
    a = torch.kthvalue(flat, lo + 1).values
    b = torch.kthvalue(flat, min(lo + 2, flat.numel())).values
    return a + (b - a) * (pos - lo).to(flat.dtype)


class QuantConv1d(Module):
    def __init__(self, weight_bit, bias_bit=None, quant_mode='none', per_channel=False, fix_bn=True):
        super().__init__()
        self.weight_bit = weight_bit
        self.bias_bit = bias_bit
        self.quantize_bias = bias_bit is not None
        self.quant_mode = quant_mode
        self.per_channel = per_channel
        self.fix_bn = fix_bn

    def set_param(self, conv):
        for a in ('in_channels', 'out_channels', 'kernel_size', 'stride', 'padding', 'dilation', 'groups'):
            setattr(self, a, getattr(conv, a))
        self.weight = Parameter(conv.weight.data.clone())
        if conv.bias is not None:
            self.bias = Parameter(conv.bias.data.clone())
            self.register_buffer('bias_integer', torch.zeros_like(self.bias))
        else:
            self.bias = None
            self.bias_integer = None
        self.register_buffer('weight_integer', torch.zeros_like(self.weight))
        self.register_buffer('conv_scaling_factor', torch.zeros(self.out_channels))
        self.conv = conv                                     # kept so fork-saved state dicts still load
        self.bn = None

    def __repr__(self):
        return (f"{self.__class__.__name__}(weight_bit={self.weight_bit}, per_channel: {self.per_channel}, "
                f"quant_mode: {self.quant_mode}")

    def fix(self):
        self.fix_bn = True

    def unfix(self):
        self.fix_bn = False

    def bn_folding(self, bn):
        self.bn = bn

    def _conv(self, x, w, b):
        return F.conv1d(x, weight=w, bias=b, stride=self.stride, padding=self.padding, dilation=self.dilation,
                        groups=self.groups)

    def folded(self):
        """(weight, bias) with BN folded in when it is attached and frozen (quant_modules.py:352-364)."""
        w = self.weight.data.detach()
        b = None if self.bias is None else self.bias.data.detach()
        if self.bn is not None and self.fix_bn:
            w, b = Q.fold_bn(w, b, self.bn.weight.detach(), self.bn.bias.detach(), self.bn.running_mean.detach(),
                             self.bn.running_var.detach(), self.bn.eps)
        return w, b

    def int_conv(self, weight, bias, x, pre_act_scaling_factor):
        """Integer conv on scale conv_sf*pre_sf (quant_modules.py:272-309) -> (float view, scale[1,C,1])."""
        if self.per_channel:
            wint, s_w = Q.weight_integers(weight, self.weight_bit)
        else:
            s_w = Q.sym_scale(self.weight_bit, weight.min(), weight.max()).reshape(1)
            wint = Q.quantize(weight, self.weight_bit, s_w)
        self.conv_scaling_factor = s_w.reshape(-1)
        self.weight_integer = wint
        bint, s_b = Q.bias_integers(bias, s_w, pre_act_scaling_factor, self.bias_bit or 32)
        if bint is not None:
            self.bias_integer = bint
        # no rounding here, as in the reference (quant_modules.py:301): the float32 quotient of x = fl32(q s) by s is q or a
        # float32 neighbour of q, and that residue is part of the float tensor the next QuantAct calibrates / ranges on
        x_int = (x / pre_act_scaling_factor).double()
        acc = self._conv(x_int, wint.double(), None if bint is None else bint.double()).float()
        sf = s_b.view(1, -1, 1)
        return acc * sf, sf

    def forward(self, x, pre_act_scaling_factor=None):
        if self.quant_mode == 'none':
            y = self._conv(x, self.weight, self.bias)
            return (self.bn(y) if self.bn is not None else y), None
        assert self.quant_mode == 'symmetric'
        w, b = self.folded()
        y, sf = self.int_conv(w, b, x, pre_act_scaling_factor)
        if self.bn is None or self.fix_bn:
            return y, sf
        # BN attached but not frozen (quantisation-aware training): update its statistics, apply it on top
        bn = self.bn
        bn.running_mean = bn.running_mean.detach() * (1 - bn.momentum) + bn.momentum * y.mean(dim=(0, 2))
        bn.running_var = bn.running_var.detach() * (1 - bn.momentum) + bn.momentum * y.var(dim=(0, 2))
        g = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).view(1, -1, 1)
        return g * (y - bn.running_mean.view(1, -1, 1)) + bn.bias.view(1, -1, 1), g * sf


class QuantLinear(Module):
    """Present for API parity only: the reference class is never instantiated on the ASR path and is
    itself broken (4 positional args into a 3-arg function, quant_modules.py:462-463)."""

    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError('QuantLinear is not on the quantised ASR path')
