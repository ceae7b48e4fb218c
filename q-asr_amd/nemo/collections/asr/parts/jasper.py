"""MaskedConv1d / JasperBlock with the reference's module tree (nemo/collections/asr/parts/jasper.py:
116-213, 293-692) so upstream-NeMo checkpoints load by key: a separable repeat occupies mconv slots
[dw, pw, BN, ReLU, Dropout], a dense one [conv, BN, ReLU, Dropout]; residual branches are
res.{j} = [1x1 conv, BN]; `bn_folding()` moves each BN into the conv in front of it.
Only what QuartzNet15x5 / Jasper10x5dr use is implemented: squeeze-excite, grouped / multi-head convs
and non-batch normalisation are out of scope (the reference itself forbids SE with quantisation, :398-399).
"""
from typing import List, Optional

import torch
import torch.nn as nn

from nemo.quantization.utils.quant_modules import QuantAct, QuantConv1d

jasper_activations = {"hardtanh": nn.Hardtanh, "relu": nn.ReLU, "selu": nn.SELU}


def init_weights(m, mode: Optional[str] = 'xavier_uniform'):
    if isinstance(m, MaskedConv1d):
        init_weights(m.conv, mode)
    if isinstance(m, (nn.Conv1d, nn.Linear, QuantConv1d)) and getattr(m, 'weight', None) is not None:
        fn = {'xavier_uniform': nn.init.xavier_uniform_, 'xavier_normal': nn.init.xavier_normal_,
              'kaiming_uniform': lambda w: nn.init.kaiming_uniform_(w, nonlinearity='relu'),
              'kaiming_normal': lambda w: nn.init.kaiming_normal_(w, nonlinearity='relu')}.get(mode)
        if mode is not None and fn is None:
            raise ValueError(f"Unknown Initialization mode: {mode}")
        if fn is not None:
            fn(m.weight)
    elif isinstance(m, nn.BatchNorm1d):
        if m.track_running_stats:
            m.running_mean.zero_()
            m.running_var.fill_(1)
            m.num_batches_tracked.zero_()
        if m.affine:
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)


def get_same_padding(kernel_size, stride, dilation):
    if stride > 1 and dilation > 1:
        raise ValueError("Only stride OR dilation may be greater than 1")
    return (dilation * kernel_size) // 2 - 1 if dilation > 1 else kernel_size // 2


class MaskedConv1d(nn.Module):
    """time mask -> QuantAct -> QuantConv1d.  Activation bits = quant_bit (+1 when `asymmetric`: inputs are
    post-ReLU, so the extra bit buys an unsigned range - jasper.py:159-163)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, heads=-1,
                 bias=False, use_mask=True, quant_mode='none', quant_bit=8, asymmetric=False):
        super().__init__()
        if heads != -1:
            raise NotImplementedError('multi-head depthwise convs are outside the quantised hot path')
        self.quant_mode = quant_mode
        self.asymmetric = asymmetric
        self.real_out_channels = out_channels
        self.use_mask = use_mask
        self.heads = heads
        conv = nn.Conv1d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, dilation=dilation,
                         groups=groups, bias=bias)
        self.act = QuantAct(quant_bit + (1 if asymmetric else 0), quant_mode=quant_mode, per_channel=False)
        self.conv = QuantConv1d(quant_bit, bias_bit=32, quant_mode=quant_mode, per_channel=True)
        self.conv.set_param(conv)

    def get_seq_len(self, lens):
        c = self.conv
        return (lens + 2 * c.padding[0] - c.dilation[0] * (c.kernel_size[0] - 1) - 1) // c.stride[0] + 1

    def forward(self, x, lens, scaling_factor=None):
        if self.use_mask:
            lens = lens.to(dtype=torch.long)
            keep = torch.arange(x.size(2), device=x.device).unsqueeze(0) < lens.to(x.device).unsqueeze(1)
            x = x * keep.unsqueeze(1).to(x.dtype)
            lens = self.get_seq_len(lens)
        x, x_sf = self.act(x, scaling_factor)
        out, out_sf = self.conv(x, x_sf)
        return out, lens, out_sf

    def bn_folding(self, bn):
        self.conv.bn_folding(bn)

    def set_quant_bit(self, quant_bit, mode='all'):
        if mode in ('all', 'act'):
            self.act.activation_bit = quant_bit + (1 if self.asymmetric else 0)
        if mode in ('all', 'weight'):
            self.conv.weight_bit = quant_bit

    def set_quant_mode(self, quant_mode):
        self.quant_mode = self.conv.quant_mode = self.act.quant_mode = quant_mode


class JasperBlock(nn.Module):
    def __init__(self, inplanes, planes, repeat=3, kernel_size=11, kernel_size_factor=1, stride=1, dilation=1,
                 padding='same', dropout=0.2, activation=None, residual=True, groups=1, separable=False, heads=-1,
                 normalization="batch", norm_groups=1, residual_mode='add', residual_panes=[], conv_mask=False,
                 se=False, se_reduction_ratio=16, se_context_window=None, se_interpolation_mode='nearest',
                 stride_last=False, quant_mode='none', quant_bit=8, layer_num=-1):
        super().__init__()
        if padding != "same":
            raise ValueError("currently only 'same' padding is supported")
        if se or groups != 1 or heads != -1 or normalization != 'batch' or float(kernel_size_factor) != 1.0:
            raise NotImplementedError('SE / grouped / multi-head / non-batch-norm blocks are outside the hot path')
        if residual_mode not in ('add', 'stride_add'):
            raise NotImplementedError(f'residual_mode {residual_mode}')
        if not conv_mask and quant_mode != 'none':
            raise AssertionError('Quantization mode only supports convolution with mask currently.')
        k = kernel_size[0] if isinstance(kernel_size, (list, tuple)) else kernel_size
        s = stride[0] if isinstance(stride, (list, tuple)) else stride
        d = dilation[0] if isinstance(dilation, (list, tuple)) else dilation
        pad = get_same_padding(k, s, d)
        self.conv_mask, self.separable, self.residual_mode = conv_mask, separable, residual_mode
        self.quant_mode, self.layer_num, self.se = quant_mode, layer_num, False
        self.convs_before_bn = []
        act = activation if activation is not None else nn.Hardtanh(min_val=0.0, max_val=20.0)

        def conv_bn(cin, cout, kk, ss, dd, pp, first):
            common = dict(use_mask=conv_mask, quant_mode=quant_mode, quant_bit=quant_bit)
            if separable and kk is not None:
                layers = [MaskedConv1d(cin, cin, kk, stride=ss, dilation=dd, padding=pp, groups=cin,
                                       asymmetric=not first, **common),
                          MaskedConv1d(cin, cout, 1, asymmetric=False, **common)]
            else:
                kk = 1 if kk is None else kk
                layers = [MaskedConv1d(cin, cout, kk, stride=ss, dilation=dd, padding=pp, asymmetric=not first,
                                       **common)]
            layers.append(nn.BatchNorm1d(cout, eps=1e-3, momentum=0.1))
            self.convs_before_bn.append((layers[-2], layers[-1]))
            return layers

        mconv = nn.ModuleList()
        cin = inplanes
        for r in range(repeat):
            last = r == repeat - 1
            ss = s if (last or not stride_last) else 1
            mconv.extend(conv_bn(cin, planes, k, ss, d, pad, first=(layer_num == 0 and r == 0 and (repeat == 1 or not last))))
            if not last:
                mconv.extend([act, nn.Dropout(p=dropout)])
            cin = planes
        self.mconv = mconv

        self.dense_residual = residual
        if residual:
            panes = list(residual_panes)
            if not panes:
                panes = [inplanes]
                self.dense_residual = False
            rs = s if residual_mode == 'stride_add' else 1
            self.res = nn.ModuleList(nn.ModuleList(conv_bn(ip, planes, None, rs, 1, 0, first=(layer_num == 0)))
                                     for ip in panes)
        else:
            self.res = None
        self.res_act = QuantAct(quant_bit, quant_mode=quant_mode, per_channel=False)
        self.mout = nn.Sequential(act, nn.Dropout(p=dropout))

    # ---- structure edits / switches ---------------------------------------------------------------
    def bn_folding(self):
        def fold(seq):
            out = nn.ModuleList()
            for l in seq:
                if isinstance(l, nn.BatchNorm1d):
                    assert isinstance(out[-1], MaskedConv1d)
                    out[-1].bn_folding(l)
                else:
                    out.append(l)
            return out

        self.mconv = fold(self.mconv)
        if self.res is not None:
            self.res = nn.ModuleList(fold(r) for r in self.res)

    def _masked_convs(self):
        for l in self.mconv:
            if isinstance(l, MaskedConv1d):
                yield l
        for r in (self.res or []):
            for l in r:
                if isinstance(l, MaskedConv1d):
                    yield l

    def set_quant_bit(self, quant_bit, mode='all'):
        for l in self._masked_convs():
            l.set_quant_bit(quant_bit, mode)
        self.res_act.activation_bit = quant_bit

    def set_quant_mode(self, quant_mode):
        self.quant_mode = quant_mode
        for l in self._masked_convs():
            l.set_quant_mode(quant_mode)
        self.res_act.quant_mode = quant_mode

    # ---- host forward (calibration / dynamic / un-quantised) ---------------------------------------
    def forward(self, input_):
        xs, lens_orig = input_ if len(input_) == 2 else (input_[0], None)
        out, out_sf = xs[-1]
        lens = lens_orig
        for l in self.mconv:
            if isinstance(l, MaskedConv1d):
                out, lens, out_sf = l(out, lens, out_sf)
            else:
                out = l(out)
        if self.res is not None:
            for i, branch in enumerate(self.res):
                r, r_sf = xs[i]
                for l in branch:
                    if isinstance(l, MaskedConv1d):
                        r, _, r_sf = l(r, lens_orig, r_sf)
                    else:
                        r = l(r)
                out, out_sf = self.res_act(out, out_sf, r, r_sf)
        out = self.mout(out)
        if self.res is not None and self.dense_residual:
            return xs + [(out, out_sf)], lens
        return [(out, out_sf)], lens
