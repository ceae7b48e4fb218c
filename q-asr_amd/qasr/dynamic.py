"""Dynamic-quantisation device path: encoder + decoder forward with every QuantAct in dynamic mode.

`qm.set_dynamic(model)` makes every QuantAct take its range from the tensor in front of it
(nemo/quantization/utils/quant_modules.py:149-167), so nothing about the activations can be packed ahead of time:
scales, fixedpoint_mul multipliers (batch_frexp, quant_utils.py:121-147) and the bias integers of the next conv
(quant_modules.py:293-299) are recomputed per batch.  The reference does that through host numpy and fp64 tensor
arithmetic with a device->host->device round trip per QuantAct; `DynamicRunner` keeps the derivation on the GPU:

    QuantAct   qasr_dyn_range (min / max of the float32 view; with qm.set_percentile in force qasr_dyn_range_percentile:
               the two torch.quantile values, quant_modules.py:158-167) -> qasr_dyn_act_params (scale, m 2^-e per channel)
               -> qasr_dyn_requant (int8 / uint8 codes, MaskedConv1d's mask applied)
    QuantConv  qasr_dyn_conv_params (output scales, bias integers) -> qasr_dw_conv_acc / qasr_pw_conv_acc (int32)

A float tensor never exists between layers: it is carried as (integers, float32 scales, pending ReLU), the form
`qasr_dyn_view` hands to the kernels.  No call synchronises with the host.  Weights are static and prepared once
(BN fold, per-channel weight integers in the kernels' layouts).

Scope: both model families - separable blocks (QuartzNet) and dense k>1 / strided convs with dense residual panes
(Jasper; accumulators from the generic dense kernel behind qasr_dense_conv_acc).
"""
import ctypes as C

import numpy as np
import torch

from . import quant_math as Q
from .engine import _check, _ptr, _stream_ptr, load_library
from .pack import fragment_order
from .topology import conv_plan


def _rup(x, m):
    return (x + m - 1) // m * m


class DynView(C.Structure):
    _fields_ = [('data', C.c_void_p), ('scale', C.c_void_p), ('is_int8', C.c_int32), ('per_channel', C.c_int32),
                ('residue_lo', C.c_void_p), ('residue_hi', C.c_void_p)]


class _Value:
    """A float32 tensor as integers x scales: `data` int32 / int8 [B, C, Tp], `scale` f32 [C] or [1]."""

    def __init__(self, data, scale, C_, T, per_channel, relu=False, feats=None, residue=None):
        self.data, self.scale, self.C, self.T, self.per_channel, self.relu, self.feats = data, scale, C_, T, per_channel, relu, feats
        self.residue = residue                                # (lo, hi) int32 accumulators: sum(w residue) in units of 2^-24

    def view(self):
        v = DynView()
        v.data, v.scale = self.data.data_ptr(), self.scale.data_ptr()
        v.is_int8, v.per_channel = int(self.data.dtype == torch.int8), int(self.per_channel)
        if self.residue is not None:
            v.residue_lo, v.residue_hi = self.residue[0].data_ptr(), self.residue[1].data_ptr()
        return v


class _Conv:
    """Static half of one QuantConv1d: weight integers in the kernel's layout, s_w, the folded float bias."""

    def __init__(self, site, w, b, wbit, dev):
        wint, s_w = Q.weight_integers(w, wbit)
        wi = wint.to(torch.int64)
        self.site, self.cout = site, w.shape[0]
        self.cout_pad = _rup(self.cout, 128)
        self.s_w = s_w.float().to(dev).contiguous()
        self.bprime = None if b is None else b.float().to(dev).contiguous()
        self.wsum128 = (128 * wi.reshape(self.cout, -1).sum(1)).to(torch.int32).to(dev).contiguous()
        self.dense = False
        if site is not None and site.role == 'dw':
            K = w.shape[2]
            self.kpad = _rup(K, 4)
            wp = torch.zeros(self.cout, self.kpad, dtype=torch.int8)
            wp[:, :K] = wi[:, 0].to(torch.int8)
            self.w = wp.to(dev)
        elif w.shape[2] > 1 or (site is not None and site.stride > 1):
            self.dense = True                                 # [cout_pad][K][cin_pad], the generic dense kernel's layout
            cin, K = w.shape[1], w.shape[2]
            self.cin_pad = _rup(cin, 128)
            wp = torch.zeros(self.cout_pad, K, self.cin_pad, dtype=torch.int8)
            wp[:self.cout, :, :cin] = wi.permute(0, 2, 1).to(torch.int8)
            self.w = wp.to(dev)
        else:
            cin = w.shape[1]
            self.cin_pad = _rup(cin, 128)
            wp = np.zeros((self.cout_pad, self.cin_pad), np.int8)
            wp[:self.cout, :cin] = wi[:, :, 0].to(torch.int8).numpy()
            self.w = torch.from_numpy(fragment_order(wp)).to(dev)


class DynamicRunner:
    """ConvASREncoder.forward + ConvASRDecoder.forward (conv_asr.py:194-206,270-275) in dynamic mode on the HIP kernels."""

    def __init__(self, cfg, state_dict, wbit=8, abit=8, device='cuda:0', percentile=None, division_residue=True):
        self.cfg, self.wbit, self.abit = cfg, wbit, abit
        # the reference's conv_int is a double conv over fl32(x / pre_sf), not over the integers (quant_modules.py:301-305):
        # carry the difference (two extra integer convs per layer) so that every range is taken over the float tensor the
        # reference's QuantAct sees, bit for bit.  False: float view = integer x scale (a range may differ in its last bit)
        self.division_residue = division_residue
        self.percentile = percentile or None                  # `if not self.percentile` (quant_modules.py:150): 0 = min / max
        self.dev = torch.device(device)
        self.lib = load_library()
        self._qws = torch.empty(self.lib.qasr_quantile_workspace_bytes(), dtype=torch.uint8, device=self.dev)
        self._xact = None                                     # scratch: x_act as float32 for the radix select
        self.plan = conv_plan(cfg)
        sd = {k: v.detach().float().cpu() for k, v in state_dict.items() if torch.is_tensor(v)}
        self.convs = {}
        for sites in self.plan:
            for s in sites:
                w = sd[f'{s.key}.conv.weight'] if f'{s.key}.conv.weight' in sd else sd[f'{s.key}.weight']
                b = sd.get(f'{s.key}.conv.bias', sd.get(f'{s.key}.bias'))
                if s.bn_key is not None:
                    w, b = Q.fold_bn(w, b, *(sd[f'{s.bn_key}.{n}'] for n in ('weight', 'bias', 'running_mean', 'running_var')))
                self.convs[s.key] = _Conv(s, w, b, wbit, self.dev)
        self.dec = _Conv(None, sd['decoder.decoder_layers.0.weight'], sd['decoder.decoder_layers.0.bias'], wbit, self.dev)
        self.trace = None                                     # tests: list of dicts per conv (acc, codes, scales)

    def close(self):
        """Drop the device tensors (the model facade calls this when its quantisation state changes, like Engine.close)."""
        self.convs, self.dec, self._xact = {}, None, None

    # ------------------------------------------------------------------ kernels
    def _range(self, a, b, lens, relu, B, T, Tp):
        mm = torch.empty(2, dtype=torch.int32, device=self.dev)
        if a.feats is not None:
            args = (None, None, _ptr(a.feats), a.feats.shape[2], _ptr(lens), 0, B, a.C, T, Tp)
        else:
            va, vb = a.view(), (b.view() if b is not None else None)
            args = (C.byref(va), C.byref(vb) if vb is not None else None, None, 0, _ptr(lens), int(relu), B, a.C, T, Tp)
        if self.percentile is None:
            _check(self.lib.qasr_dyn_range(_stream_ptr(), *args, _ptr(mm)), 'qasr_dyn_range')
            return mm
        n = B * a.C * T
        if self._xact is None or self._xact.numel() < n:
            self._xact = torch.empty(n, dtype=torch.float32, device=self.dev)
        # torch.tensor(1 - p / 100) / torch.tensor(p / 100) (quant_modules.py:161,165): float32 roundings of the doubles
        _check(self.lib.qasr_dyn_range_percentile(_stream_ptr(), *args, C.c_float(1 - self.percentile / 100),
                                                  C.c_float(self.percentile / 100), _ptr(self._xact), _ptr(self._qws),
                                                  self._qws.numel(), _ptr(mm)), 'qasr_dyn_range_percentile')
        return mm

    def _quant_act(self, v, bits, unsigned, lens, B, ident=None, mask=True):
        """QuantAct.forward, dynamic (quant_modules.py:149-194) on value `v` (+ identity `ident`): -> (codes, s [1])."""
        T = v.T
        Tp = _rup(T, 64)
        lens_arg = lens if mask else None
        mm = self._range(v, ident, lens_arg, v.relu, B, T, Tp)
        s = torch.empty(1, dtype=torch.float32, device=self.dev)
        out = torch.empty(B, v.C, Tp, dtype=torch.int8, device=self.dev)
        lo, hi = Q.qrange(bits)
        if unsigned:                                          # post-ReLU tensors: codes in [0, 2^(bits-1) - 1], stored as u8
            lo = 0
        if v.feats is not None:                               # first layer: SymmetricQuantFunction on the float input
            _check(self.lib.qasr_dyn_quant_in(_stream_ptr(), _ptr(v.feats), v.feats.shape[2], _ptr(mm), _ptr(lens), bits, B,
                                              v.C, T, Tp, _ptr(s), _ptr(out)), 'qasr_dyn_quant_in')
            return out, s
        Ma = torch.empty(v.C, dtype=torch.float64, device=self.dev)
        Mb = torch.empty(v.C, dtype=torch.float64, device=self.dev) if ident is not None else None
        _check(self.lib.qasr_dyn_act_params(_stream_ptr(), _ptr(mm), bits, v.C, _ptr(v.scale), int(v.per_channel),
                                            _ptr(ident.scale) if ident is not None else None,
                                            int(ident.per_channel) if ident is not None else 0, _ptr(s), _ptr(Ma), _ptr(Mb)),
               'qasr_dyn_act_params')
        va = v.view()
        vb = ident.view() if ident is not None else None
        _check(self.lib.qasr_dyn_requant(_stream_ptr(), C.byref(va), _ptr(Ma), C.byref(vb) if vb is not None else None,
                                         _ptr(Mb), _ptr(lens_arg), int(v.relu), B, v.C, T, Tp, lo, hi, _ptr(out)),
               'qasr_dyn_requant')
        return out, s

    def _conv(self, cv, codes, s_x, unsigned, B, T, stride=1, dilation=1, padding=0, kernel=1):
        """QuantConv1d.int_conv (quant_modules.py:272-309): -> _Value(acc int32, sf per channel)."""
        sf = torch.empty(cv.cout_pad, dtype=torch.float32, device=self.dev)
        bias = torch.empty(cv.cout_pad, dtype=torch.int32, device=self.dev)
        _check(self.lib.qasr_dyn_conv_params(_stream_ptr(), _ptr(s_x), _ptr(cv.s_w), _ptr(cv.bprime),
                                             _ptr(cv.wsum128) if unsigned else None, cv.cout, cv.cout_pad, _ptr(sf),
                                             _ptr(bias)), 'qasr_dyn_conv_params')
        Tp = codes.shape[2]
        T_out = (T + 2 * padding - dilation * (kernel - 1) - 1) // stride + 1
        Tpo = _rup(T_out, 64)

        def int_conv(x, x_unsigned, b):
            acc = torch.zeros(B, cv.cout, Tpo, dtype=torch.int32, device=self.dev)
            if cv.dense:
                _check(self.lib.qasr_dense_conv_acc(_stream_ptr(), _ptr(x), int(x_unsigned), _ptr(cv.w), _ptr(b), B, x.shape[1],
                                                    cv.cin_pad, cv.cout, kernel, stride, dilation, padding, T, Tp, T_out, Tpo,
                                                    _ptr(acc)), 'qasr_dense_conv_acc')
            elif cv.site is not None and cv.site.role == 'dw':
                _check(self.lib.qasr_dw_conv_acc(_stream_ptr(), _ptr(x), int(x_unsigned), _ptr(cv.w), _ptr(b), B, cv.cout,
                                                 kernel, cv.kpad, stride, dilation, padding, T, Tp, T_out, Tpo, _ptr(acc)),
                       'qasr_dw_conv_acc')
            else:
                _check(self.lib.qasr_pw_conv_acc(_stream_ptr(), _ptr(x), int(x_unsigned), _ptr(cv.w), _ptr(b), B,
                                                 x.shape[1], cv.cin_pad, cv.cout, T, Tp, _ptr(acc)), 'qasr_pw_conv_acc')
            return acc

        acc = int_conv(codes, unsigned, bias)
        residue = None
        if self.division_residue:
            lo, hi = torch.empty_like(codes), torch.empty_like(codes)
            _check(self.lib.qasr_dyn_residue_codes(_stream_ptr(), _ptr(codes), int(unsigned), _ptr(s_x), codes.numel(), _ptr(lo),
                                                   _ptr(hi)), 'qasr_dyn_residue_codes')
            residue = (int_conv(lo, False, None), int_conv(hi, False, None))
        if self.trace is not None:
            self.trace.append(dict(key=cv.site.key if cv.site is not None else 'decoder', acc=acc[:, :, :T_out], codes=codes[:, :, :T],
                                   unsigned=unsigned, s_x=s_x, s_b=sf[:cv.cout]))
        return _Value(acc, sf, cv.cout, T_out, True, residue=residue)

    def _masked_conv(self, s, v, lens, B):
        """MaskedConv1d.forward (jasper.py:175-194): mask -> QuantAct -> QuantConv1d; lens -> get_seq_len."""
        bits = self.abit + (1 if s.asymmetric else 0)
        codes, s_x = self._quant_act(v, bits, s.asymmetric, lens, B)
        out = self._conv(self.convs[s.key], codes, s_x, s.asymmetric, B, v.T, s.stride, s.dilation, s.padding, s.kernel)
        new_lens = torch.div(lens + 2 * s.padding - s.dilation * (s.kernel - 1) - 1, s.stride, rounding_mode='floor') + 1
        return out, new_lens.to(torch.int32)

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, feats, lens):
        """feats f32 [B, feat_in, T] (cuda), lens [B] -> dict(log_probs [B, T', V], tokens, enc_len, logits)."""
        feats = feats.to(self.dev, torch.float32).contiguous()
        lens = lens.to(self.dev, torch.int32).contiguous()
        B, _, T = feats.shape
        xs = [_Value(None, None, feats.shape[1], T, False, feats=feats)]   # block inputs (dense residual keeps them all)
        for bi, sites in enumerate(self.plan):
            lens_in = lens
            v, cl = xs[-1], lens
            for s in (s for s in sites if s.role != 'res'):
                v, cl = self._masked_conv(s, v, cl, B)
                v.relu = s.relu_after
            rs = [s for s in sites if s.role == 'res']
            for s in rs:
                r, _ = self._masked_conv(s, xs[s.pane], lens_in, B)
                # res_act(out, out_sf, res_out, res_sf) (jasper.py:664-682): a dynamic QuantAct on identity + x after
                # EVERY pane, no mask; from the second pane on x is the previous res_act's codes on its per-tensor scale
                codes, S = self._quant_act(v, self.abit, False, cl, B, ident=r, mask=False)
                v = _Value(codes, S, v.C, v.T, False)
            v.relu = True                                     # self.mout (jasper.py:687)
            xs = xs + [v] if (rs and self.cfg.blocks[bi].residual_dense) else [v]
            lens = cl
        cur = xs[-1]
        # decoder (conv_asr.py:270-275): QuantAct (signed) -> 1x1 conv with bias -> log_softmax
        codes, S = self._quant_act(cur, self.abit, False, lens, B, mask=False)
        out = self._conv(self.dec, codes, S, False, B, cur.T)
        z = out.data[:, :, :out.T]
        if out.residue is not None:                           # conv_int.type(torch.float): integers + residue, rounded once
            r = out.residue[0][:, :, :out.T].double() + 128.0 * out.residue[1][:, :, :out.T].double()
            z = z.double() + r * 2.0 ** -24
        logits = z.float() * out.scale[:out.C].view(1, -1, 1)
        logp = torch.log_softmax(logits.transpose(1, 2), dim=-1)
        return dict(log_probs=logp, tokens=logp.argmax(-1), enc_len=lens, logits=logits)
