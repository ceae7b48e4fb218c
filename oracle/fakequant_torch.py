"""CPU baseline port (TEST / BENCH INFRASTRUCTURE — never imported by the product path).

A PyTorch-CPU restatement of the op sequence the reference executes for one evaluate-mode
forward of encoder + decoder — the "reference CPU path" of BASELINE.md §3: fake-quant in
float32/float64 with per-forward BN folding and weight re-quantisation
(nemo/quantization/utils/quant_modules.py:272-367), fp64 `F.conv1d` on integer-valued tensors
(:301-305), `fixedpoint_mul` with the host-side `batch_frexp` Decimal loop on every QuantAct
(quant_utils.py:121-216), masking per conv (nemo/collections/asr/parts/jasper.py:175-183) and the
block structure of JasperBlock.forward (:641-692).  It is what bench.py times as
`cpu_baseline` (kind "port") on the GPU box's host cores, and it is pinned to the reference by
tests/test_oracle_golden.py (tokens / accumulators of the golden fixtures).
"""
import decimal
from decimal import Decimal

import numpy as np
import torch
import torch.nn.functional as F


def _scale(bits, lo, hi):
    n = 2 ** (bits - 1) - 1
    return torch.clamp(torch.maximum(lo.abs(), hi.abs()), min=1e-8) / n


def _quant(x, bits, scale):
    n = 2 ** (bits - 1) - 1
    return torch.clamp(torch.round(1. / scale * x + 0.), -n, n - 1)


def _batch_frexp(r, max_bit=31):
    shape = r.size()
    mant, ex = np.frexp(r.view(-1).cpu().numpy())
    m = np.array([int(Decimal(v * (2 ** max_bit)).quantize(Decimal('1'), rounding=decimal.ROUND_HALF_UP))
                  for v in mant])
    return torch.from_numpy(m).view(shape), torch.from_numpy(float(max_bit) - ex).view(shape)


def _fixedpoint(x, pre_sf, bits, z_sf, identity=None, id_sf=None):
    n = 2 ** (bits - 1) - 1

    def one(v, sf):
        sf = sf if sf.dim() == 3 else sf.view(1, 1, -1)
        z = torch.round(v / sf)
        new = sf.double() / z_sf.float().double()
        m, e = _batch_frexp(new)
        return torch.round(z.double() * m.double() / (2.0 ** e))

    out = one(x, pre_sf)
    if identity is not None:
        out = one(identity, id_sf) + out
    return torch.clamp(out.float(), -n - 1, n)


def _sqrt_ieee(v):
    # see oracle/int_oracle.host_sqrt_f32: torch's CPU sqrt is host dependent; fixtures pin the IEEE value
    return torch.from_numpy(np.sqrt(v.numpy().astype(np.float64)).astype(np.float32))


class FakeQuantNet:
    def __init__(self, plan, cfg, state_dict, act_min, act_max, wbit=8, abit=8):
        self.plan, self.cfg = plan, cfg
        self.sd = {k: torch.from_numpy(np.asarray(v)) for k, v in state_dict.items()}
        self.amin = torch.from_numpy(np.asarray(act_min, np.float32))
        self.amax = torch.from_numpy(np.asarray(act_max, np.float32))
        self.wbit, self.abit = wbit, abit
        self.acc = []

    def _act(self, ai, bits, x, pre_sf, identity=None, id_sf=None):
        sf = _scale(bits, self.amin[ai].view(1, 1, 1), self.amax[ai].view(1, 1, 1))
        if pre_sf is None:
            x = _quant(x, bits, sf) * sf
            pre_sf = sf
        q = _fixedpoint(x, pre_sf, bits, sf, identity, id_sf)
        return q * sf, sf

    def _conv(self, key, bn_key, x, pre_sf, stride, padding, dilation, groups):
        w = self.sd[f'{key}.conv.weight'] if f'{key}.conv.weight' in self.sd else self.sd[f'{key}.weight']
        b = self.sd.get(f'{key}.conv.bias', self.sd.get(f'{key}.bias'))
        if bn_key is not None:                                   # re-folded every forward, like the reference
            g = self.sd[f'{bn_key}.weight'] / _sqrt_ieee(self.sd[f'{bn_key}.running_var'] + 1e-3)
            w = w * g.reshape(-1, 1, 1)
            b0 = torch.zeros_like(g) if b is None else b
            b = (b0 - self.sd[f'{bn_key}.running_mean']) * g + self.sd[f'{bn_key}.bias']
        wmin = w.min(dim=-1).values.min(dim=-1).values.view(-1, 1, 1)
        wmax = w.max(dim=-1).values.max(dim=-1).values.view(-1, 1, 1)
        s_w = _scale(self.wbit, wmin, wmax)
        wint = _quant(w, self.wbit, s_w)
        s_b = s_w * pre_sf
        bint = None if b is None else _quant(b, 32, s_b.reshape(-1)).double()
        x_int = (x / pre_sf).double()
        conv = F.conv1d(x_int, wint.double(), bint, stride=stride, padding=padding, dilation=dilation, groups=groups)
        self.acc.append(conv)
        sf = s_b.view(1, -1, 1)
        return conv.float() * sf, sf

    def _masked(self, s, x, lens, pre_sf, ai):
        mask = torch.arange(x.size(2)).expand(len(lens), x.size(2)) >= lens.unsqueeze(1)
        x = x.masked_fill(mask.unsqueeze(1), 0)
        bits = self.abit + (1 if s.asymmetric else 0)
        x, sf = self._act(ai, bits, x, pre_sf)
        y, ysf = self._conv(s.key, s.bn_key, x, sf, s.stride, s.padding, s.dilation, s.groups)
        new_lens = (lens + 2 * s.padding - s.dilation * (s.kernel - 1) - 1) // s.stride + 1
        return y, ysf, new_lens

    @torch.no_grad()
    def forward(self, feats, lens):
        self.acc = []
        xs = [(torch.as_tensor(feats, dtype=torch.float32), None)]
        lens = torch.as_tensor(lens, dtype=torch.long)
        ai = 0
        for bi, sites in enumerate(self.plan):
            blk = self.cfg.blocks[bi]
            lens_orig = lens
            y, sf = xs[-1]
            cur = lens
            rsites = [s for s in sites if s.role == 'res']
            for s in sites:
                if s.role == 'res':
                    continue
                y, sf, cur = self._masked(s, y, cur, sf, ai)
                ai += 1
                if s.relu_after:
                    y = torch.relu(y)
            for s in rsites:
                r, rsf = xs[s.pane]
                r, rsf, _ = self._masked(s, r, lens_orig, rsf, ai)
                ai += 1
                y, sf = self._act(ai + (len(rsites) - 1 - rsites.index(s)), self.abit, y, sf, r, rsf)
            ai += 1
            y = torch.relu(y)
            lens = cur
            xs = xs + [(y, sf)] if (rsites and blk.residual_dense) else [(y, sf)]
        y, sf = xs[-1]
        y, sf = self._act(ai, self.abit, y, sf)
        enc_codes = torch.round(y / sf).to(torch.int32)          # the decoder QuantAct's integers (conv_asr.py:270)
        logits, _ = self._conv('decoder.decoder_layers.0', None, y, sf, 1, 0, 1, 1)
        logp = torch.log_softmax(logits.transpose(1, 2), dim=-1)
        return dict(log_probs=logp, tokens=logp.argmax(-1), enc_len=lens, logits=logits, enc_codes=enc_codes)
