"""Pack step: calibrated float model -> immutable blob for the HIP engine (include/qasr.h).

Runs once after calibration on the host (PyTorch CPU), hoisting what the reference
recomputes on every forward: BN fold, weight / bias integers, and the
`batch_frexp` requant multipliers (quant_utils.py:121-147) of every QuantAct.  Because
activation ranges are static in `evaluate` mode, each QuantAct's requantisation is
attached to the op that PRODUCES its input (an `out` of that op), so no stand-alone
requant pass runs on device; values with more than QASR_MAX_OUTS consumers (Jasper's
dense residual) are stored raw once and requantised by small REQUANT ops.

Graph walk mirrors ConvASREncoder.forward / JasperBlock.forward
(conv_asr.py:194-206, jasper.py:641-692); the order of `act_min/act_max` is: per block
[mconv sites, residual sites, res_act], then the decoder's QuantAct.
"""
import struct
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch

from . import quant_math as Q
from .topology import ModelCfg, conv_plan

MAGIC, VERSION = 0x52534151, 6
OP_QUANT_IN, OP_DW, OP_PW, OP_DENSE, OP_LOGSOFTMAX, OP_REQUANT = range(6)
F_RELU, F_MASK_OUT, F_EXACT_Z, F_LOGITS, F_RESADD, F_TAPMAJOR, F_WIDE_RQ, F_W6PACK = 1, 2, 4, 8, 16, 32, 64, 128
DT_S8, DT_U8, DT_F32, DT_I32 = range(4)
MAX_PANES, MAX_OUTS = 12, 3
COUT_ALIGN, CIN_ALIGN = 128, 128
Z_EXACT_LIMIT = (1 << 22) - 1       # below this z == acc is a theorem
RQ_NARROW_LIMIT = float(1 << 30)    # |acc * M| below this: the low word of fma(acc, M, 1.5*2^52) is the rounded product


def fragment_order(w: np.ndarray) -> np.ndarray:
    """1x1 conv weights [cout_pad(128k)][cin_pad(64k)] -> MFMA B-fragment order (include/qasr.h):
    byte ((tile*NKS + ks)*64 + lane)*16 + j  =  W[32*tile + (lane & 31)][32*ks + 16*(lane >> 5) + j],
    so one wave instruction (64 lanes x 16 B) reads one contiguous 1 KiB block."""
    cp, cinp = w.shape
    assert cp % 32 == 0 and cinp % 32 == 0
    v = w.reshape(cp // 32, 32, cinp // 32, 2, 16)          # [tile, r, ks, h, j]
    return np.ascontiguousarray(v.transpose(0, 2, 3, 1, 4)).reshape(-1)   # [tile, ks, h, r, j]


def pack6(a: np.ndarray) -> np.ndarray:
    """Sub-byte weight storage (BASELINE config 3): int8 codes in [-32, 31] (quant_utils.py:57-79 at 6 bits gives
    [-31, 30]) -> 4 codes per 3 bytes, two's-complement 6-bit fields, little end first:
    b0 = c0 | c1 << 6,  b1 = c1 >> 2 | c2 << 4,  b2 = c2 >> 4 | c3 << 2.  The element order (e.g. MFMA fragment order) is
    kept; the engine expands the arrays back to int8 on the device when it loads the blob (qasr_engine_create)."""
    c = np.ascontiguousarray(a).reshape(-1)
    assert c.dtype == np.int8 and c.size % 4 == 0 and c.min() >= -32 and c.max() <= 31
    u = (c.astype(np.int16) & 0x3f).astype(np.uint32).reshape(-1, 4)
    word = u[:, 0] | (u[:, 1] << 6) | (u[:, 2] << 12) | (u[:, 3] << 18)
    out = np.empty((word.size, 3), np.uint8)
    out[:, 0], out[:, 1], out[:, 2] = word & 0xff, (word >> 8) & 0xff, (word >> 16) & 0xff
    return out.reshape(-1)


def unpack6(b: np.ndarray) -> np.ndarray:
    """Inverse of pack6 (host restatement of the device expansion, for the tests)."""
    t = np.ascontiguousarray(b).reshape(-1, 3).astype(np.uint32)
    word = t[:, 0] | (t[:, 1] << 8) | (t[:, 2] << 16)
    c = np.stack([(word >> (6 * i)) & 0x3f for i in range(4)], axis=1).astype(np.int16)
    return ((c ^ 0x20) - 0x20).astype(np.int8).reshape(-1)


def _t(a):
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.asarray(a))


def _rup(x, m):
    return (x + m - 1) // m * m


@dataclass
class _Consumer:
    s_x: torch.Tensor           # scalar f32 scale of the consuming QuantAct
    lo: int
    hi: int
    tensor: int = -1            # filled in pass 2


@dataclass
class _Value:
    """A float-view quantity of the reference: conv accumulator (per-channel scale) or
    res_act result (scalar scale)."""
    kind: str                   # 'first' | 'acc' | 'q'
    op: int                     # producing op (index into Packer.ops)
    channels: int
    domain: int
    scale: Optional[torch.Tensor]   # [C] for 'acc', [1] for 'q'
    consumers: List[_Consumer] = field(default_factory=list)
    tensor: int = -1            # 'first': the s8 tensor itself


class Packer:
    def __init__(self, cfg: ModelCfg, state_dict, act_min, act_max, wbit=8, abit=8):
        self.cfg = cfg
        self.sd = {k: _t(v) for k, v in state_dict.items()}
        self.amin = _t(act_min).float().reshape(-1)
        self.amax = _t(act_max).float().reshape(-1)
        self.wbit, self.abit = wbit, abit
        self.ops, self.values, self.tensors, self.domains = [], [], [], []
        self.data = bytearray(16)       # offset 0 means "absent" in the op tables
        self.sites = []              # (op index, pane index or -1) per conv, in reference call order

    # ------------------------------------------------------------------ data section
    def _put(self, arr: np.ndarray) -> int:
        pad = (-len(self.data)) % 16
        self.data += b'\0' * pad
        off = len(self.data)
        self.data += np.ascontiguousarray(arr).tobytes()
        return off                   # relative to data section; rebased in serialise()

    def _put_w(self, arr: np.ndarray) -> int:
        """int8 weight array: stored sub-byte (pack6) when the model's weights have <= 6 bits."""
        return self._put(pack6(arr) if self.wbit <= 6 else arr)

    def _tensor(self, channels, dtype, domain, producer):
        self.tensors.append(dict(channels=channels, dtype=dtype, domain=domain, producer=producer, last_use=-1))
        return len(self.tensors) - 1

    # ------------------------------------------------------------------ per-conv host math
    def _conv_ints(self, key, bn_key, s_x, in_unsigned, in_absmax):
        w = self.sd[f'{key}.conv.weight'].float() if f'{key}.conv.weight' in self.sd else self.sd[f'{key}.weight'].float()
        b = self.sd.get(f'{key}.conv.bias', self.sd.get(f'{key}.bias'))
        b = None if b is None else b.float()
        if bn_key is not None:
            w, b = Q.fold_bn(w, b, *(self.sd[f'{bn_key}.{n}'].float() for n in
                                      ('weight', 'bias', 'running_mean', 'running_var')))
        wint, s_w = Q.weight_integers(w, self.wbit)
        bint, s_b = Q.bias_integers(b, s_w, s_x)
        wi = wint.to(torch.int64)
        bi = torch.zeros(w.shape[0], dtype=torch.int64) if bint is None else bint.to(torch.int64)
        absw = wi.abs().reshape(w.shape[0], -1).sum(1)
        bound = int((absw * in_absmax + bi.abs()).max())
        assert bound < 2 ** 31 - 2 ** 24, f'{key}: int32 accumulator could overflow ({bound})'
        if in_unsigned:               # u8 input is fed as (x - 128): fold 128*sum(W) into the bias
            bi = bi + 128 * wi.reshape(w.shape[0], -1).sum(1)
        return wi, bi, s_b, bound

    def _act(self, ai, bits):
        s = Q.sym_scale(bits, self.amin[ai], self.amax[ai]).reshape(1)
        lo, hi = Q.qrange(bits)
        return s, lo, hi

    def _consume(self, value: _Value, ai, bits) -> _Consumer:
        s, lo, hi = self._act(ai, bits)
        c = _Consumer(s, lo, hi)
        if value.kind == 'first':
            c.tensor = value.tensor
        value.consumers.append(c)
        return c

    # ------------------------------------------------------------------ pass 1: graph
    def build_graph(self):
        cfg = self.cfg
        plan = conv_plan(cfg)
        self.domains.append(dict(parent=-1, kernel=1, stride=1, dilation=1, padding=0))
        t_in = self._tensor(cfg.feat_in, DT_F32, 0, -1)
        ai = 0
        first_site = plan[0][0]
        assert not first_site.asymmetric
        s0, lo0, hi0 = self._act(0, self.abit)
        n0 = 2 ** (self.abit - 1) - 1
        qop = dict(kind=OP_QUANT_IN, flags=F_MASK_OUT, in_tensor=t_in, cin=cfg.feat_in, cout=cfg.feat_in,
                   inv_scale=float((1.0 / s0)[0]), qlo=-n0, qhi=n0 - 1, consumers_of=None)
        self.ops.append(qop)
        vfirst = _Value('first', 0, cfg.feat_in, 0, s0)
        vfirst.tensor = self._tensor(cfg.feat_in, DT_S8, 0, 0)
        qop['out_tensor'] = vfirst.tensor
        self.values.append(vfirst)

        xs = [vfirst]
        for bi, sites in enumerate(plan):
            blk = cfg.blocks[bi]
            msites = [s for s in sites if s.role != 'res']
            rsites = [s for s in sites if s.role == 'res']
            cur = xs[-1]
            for si, s in enumerate(msites):
                bits = self.abit + (1 if s.asymmetric else 0)
                if cur.kind == 'first':
                    assert ai == 0 and not s.asymmetric
                cons = self._consume(cur, ai, bits)
                ai += 1
                in_unsigned = cons.hi > 127
                wi, bint, s_b, bound = self._conv_ints(s.key, s.bn_key, cons.s_x, in_unsigned,
                                                       max(abs(cons.lo), abs(cons.hi)))
                exact = bound > Z_EXACT_LIMIT
                dom = cur.domain
                if not (s.stride == 1 and 2 * s.padding == s.dilation * (s.kernel - 1)):
                    self.domains.append(dict(parent=dom, kernel=s.kernel, stride=s.stride, dilation=s.dilation,
                                             padding=s.padding))
                    dom = len(self.domains) - 1
                kind = OP_DW if s.role == 'dw' else (OP_PW if s.kernel == 1 else OP_DENSE)
                op = dict(kind=kind, flags=F_MASK_OUT | (F_RELU if s.relu_after else 0) | (F_EXACT_Z if exact else 0),
                          site=s, inp=cons, wi=wi, bint=bint, s_b=s_b, panes=[], in_unsigned=in_unsigned, bound=bound)
                if kind == OP_DENSE and s.stride == 1 and (s.kernel & 1) and 2 * s.padding == s.dilation * (s.kernel - 1):
                    op['flags'] |= F_TAPMAJOR                # runs as taps shifted 1x1 GEMMs on the tile kernel
                self.ops.append(op)
                self.sites.append((len(self.ops) - 1, -1))
                cur = _Value('acc', len(self.ops) - 1, s.cout, dom, s_b)
                self.values.append(cur)
            main_op = self.ops[cur.op]
            if rsites:
                S, qlo, qhi = self._act(ai + len(rsites), self.abit)
                main_op['flags'] |= F_RESADD
                main_op['S'] = S
                main_op['qlo'], main_op['qhi'] = qlo, qhi
                for s in rsites:
                    src = xs[s.pane]
                    bits = self.abit + (1 if s.asymmetric else 0)
                    cons = self._consume(src, ai, bits)
                    ai += 1
                    in_unsigned = cons.hi > 127
                    wi, bint, s_b, bound = self._conv_ints(s.key, s.bn_key, cons.s_x, in_unsigned,
                                                           max(abs(cons.lo), abs(cons.hi)))
                    exact = bound > Z_EXACT_LIMIT
                    if exact:
                        main_op['flags'] |= F_EXACT_Z
                    main_op['panes'].append(dict(site=s, inp=cons, wi=wi, bint=bint, s_b=s_b,
                                                 in_unsigned=in_unsigned, bound=bound))
                    self.sites.append((cur.op, len(main_op['panes']) - 1))
                assert len(main_op['panes']) <= MAX_PANES
                cur = _Value('q', cur.op, main_op['site'].cout, cur.domain, S)
                self.values.append(cur)
            ai += 1                                        # res_act slot
            main_op['flags'] |= F_RELU                     # JasperBlock.mout (jasper.py:687)
            xs = xs + [cur] if (rsites and blk.residual_dense) else [cur]

        # decoder: QuantAct(abit, signed) -> 1x1 conv with real bias -> logits (conv_asr.py:270-275)
        enc = xs[-1]
        self.ops[enc.op]['flags'] &= ~F_MASK_OUT           # the decoder does not mask its input
        cons = self._consume(enc, ai, self.abit)
        ai += 1
        assert ai == len(self.amin), (ai, len(self.amin))
        wi, bint, s_b, _ = self._conv_ints('decoder.decoder_layers.0', None, cons.s_x, False,
                                           max(abs(cons.lo), abs(cons.hi)))
        ncls = wi.shape[0]
        dsite = type('S', (), dict(key='decoder', cin=enc.channels, cout=ncls, kernel=1, stride=1, dilation=1,
                                   padding=0, groups=1, role='dec'))()
        dop = dict(kind=OP_PW, flags=F_LOGITS, site=dsite, inp=cons, wi=wi, bint=bint, s_b=s_b, panes=[],
                   in_unsigned=False)
        self.ops.append(dop)
        self.sites.append((len(self.ops) - 1, -1))
        self.dec_value = _Value('logits', len(self.ops) - 1, ncls, enc.domain, s_b)
        self.out_domain = enc.domain

    # ------------------------------------------------------------------ pass 2: outs, tensors, order
    def resolve(self):
        final_ops = []
        remap = {}
        extra = {}                # producer op -> list of REQUANT op dicts
        for v in self.values:
            if v.kind == 'first':
                continue
            prod = self.ops[v.op]
            outs = prod.setdefault('outs', [])
            if v.kind == 'acc' and (prod['flags'] & F_RESADD):
                continue          # the accumulator of a RESADD op is consumed inside the op; its 'q' value has the outs
            mode = 1 if v.kind == 'acc' else 0
            # Consumers whose QuantAct ended up with the SAME range and width read the same integers: a block's output
            # feeds the next block's first conv and the residual convs of later blocks, each behind a QuantAct of its own
            # that was calibrated on that very tensor (jasper.py:664-676) - identical x_min / x_max, identical
            # (multiplier, clamp).  Such consumers share ONE stored tensor (compared on the exact float64 multipliers and
            # clamp bounds: a checkpoint whose ranges differ keeps them apart).
            groups = []                                    # [(key, dtype, lo, hi, M, [consumers])]
            for c in v.consumers:
                dt = DT_U8 if c.hi > 127 else DT_S8
                M = Q.requant_multiplier(v.scale, c.s_x)
                lo = max(c.lo, 0) if self._nonneg(prod) else c.lo
                key = (dt, lo, c.hi, np.asarray(M, dtype=np.float64).tobytes())
                for g in groups:
                    if g[0] == key:
                        g[5].append(c)
                        break
                else:
                    groups.append((key, dt, lo, c.hi, M, [c]))
            if len(groups) <= MAX_OUTS:
                for _, dt, lo, hi, M, members in groups:
                    t = self._tensor(v.channels, dt, v.domain, v.op)
                    for c in members:
                        c.tensor = t
                    outs.append(dict(tensor=t, lo=lo, hi=hi, mode=mode, M=M))
            else:
                raw_dt = DT_I32 if v.kind == 'acc' else DT_S8
                raw = self._tensor(v.channels, raw_dt, v.domain, v.op)
                outs.append(dict(tensor=raw, lo=0, hi=0, mode=3 if v.kind == 'acc' else 2, M=None))
                for _, dt, lo, hi, M, members in groups:
                    t = self._tensor(v.channels, dt, v.domain, -2)
                    for c in members:
                        c.tensor = t
                    extra.setdefault(v.op, []).append(dict(
                        kind=OP_REQUANT, flags=(prod['flags'] & (F_MASK_OUT | F_EXACT_Z)), in_tensor=raw,
                        cin=v.channels, cout=v.channels, s_b=v.scale if v.kind == 'acc' else None,
                        outs=[dict(tensor=t, lo=lo, hi=hi, mode=mode, M=M)]))
        for i, op in enumerate(self.ops):
            remap[i] = len(final_ops)
            final_ops.append(op)
            for r in extra.get(i, []):
                final_ops.append(r)
        self.sites = [(remap[o], p) for (o, p) in self.sites]
        # logits + log-softmax
        t_logits = self._tensor(self.dec_value.channels, DT_F32, self.out_domain, remap[self.dec_value.op])
        self.ops[self.dec_value.op]['outs'] = [dict(tensor=t_logits, lo=0, hi=0, mode=3, M=None)]
        final_ops.append(dict(kind=OP_LOGSOFTMAX, flags=0, in_tensor=t_logits, cin=self.dec_value.channels,
                              cout=self.dec_value.channels, outs=[]))
        self.final_ops = final_ops
        # producer / last_use bookkeeping
        for oi, op in enumerate(final_ops):
            for t in self._op_inputs(op):
                self.tensors[t]['last_use'] = max(self.tensors[t]['last_use'], oi)
            for o in op.get('outs', []):
                self.tensors[o['tensor']]['producer'] = oi
            if 'out_tensor' in op:
                self.tensors[op['out_tensor']]['producer'] = oi

    @staticmethod
    def _exact_z_matters(op):
        """QASR_F_EXACT_Z asks the kernels for fixedpoint_mul's float32 round trip z = rint(fl32(fl32(acc) s) / s)
        (quant_utils.py:187) wherever |acc| may exceed 2^22 - 1, because only below that bound z == acc is a theorem.
        Above it |z - acc| <= 2, i.e. the product z M moves by at most 2 M.  When even the smallest multiplier of every
        consumer already drives such an accumulator past the clamp range, (2^22 - 3) M - 1/2 >= max(|lo|, |hi|), both z and
        acc requantise to the same saturated code and the round trip cannot change a result: the flag is dropped
        (ordinary layers: M ~ 1e-3, i.e. 2^22 M ~ 4000 against a range of 128 / 255).  res_act sums two unbounded terms
        and raw / identity outputs keep the accumulator itself, so those ops keep the flag."""
        if op['flags'] & F_RESADD:
            return True
        outs = op.get('outs', [])
        if not outs or any(o['mode'] != 1 for o in outs):
            return True
        for o in outs:
            m_min = float(_t(o['M']).min())
            if (Z_EXACT_LIMIT - 2) * m_min - 0.5 < max(abs(o['lo']), abs(o['hi'])):
                return True
        return False

    @staticmethod
    def _wide_requant(op):
        """True when some |acc * M| of the op may reach 2^30: k_sep2 takes the rounded product from the low mantissa
        word (exact below 2^31), k_sep clamps in the double domain and takes such ops instead."""
        b = float(op.get('bound', 0))
        ms = []
        if op['flags'] & F_RESADD:
            ms.append((b, Q.requant_multiplier(op['s_b'], op['S'])))
            ms += [(float(p['bound']), Q.requant_multiplier(p['s_b'], op['S'])) for p in op['panes']]
        else:
            ms += [(b, o['M']) for o in op.get('outs', []) if o['mode'] == 1 and o['M'] is not None]
        return any(bb * float(_t(m).abs().max()) >= RQ_NARROW_LIMIT for bb, m in ms)

    @staticmethod
    def _nonneg(prod):
        # after ReLU the requantised value is >= 0: fold max(.,0) into the clamp's lower bound
        return bool(prod['flags'] & F_RELU)

    @staticmethod
    def _op_inputs(op):
        if 'in_tensor' in op:
            return [op['in_tensor']]
        return [op['inp'].tensor] + [p['inp'].tensor for p in op['panes']]

    # ------------------------------------------------------------------ serialise
    def _pad_rows(self, a: torch.Tensor, rows, fill=0):
        out = torch.full((rows,) + tuple(a.shape[1:]), fill, dtype=a.dtype)
        out[:a.shape[0]] = a
        return out

    def _pack_weights(self, kind, wi, tap_major=False):
        cout = wi.shape[0]
        cp = _rup(cout, COUT_ALIGN)
        if kind == OP_DW:
            k = wi.shape[2]
            kp = _rup(k, 4)
            w = torch.zeros(cout, kp, dtype=torch.int8)
            w[:, :k] = wi[:, 0, :].to(torch.int8)
            return self._put_w(w.numpy())
        cin, k = wi.shape[1], wi.shape[2]
        cinp = _rup(cin, CIN_ALIGN)
        w = torch.zeros(cp, k, cinp, dtype=torch.int8)
        w[:cout, :, :cin] = wi.permute(0, 2, 1).to(torch.int8)
        if kind == OP_PW:
            return self._put_w(fragment_order(w[:, 0, :].numpy()))
        if tap_major:                                        # one fragment-ordered [cout_pad][cin_pad] matrix per tap
            return self._put_w(np.concatenate([fragment_order(np.ascontiguousarray(w[:, t, :].numpy())) for t in range(k)]))
        return self._put_w(w.numpy())

    def _vec(self, t, rows, dtype, fill=0):
        a = np.full(rows, fill, dtype=dtype)
        a[:t.numel()] = t.reshape(-1).numpy().astype(dtype)
        return self._put(a)

    def serialise(self):
        OUT = struct.Struct('<iiiIQd')                    # qasr_out: 32 B
        PANE = struct.Struct('<iIQQQQ')                   # qasr_pane: 40 B
        HEAD = struct.Struct('<IIiIIIIIII QQQQ ii f I')   # fixed part of qasr_op_desc
        recs = []
        for op in self.final_ops:
            kind = op['kind']
            site = op.get('site')
            cout = op['cout'] if site is None else site.cout
            cp = _rup(cout, COUT_ALIGN)
            w_off = bias_off = m_off = sb_off = 0
            if kind in (OP_DW, OP_PW, OP_DENSE) and self._wide_requant(op):
                op['flags'] |= F_WIDE_RQ
            if kind in (OP_DW, OP_PW, OP_DENSE) and (op['flags'] & F_EXACT_Z) and not self._exact_z_matters(op):
                op['flags'] &= ~F_EXACT_Z
            if kind in (OP_DW, OP_PW, OP_DENSE) and self.wbit <= 6:
                op['flags'] |= F_W6PACK                      # weights (and panes' weights) are pack6 arrays
            if kind in (OP_DW, OP_PW, OP_DENSE):
                w_off = self._pack_weights(kind, op['wi'], bool(op['flags'] & F_TAPMAJOR))
                bias_off = self._vec(op['bint'].to(torch.int32), cp, np.int32)
                sb_off = self._vec(op['s_b'], cp, np.float32, 1.0)
                if op['flags'] & F_RESADD:
                    m_off = self._vec(Q.requant_multiplier(op['s_b'], op['S']), cp, np.float64, 0.0)
                elif kind == OP_DW and self.wbit <= 6:
                    m_off = 0                                # sub-byte blobs: the engine derives the zero-margined rows on load
                elif kind == OP_DW:                          # zero-margined tap rows for the MFMA depthwise stage
                    wi = op['wi']
                    k = wi.shape[2]
                    w2 = np.zeros((wi.shape[0], _rup(k, 4) + 32), dtype=np.int8)
                    w2[:, 8:8 + k] = wi[:, 0, :].to(torch.int8).numpy()
                    m_off = self._put(w2)
            elif kind == OP_REQUANT and op.get('s_b') is not None:
                sb_off = self._vec(op['s_b'], cp, np.float32, 1.0)
            outs = b''
            for j in range(MAX_OUTS):
                if j < len(op.get('outs', [])):
                    o = op['outs'][j]
                    mo, ms = 0, 0.0
                    if o['mode'] == 1:
                        mo = self._vec(o['M'], cp, np.float64, 0.0)
                    elif o['mode'] == 0:
                        ms = float(o['M'].reshape(-1)[0])
                    outs += OUT.pack(o['tensor'], o['lo'], o['hi'], o['mode'], mo, ms)
                else:
                    outs += OUT.pack(-1, 0, 0, 0, 0, 0.0)
            panes = b''
            for j in range(MAX_PANES):
                if j < len(op.get('panes', [])):
                    p = op['panes'][j]
                    ps = p['site']
                    panes += PANE.pack(p['inp'].tensor, ps.cin, self._pack_weights(OP_PW, p['wi']),
                                       self._vec(p['bint'].to(torch.int32), cp, np.int32),
                                       self._vec(Q.requant_multiplier(p['s_b'], op['S']), cp, np.float64, 0.0),
                                       self._vec(p['s_b'], cp, np.float32, 1.0))
                else:
                    panes += PANE.pack(-1, 0, 0, 0, 0, 0)
            in_t = op['in_tensor'] if 'in_tensor' in op else op['inp'].tensor
            if site is not None:
                geo = (site.cin, site.cout, site.kernel, site.stride, site.dilation, site.padding)
            else:
                geo = (op['cin'], op['cout'], 1, 1, 1, 0)
            head = HEAD.pack(kind, op['flags'], in_t, *geo, len(op.get('panes', [])),
                             w_off, bias_off, m_off, sb_off,
                             op.get('qlo', 0), op.get('qhi', 0), op.get('inv_scale', 0.0), 0)
            rec = head + outs + panes
            recs.append(rec)
        if self.final_ops[0]['kind'] == OP_QUANT_IN:       # QUANT_IN writes its tensor through outs[0] too
            pass
        op_size = len(recs[0])
        assert all(len(r) == op_size for r in recs)
        TENS = struct.Struct('<IIIIii')
        DOM = struct.Struct('<iIIII3I')
        tens = b''.join(TENS.pack(t['channels'], t['dtype'], t['domain'], 0, t['producer'], t['last_use'])
                        for t in self.tensors)
        doms = b''.join(DOM.pack(d['parent'], d['kernel'], d['stride'], d['dilation'], d['padding'], 0, 0, 0)
                        for d in self.domains)
        HDR = struct.Struct('<IIIIIIIIII QQQQQ')
        tensors_off = _rup(HDR.size, 16)
        ops_off = _rup(tensors_off + len(tens), 16)
        domains_off = _rup(ops_off + op_size * len(recs), 16)
        data_off = _rup(domains_off + len(doms), 256)
        total = data_off + len(self.data)
        ncls = self.dec_value.channels
        hdr = HDR.pack(MAGIC, VERSION, len(self.tensors), len(recs), self.cfg.feat_in, ncls, self.wbit, self.abit,
                       len(self.domains), op_size, tensors_off, ops_off, domains_off, data_off, total)
        blob = bytearray(total)
        blob[:len(hdr)] = hdr
        blob[tensors_off:tensors_off + len(tens)] = tens
        blob[ops_off:ops_off + op_size * len(recs)] = b''.join(recs)
        blob[domains_off:domains_off + len(doms)] = doms
        blob[data_off:] = self.data
        return bytes(blob)

    def pack(self):
        self.build_graph()
        self.resolve()
        # QUANT_IN's single output
        q = self.final_ops[0]
        q['outs'] = [dict(tensor=q['out_tensor'], lo=q['qlo'], hi=q['qhi'], mode=2, M=None)]
        blob = self.serialise()
        meta = dict(sites=self.sites, n_ops=len(self.final_ops), n_tensors=len(self.tensors),
                    kinds=[o['kind'] for o in self.final_ops],
                    tensors=[dict(t) for t in self.tensors], domains=[dict(d) for d in self.domains],
                    # the decoder's input: the final encoder codes (block 17's output behind the decoder's QuantAct)
                    dec_in=next((o['in_tensor'] if 'in_tensor' in o else o['inp'].tensor)
                                for o in self.final_ops if o['kind'] == OP_PW and o['flags'] & F_LOGITS))
        return blob, meta


def pack_model(cfg, state_dict, act_min, act_max, wbit=8, abit=8):
    """-> (blob bytes, meta).  meta['sites'][i] = (op, pane) of the i-th conv in reference call order."""
    return Packer(cfg, state_dict, act_min, act_max, wbit, abit).pack()
