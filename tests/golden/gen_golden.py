#!/usr/bin/env python3
"""Golden-vector generator.  Runs ONLY in the build container (needs /root/reference).

It imports the reference's own operator code (nemo/quantization/utils/*.py and
nemo/collections/asr/parts/jasper.py, unmodified, from /root/reference) on CPU using
the recipe of SURVEY.md Appendix C, feeds it the deterministic synthetic checkpoints
of qasr.synth, and records inputs / expected outputs as small .npz fixtures next to
this script.  The fixtures are data only; no reference source travels.

    python tests/golden/gen_golden.py            # regenerate every fixture

What is captured (all from the reference's forward, nothing recomputed here):
  ops.npz        batch_frexp, QuantAct (first layer / requant / residual incl. a
                 saturating case), QuantConv1d (dw k33 s2, dw k15 d2, pw+BN, decoder+bias)
  net_*.npz      whole encoder+decoder runs: calibrated x_min/x_max of every QuantAct,
                 every conv's rint(conv_int) accumulator, W_int, bias_int, scales, final
                 logits / log-probs / greedy tokens, for mini nets (full tensors) and the
                 full QuartzNet15x5 / Jasper10x5dr (checksums + final outputs)
  wer.json       known answers quoted from the reference's own unit tests
"""
import hashlib
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
os.environ['NEMO_PACKAGE_BUILDING'] = '1'
sys.dont_write_bytecode = True
sys.path.insert(0, '/root/reference')
sys.path.append(os.path.join(ROOT, 'q-asr_amd'))

import numpy as np
import torch
import torch.nn as nn

torch.Tensor.cuda = lambda self, *a, **k: self       # CPU only: neutralise hard-coded .cuda()

# torch's CPU float32 sqrt (MKL VML) is not correctly rounded AND differs between host CPUs
# (~0.7 % of inputs 1 ulp low on this container's Xeon; other values on the GPU box's host), so
# BN folding (quant_modules.py:353) would make the fixtures host-dependent.  Pin it to the IEEE
# correctly-rounded result - what the reference computes on its own (CUDA) platform:
# float32(sqrt(float64(x))) is correctly rounded for float32 (2*24+2 <= 53).
_torch_sqrt = torch.sqrt


def _ieee_sqrt(x, *a, **k):
    if isinstance(x, torch.Tensor) and x.dtype == torch.float32 and not a and not k and not x.requires_grad:   # (autograd: gen_distill)
        return torch.from_numpy(np.sqrt(x.detach().numpy().astype(np.float64)).astype(np.float32))
    return _torch_sqrt(x, *a, **k)


torch.sqrt = _ieee_sqrt
import nemo  # noqa: E402  (light: package_info only)

for _n in ['nemo.collections', 'nemo.collections.asr', 'nemo.collections.asr.parts']:
    _m = types.ModuleType(_n)
    _m.__path__ = ['/root/reference/' + _n.replace('.', '/')]
    sys.modules[_n] = _m
from nemo.collections.asr.parts.jasper import JasperBlock, MaskedConv1d  # noqa: E402
import nemo.quantization.utils.quantize_model as qm  # noqa: E402
import nemo.quantization.utils.quant_modules as ref_qmod  # noqa: E402
import nemo.quantization.utils.quant_utils as ref_qutil  # noqa: E402
from nemo.quantization.utils.quant_modules import QuantAct, QuantConv1d  # noqa: E402

from qasr import synth, topology  # noqa: E402

torch.set_grad_enabled(False)
torch.set_num_threads(8)


# --------------------------------------------------------------------------- helpers
class ConvTap:
    """Records every F.conv1d issued by QuantConv1d.int_conv (quant_modules.py:304)."""

    def __init__(self):
        self.calls = []
        self.enabled = False
        self._orig = ref_qmod.F.conv1d

    def __enter__(self):
        tap = self

        def conv1d(x, weight=None, bias=None, **kw):
            out = tap._orig(x, weight=weight, bias=bias, **kw)
            if tap.enabled and x.dtype == torch.float64:
                tap.calls.append(dict(x=x.clone(), w=weight.clone(),
                                      b=None if bias is None else bias.clone(), y=out.clone(), kw=kw))
            return out

        ref_qmod.F.conv1d = conv1d
        return self

    def __exit__(self, *a):
        ref_qmod.F.conv1d = self._orig


def build_reference_model(cfg, sd, wbit, abit, percentile, fold=True):
    """Restates the 40-line assembly loop of ConvASREncoder.__init__ (conv_asr.py:136-192)
    and ConvASRDecoder.__init__ (:247-268) around the reference's own JasperBlock /
    QuantAct / QuantConv1d, then loads the synthetic checkpoint by its NeMo keys."""
    blocks, panes, feat_in = [], [], cfg.feat_in
    for i, b in enumerate(cfg.blocks):
        dense = []
        if b.residual_dense:
            panes.append(feat_in)
            dense = panes
        blocks.append(JasperBlock(
            feat_in, b.filters, repeat=b.repeat, kernel_size=[b.kernel], stride=[b.stride],
            dilation=[b.dilation], dropout=0.0, residual=b.residual, groups=1, separable=b.separable,
            heads=-1, residual_mode='add', normalization='batch', norm_groups=-1,
            activation=nn.ReLU(), residual_panes=dense, conv_mask=True, se=False,
            quant_mode='symmetric', quant_bit=8, layer_num=i))
        feat_in = b.filters

    class Enc(nn.Module):
        def __init__(self):
            super().__init__()
            self.encoder = nn.Sequential(*blocks)

    class Dec(nn.Module):
        def __init__(self):
            super().__init__()
            self.act = QuantAct(8, quant_mode='symmetric', per_channel=False)
            conv = nn.Conv1d(feat_in, cfg.num_classes + 1, kernel_size=1, bias=True)
            q = QuantConv1d(8, bias_bit=32, quant_mode='symmetric', per_channel=True)
            q.set_param(conv)
            self.decoder_layers = nn.Sequential(q)

    class Model(nn.Module):
        def __init__(self):
            super().__init__()
            self.encoder = Enc()
            self.decoder = Dec()

    model = Model()
    tsd = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    missing, unexpected = model.load_state_dict(tsd, strict=False)
    assert not unexpected, unexpected
    for k in missing:  # only the fork's extra buffers / duplicate inner conv may be missing
        assert any(t in k for t in ('.conv.conv.', 'decoder_layers.0.conv.', 'weight_integer', 'bias_integer', 'conv_scaling_factor',
                                    'x_min', 'x_max', 'act_scaling_factor')), k
    model.eval()
    for blk in blocks:
        blk.set_quant_bit(wbit, 'weight')
        blk.set_quant_bit(abit, 'act')
    model.decoder.act.activation_bit = abit
    model.decoder.decoder_layers[0].weight_bit = wbit
    if percentile is not None:
        qm.set_percentile(model, percentile)
    if fold:
        for blk in blocks:
            blk.bn_folding()
    return model, blocks


def encoder_forward(blocks, x, length):
    s_input = [(x, None)]
    for blk in blocks:
        s_input, length = blk((s_input, length))
    out, sf = s_input[-1]
    return out, length, sf


def decoder_forward(dec, enc_out, enc_sf):
    out, sf = dec.act(enc_out, enc_sf)
    logits, _ = dec.decoder_layers[0](out, sf)
    return logits, torch.log_softmax(logits.transpose(1, 2), dim=-1)


def quant_acts_in_order(model, blocks):
    """QuantAct modules in the order SURVEY Appendix A / qasr.topology.conv_plan uses:
    per block mconv sites, then residual sites, then res_act; decoder act last."""
    acts = []
    for blk in blocks:
        for l in blk.mconv:
            if isinstance(l, MaskedConv1d):
                acts.append(l.act)
        if blk.res is not None:
            for lst in blk.res:
                for l in lst:
                    if isinstance(l, MaskedConv1d):
                        acts.append(l.act)
        acts.append(blk.res_act)
    acts.append(model.decoder.act)
    return acts


def checksum(a):
    """Order-sensitive 64-bit checksum of an integer tensor (also used by the tests)."""
    a = np.ascontiguousarray(a.astype(np.int64)).ravel()
    idx = (np.arange(a.size, dtype=np.int64) % 65521) + 1
    return np.array([a.sum(), (a * idx).sum()], dtype=np.int64)


def run_net(name, cfg, seed, wbit, abit, percentile, batch, frames, lengths, ncal, cal_batch,
            full_tensors, dynamic=False):
    sd = synth.make_state_dict(cfg, seed)
    model, blocks = build_reference_model(cfg, sd, wbit, abit, percentile)
    if not dynamic:
        cal = synth.make_calibration(ncal, cal_batch, cfg.feat_in, frames, seed)
        qm.calibrate(model)
        clen = torch.tensor([frames] * cal_batch)
        for c in cal:
            o, l, sf = encoder_forward(blocks, torch.from_numpy(c), clen)
            decoder_forward(model.decoder, o, sf)
    qm.evaluate(model)
    qm.set_dynamic(model, dynamic)      # inference.py:101; dynamic: every QuantAct ranges over the batch in front of it

    x = synth.make_features(batch, cfg.feat_in, frames, seed)
    lens = torch.tensor(lengths)
    with ConvTap() as tap:
        tap.enabled = True
        enc, enc_len, enc_sf = encoder_forward(blocks, torch.from_numpy(x), lens)
        logits, logp = decoder_forward(model.decoder, enc, enc_sf)
    tokens = logp.argmax(-1)

    acts = quant_acts_in_order(model, blocks)
    out = dict(
        meta=np.array(json.dumps(dict(model=name, seed=seed, wbit=wbit, abit=abit, percentile=percentile,
                                      batch=batch, frames=frames, lengths=list(lengths), ncal=ncal,
                                      cal_batch=cal_batch, nconv=len(tap.calls), dynamic=bool(dynamic)))),
        act_min=np.array([float(a.x_min) for a in acts], dtype=np.float32),
        act_max=np.array([float(a.x_max) for a in acts], dtype=np.float32),
        act_sf=np.array([float(a.act_scaling_factor.reshape(-1)[0]) for a in acts], dtype=np.float32),
        enc_len=enc_len.numpy().astype(np.int64),
        enc_sf=enc_sf.reshape(-1).numpy().astype(np.float32),
        logits=logits.numpy().astype(np.float32),
        log_probs=logp.numpy().astype(np.float32),
        tokens=tokens.numpy().astype(np.int64),
    )
    dev = 0.0
    sums = []
    for i, c in enumerate(tap.calls):
        y = c['y'].numpy()
        yi = np.rint(y)
        dev = max(dev, float(np.abs(y - yi).max()))
        xi = np.rint(c['x'].numpy())
        sums.append(np.concatenate([checksum(yi), checksum(xi), checksum(c['w'].numpy())]))
        if full_tensors:
            out[f'acc_{i}'] = yi.astype(np.int32)
            out[f'xint_{i}'] = xi.astype(np.int16)
            out[f'wint_{i}'] = c['w'].numpy().astype(np.int8)
            if c['b'] is not None:
                out[f'bint_{i}'] = c['b'].numpy().astype(np.int64)
    out['conv_checksums'] = np.stack(sums)
    out['conv_int_max_dev'] = np.array(dev)
    enc_int = np.rint((enc / enc_sf).numpy())
    out['enc_int_checksum'] = checksum(enc_int)
    if full_tensors:
        out['enc_int'] = enc_int.astype(np.int32)
    np.savez_compressed(os.path.join(HERE, f'{name}.npz'), **out)
    print(f'{name}: {len(tap.calls)} convs, max |conv_int - rint| = {dev:.3g}, '
          f'tokens[0][:12]={tokens[0][:12].tolist()}')


# --------------------------------------------------------------------------- op fixtures
def gen_ops():
    rng = np.random.Generator(np.random.PCG64(7))
    out = {}
    # batch_frexp (quant_utils.py:121-147): random ratios + exact powers of two + tiny/large
    r = np.concatenate([np.exp(rng.uniform(np.log(1e-6), np.log(50.0), 900)),
                        2.0 ** np.arange(-20, 6), [1.0, 0.75, 0.5000000001, 0.9999999999, 3e-9, 1e4]])
    m, e = ref_qutil.batch_frexp(torch.from_numpy(r).view(1, -1, 1))
    out['frexp_in'] = r
    out['frexp_m'] = m.view(-1).numpy().astype(np.int64)
    out['frexp_e'] = e.view(-1).numpy().astype(np.float64)

    def act_case(tag, bits, x, pre_sf, lo, hi, identity=None, id_sf=None):
        a = QuantAct(bits, quant_mode='symmetric', per_channel=False)
        a.fix()
        a.x_min = torch.tensor([lo], dtype=torch.float32)
        a.x_max = torch.tensor([hi], dtype=torch.float32)
        y, sf = a(torch.from_numpy(x), None if pre_sf is None else torch.from_numpy(pre_sf),
                  None if identity is None else torch.from_numpy(identity),
                  None if id_sf is None else torch.from_numpy(id_sf))
        out[f'{tag}_x'] = x
        if pre_sf is not None:
            out[f'{tag}_pre_sf'] = pre_sf
        if identity is not None:
            out[f'{tag}_id'] = identity
            out[f'{tag}_id_sf'] = id_sf
        out[f'{tag}_range'] = np.array([lo, hi], dtype=np.float32)
        out[f'{tag}_sf'] = sf.reshape(-1).numpy().astype(np.float32)
        out[f'{tag}_q'] = np.rint((y / sf).numpy()).astype(np.int32)
        out[f'{tag}_y'] = y.numpy().astype(np.float32)

    B, C, T = 2, 24, 40
    for bits in (8, 6):
        x = rng.standard_normal((B, C, T)).astype(np.float32) * 1.3
        act_case(f'first{bits}', bits, x, None, -2.9, 3.1)
    for bits in (8, 9, 6, 7):
        pre = np.exp(rng.uniform(np.log(1e-4), np.log(3e-3), (1, C, 1))).astype(np.float32)
        acc = rng.integers(-40000, 40000, (B, C, T)).astype(np.float32)
        if bits in (9, 7):
            acc = np.maximum(acc, 0)
        x = (acc * pre).astype(np.float32)
        act_case(f'requant{bits}', bits, x, pre, float(x.min()) * 0.8, float(x.max()) * 0.8)
    # large accumulators (|acc| up to 2^24): exercises the float round trip z != acc regime
    pre = np.exp(rng.uniform(np.log(1e-6), np.log(3e-5), (1, C, 1))).astype(np.float32)
    acc = rng.integers(-(1 << 24) + 1, (1 << 24) - 1, (B, C, T)).astype(np.float32)
    x = (acc * pre).astype(np.float32)
    act_case('requant_big', 8, x, pre, float(x.min()) * 0.7, float(x.max()) * 0.7)
    out['requant_big_acc'] = acc.astype(np.int32)
    # residual add: normal and saturating (range far too small -> both operands clamp)
    for tag, shrink in (('res', 0.9), ('res_sat', 0.15)):
        pre = np.exp(rng.uniform(np.log(1e-4), np.log(3e-3), (1, C, 1))).astype(np.float32)
        ipre = np.exp(rng.uniform(np.log(1e-4), np.log(3e-3), (1, C, 1))).astype(np.float32)
        x = (rng.integers(-30000, 30000, (B, C, T)).astype(np.float32) * pre).astype(np.float32)
        idn = (rng.integers(-30000, 30000, (B, C, T)).astype(np.float32) * ipre).astype(np.float32)
        s = x + idn
        act_case(tag, 8, x, pre, float(s.min()) * shrink, float(s.max()) * shrink, idn, ipre)

    def conv_case(tag, cin, cout, k, stride, dil, pad, groups, bias, bn, wbit, xbits_unsigned):
        conv = nn.Conv1d(cin, cout, k, stride=stride, padding=pad, dilation=dil, groups=groups, bias=bias)
        w = rng.uniform(-0.3, 0.3, conv.weight.shape).astype(np.float32)
        conv.weight.data = torch.from_numpy(w)
        if bias:
            conv.bias.data = torch.from_numpy(rng.normal(0, 0.5, cout).astype(np.float32))
        q = QuantConv1d(wbit, bias_bit=32, quant_mode='symmetric', per_channel=True)
        q.set_param(conv)
        if bn:
            b = nn.BatchNorm1d(cout, eps=1e-3)
            b.weight.data = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32))
            b.bias.data = torch.from_numpy(rng.normal(0, 0.3, cout).astype(np.float32))
            b.running_mean = torch.from_numpy(rng.normal(0, 0.1, cout).astype(np.float32))
            b.running_var = torch.from_numpy(rng.uniform(0.3, 0.7, cout).astype(np.float32))
            b.eval()
            q.bn_folding(b)
            for nme in ('weight', 'bias', 'running_mean', 'running_var'):
                out[f'{tag}_bn_{nme}'] = getattr(b, nme).detach().numpy().astype(np.float32)
        q.eval()
        sx = np.float32(0.0123)
        lo, hi = (0, 255) if xbits_unsigned else (-128, 127)
        xi = rng.integers(lo, hi + 1, (2, cin, 50)).astype(np.float32)
        x = (xi * sx).astype(np.float32)
        with ConvTap() as tap:
            tap.enabled = True
            y, sf = q(torch.from_numpy(x), torch.tensor([[[sx]]]))
        c = tap.calls[0]
        out[f'{tag}_w'] = w
        if bias:
            out[f'{tag}_b'] = conv.bias.detach().numpy().astype(np.float32)
        out[f'{tag}_xint'] = xi.astype(np.int16)
        out[f'{tag}_sx'] = np.array(sx)
        out[f'{tag}_wint'] = c['w'].numpy().astype(np.int8)
        if c['b'] is not None:
            out[f'{tag}_bint'] = c['b'].numpy().astype(np.int64)
        out[f'{tag}_acc'] = np.rint(c['y'].numpy()).astype(np.int32)
        out[f'{tag}_wsf'] = q.conv_scaling_factor.numpy().astype(np.float32)
        out[f'{tag}_osf'] = sf.reshape(-1).numpy().astype(np.float32)
        out[f'{tag}_y'] = y.numpy().astype(np.float32)
        out[f'{tag}_cfg'] = np.array([cin, cout, k, stride, dil, pad, groups, wbit], dtype=np.int64)

    conv_case('dw_k33_s2', 32, 32, 33, 2, 1, 16, 32, False, False, 8, False)
    conv_case('dw_k15_d2', 32, 32, 15, 1, 2, 14, 32, False, False, 8, True)
    conv_case('pw_bn', 48, 64, 1, 1, 1, 0, 1, False, True, 8, False)
    conv_case('pw_bn_w6', 48, 64, 1, 1, 1, 0, 1, False, True, 6, False)
    conv_case('dense_k5_bn', 24, 40, 5, 1, 1, 2, 1, False, True, 8, True)
    conv_case('dec_bias', 64, 29, 1, 1, 1, 0, 1, True, False, 8, False)
    np.savez_compressed(os.path.join(HERE, 'ops.npz'), **out)
    print('ops.npz:', len(out), 'arrays')


def gen_frontend():
    """FilterbankFeatures.forward of the reference (features.py:334-397) on seeded audio, dither off.
    librosa / torch_stft are absent: `librosa.filters.mel` is served by qasr.melbank (so the mel matrix
    itself stays unpinned - real checkpoints carry it), the other imports are inert stubs."""
    from qasr.melbank import mel_filterbank
    lib = types.ModuleType('librosa')
    lib.filters = types.ModuleType('librosa.filters')
    lib.filters.mel = lambda sr, n_fft, n_mels=64, fmin=0, fmax=None: mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
    lib.util = types.ModuleType('librosa.util')
    lib.util.tiny = lambda x: np.finfo(np.float32).tiny
    stubs = {'librosa': lib, 'librosa.filters': lib.filters, 'librosa.util': lib.util}
    ts = types.ModuleType('torch_stft')
    ts.STFT = type('STFT', (torch.nn.Module,), {})
    stubs['torch_stft'] = ts
    nu = types.ModuleType('nemo.utils')
    nu.logging = types.SimpleNamespace(info=lambda *a, **k: None, debug=lambda *a, **k: None,
                                       warning=lambda *a, **k: None)
    stubs['nemo.utils'] = nu
    for n, attr in (('nemo.collections.asr.parts.perturb', 'AudioAugmentor'),
                    ('nemo.collections.asr.parts.segment', 'AudioSegment')):
        m = types.ModuleType(n)
        setattr(m, attr, type(attr, (), {}))
        stubs[n] = m
    sys.modules.update(stubs)
    from nemo.collections.asr.parts.features import FilterbankFeatures
    fe = FilterbankFeatures(sample_rate=16000, n_window_size=320, n_window_stride=160, window='hann',
                            normalize='per_feature', n_fft=512, preemph=0.97, nfilt=64, dither=0.0, pad_to=16)
    fe.eval()
    audio = synth.make_audio(3, 24000, seed=9)
    lens = np.array([24000, 17321, 8000], dtype=np.int64)
    for i, n in enumerate(lens):
        audio[i, n:] = 0
    y, seq = fe(torch.from_numpy(audio.copy()), torch.from_numpy(lens))
    np.savez_compressed(os.path.join(HERE, 'frontend.npz'), audio=audio, lens=lens, feats=y.numpy().astype(np.float32),
                        seq_len=seq.numpy().astype(np.int64), fb=fe.fb[0].numpy().astype(np.float32),
                        window=fe.window.numpy().astype(np.float32))
    print('frontend.npz:', tuple(y.shape), seq.tolist())


def gen_distill():
    """Zero-shot calibration data: the reference's OWN nemo/quantization/utils/distill_data.py
    (get_synthetic_data, _kl_loss; imported unmodified) on a float ('none' mode, BatchNorm not folded) MiniQuartzNet built
    from the reference's JasperBlocks.  Its random start (_get_random_data: 32 loader workers drawing uniform noise) is
    replaced by a fixed seeded tensor; 3 Adam iterations on 2 batches.  Stored: the start, every _kl_loss call's inputs'
    result, the loss of every iteration, the refined batches, and stand-alone _kl_loss known answers."""
    import nemo.quantization.utils.distill_data as ref_dd
    cfg = topology.MODELS['MiniQuartzNet']()
    seed, B, T, iters, nb, lr = 3, 2, 48, 3, 2, 0.05
    sd = synth.make_state_dict(cfg, seed)
    model, blocks = build_reference_model(cfg, sd, 8, 8, None, fold=False)
    for blk in blocks:
        blk.set_quant_mode('none')
    model.decoder.act.quant_mode = 'none'
    model.decoder.decoder_layers[0].quant_mode = 'none'

    class Teacher(nn.Module):                                  # ConvASREncoder's forward + convs_before_bn (conv_asr.py:134,185,194-206)
        def __init__(self):
            super().__init__()
            self.encoder = nn.Sequential(*blocks)
            self.convs_before_bn = [cb for blk in blocks for cb in blk.convs_before_bn]

        def forward(self, x, length):
            return encoder_forward(blocks, x, length)

    class TeacherDecoder(nn.Module):
        def __init__(self):
            super().__init__()
            self.dec = model.decoder

        def forward(self, encoder_output, encoder_output_scaling_factor=None):
            return decoder_forward(self.dec, encoder_output, encoder_output_scaling_factor)[1]

    g = torch.Generator().manual_seed(1234)
    start = [torch.rand(B, cfg.feat_in, T, generator=g) * 0.6 - 0.3 for _ in range(nb)]
    ref_dd._get_random_data = lambda batch_size, dim, seqlen: [t.clone() for t in start]
    ref_dd.tqdm = lambda it: it
    kl_calls = []
    kl_orig = ref_dd._kl_loss

    def kl_rec(*a):
        v = kl_orig(*a)
        kl_calls.append(float(v))
        return v
    ref_dd._kl_loss = kl_rec
    # ReduceLROnPlateau(verbose=True) is rejected by torch 2.10: same scheduler without the keyword
    rl_orig = ref_dd.optim.lr_scheduler.ReduceLROnPlateau
    ref_dd.optim.lr_scheduler.ReduceLROnPlateau = lambda opt, **kw: rl_orig(opt, **{k: v for k, v in kw.items() if k != 'verbose'})
    with torch.enable_grad():
        out = ref_dd.get_synthetic_data(Teacher(), TeacherDecoder(), batch_size=B, dim=cfg.feat_in, seqlen=T, train_iter=iters,
                                        num_batch=nb, lr=lr)
    ref_dd.optim.lr_scheduler.ReduceLROnPlateau = rl_orig
    n_hooks = len(kl_calls) // (iters * nb)
    losses = np.array(kl_calls, np.float64).reshape(nb * iters, n_hooks).sum(1)
    one = torch.ones(5)
    kl_known = np.array([float(kl_orig(one * 0.3, one * 2, one * 0.3, one * 2)), float(kl_orig(one * 0.3, one * 2, one * 0.5, one * 1.5)),
                         float(kl_orig(torch.tensor([0.1, -0.2]), torch.tensor([1.0, 0.5]), torch.tensor([0.0, 0.3]), torch.tensor([2.0, 0.25])))])
    np.savez_compressed(os.path.join(HERE, 'distill.npz'), start=np.stack([t.numpy() for t in start]),
                        refined=np.stack([t.numpy() for t in out]), kl_calls=np.array(kl_calls, np.float64), losses=losses,
                        kl_known=kl_known,
                        meta=json.dumps(dict(model='MiniQuartzNet', seed=seed, batch=B, frames=T, train_iter=iters, num_batch=nb,
                                             lr=lr, n_hooks=n_hooks)))
    print('distill.npz:', losses.round(4).tolist(), 'max |refined - start| =', float(np.abs(np.stack([t.numpy() for t in out]) - np.stack([t.numpy() for t in start])).max()))


def gen_wer():
    """Known answers quoted from /root/reference/tests/collections/asr/test_asr_metrics.py:94-111."""
    cases = [dict(hyp=['cat'], ref=['cot'], wer=1.0),
             dict(hyp=['GPU'], ref=['G P U'], wer=1.0),
             dict(hyp=['G P U'], ref=['GPU'], wer=3.0),
             dict(hyp=['ducati motorcycle'], ref=['motorcycle'], wer=1.0),
             dict(hyp=['ducati motorcycle'], ref=['ducuti motorcycle'], wer=0.5),
             dict(hyp=['a B c'], ref=['a b c'], wer=1.0 / 3.0)]
    with open(os.path.join(HERE, 'wer.json'), 'w') as f:
        json.dump(cases, f, indent=1)


if __name__ == '__main__':
    which = set(sys.argv[1:])
    if not which or 'ops' in which:
        gen_ops()
        gen_wer()
    if not which or 'frontend' in which:
        gen_frontend()
    if not which or 'distill' in which:
        gen_distill()
    M = topology.MODELS
    if not which or 'mini' in which:
        run_net('net_miniq_w8a8', M['MiniQuartzNet'](), 1, 8, 8, None, 3, 96, (96, 71, 40), 3, 4, True)
        run_net('net_miniq_w8a8_pct', M['MiniQuartzNet'](), 2, 8, 8, 99.9, 3, 96, (96, 64, 33), 3, 4, True)
        run_net('net_miniq_w6a6', M['MiniQuartzNet'](), 3, 6, 6, None, 3, 96, (96, 71, 40), 3, 4, True)
        run_net('net_minij_w8a8', M['MiniJasper'](), 4, 8, 8, None, 3, 96, (96, 80, 37), 3, 4, True)
    if not which or 'dynamic' in which:
        run_net('net_miniq_dyn_w8a8', M['MiniQuartzNet'](), 7, 8, 8, None, 3, 96, (96, 71, 40), 0, 0, True, dynamic=True)
        run_net('net_miniq_dyn_w6a6', M['MiniQuartzNet'](), 8, 6, 6, None, 2, 80, (80, 33), 0, 0, True, dynamic=True)
        run_net('net_quartznet_dyn_w8a8', M['QuartzNet15x5Base-En'](), 9, 8, 8, None, 2, 64, (64, 41), 0, 0, False, dynamic=True)
    if not which or 'dynamic_pct' in which:               # --dynamic --percentile: torch.quantile ranges per batch
        run_net('net_miniq_dynp_w8a8', M['MiniQuartzNet'](), 12, 8, 8, 99.0, 3, 96, (96, 71, 40), 0, 0, True, dynamic=True)
        run_net('net_minij_dynp_w6a6', M['MiniJasper'](), 13, 6, 6, 99.9, 2, 80, (80, 37), 0, 0, True, dynamic=True)
        run_net('net_quartznet_dynp_w8a8', M['QuartzNet15x5Base-En'](), 14, 8, 8, 99.9, 2, 64, (64, 41), 0, 0, False, dynamic=True)
    if not which or 'dynamic_jasper' in which:
        run_net('net_minij_dyn_w8a8', M['MiniJasper'](), 10, 8, 8, None, 3, 96, (96, 80, 37), 0, 0, True, dynamic=True)
        run_net('net_jasper_dyn_w8a8', M['Jasper10x5Dr-En'](), 11, 8, 8, None, 2, 64, (64, 41), 0, 0, False, dynamic=True)
    if not which or 'full' in which:
        run_net('net_quartznet_w8a8', M['QuartzNet15x5Base-En'](), 5, 8, 8, 99.996, 2, 64, (64, 41), 2, 2, False)
        run_net('net_quartznet_w6a6', M['QuartzNet15x5Base-En'](), 5, 6, 6, 99.996, 2, 64, (64, 41), 2, 2, False)
        run_net('net_jasper_w8a8', M['Jasper10x5Dr-En'](), 6, 8, 8, None, 2, 64, (64, 41), 2, 2, False)
