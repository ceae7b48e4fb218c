"""Pins oracle/int_oracle.py bit-exactly to outputs of the reference's own modules
(fixtures from tests/golden/gen_golden.py, generated in the build container)."""
import json
import os

import numpy as np
import pytest

from oracle import int_oracle as O
from qasr import synth, topology

NETS = ['net_miniq_w8a8', 'net_miniq_w8a8_pct', 'net_miniq_w6a6', 'net_minij_w8a8', 'net_miniq_dyn_w8a8', 'net_miniq_dyn_w6a6', 'net_minij_dyn_w8a8', 'net_miniq_dynp_w8a8',
        'net_minij_dynp_w6a6']
FULL = ['net_quartznet_w8a8', 'net_quartznet_w6a6', 'net_jasper_w8a8', 'net_quartznet_dyn_w8a8', 'net_jasper_dyn_w8a8', 'net_quartznet_dynp_w8a8']


def load(golden_dir, name):
    d = np.load(os.path.join(golden_dir, name + '.npz'))
    return d, json.loads(str(d['meta']))


def run_oracle(d, meta):
    cfg = topology.MODELS[meta['model'] if meta['model'] in topology.MODELS else None]() \
        if meta['model'] in topology.MODELS else None
    return cfg


def _model_cfg(name):
    if 'miniq' in name:
        return topology.mini_quartznet()
    if 'minij' in name:
        return topology.mini_jasper()
    if 'quartznet' in name:
        return topology.quartznet15x5()
    return topology.jasper10x5dr()


def _forward(golden_dir, name):
    d, meta = load(golden_dir, name)
    cfg = _model_cfg(name)
    sd = synth.make_state_dict(cfg, meta['seed'])
    net = O.OracleNet(topology.conv_plan(cfg), cfg, sd, d['act_min'], d['act_max'], meta['wbit'], meta['abit'],
                      dynamic=meta.get('dynamic', False), percentile=meta.get('percentile'),
                      division_residue=meta.get('dynamic', False))
    x = synth.make_features(meta['batch'], cfg.feat_in, meta['frames'], meta['seed'])
    out = net.forward(x, meta['lengths'])
    return d, meta, net, out


def test_frexp(golden_dir):
    d = np.load(os.path.join(golden_dir, 'ops.npz'))
    m, e = O.frexp_me(d['frexp_in'])
    assert np.array_equal(m, d['frexp_m'])
    assert np.array_equal(e.astype(np.float64), d['frexp_e'])


@pytest.mark.parametrize('bits', [8, 6])
def test_first_layer_act(golden_dir, bits):
    d = np.load(os.path.join(golden_dir, 'ops.npz'))
    t = f'first{bits}'
    s = O.sym_scale(bits, d[t + '_range'][0], d[t + '_range'][1])
    assert s == d[t + '_sf'][0]
    q = O.act_clamp(O.quantize(d[t + '_x'], bits, s), bits)
    assert np.array_equal(q, d[t + '_q'])


@pytest.mark.parametrize('tag,bits', [('requant8', 8), ('requant9', 9), ('requant6', 6), ('requant7', 7),
                                      ('requant_big', 8)])
def test_requant(golden_dir, tag, bits):
    d = np.load(os.path.join(golden_dir, 'ops.npz'))
    s = O.sym_scale(bits, d[tag + '_range'][0], d[tag + '_range'][1])
    assert s == d[tag + '_sf'][0]
    q, z = O.fixedpoint_requant(d[tag + '_x'], d[tag + '_pre_sf'].reshape(-1), s)
    assert np.array_equal(O.act_clamp(q, bits), d[tag + '_q'])
    if tag == 'requant_big':        # the float round trip does NOT return acc for |acc| >= 2^22
        assert (z != d['requant_big_acc']).any()
        small = np.abs(d['requant_big_acc']) < (1 << 22)
        assert np.array_equal(z[small], d['requant_big_acc'][small])


@pytest.mark.parametrize('tag', ['res', 'res_sat'])
def test_residual_act(golden_dir, tag):
    d = np.load(os.path.join(golden_dir, 'ops.npz'))
    s = O.sym_scale(8, d[tag + '_range'][0], d[tag + '_range'][1])
    q1, _ = O.fixedpoint_requant(d[tag + '_x'], d[tag + '_pre_sf'].reshape(-1), s)
    q2, _ = O.fixedpoint_requant(d[tag + '_id'], d[tag + '_id_sf'].reshape(-1), s)
    q = O.act_clamp(q1 + q2, 8)
    assert np.array_equal(q, d[tag + '_q'])
    if tag == 'res_sat':
        assert (q == 127).any() and (q == -128).any()


@pytest.mark.parametrize('tag', ['dw_k33_s2', 'dw_k15_d2', 'pw_bn', 'pw_bn_w6', 'dense_k5_bn', 'dec_bias'])
def test_quant_conv(golden_dir, tag):
    d = np.load(os.path.join(golden_dir, 'ops.npz'))
    cin, cout, k, stride, dil, pad, groups, wbit = d[tag + '_cfg'].tolist()
    w = d[tag + '_w']
    b = d[tag + '_b'] if tag + '_b' in d else None
    if tag + '_bn_weight' in d:
        w, b = O.fold_bn(w, b, tuple(d[f'{tag}_bn_{n}'] for n in ('weight', 'bias', 'running_mean', 'running_var')))
    wint, s_w = O.weight_ints(w, wbit)
    assert np.array_equal(wint, d[tag + '_wint'])
    assert np.array_equal(s_w, d[tag + '_wsf'])
    bint, s_b = O.bias_ints(b, s_w, d[tag + '_sx'])
    assert np.array_equal(s_b, d[tag + '_osf'])
    if bint is not None:
        assert np.array_equal(bint, d[tag + '_bint'])
    acc = O.conv1d_int(d[tag + '_xint'].astype(np.int64), wint, bint, stride, pad, dil, groups)
    assert np.array_equal(acc, d[tag + '_acc'])
    y = O.float_view(acc, s_b)
    # float view: the reference's conv_int carries off-integer x_int noise (<= 0.05 acc units)
    assert np.all(np.abs(y - d[tag + '_y']) <= 0.05 * s_b.reshape(1, -1, 1) + 1e-6 * np.abs(y))


@pytest.mark.parametrize('name', NETS)
def test_mini_net_bit_exact(golden_dir, name):
    d, meta, net, out = _forward(golden_dir, name)
    assert len(net.trace) == meta['nconv']
    for i, t in enumerate(net.trace):
        assert np.array_equal(t['wint'], d[f'wint_{i}']), (i, t['key'])
        assert np.array_equal(t['xint'], d[f'xint_{i}']), (i, t['key'])
        if t['bint'] is not None:
            assert np.array_equal(t['bint'], d[f'bint_{i}']), (i, t['key'])
        assert np.array_equal(t['acc'], d[f'acc_{i}']), (i, t['key'])
    assert np.array_equal(out['enc_len'], d['enc_len'])
    assert np.array_equal(out['enc_sf'].reshape(-1), d['enc_sf'])
    assert np.array_equal(np.rint(out['enc_y'] / out['enc_sf'].reshape(1, -1, 1)).astype(np.int32), d['enc_int'])
    assert np.array_equal(out['tokens'], d['tokens'])
    # float logits: accumulators are bit-exact, floats differ by the reference's off-integer
    # x_int noise surviving conv_int.float() (SURVEY Appendix A validation note)
    np.testing.assert_allclose(out['logits'], d['logits'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out['log_probs'], d['log_probs'], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('name', FULL)
def test_full_net_checksums(golden_dir, name):
    d, meta, net, out = _forward(golden_dir, name)
    assert len(net.trace) == meta['nconv']
    for i, t in enumerate(net.trace):
        got = np.concatenate([O.checksum(t['acc']), O.checksum(t['xint']), O.checksum(t['wint'])])
        assert np.array_equal(got, d['conv_checksums'][i]), (i, t['key'])
    assert np.array_equal(out['tokens'], d['tokens'])
    assert np.array_equal(out['enc_len'], d['enc_len'])
    np.testing.assert_allclose(out['logits'], d['logits'], rtol=1e-5, atol=1e-5)


def test_wer_known_answers(golden_dir):
    for c in json.load(open(os.path.join(golden_dir, 'wer.json'))):
        assert abs(O.word_error_rate(c['hyp'], c['ref']) - c['wer']) < 1e-12


def test_ctc_greedy_decode():
    vocab = topology.VOCABULARY
    blank = len(vocab)
    toks = np.array([[3, 3, blank, 3, 1, 1, blank, blank, 20], [blank] * 9])
    assert O.ctc_greedy_decode(toks, vocab) == ['ccat', '']


@pytest.mark.parametrize('name', [n for n in NETS if '_dyn' not in n] + ['net_quartznet_w8a8'])
def test_cpu_baseline_port_matches_reference(golden_dir, name):
    """oracle/fakequant_torch.py (the op sequence bench.py times as cpu_baseline) against the fixtures."""
    from oracle.fakequant_torch import FakeQuantNet
    d, meta = load(golden_dir, name)
    cfg = _model_cfg(name)
    sd = synth.make_state_dict(cfg, meta['seed'])
    net = FakeQuantNet(topology.conv_plan(cfg), cfg, sd, d['act_min'], d['act_max'], meta['wbit'], meta['abit'])
    x = synth.make_features(meta['batch'], cfg.feat_in, meta['frames'], meta['seed'])
    out = net.forward(x, meta['lengths'])
    assert len(net.acc) == meta['nconv']
    assert np.array_equal(out['tokens'].numpy(), d['tokens'])
    assert np.array_equal(out['enc_len'].numpy(), d['enc_len'])
    np.testing.assert_allclose(out['log_probs'].numpy(), d['log_probs'], rtol=1e-5, atol=1e-6)
    for i, a in enumerate(net.acc):
        yi = np.rint(a.numpy())
        if f'acc_{i}' in d:
            assert np.array_equal(yi.astype(np.int32), d[f'acc_{i}']), i
        else:
            assert np.array_equal(O.checksum(yi), d['conv_checksums'][i][:2]), i


def test_quantile_restatement_matches_torch():
    """oracle.quantile_f32 restates torch.quantile (what QuantAct's percentile calibration calls) bit for bit."""
    import torch
    rng = np.random.default_rng(7)
    cases = [rng.standard_normal(100003).astype(np.float32), rng.standard_normal(4096).astype(np.float32) * 50,
             np.abs(rng.standard_normal(65536)).astype(np.float32), np.array([3.0, -1.0], np.float32),
             np.full(1000, 2.5, np.float32), np.round(rng.standard_normal(50000) * 4).astype(np.float32)]
    for x in cases:
        for q in (1 - 99.996 / 100, 99.996 / 100, 0.0, 1.0, 0.5, 0.001, 0.9999):
            want = torch.quantile(torch.from_numpy(x), torch.tensor(q, dtype=torch.float32)).numpy()
            got = O.quantile_f32(x, np.float32(q))
            assert got == want, (x.size, q, got, want)
    # two-element inputs make w = q: many of these separate a fused lerp from an unfused one
    for _ in range(3000):
        a = np.float32(rng.standard_normal() * 10)
        x = np.array([a, np.float32(a + abs(rng.standard_normal()) * 0.1)], np.float32)
        q = np.float32(rng.random())
        want = torch.quantile(torch.from_numpy(x), torch.tensor(q)).numpy()
        assert O.quantile_f32(x, q) == want, (x, q)
