"""GPU tests of the dynamic-quantisation device path: qasr_dyn_* kernels + qasr.dynamic.DynamicRunner
against fixtures the reference's own modules produced with qm.set_dynamic(model, True) (tests/golden/gen_golden.py
`dynamic`), and against the oracle's dynamic mode on inputs of the tests' own making."""
import json
import os

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

import nemo.quantization.utils.quantize_model as qm  # noqa: E402
from nemo.collections.asr.models import EncDecCTCModel  # noqa: E402
from oracle import int_oracle as O  # noqa: E402
from qasr import dynamic, synth, topology  # noqa: E402


@pytest.fixture(scope='module', autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    torch.set_grad_enabled(False)


def _load(golden_dir, name):
    d = np.load(os.path.join(golden_dir, name + '.npz'))
    return d, json.loads(str(d['meta']))


def _cfg(name):
    if 'miniq' in name:
        return topology.mini_quartznet()
    if 'minij' in name:
        return topology.mini_jasper()
    return topology.quartznet15x5() if 'quartznet' in name else topology.jasper10x5dr()


def _run(golden_dir, name):
    d, meta = _load(golden_dir, name)
    cfg = _cfg(name)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, meta['seed']).items()}
    r = dynamic.DynamicRunner(cfg, sd, meta['wbit'], meta['abit'], 'cuda:0', percentile=meta.get('percentile'))
    r.trace = []
    x = synth.make_features(meta['batch'], cfg.feat_in, meta['frames'], meta['seed'])
    out = r.forward(torch.from_numpy(x).cuda(), torch.tensor(meta['lengths']))
    return d, meta, r, out


def _codes(t):
    c = t['codes']
    return (c.view(torch.uint8) if t['unsigned'] else c).cpu().numpy().astype(np.int64)


@pytest.mark.parametrize('name', ['net_miniq_dyn_w8a8', 'net_miniq_dyn_w6a6', 'net_minij_dyn_w8a8', 'net_miniq_dynp_w8a8',
                                  'net_minij_dynp_w6a6'])
def test_dynamic_mini_net_every_accumulator(golden_dir, name):
    """Every conv's input codes and int32 accumulator equal the reference's rint(x_int) / rint(conv_int) in dynamic
    mode (ragged lengths: the mask is part of what each QuantAct ranges over), then lengths, tokens and logits.  `dynp`:
    --dynamic with --percentile (every range = two torch.quantile values of the batch, quant_modules.py:158-167)."""
    d, meta, r, out = _run(golden_dir, name)
    assert len(r.trace) == meta['nconv']
    for i, t in enumerate(r.trace):
        assert np.array_equal(_codes(t), d[f'xint_{i}'].astype(np.int64)), (i, t['key'], 'codes')
        assert np.array_equal(t['acc'].cpu().numpy(), d[f'acc_{i}']), (i, t['key'], 'acc')
    assert np.array_equal(out['enc_len'].cpu().numpy(), d['enc_len'])
    assert np.array_equal(out['tokens'].cpu().numpy(), d['tokens'])
    np.testing.assert_allclose(out['logits'].cpu().numpy(), d['logits'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out['log_probs'].cpu().numpy(), d['log_probs'], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('name', ['net_quartznet_dyn_w8a8', 'net_jasper_dyn_w8a8', 'net_quartznet_dynp_w8a8'])
def test_dynamic_full_net_checksums(golden_dir, name):
    """All 171 convs of QuartzNet15x5 / all 109 of Jasper10x5dr (dense k>1 and strided convs, up to ten sequentially
    re-ranged residual panes per block) in dynamic mode: checksums of accumulators and input codes, tokens."""
    d, meta, r, out = _run(golden_dir, name)
    assert len(r.trace) == meta['nconv']
    for i, t in enumerate(r.trace):
        got = np.concatenate([O.checksum(t['acc'].cpu().numpy()), O.checksum(_codes(t))])
        assert np.array_equal(got, d['conv_checksums'][i][:4]), (i, t['key'])
    assert np.array_equal(out['tokens'].cpu().numpy(), d['tokens'])
    np.testing.assert_allclose(out['logits'].cpu().numpy(), d['logits'], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('percentile', [None, 99.5])
@pytest.mark.parametrize('residue', [True, False])
def test_dynamic_batch1_against_oracle(percentile, residue):
    """BASELINE.json config 1 shape (B = 1, --dynamic, with and without --percentile) on inputs the fixtures do not hold:
    the oracle's dynamic mode (pinned by the fixtures above on CPU) is the checker.  residue: with the reference's float32
    quotient residue in every float view (what the reference computes, the runner's default) and without it (float view =
    integer x scale) - the two arithmetics the oracle and the runner both implement."""
    cfg = topology.mini_quartznet()
    sdn = synth.make_state_dict(cfg, 21)
    x = synth.make_features(1, cfg.feat_in, 200, 5)
    net = O.OracleNet(topology.conv_plan(cfg), cfg, sdn, None, None, 8, 8, dynamic=True, percentile=percentile,
                      division_residue=residue)
    want = net.forward(x, [200])
    r = dynamic.DynamicRunner(cfg, {k: torch.from_numpy(v) for k, v in sdn.items()}, 8, 8, 'cuda:0', percentile=percentile,
                              division_residue=residue)
    r.trace = []
    out = r.forward(torch.from_numpy(x).cuda(), torch.tensor([200]))
    for i, (t, w) in enumerate(zip(r.trace, net.trace)):
        assert float(t['s_x'][0]) == float(w['s_x']), (i, t['key'])
        assert np.array_equal(t['acc'].cpu().numpy(), w['acc']), (i, t['key'])
    assert np.array_equal(out['tokens'].cpu().numpy(), want['tokens'])
    np.testing.assert_array_equal(out['logits'].cpu().numpy(), want['logits'])


def test_dynamic_residue_codes():
    """qasr_dyn_residue_codes: fl32(fl32(q s) / s) - q for every code value (signed and u8) and a spread of scales against
    the oracle's division_residue, reassembled from the (lo, hi) bytes."""
    from qasr import engine
    lib = engine.load_library()
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(3)
    scales = np.concatenate([np.exp(rng.uniform(-20, 2, 60)), [1e-8 / 127, 1.0, 3.0, 1 / 3]]).astype(np.float32)
    nonzero = 0
    for unsigned in (0, 1):
        q = np.arange(256, dtype=np.uint8) if unsigned else np.arange(-128, 128).astype(np.int8)
        codes = torch.from_numpy(q.view(np.int8).copy()).to(dev)
        for s in scales:
            lo, hi = torch.empty_like(codes), torch.empty_like(codes)
            sx = torch.tensor([s], dtype=torch.float32, device=dev)
            engine._check(lib.qasr_dyn_residue_codes(engine._stream_ptr(), engine._ptr(codes), unsigned, engine._ptr(sx), 256,
                                                     engine._ptr(lo), engine._ptr(hi)), 'residue_codes')
            lo, hi = lo.cpu().numpy().astype(np.int64), hi.cpu().numpy().astype(np.int64)
            assert np.abs(lo).max() <= 64 and np.abs(hi).max() <= 2
            want = O.division_residue(q.astype(np.int64), s)
            assert np.array_equal((lo + 128 * hi).astype(np.float64) * 2.0 ** -24, want), (unsigned, s)
            nonzero += int((want != 0).sum())
    assert nonzero > 1000                                     # the residue is common, not a corner case


def test_dynamic_kernels_edge_cases():
    """qasr_dyn_range / act_params on tensors with a known answer: all-zero input (scale floor 1e-8 / n), negative-only
    values, masked rows, and the identity operand."""
    import ctypes as C
    from qasr import engine
    lib = engine.load_library()
    dev = torch.device('cuda:0')
    B, Cc, T, Tp = 2, 5, 70, 128
    rng = np.random.default_rng(0)
    acc = torch.zeros(B, Cc, Tp, dtype=torch.int32, device=dev)
    acc[:, :, :T] = torch.from_numpy(rng.integers(-50000, -10, (B, Cc, T)).astype(np.int32)).to(dev)
    sf = torch.from_numpy(np.exp(rng.uniform(-9, -5, Cc)).astype(np.float32)).to(dev)
    lens = torch.tensor([70, 31], dtype=torch.int32, device=dev)
    mm = torch.zeros(2, dtype=torch.int32, device=dev)
    s = torch.zeros(1, device=dev)
    M = torch.zeros(Cc, dtype=torch.float64, device=dev)

    def view(t, sc, per_channel=True):
        v = dynamic.DynView()
        v.data, v.scale, v.is_int8, v.per_channel = t.data_ptr(), sc.data_ptr(), int(t.dtype == torch.int8), int(per_channel)
        return v

    def rng_of(a, b, lens_, relu):
        va = view(acc, sf)
        engine._check(lib.qasr_dyn_range(engine._stream_ptr(), C.byref(va), None, None, 0, engine._ptr(lens_), int(relu), B, Cc, T,
                                         Tp, engine._ptr(mm)), 'range')
        engine._check(lib.qasr_dyn_act_params(engine._stream_ptr(), engine._ptr(mm), 8, Cc, engine._ptr(sf), 1, None, 0,
                                              engine._ptr(s), engine._ptr(M), None), 'params')
        return float(s[0]), M.cpu().numpy()

    y = (acc[:, :, :T].float() * sf.view(1, -1, 1)).cpu().numpy()
    # negative-only tensor, no mask: scale from |min|
    got, Mg = rng_of(acc, None, None, False)
    want = O.sym_scale(8, y.min(), y.max())
    assert got == float(want)
    m, e = O.requant_multiplier(sf.cpu().numpy(), want)
    assert np.array_equal(Mg, m.astype(np.float64) * np.power(2.0, -e.astype(np.float64)))
    # masked: utterance 1 contributes zeros beyond 31 frames, so max = 0 with a mask, < 0 without
    ym = y.copy()
    ym[1, :, 31:] = 0
    got, _ = rng_of(acc, None, lens, False)
    assert got == float(O.sym_scale(8, ym.min(), ym.max()))
    # ReLU of a negative-only tensor is all zero: the 1e-8 floor
    got, _ = rng_of(acc, None, None, True)
    assert got == float(O.sym_scale(8, 0.0, 0.0)) and got == float(np.float32(1e-8) / np.float32(127))


def test_model_dynamic_mode_runs_on_device():
    """EncDecCTCModel with qm.set_dynamic on cuda routes through DynamicRunner (ctc_models.forward) and agrees with the
    host modules in dynamic mode on tokens."""
    m = EncDecCTCModel.from_synthetic('MiniQuartzNet', seed=3).cuda()
    m.eval()
    m.set_quant_bit(8, mode='all')
    m.encoder.bn_folding()
    qm.evaluate(m)
    qm.set_dynamic(m, True)
    assert m.dynamic_ready() and not m.engine_ready()
    x = torch.from_numpy(synth.make_features(2, 16, 96, 9)).cuda()
    L = torch.tensor([96, 50]).cuda()
    lp, el, tok = m(processed_signal=x, processed_signal_length=L)
    assert isinstance(m._engine, dynamic.DynamicRunner)
    enc, enc_len, sf = m.encoder(audio_signal=x, length=L)              # host modules, dynamic
    ref = m.decoder(encoder_output=enc, encoder_output_scaling_factor=sf)
    assert torch.equal(el.cpu(), enc_len.cpu().long())
    agree = (ref.argmax(-1) == tok).float().mean().item()
    assert agree >= 0.99, agree
    # --dynamic --percentile (inference.py:101,110-112): same route, ranges from the radix select
    qm.set_percentile(m, 99.9)
    assert m.dynamic_ready()
    lp, el, tok = m(processed_signal=x, processed_signal_length=L)
    assert isinstance(m._engine, dynamic.DynamicRunner) and m._engine.percentile == 99.9
    enc, enc_len, sf = m.encoder(audio_signal=x, length=L)
    ref = m.decoder(encoder_output=enc, encoder_output_scaling_factor=sf)
    agree = (ref.argmax(-1) == tok).float().mean().item()
    assert agree >= 0.99, agree
