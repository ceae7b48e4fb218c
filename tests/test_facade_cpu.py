"""Host-side drop-in surface (nemo.* façade, pack step) on CPU against the reference fixtures."""
import json
import os
import struct

import numpy as np
import pytest
import torch

import nemo.quantization.utils.quantize_model as qm
from nemo.collections.asr.metrics.wer import WER, word_error_rate
from nemo.collections.asr.models import EncDecCTCModel
from nemo.quantization.utils.quant_modules import QuantAct, QuantConv1d
from nemo.quantization.utils.quant_utils import batch_frexp, fixedpoint_mul
from qasr import configs, pack, synth, topology

torch.set_grad_enabled(False)


def _mini(name):
    return topology.mini_jasper() if 'minij' in name else topology.mini_quartznet()


def _calibrated(golden_dir, name):
    d = np.load(os.path.join(golden_dir, name + '.npz'))
    meta = json.loads(str(d['meta']))
    cfg = _mini(name)
    m = EncDecCTCModel(configs.model_config(cfg))
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_state_dict(cfg, meta['seed']).items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected
    # only the fork's extra buffers, the duplicated inner conv and the front-end buffers may be absent
    for k in missing:
        assert any(t in k for t in ('x_min', 'x_max', 'scaling_factor', '_integer', 'featurizer')), k
    m.eval()
    m.set_quant_bit(meta['wbit'], mode='weight')
    m.set_quant_bit(meta['abit'], mode='act')
    if meta['percentile'] is not None:
        qm.set_percentile(m, meta['percentile'])
    m.encoder.bn_folding()
    qm.calibrate(m)
    L = torch.tensor([meta['frames']] * meta['cal_batch'])
    for c in synth.make_calibration(meta['ncal'], meta['cal_batch'], cfg.feat_in, meta['frames'], meta['seed']):
        e, _, sf = m.encoder(audio_signal=torch.from_numpy(c), length=L)
        m.decoder(encoder_output=e, encoder_output_scaling_factor=sf)
    qm.evaluate(m)
    qm.set_dynamic(m, False)
    return d, meta, cfg, m


@pytest.mark.parametrize('name', ['net_miniq_w8a8', 'net_miniq_w8a8_pct', 'net_miniq_w6a6', 'net_minij_w8a8'])
def test_calibration_and_forward_match_reference(golden_dir, name):
    """Same CLI sequence as inference.py:105-138 on the façade; ranges and tokens vs the reference's run."""
    d, meta, cfg, m = _calibrated(golden_dir, name)
    _, sd, amin, amax, wb, ab = m.export_pack_inputs()
    assert (wb, ab) == (meta['wbit'], meta['abit'])
    np.testing.assert_array_equal(amin, d['act_min'])         # bit for bit, percentile calibration (torch.quantile) included
    np.testing.assert_array_equal(amax, d['act_max'])
    x = torch.from_numpy(synth.make_features(meta['batch'], cfg.feat_in, meta['frames'], meta['seed']))
    e, l, sf = m.encoder(audio_signal=x, length=torch.tensor(meta['lengths']))
    lp = m.decoder(encoder_output=e, encoder_output_scaling_factor=sf)
    assert np.array_equal(l.numpy(), d['enc_len'])
    assert np.array_equal(lp.argmax(-1).numpy(), d['tokens'])
    np.testing.assert_allclose(lp.numpy(), d['log_probs'], rtol=1e-4, atol=2e-5)
    # the evaluate-mode model refuses to run the integer path on CPU: no silent fallback
    assert m.engine_ready()
    with pytest.raises(RuntimeError):
        m(processed_signal=x, processed_signal_length=torch.tensor(meta['lengths']))


def test_pack_blob_layout(golden_dir):
    d, meta, cfg, m = _calibrated(golden_dir, 'net_miniq_w8a8')
    blob, pm = pack.pack_model(*m.export_pack_inputs())
    blob2, _ = pack.pack_model(cfg, synth.make_state_dict(cfg, meta['seed']), d['act_min'], d['act_max'], 8, 8)
    assert blob == blob2                          # live model and checkpoint+ranges pack to the same bytes
    magic, version, n_t, n_ops, feat, ncls, wb, ab, ndom, opsz = struct.unpack_from('<10I', blob, 0)
    assert magic == 0x52534151 and feat == cfg.feat_in and ncls == 29 and (wb, ab) == (8, 8)
    assert n_ops == pm['n_ops'] and len(pm['sites']) == meta['nconv']


def test_engine_path_fails_loudly_without_gpu(golden_dir):
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from qasr import engine
    d = np.load(os.path.join(golden_dir, 'net_miniq_w8a8.npz'))
    cfg = topology.mini_quartznet()
    blob, _ = pack.pack_model(cfg, synth.make_state_dict(cfg, 1), d['act_min'], d['act_max'], 8, 8)
    with pytest.raises(engine.QasrError):
        engine.Engine(blob, 0)


def test_quant_modes_and_switches():
    m = EncDecCTCModel(configs.model_config(topology.mini_quartznet()))
    acts = [a for a in m.modules() if isinstance(a, QuantAct)]
    convs = [c for c in m.modules() if isinstance(c, QuantConv1d)]
    assert all(a.quant_mode == 'symmetric' for a in acts)        # forced by the model (ctc_models.py:103-107)
    qm.calibrate(m)
    assert all(a.running_stat for a in acts) and all(c.fix_bn for c in convs)
    qm.train(m)
    assert all(a.running_stat for a in acts) and not any(c.fix_bn for c in convs)
    qm.evaluate(m)
    assert not any(a.running_stat for a in acts) and all(c.fix_bn for c in convs)
    qm.set_percentile(m, 99.9)
    assert all(a.percentile == 99.9 for a in acts)
    qm.set_dynamic(m, True)
    assert all(a.dynamic for a in acts)
    m.set_quant_bit(6, mode='act')
    asym = [mc for mc in m._masked_convs() if mc.asymmetric]
    assert asym and all(mc.act.activation_bit == 7 for mc in asym)
    assert all(b.res_act.activation_bit == 6 for b in m.encoder.encoder_layers)
    m.set_quant_mode('none')
    x = torch.randn(2, 16, 64)
    out = m.encoder(audio_signal=x, length=torch.tensor([64, 50]))
    assert out[0].shape == (2, 64, 32)


def test_dynamic_mode_cpu_plumbing():
    """BASELINE config 1: --dynamic, batch 1, CPU: per-batch min/max, no calibration."""
    m = EncDecCTCModel(configs.model_config(topology.mini_quartznet()))
    m.eval()
    m.encoder.bn_folding()
    qm.evaluate(m)
    qm.set_dynamic(m, True)
    assert not m.engine_ready()
    x = torch.from_numpy(synth.make_features(1, 16, 96, 3))
    lp, el, tok = m(processed_signal=x, processed_signal_length=torch.tensor([96]))
    assert lp.shape == (1, 48, 29) and tok.shape == (1, 48) and int(el[0]) == 48
    assert torch.isfinite(lp).all()


def test_batch_frexp_and_fixedpoint(golden_dir):
    d = np.load(os.path.join(golden_dir, 'ops.npz'))
    m, e = batch_frexp(torch.from_numpy(d['frexp_in']).view(1, -1, 1))
    assert np.array_equal(m.view(-1).numpy().astype(np.int64), d['frexp_m'])
    assert np.array_equal(e.view(-1).numpy(), d['frexp_e'])
    for tag, bits in (('requant8', 8), ('requant9', 9), ('requant_big', 8), ('res', 8), ('res_sat', 8)):
        sf = torch.from_numpy(d[tag + '_sf'])
        idn = torch.from_numpy(d[tag + '_id']) if tag + '_id' in d else None
        idsf = torch.from_numpy(d[tag + '_id_sf']) if idn is not None else None
        q = fixedpoint_mul.apply(torch.from_numpy(d[tag + '_x']), torch.from_numpy(d[tag + '_pre_sf']), bits,
                                 'symmetric', sf, idn, idsf)
        assert np.array_equal(q.numpy().astype(np.int32), d[tag + '_q']), tag


def test_quant_act_module_against_reference(golden_dir):
    d = np.load(os.path.join(golden_dir, 'ops.npz'))
    for tag, bits in (('first8', 8), ('first6', 6), ('requant7', 7)):
        a = QuantAct(bits, quant_mode='symmetric')
        a.fix()
        a.x_min = torch.tensor([d[tag + '_range'][0]])
        a.x_max = torch.tensor([d[tag + '_range'][1]])
        pre = torch.from_numpy(d[tag + '_pre_sf']) if tag + '_pre_sf' in d else None
        y, sf = a(torch.from_numpy(d[tag + '_x']), pre)
        assert np.array_equal(sf.reshape(-1).numpy(), d[tag + '_sf'])
        assert np.array_equal(y.numpy(), d[tag + '_y'])


def test_frontend_host_matches_reference(golden_dir):
    d = np.load(os.path.join(golden_dir, 'frontend.npz'))
    m = EncDecCTCModel(configs.model_config('QuartzNet15x5Base-En'))
    f = m.preprocessor.featurizer
    f.dither = 0.0
    assert np.allclose(f.fb[0].numpy(), d['fb']) and np.allclose(f.window.numpy(), d['window'])
    y, seq = m.preprocessor(input_signal=torch.from_numpy(d['audio']), length=torch.from_numpy(d['lens']))
    assert np.array_equal(seq.numpy(), d['seq_len'])
    np.testing.assert_allclose(y.numpy(), d['feats'], rtol=0, atol=2e-4)


def test_wer_known_answers_and_decode(golden_dir):
    for c in json.load(open(os.path.join(golden_dir, 'wer.json'))):
        assert abs(word_error_rate(c['hyp'], c['ref']) - c['wer']) < 1e-12
    with pytest.raises(ValueError):
        word_error_rate(['a'], ['a', 'b'])
    vocab = topology.VOCABULARY
    w = WER(vocabulary=vocab)
    blank = len(vocab)
    toks = torch.tensor([[3, 3, blank, 3, 1, 1, blank, blank, 20], [blank] * 9])
    assert w.ctc_decoder_predictions_tensor(toks) == ['ccat', '']
    # randomized cross-check of the metric against the function (mirrors test_asr_metrics.py:114-135)
    import random
    rnd = random.Random(0)
    for _ in range(32):
        s1 = ''.join(rnd.choice(''.join(vocab)) for _ in range(rnd.randint(1, 80)))
        s2 = ''.join(rnd.choice(''.join(vocab)) for _ in range(rnd.randint(1, 80)))
        if not s2.strip():
            continue
        w.reset()
        enc = lambda s: torch.tensor([[vocab.index(c) for c in s]])
        pred = []
        for c in s1:                                     # CTC-expand so repeats survive the collapse
            pred += [vocab.index(c), blank]
        w.update(torch.tensor([pred]), enc(s2), torch.tensor([len(s2)]))
        assert abs(float(w.compute()[0]) - word_error_rate([s1], [s2])) < 1e-6


def test_nemo_archive_round_trip(tmp_path):
    m = EncDecCTCModel.from_synthetic('MiniQuartzNet', seed=4) if 'MiniQuartzNet' in topology.MODELS else None
    path = str(tmp_path / 'mini.nemo')
    m.save_to(path)
    m2 = EncDecCTCModel.restore_from(path)
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    with pytest.raises(FileNotFoundError):
        EncDecCTCModel.from_pretrained('QuartzNet15x5Base-En')


def test_load_reads_reference_pickle_and_rejects_code(tmp_path):
    """`--load` must read what the reference's synthesize.py writes - pickle.dump([x.cpu() ...]) (synthesize.py:103-104) -
    and nothing that would run code from the file."""
    import pickle

    from qasr.calib_io import load_synthetic
    g = torch.Generator().manual_seed(0)
    batches = [torch.rand(4, 64, 50, generator=g) - 0.5, torch.rand(4, 64, 50, generator=g)[:, :, ::2]]   # one non-contiguous
    p = tmp_path / 'synthetic.pkl'
    with open(p, 'wb') as f:
        pickle.dump([x.cpu() for x in batches], f)
    got = load_synthetic(str(p))
    assert len(got) == 2 and all(torch.equal(a, b) for a, b in zip(got, batches))
    pt = tmp_path / 'synthetic.pt'
    torch.save(batches, pt)
    assert all(torch.equal(a, b) for a, b in zip(load_synthetic(str(pt)), batches))
    npz = tmp_path / 'synthetic.npz'
    np.savez(npz, a=batches[0].numpy(), b=batches[1].numpy())
    assert all(torch.equal(a, b) for a, b in zip(load_synthetic(str(npz)), batches))

    class Evil:
        def __reduce__(self):
            import os
            return (os.system, ('echo pwned > /dev/null',))
    bad = tmp_path / 'evil.pkl'
    with open(bad, 'wb') as f:
        pickle.dump([batches[0], Evil()], f)
    with pytest.raises(pickle.UnpicklingError, match='not allowed'):
        load_synthetic(str(bad))
    notlist = tmp_path / 'str.pkl'
    with open(notlist, 'wb') as f:
        pickle.dump(['text'], f)
    with pytest.raises(ValueError):
        load_synthetic(str(notlist))


def test_mel_filterbank_matches_slaney_reference():
    """FilterbankFeatures takes its matrix from librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) (features.py:281-283:
    Slaney scale, Slaney norm).  librosa is absent here; transformers' mel_filter_bank implements the same published
    construction and is what the survey's reference run used as the stand-in - qasr.melbank equals it bit for bit."""
    from transformers.audio_utils import mel_filter_bank

    from qasr.melbank import mel_filterbank
    for n_mels, fmax in ((64, 8000.0), (80, 8000.0), (64, 7600.0)):
        want = mel_filter_bank(257, n_mels, 0.0, fmax, 16000, norm='slaney', mel_scale='slaney').T.astype(np.float32)
        got = mel_filterbank(16000, 512, n_mels, 0, fmax)
        assert got.shape == want.shape and np.array_equal(got, want), np.abs(got - want).max()


def test_engine_frontend_guard_rejects_other_featurizers():
    """The HIP front-end is built for the 20 ms hann / 10 ms hop / n_fft 512 preprocessor; any other featurizer
    configuration must route to the host module (ctc_models._frontend_hip_supported), never to the kernels."""
    base = configs.model_config('MiniQuartzNet')
    assert EncDecCTCModel(base)._frontend_hip_supported()
    for patch in (dict(window_size=0.025), dict(normalize='all_features'), dict(window_stride=0.02), dict(n_fft=1024),
                  dict(log_zero_guard_value=1e-5), dict(mag_power=1.0)):
        cfg = json.loads(json.dumps(base))
        cfg['preprocessor'].update(patch)
        assert not EncDecCTCModel(cfg)._frontend_hip_supported(), patch


def test_en_char_parser_normalisation():
    """ENCharParser / cleaners.clean_text (parsers.py:101-145, cleaners.py:93-215): lower-casing, whitespace, numbers to
    words, abbreviations, + & % replaced by words, other punctuation to spaces - expected strings worked out by hand from
    the reference's regexes (inflect / unidecode are absent: their restatement is documented as unpinned)."""
    from nemo.collections.asr.parts import parsers
    vocab = list(" abcdefghijklmnopqrstuvwxyz'")
    p = parsers.make_parser(labels=vocab, name='en', unk_id=-1, blank_id=-1, do_normalize=True)
    cases = {
        "HELLO   World": "hello world",
        "Mr. Smith & Co. pay 20% + tax": "mister smith and company pay twenty percent plus tax",
        "it's 1234 now": "it's one thousand two hundred and thirty four now",
        "the 21st of Jan. at 7:30pm": "the twenty first of january at seven thirty pm",
        "café, naïve -- déjà vu!": "cafe naive deja vu",
        "pi is 3.14": "pi is three point one four",
    }
    for raw, want in cases.items():
        assert p._normalize(raw) == want, (raw, p._normalize(raw))
    ids = p("Don't stop")
    assert ids == [vocab.index(c) for c in "don't stop"]
    base = parsers.make_parser(labels=vocab, name='base', unk_id=-1, blank_id=-1)
    assert base("A-b") == [vocab.index('a'), vocab.index('b')]          # unknown '-' -> unk_id == blank_id -> dropped


def test_distill_matches_reference_fixture(golden_dir):
    """tests/golden/distill.npz = the reference's own distill_data.get_synthetic_data / _kl_loss (imported unmodified by
    gen_golden.py) on a float MiniQuartzNet: 3 Adam iterations on 2 fixed start batches.  This repo's get_synthetic_data
    from the same start must give the same per-iteration losses and the same refined batches (rtol 1e-5), _kl_loss the
    same known answers."""
    import json

    from nemo.quantization.utils import distill_data
    d = np.load(os.path.join(golden_dir, 'distill.npz'))
    meta = json.loads(str(d['meta']))
    m = EncDecCTCModel.from_synthetic(meta['model'], seed=meta['seed'])
    m.set_quant_mode('none')
    assert len(m.encoder.convs_before_bn) == meta['n_hooks']
    hist = []
    out = distill_data.get_synthetic_data(m.encoder, m.decoder, batch_size=meta['batch'], dim=d['start'].shape[2], seqlen=meta['frames'],
                                          train_iter=meta['train_iter'], num_batch=meta['num_batch'], lr=meta['lr'], verbose=False,
                                          history=hist, init=[torch.from_numpy(t) for t in d['start']])
    np.testing.assert_allclose(np.array(hist), d['losses'], rtol=1e-5)
    assert d['losses'][2] < d['losses'][0] and d['losses'][5] < d['losses'][3]
    got = np.stack([t.numpy() for t in out])
    assert np.abs(d['refined'] - d['start']).max() > 0.1          # the optimiser moved the data ...
    np.testing.assert_allclose(got, d['refined'], rtol=1e-5, atol=1e-6)   # ... to where the reference moved it
    one = torch.ones(5)
    kl = [float(distill_data._kl_loss(one * 0.3, one * 2, one * 0.3, one * 2)), float(distill_data._kl_loss(one * 0.3, one * 2, one * 0.5, one * 1.5)),
          float(distill_data._kl_loss(torch.tensor([0.1, -0.2]), torch.tensor([1.0, 0.5]), torch.tensor([0.0, 0.3]), torch.tensor([2.0, 0.25])))]
    np.testing.assert_allclose(kl, d['kl_known'], rtol=1e-6, atol=1e-7)
    assert all(p.requires_grad for p in m.encoder.parameters())   # the weights' requires_grad is restored
    assert all(len(c._forward_hooks) == 0 for c, _ in m.encoder.convs_before_bn)


def test_synthesize_zero_shot_calibration_data(tmp_path):
    """synthesize.py / distill_data.get_synthetic_data (distill_data.py:71-162): the BN-statistics loss of the float mini
    model goes down under Adam on the input, and the dumped pickle is what `inference.py --load` reads."""
    import importlib.util
    import sys

    from nemo.quantization.utils import distill_data
    from qasr.calib_io import load_synthetic
    torch.manual_seed(0)
    m = EncDecCTCModel.from_synthetic('MiniQuartzNet', seed=3)
    m.set_quant_mode('none')
    hist = []
    out = distill_data.get_synthetic_data(m.encoder, m.decoder, batch_size=2, dim=16, seqlen=48, train_iter=12, num_batch=2,
                                          lr=0.05, seed=1, verbose=False, history=hist)
    assert len(out) == 2 and out[0].shape == (2, 16, 48) and not out[0].requires_grad
    assert hist[11] < hist[0] and hist[23] < hist[12], hist      # the loss of each batch decreases
    assert len(m.encoder.convs_before_bn) > 0 and all(len(c._forward_hooks) == 0 for c, _ in m.encoder.convs_before_bn)
    # _kl_loss is 0 for identical Gaussians and positive otherwise
    one = torch.ones(4)
    assert float(distill_data._kl_loss(one * 0.3, one * 2, one * 0.3, one * 2)) == 0.0
    assert float(distill_data._kl_loss(one * 0.3, one * 2, one * 0.5, one * 1.5)) > 0
    # the CLI end to end on the CPU
    spec = importlib.util.spec_from_file_location(
        'synthesize_cli', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'q-asr_amd', 'examples',
                                       'asr', 'quantization', 'synthesize.py'))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    path = cli.main(['--asr_model', 'MiniQuartzNet', '--synthetic_model', '--cpu', '--num_batch', '2', '--batch_size', '2',
                     '--seqlen', '32', '--train_iter', '3', '--dump_path', str(tmp_path), '--seed', '5'])
    assert os.path.basename(path) == 'syn_nb2_iter3_lr0.010.pkl'
    data = load_synthetic(path)
    assert len(data) == 2 and data[0].shape == (2, 16, 32)


def test_w6_blob_is_sub_byte_packed(golden_dir):
    """BASELINE config 3: 6-bit weights are stored 4 codes per 3 bytes (pack6, element order unchanged) and the depthwise
    tap rows are derived on load, so the w6a6 blob of QuartzNet15x5 is <= 0.78x the w8a8 one; unpack6 (the host
    restatement of the device expansion) returns the exact integers."""
    from qasr.pack import F_W6PACK, Packer, pack6, unpack6
    rng = np.random.default_rng(1)
    a = rng.integers(-32, 32, 4096).astype(np.int8)
    assert pack6(a).size == a.size * 3 // 4 and np.array_equal(unpack6(pack6(a)), a)
    with pytest.raises(AssertionError):
        pack6(np.array([40, 0, 0, 0], np.int8))
    sizes = {}
    for name in ('net_quartznet_w8a8', 'net_quartznet_w6a6'):
        d = np.load(os.path.join(golden_dir, name + '.npz'))
        meta = json.loads(str(d['meta']))
        cfg = topology.quartznet15x5()
        p = Packer(cfg, synth.make_state_dict(cfg, meta['seed']), d['act_min'], d['act_max'], meta['wbit'], meta['abit'])
        blob, _ = p.pack()
        sizes[name] = len(blob)
        packed = [bool(o['flags'] & F_W6PACK) for o in p.final_ops if o['kind'] in (1, 2, 3)]
        assert all(packed) if meta['wbit'] <= 6 else not any(packed)
    assert sizes['net_quartznet_w6a6'] <= 0.78 * sizes['net_quartznet_w8a8'], sizes


def test_dynamic_runner_static_halves_on_cpu():
    """qasr.dynamic._Conv (the static half of a QuantConv1d in the dynamic device path): weight integers in the three kernel
    layouts (depthwise [C][kpad4], 1x1 fragment order, dense [cout_pad][K][cin_pad]), weight scales, 128 * sum(W)."""
    import torch
    from qasr import dynamic, quant_math as Q
    from qasr.pack import fragment_order
    from qasr.topology import ConvSite
    g = torch.Generator().manual_seed(0)
    # depthwise k = 11
    w = torch.randn(24, 1, 11, generator=g)
    s = ConvSite('x.0', None, 0, 'dw', 24, 24, 11, 1, 1, 5, 24, True, False)
    c = dynamic._Conv(s, w, None, 8, 'cpu')
    wi, sw = Q.weight_integers(w, 8)
    assert c.kpad == 12 and c.w.shape == (24, 12) and not c.dense
    assert torch.equal(c.w[:, :11].long(), wi[:, 0].long()) and int(c.w[:, 11].abs().sum()) == 0
    assert torch.equal(c.s_w, sw) and torch.equal(c.wsum128.long(), 128 * wi.long().reshape(24, -1).sum(1))
    # 1x1 40 -> 72 with bias
    w = torch.randn(72, 40, 1, generator=g)
    b = torch.randn(72, generator=g)
    c = dynamic._Conv(ConvSite('x.1', None, 0, 'pw', 40, 72, 1, 1, 1, 0, 1, False, True), w, b, 8, 'cpu')
    wi, _ = Q.weight_integers(w, 8)
    ref = np.zeros((128, 128), np.int8)
    ref[:72, :40] = wi[:, :, 0].numpy().astype(np.int8)
    assert c.cout_pad == 128 and c.cin_pad == 128 and not c.dense
    assert np.array_equal(c.w.numpy(), fragment_order(ref)) and torch.equal(c.bprime, b)
    # dense k = 5, stride 2
    w = torch.randn(32, 16, 5, generator=g)
    c = dynamic._Conv(ConvSite('x.2', None, 0, 'dense', 16, 32, 5, 2, 1, 2, 1, False, True), w, None, 6, 'cpu')
    wi, _ = Q.weight_integers(w, 6)
    assert c.dense and c.w.shape == (128, 5, 128) and int(wi.abs().max()) <= 31
    assert torch.equal(c.w[:32, :, :16].long(), wi.permute(0, 2, 1).long()) and int(c.w[32:].abs().sum()) == 0


def test_transcribe_and_trim_silence(tmp_path):
    """EncDecCTCModel.transcribe (ctc_models.py:148-212): transcripts in input order, per-file log-probs with `logprobs=True`,
    dither / pad_to / training mode restored, {} for no files; and the `trim_silence` step of its data loader (segment.py:60-61:
    librosa.effects.trim at 60 dB, restated - librosa is not importable, parity unpinned) on a signal with silent ends."""
    import wave
    from nemo.collections.asr.data.audio_to_text import trim_silence
    rng = np.random.default_rng(3)
    x = np.zeros(48000, np.float32)
    x[10000:30000] = 0.1 * rng.standard_normal(20000)
    y = trim_silence(x)
    assert 20000 <= y.size <= 20000 + 2 * 2048 and np.abs(y).max() > 0
    assert trim_silence(np.zeros(1000, np.float32)).size == 0
    loud = 0.1 * rng.standard_normal(5000).astype(np.float32)
    assert np.array_equal(trim_silence(loud), loud)

    m = EncDecCTCModel.from_synthetic('MiniQuartzNet', seed=4)
    m.set_quant_mode('none')                                   # float modules on the CPU (inference.py --no_quant)
    m.train()
    f = m.preprocessor.featurizer
    f.dither, f.pad_to = 1e-5, 16
    paths = []
    for i, n in enumerate((16000, 9000, 12345)):
        a = (0.1 * rng.standard_normal(n)).astype(np.float32)
        p = str(tmp_path / f't{i}.wav')
        with wave.open(p, 'wb') as w:
            w.setnchannels(1)
            w.setsampwidth(2)
            w.setframerate(16000)
            w.writeframes((np.clip(a, -1, 1) * 32767).astype('<i2').tobytes())
        paths.append(p)
    assert m.transcribe([]) == {} and m.transcribe(None) == {}
    hyps = m.transcribe(paths, batch_size=2)
    assert len(hyps) == 3 and all(isinstance(h, str) for h in hyps)
    assert m.training and f.dither == 1e-5 and f.pad_to == 16    # everything restored
    lps = m.transcribe(paths, batch_size=8, logprobs=True)
    assert [tuple(t.shape)[1] for t in lps] == [29, 29, 29] and lps[0].shape[0] > lps[1].shape[0]
    # File 0 is the longest of every batch it sat in: its row carries no pad frames and its STFT sees its own reflected end in both
    # calls, so its transcript and its log-probs tell the same story.  (A SHORTER file's last frames and pad frames depend on the
    # batch it sits in - the greedy decoder walks the whole padded row like the reference's, wer.py:117-136, and the centred STFT
    # pads the batch tensor, not the utterance - so no such equality holds for files 1 and 2, here or in the reference.)
    from nemo.collections.asr.metrics.wer import WER
    again = WER(vocabulary=m.decoder.vocabulary).ctc_decoder_predictions_tensor(lps[0].argmax(-1).unsqueeze(0))
    # (log-probs are cut at the encoded length, the transcript walks every frame the encoder emitted: at most a trailing token more)
    assert hyps[0].startswith(again[0]) and len(hyps[0]) - len(again[0]) <= 2
