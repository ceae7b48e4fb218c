"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI
(qasr.engine -> libqasr_hip.so), against the CPU oracle and the committed golden fixtures.
Bar: bit-exact for every integer (accumulators, requantised activations, tokens);
float logits within 1e-5 relative of the reference (its own x_int noise)."""
import json
import os

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import int_oracle as O  # noqa: E402
from qasr import pack, synth, topology  # noqa: E402


@pytest.fixture(scope='module')
def eng():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from qasr import engine
    engine.load_library()          # raises if the extension was not built: no silent fallback
    return engine


def _cfg(name):
    if 'miniq' in name:
        return topology.mini_quartznet()
    if 'minij' in name:
        return topology.mini_jasper()
    if 'quartznet' in name:
        return topology.quartznet15x5()
    return topology.jasper10x5dr()


def _load(golden_dir, name):
    d = np.load(os.path.join(golden_dir, name + '.npz'))
    return d, json.loads(str(d['meta']))


# ---------------------------------------------------------------------------------- operators
@pytest.mark.parametrize('cin,cout,T,unsigned', [(64, 128, 64, False), (256, 256, 250, False), (512, 512, 250, True),
                                                 (48, 29, 33, False), (1024, 29, 250, False), (16, 32, 7, True)])
def test_pw_conv_acc(eng, cin, cout, T, unsigned):
    rng = np.random.default_rng(cin * 1000 + cout)
    x = rng.integers(0 if unsigned else -128, 256 if unsigned else 128, (3, cin, T))
    w = rng.integers(-127, 127, (cout, cin, 1))
    b = rng.integers(-50000, 50000, cout)
    want = O.conv1d_int(x.astype(np.int64), w.astype(np.int64), b, 1, 0, 1, 1)
    xt = torch.from_numpy(x.astype(np.uint8 if unsigned else np.int8)).cuda()
    got = eng.pw_conv_acc(xt, torch.from_numpy(w[:, :, 0].astype(np.int8)), torch.from_numpy(b.astype(np.int32)),
                          x_unsigned=unsigned).cpu().numpy()
    assert np.array_equal(got, want)


def test_pw_mfma_layout_asymmetric(eng):
    """A = I-style check with asymmetric operands: catches swapped rows/cols or a k permutation mismatch."""
    cin = cout = 128
    T = 64
    x = np.zeros((1, cin, T), np.int64)
    for t in range(T):
        x[0, (3 * t + 1) % cin, t] = t + 1                  # one distinct non-zero channel per time step
    w = (np.arange(cout)[:, None] * 3 + np.arange(cin)[None, :] * 5) % 251 - 125
    want = O.conv1d_int(x, w[:, :, None].astype(np.int64), None, 1, 0, 1, 1)
    got = eng.pw_conv_acc(torch.from_numpy(x.astype(np.int8)).cuda(), torch.from_numpy(w.astype(np.int8))).cpu().numpy()
    assert np.array_equal(got, want)


@pytest.mark.parametrize('K,stride,dil', [(33, 1, 1), (39, 1, 1), (51, 1, 1), (63, 1, 1), (75, 1, 1), (11, 1, 1),
                                          (13, 1, 1), (15, 1, 2), (33, 2, 1), (87, 1, 2), (5, 1, 1)])
@pytest.mark.parametrize('T', [250, 64, 301])
def test_dw_conv_acc(eng, K, stride, dil, T):
    rng = np.random.default_rng(K * 7 + T)
    C = 12
    pad = topology.same_padding(K, stride, dil)
    x = rng.integers(-128, 128, (2, C, T))
    w = rng.integers(-127, 127, (C, 1, K))
    want = O.conv1d_int(x.astype(np.int64), w.astype(np.int64), None, stride, pad, dil, C)
    got = eng.dw_conv_acc(torch.from_numpy(x.astype(np.int8)).cuda(), torch.from_numpy(w[:, 0].astype(np.int8)),
                          stride, dil, pad).cpu().numpy()
    assert got.shape == want.shape
    assert np.array_equal(got, want)


@pytest.mark.parametrize('tag,bits', [('requant8', 8), ('requant9', 9), ('requant6', 6), ('requant7', 7)])
def test_requant_golden(eng, golden_dir, tag, bits):
    """fixedpoint_mul against the reference's own outputs (ops.npz)."""
    d = np.load(os.path.join(golden_dir, 'ops.npz'))
    pre = d[tag + '_pre_sf'].reshape(-1)
    acc = np.rint(d[tag + '_x'] / pre.reshape(1, -1, 1)).astype(np.int32)
    from qasr import quant_math as Q
    M = Q.requant_multiplier(torch.from_numpy(pre), torch.from_numpy(d[tag + '_sf']))
    lo, hi = Q.qrange(bits)
    got = eng.requant(torch.from_numpy(acc).cuda(), M, lo, hi).cpu().numpy()
    if bits == 9:                   # 9-bit 'asymmetric' activations are >= 0 and stored as u8
        got = got.view(np.uint8)
    assert np.array_equal(got, d[tag + '_q'])


def test_requant_exact_z_golden(eng, golden_dir):
    """|acc| up to 2^24: z must go through the float32 round trip (QASR_F_EXACT_Z)."""
    d = np.load(os.path.join(golden_dir, 'ops.npz'))
    pre = d['requant_big_pre_sf'].reshape(-1)
    from qasr import quant_math as Q
    M = Q.requant_multiplier(torch.from_numpy(pre), torch.from_numpy(d['requant_big_sf']))
    acc = torch.from_numpy(d['requant_big_acc']).cuda()
    got = eng.requant(acc, M, -128, 127, sb=torch.from_numpy(pre), exact_z=True).cpu().numpy()
    assert np.array_equal(got, d['requant_big_q'])
    # without QASR_F_EXACT_Z the kernel requantises the accumulator itself: clamp(rint(f64(acc) * M)) - which is NOT
    # fixedpoint_mul's result beyond 2^22 (the packer sets the flag there); pinned here against numpy
    fast = eng.requant(acc, M, -128, 127).cpu().numpy()
    want_fast = np.clip(np.rint(d['requant_big_acc'].astype(np.float64) * M.numpy().reshape(1, -1, 1)), -128, 127)
    assert np.array_equal(fast, want_fast.astype(np.int8))


# ---------------------------------------------------------------------------------- networks
def _run_engine(eng, golden_dir, name, debug=True, wide_tiles=False):
    d, meta = _load(golden_dir, name)
    cfg = _cfg(name)
    sd = synth.make_state_dict(cfg, meta['seed'])
    blob, pm = pack.pack_model(cfg, sd, d['act_min'], d['act_max'], meta['wbit'], meta['abit'])
    e = eng.Engine(blob, 0, debug=debug, wide_tiles=wide_tiles)
    x = synth.make_features(meta['batch'], cfg.feat_in, meta['frames'], meta['seed'])
    logp, tokens, enc_len = e.forward(torch.from_numpy(x).cuda(), torch.tensor(meta['lengths']))
    torch.cuda.synchronize()
    return d, meta, cfg, pm, e, logp.cpu().numpy(), tokens.cpu().numpy(), enc_len.cpu().numpy()


def _site_dims(cfg):
    dims = []
    for sites in topology.conv_plan(cfg):
        for s in sites:
            dims.append(s.cout)
    dims.append(cfg.num_classes + 1)
    return dims


KERNEL_FAMILIES = [dict(), dict(wide_tiles=True)]
KERNEL_IDS = ['k_sep32', 'k_sep64']


@pytest.mark.parametrize('family', KERNEL_FAMILIES, ids=KERNEL_IDS)
@pytest.mark.parametrize('name', ['net_miniq_w8a8', 'net_miniq_w8a8_pct', 'net_miniq_w6a6', 'net_minij_w8a8'])
def test_mini_net_every_accumulator(eng, golden_dir, name, family):
    d, meta, cfg, pm, e, logp, tokens, enc_len = _run_engine(eng, golden_dir, name, **family)
    couts = _site_dims(cfg)
    for i, (op, pane) in enumerate(pm['sites']):
        want = d[f'acc_{i}']                               # rint(conv_int) of the reference itself
        got = e.read_acc(op, pane, couts[i], want.shape[2])
        assert np.array_equal(got, want), f'conv {i} (op {op}, pane {pane})'
    assert np.array_equal(tokens, d['tokens'])
    assert np.array_equal(enc_len, d['enc_len'])
    np.testing.assert_allclose(logp, d['log_probs'], rtol=1e-4, atol=2e-5)
    e.close()


@pytest.mark.parametrize('family', KERNEL_FAMILIES, ids=KERNEL_IDS)
@pytest.mark.parametrize('name', ['net_quartznet_w8a8', 'net_quartznet_w6a6', 'net_jasper_w8a8'])
def test_full_net_checksums(eng, golden_dir, name, family):
    d, meta, cfg, pm, e, logp, tokens, enc_len = _run_engine(eng, golden_dir, name, **family)
    couts = _site_dims(cfg)
    T_out = d['tokens'].shape[1]
    for i, (op, pane) in enumerate(pm['sites']):
        got = e.read_acc(op, pane, couts[i], T_out)
        assert np.array_equal(O.checksum(got), d['conv_checksums'][i][:2]), f'conv {i} (op {op}, pane {pane})'
    assert np.array_equal(tokens, d['tokens'])
    np.testing.assert_allclose(logp, d['log_probs'], rtol=1e-4, atol=5e-5)
    e.close()


def test_release_engine_matches_debug_engine(eng, golden_dir):
    """Buffer reuse (non-debug arena) must not change results."""
    d, meta, cfg, pm, e, logp, tokens, enc_len = _run_engine(eng, golden_dir, 'net_quartznet_w8a8', debug=False)
    assert np.array_equal(tokens, d['tokens'])
    e.close()


def test_bench_size_properties(eng, golden_dir):
    """BASELINE config 2 (QuartzNet15x5 w8a8, B=32, T=500): size-independent properties.
    (a) each utterance's result is independent of its batch neighbours and of batch padding;
    (b) re-running is bit-reproducible; (c) a short utterance agrees with the oracle."""
    d, meta = _load(golden_dir, 'net_quartznet_w8a8')
    cfg = topology.quartznet15x5()
    sd = synth.make_state_dict(cfg, meta['seed'])
    blob, _ = pack.pack_model(cfg, sd, d['act_min'], d['act_max'], 8, 8)
    e = eng.Engine(blob, 0, debug=False)
    B, T = 32, 500
    x = torch.from_numpy(synth.make_features(B, 64, T, 11)).cuda()
    lens = torch.tensor([T - 7 * (i % 9) for i in range(B)])
    lp1, tk1, el1 = e.forward(x, lens)
    tk1, el1, lp1 = tk1.cpu().numpy(), el1.cpu().numpy(), lp1.cpu().numpy()
    lp2, tk2, _ = e.forward(x, lens)
    assert np.array_equal(tk1, tk2.cpu().numpy()) and np.array_equal(lp1, lp2.cpu().numpy())
    ew = eng.Engine(blob, 0, debug=False, wide_tiles=True)     # 64-frame tiles: identical integers
    lpw, tkw, _ = ew.forward(x, lens)
    assert np.array_equal(tk1, tkw.cpu().numpy()) and np.array_equal(lp1, lpw.cpu().numpy())
    ew.close()
    assert np.array_equal(el1, (lens.numpy() + 1) // 2)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(0))
    _, tk3, _ = e.forward(x[perm].contiguous(), lens[perm])
    assert np.array_equal(tk3.cpu().numpy(), tk1[perm.numpy()])
    # utterance 5 alone, cropped to its own length: identical tokens on its valid frames
    i, L = 5, int(lens[5])
    lp4, tk4, el4 = e.forward(x[i:i + 1, :, :L].contiguous(), lens[i:i + 1])
    n = int(el4[0])
    assert np.array_equal(tk4.cpu().numpy()[0, :n], tk1[i, :n])
    np.testing.assert_array_equal(lp4.cpu().numpy()[0, :n], lp1[i, :n])
    # (c) a short utterance against the ORACLE (OracleNet on the CPU): the first 150 frames of utterance 5 as an utterance
    # of their own - tokens, encoded length, log-probs
    xs = x[i:i + 1, :, :150].contiguous()
    lp5, tk5, el5 = e.forward(xs, torch.tensor([150]))
    net = O.OracleNet(topology.conv_plan(cfg), cfg, sd, d['act_min'], d['act_max'], 8, 8)
    want = net.forward(xs.cpu().numpy(), [150])
    n = int(want['enc_len'][0])
    assert int(el5[0]) == n == 75
    assert np.array_equal(tk5.cpu().numpy()[0, :n], want['tokens'][0, :n])
    np.testing.assert_allclose(lp5.cpu().numpy()[0, :n], want['log_probs'][0, :n], rtol=1e-4, atol=5e-5)
    e.close()


def test_fused_stem_and_decoder_match_unfused(eng, golden_dir):
    """k_stem (lengths + first-layer quantisation + strided depthwise + 1x1 conv of block 0) and k_dec (decoder conv +
    log-softmax + argmax + lengths) against the four- / two-launch forms they replace: identical log-probs, tokens and
    lengths at bench size with ragged lengths, and the labels say which form ran."""
    d, meta = _load(golden_dir, 'net_quartznet_w8a8')
    cfg = topology.quartznet15x5()
    sd = synth.make_state_dict(cfg, meta['seed'])
    blob, _ = pack.pack_model(cfg, sd, d['act_min'], d['act_max'], 8, 8)
    cases = [(32, 500, [500 - 13 * (i % 11) for i in range(32)]),       # bench size, ragged
             (3, 501, [501, 2, 77]), (2, 33, [33, 1]), (1, 129, [100])]    # odd frame counts, 1- and 2-frame utterances
    outs = []
    for stem, dec in ((1, 1), (0, 0)):
        e = eng.Engine(blob, 0, fuse_stem=bool(stem), fuse_decoder=bool(dec))      # qasr_engine_opts, not the environment
        res = []
        for B, T, ls in cases:
            x = torch.from_numpy(synth.make_features(B, 64, T, 13 + T)).cuda()
            lp, tk, el = e.forward(x, torch.tensor(ls))
            labels = e.op_labels()
            assert ('k_stem' in labels) == bool(stem) and ('k_dec' in labels) == bool(dec), labels[:4] + labels[-3:]
            assert ('k_quant_in' in labels) != bool(stem) and ('k_logsoftmax' in labels) != bool(dec)
            res.append((lp.cpu().numpy(), tk.cpu().numpy(), el.cpu().numpy()))
        outs.append(res)
        e.close()
    for (B, T, ls), ra, rb in zip(cases, outs[0], outs[1]):
        for a, b in zip(ra, rb):
            assert np.array_equal(a, b), (B, T)
        assert np.array_equal(ra[2], (np.array(ls) + 1) // 2)


def test_long_utterances_tile_sizes_agree(eng, golden_dir):
    """30-second utterances (2999 mel frames -> 1500 encoder frames: 47 / 24 / 12 time tiles per utterance, the last
    one partial) with ragged lengths: the three tile sizes and both kernel generations produce identical log-probs,
    tokens and lengths, and re-running is bit-reproducible."""
    d, meta = _load(golden_dir, 'net_quartznet_w8a8')
    cfg = topology.quartznet15x5()
    sd = synth.make_state_dict(cfg, meta['seed'])
    blob, _ = pack.pack_model(cfg, sd, d['act_min'], d['act_max'], 8, 8)
    B, T = 3, 2999
    x = torch.from_numpy(synth.make_features(B, 64, T, 17)).cuda()
    lens = torch.tensor([2999, 1777, 64])
    ref = None
    for fam in (dict(gen=2), dict(gen=2, wide_tiles=True, tile128=0), dict(gen=2, wide_tiles=True, tile128=1), dict(gen=1)):
        fam = dict(fam)
        e = _engine_gen(eng, blob, fam.pop('gen'), **fam)
        lp, tk, el = e.forward(x, lens)
        lp2, tk2, _ = e.forward(x, lens)
        out = (lp.cpu().numpy(), tk.cpu().numpy(), el.cpu().numpy())
        assert np.array_equal(out[1], tk2.cpu().numpy()) and np.array_equal(out[0], lp2.cpu().numpy())
        e.close()
        assert np.array_equal(out[2], (lens.numpy() + 1) // 2)
        if ref is None:
            ref = out
        else:
            for a, b_ in zip(ref, out):
                assert np.array_equal(a, b_), fam


def test_jasper_long_utterances_dense2_against_oracle_and_k_sep(eng, golden_dir):
    """k_dense2 beyond one 256-frame work-group (grid.z > 1, the last time tile partial), frame counts that are not a
    multiple of 64, ragged lengths: (a) the mini Jasper (dense residual, panes) against OracleNet on 700 / 333-frame
    utterances - tokens, lengths, log-probs; (b) Jasper10x5dr at 2 x 1101 frames (551 encoder frames: three time tiles for the
    plain layers, five for the block ends): k_dense2 (kernel generation 2) and round 1's k_sep (generation 1) give identical
    log-probs, tokens and lengths, and the labels say which ran."""
    d, meta = _load(golden_dir, 'net_minij_w8a8')
    cfg = topology.mini_jasper()
    sd = synth.make_state_dict(cfg, meta['seed'])
    blob, _ = pack.pack_model(cfg, sd, d['act_min'], d['act_max'], 8, 8)
    net = O.OracleNet(topology.conv_plan(cfg), cfg, sd, d['act_min'], d['act_max'], 8, 8)
    for T, ls in ((700, [700, 333, 5]), (333, [333, 130])):
        x = synth.make_features(len(ls), cfg.feat_in, T, 29 + T)
        want = net.forward(x, ls)
        e = eng.Engine(blob, 0)
        lp, tk, el = e.forward(torch.from_numpy(x).cuda(), torch.tensor(ls))
        assert any(l.startswith('k_dense2<') and l.endswith('true>') for l in e.op_labels()) and \
            any(l.startswith('k_dense2<') and l.endswith('false>') for l in e.op_labels()), e.op_labels()
        assert np.array_equal(el.cpu().numpy(), want['enc_len'])
        for b in range(len(ls)):
            n = int(want['enc_len'][b])
            assert np.array_equal(tk.cpu().numpy()[b, :n], want['tokens'][b, :n]), (T, b)
            np.testing.assert_allclose(lp.cpu().numpy()[b, :n], want['log_probs'][b, :n], rtol=1e-4, atol=5e-5)
        e.close()
    d, meta = _load(golden_dir, 'net_jasper_w8a8')
    cfg = topology.jasper10x5dr()
    sd = synth.make_state_dict(cfg, meta['seed'])
    blob, _ = pack.pack_model(cfg, sd, d['act_min'], d['act_max'], 8, 8)
    x = torch.from_numpy(synth.make_features(2, 64, 1101, 31)).cuda()
    lens = torch.tensor([1101, 642])
    outs = []
    for gen in (2, 1):
        e = eng.Engine(blob, 0, sep_gen=gen, tile=128)
        lp, tk, el = e.forward(x, lens)
        labels = e.op_labels()
        assert (sum(l.startswith('k_dense2<') for l in labels) >= 50) == (gen == 2), labels
        outs.append((lp.cpu().numpy(), tk.cpu().numpy(), el.cpu().numpy()))
        e.close()
    for a, b_ in zip(*outs):
        assert np.array_equal(a, b_)


def test_steps_in_flight_match_serial(eng, golden_dir):
    """bench.py keeps several steps in flight (one engine + HIP stream each): kernels of different steps interleave on
    the CUs, which exposes any tensor whose arena slot is recycled before its last reader (the halo reads of a fused
    depthwise stage happen in the FOLLOWING launch).  Every concurrent result must equal the serial one."""
    d, meta = _load(golden_dir, 'net_quartznet_w8a8')
    cfg = topology.quartznet15x5()
    sd = synth.make_state_dict(cfg, meta['seed'])
    blob, _ = pack.pack_model(cfg, sd, d['act_min'], d['act_max'], 8, 8)
    S = 4
    engs = [eng.Engine(blob, 0, wide_tiles=bool(k & 1)) for k in range(S)]   # both tile sizes in flight together
    streams = [torch.cuda.Stream() for _ in range(S)]
    B, T = 32, 512
    x = torch.from_numpy(synth.make_features(B, 64, T, 1)).cuda()
    lens = torch.full((B,), 500)
    lp0, tk0, _ = engs[0].forward(x, lens)
    torch.cuda.synchronize()
    lp0, tk0 = lp0.cpu().numpy(), tk0.cpu().numpy()
    outs = []
    for i in range(24):
        with torch.cuda.stream(streams[i % S]):
            lp, tk, _ = engs[i % S].forward(x, lens)
            outs.append((lp, tk))
    torch.cuda.synchronize()
    for i, (lp, tk) in enumerate(outs):
        assert np.array_equal(tk.cpu().numpy(), tk0), f'step {i} (engine {i % S}) tokens differ from the serial run'
        assert np.array_equal(lp.cpu().numpy(), lp0), f'step {i} (engine {i % S}) log-probs differ from the serial run'
    for e in engs:
        e.close()


def test_device_quantile_matches_torch_cpu(eng):
    """qasr_quantile2 (radix select on device) against torch.quantile on the CPU and the oracle restatement: bit-exact,
    including ties, negative values, tiny inputs and the bench-size activation tensor (32 x 1024 x 256)."""
    from oracle import int_oracle as O
    g = torch.Generator().manual_seed(3)
    cases = [torch.randn(100003, generator=g), torch.randn(32, 256, 250, generator=g) * 7, torch.randn(4, generator=g),
             torch.tensor([3.0, -1.0]), torch.full((1000,), 2.5), torch.round(torch.randn(50000, generator=g) * 4),
             torch.relu(torch.randn(32, 1024, 256, generator=g))]
    for x in cases:
        for p in (99.996, 99.9, 100.0, 50.0):
            ql, qh = 1 - p / 100, p / 100
            got = eng.quantile2(x.cuda(), ql, qh).cpu().numpy()
            flat = x.reshape(-1)
            want = [torch.quantile(flat, torch.tensor(q, dtype=torch.float32)).item() if flat.numel() <= 16_000_000
                    else float(O.quantile_f32(flat.numpy(), np.float32(q))) for q in (ql, qh)]
            assert got[0] == np.float32(want[0]) and got[1] == np.float32(want[1]), (tuple(x.shape), p, got, want)
            if flat.numel() <= 200000:
                assert got[0] == O.quantile_f32(flat.numpy(), np.float32(ql))
    rng = np.random.default_rng(5)                          # two-element inputs: w = q, separates fused from unfused lerp
    for _ in range(200):
        a = np.float32(rng.standard_normal() * 10)
        x = np.array([a, np.float32(a + abs(rng.standard_normal()) * 0.1), 0, 0], np.float32)[:2]
        q = float(np.float32(rng.random()))
        got = eng.quantile2(torch.from_numpy(x.copy()).cuda(), q, q).cpu().numpy()
        assert got[0] == O.quantile_f32(x, np.float32(q)) and got[1] == got[0], (x, q, got)


@pytest.mark.parametrize('family', KERNEL_FAMILIES, ids=KERNEL_IDS)
def test_odd_sizes_against_oracle(eng, golden_dir, family):
    """Edge shapes the reference's MaskedConv1d handles (jasper.py:170-183): a single short utterance, lengths of
    1 frame and of 0 frames (all of its output is masked, enc_len 0), lengths that are not multiples of any tile, T beyond
    one 256-frame window, every utterance shorter than the padded batch.  Tokens and integer log-prob inputs must equal the oracle on the valid frames."""
    from oracle import int_oracle as O
    d, meta = _load(golden_dir, 'net_miniq_w8a8')
    cfg = _cfg('net_miniq_w8a8')
    sd = synth.make_state_dict(cfg, meta['seed'])
    blob, pm = pack.pack_model(cfg, sd, d['act_min'], d['act_max'], 8, 8)
    net = O.OracleNet(topology.conv_plan(cfg), cfg, sd, d['act_min'], d['act_max'], 8, 8)
    e = eng.Engine(blob, 0, **family)
    cases = [(1, 16, [16]), (1, 33, [33]), (3, 40, [1, 17, 40]), (2, 70, [2, 69]), (2, 600, [600, 301]),
             (5, 130, [129, 64, 65, 1, 128]), (4, 96, [50, 50, 50, 50]), (3, 40, [0, 17, 40]), (2, 48, [0, 0])]
    for B, T, lens in cases:
        x = synth.make_features(B, cfg.feat_in, T, 100 + T)
        want = net.forward(x, lens)
        logp, tok, elen = e.forward(torch.from_numpy(x).cuda(), torch.tensor(lens))
        tok, elen, logp = tok.cpu().numpy(), elen.cpu().numpy(), logp.cpu().numpy()
        assert np.array_equal(elen, want['enc_len']), (B, T, lens)
        for b in range(B):
            n = int(want['enc_len'][b])
            assert np.array_equal(tok[b, :n], want['tokens'][b, :n]), (B, T, lens, b)
            np.testing.assert_allclose(logp[b, :n], want['log_probs'][b, :n], rtol=1e-4, atol=5e-5)
    # an empty batch or zero frames is an argument error at the boundary, not a launch with an empty grid
    for B, T in ((0, 40), (2, 0)):
        with pytest.raises(eng.QasrError):
            e.forward(torch.zeros(B, cfg.feat_in, T, device='cuda'), torch.zeros(B, dtype=torch.int32))
    e.close()


def test_engine_rejects_foreign_blob(eng, golden_dir):
    """A blob of another format version (or garbage) must fail loudly at create time, not run with a wrong layout."""
    d, meta = _load(golden_dir, 'net_miniq_w8a8')
    cfg = _cfg('net_miniq_w8a8')
    blob, _ = pack.pack_model(cfg, synth.make_state_dict(cfg, meta['seed']), d['act_min'], d['act_max'], 8, 8)
    old = bytearray(blob)
    old[4:8] = (int.from_bytes(blob[4:8], 'little') - 1).to_bytes(4, 'little')      # header: magic, version, ...
    with pytest.raises(eng.QasrError):
        eng.Engine(bytes(old), 0)
    with pytest.raises(eng.QasrError):
        eng.Engine(b'\0' * 4096, 0)


def test_graph_replay_matches_direct_launches(eng, golden_dir):
    """Engine flag 16: the forward is captured into a hipGraph on the second call with the same buffers and replayed
    afterwards; new buffers or a new shape start over.  Results must equal the kernel-by-kernel engine every time."""
    d, meta = _load(golden_dir, 'net_quartznet_w8a8')
    cfg = topology.quartznet15x5()
    sd = synth.make_state_dict(cfg, meta['seed'])
    blob, _ = pack.pack_model(cfg, sd, d['act_min'], d['act_max'], 8, 8)
    ref, g = eng.Engine(blob, 0), eng.Engine(blob, 0, graph=True, wide_tiles=True)
    for B, T in ((8, 256), (8, 256), (3, 130)):
        x = torch.from_numpy(synth.make_features(B, 64, T, 7 + T)).cuda()
        lens = torch.tensor([T - 3 * i for i in range(B)], dtype=torch.int32).cuda()
        lp0, tk0, el0 = ref.forward(x, lens)
        To = g.out_frames(T)
        out = (torch.empty(B, To, 29, device='cuda'), torch.empty(B, To, dtype=torch.int32, device='cuda'),
               torch.empty(B, dtype=torch.int32, device='cuda'))
        side = torch.cuda.Stream()
        torch.cuda.synchronize()
        for it in range(6):                                   # direct, capture, replay, replay, new pointer, default stream
            x2 = x.clone() if it == 4 else x                 # a new input pointer forces a fresh capture cycle
            out[1].zero_()
            torch.cuda.synchronize()
            if it < 5:
                with torch.cuda.stream(side):
                    lp, tk, el = g.forward(x2, lens, out=out)
            else:                                             # the legacy default stream cannot be captured: direct launches
                lp, tk, el = g.forward(x2, lens, out=out)
            torch.cuda.synchronize()
            assert torch.equal(tk, tk0) and torch.equal(lp, lp0) and torch.equal(el, el0), (B, T, it)
    ref.close()
    g.close()


def test_requant_fast_path_adversarial(eng):
    """qasr_requant runs the production requant_batch: t = fma(f64(z), M, 1.5 * 2^52), clamp in the double domain, low
    mantissa word (csrc/qasr_device.h; fixedpoint_mul, quant_utils.py:196-198).  Inputs built to sit ON and next to rounding
    ties (z*M = k + 1/2 exactly, and the nearest integers z on either side of it), beyond the clamp range, and with
    (m, e) multipliers as batch_frexp makes them - against numpy fp64 round-half-even."""
    rng = np.random.default_rng(11)
    C, T = 64, 256
    M = np.empty(C)
    M[:16] = 0.5                                              # z odd -> exact ties
    M[16:32] = 2.0 ** -rng.integers(1, 12, 16) * rng.integers(1, 64, 16)          # dyadic: many exact ties
    M[32:48] = (rng.integers(1 << 30, 1 << 31, 16).astype(np.float64)) * 2.0 ** -rng.integers(31, 45, 16)  # (m, e) like batch_frexp
    M[48:] = rng.random(16) * 0.05
    z = rng.integers(-(1 << 22), 1 << 22, size=(2, C, T)).astype(np.int32)
    z[:, :, :64] = rng.integers(-4000, 4000, size=(2, C, 64))                      # small: products inside the ranges
    k = rng.integers(-200, 200, size=(2, 16, 64))
    z[:, 32:48, 64:128] = np.rint((k + 0.5) / M[32:48, None]).astype(np.int32)     # as close to a tie as integers allow
    for lo, hi in ((-128, 127), (0, 255), (-32, 31), (-256, 255)):
        want = np.clip(np.rint(z.astype(np.float64) * M[None, :, None]), lo, hi)
        got = eng.requant(torch.from_numpy(z).cuda(), torch.from_numpy(M), lo, hi).cpu().numpy()
        got = got.view(np.uint8).astype(np.int64) if hi > 127 else got.astype(np.int64)
        if lo < 0 and hi > 127:                              # 9-bit signed range does not fit a byte: compare where it does
            ok = (want >= -128) & (want <= 127)
            assert np.array_equal(got.astype(np.int8)[ok], want.astype(np.int64)[ok].astype(np.int8))
        else:
            assert np.array_equal(got, want.astype(np.int64)), (lo, hi, int((got != want).sum()))


# ---------------------------------------------------------------------------------- production shape
GENERATIONS = [dict(gen=2), dict(gen=2, wide_tiles=True, tile128=0), dict(gen=2, wide_tiles=True, tile128=1), dict(gen=1),
               dict(gen=1, wide_tiles=True)]
GEN_IDS = ['k_sep2_32', 'k_sep2_64', 'k_sep2_128', 'k_sep_32', 'k_sep_64']


def _engine_gen(eng, blob, gen, tile128=None, wide_tiles=False, **kw):
    """Kernel family and tile size through qasr_engine_opts (include/qasr.h): gen 1 keeps every separable layer on k_sep,
    2 (default) routes the stride-1 layers with 256 / 512 input channels to k_sep2; with wide tiles, tile128 = 1 puts
    k_sep2's separable layers (and Jasper's plain dense convs) on 128-frame tiles, 0 keeps everything on 64."""
    tile = 32 if not wide_tiles else (64 if tile128 == 0 else 128)
    return eng.Engine(blob, 0, sep_gen=gen, tile=tile, **kw)


@pytest.fixture(scope='module')
def oracle_quartznet_t500(golden_dir):
    """OracleNet on 2 utterances x 500 frames (one full, one ragged) of QuartzNet15x5 w8a8: every accumulator."""
    d, meta = _load(golden_dir, 'net_quartznet_w8a8')
    cfg = topology.quartznet15x5()
    sd = synth.make_state_dict(cfg, meta['seed'])
    net = O.OracleNet(topology.conv_plan(cfg), cfg, sd, d['act_min'], d['act_max'], 8, 8)
    x = synth.make_features(2, 64, 500, 3)
    lens = [500, 437]
    want = net.forward(x, lens)
    blob, pm = pack.pack_model(cfg, sd, d['act_min'], d['act_max'], 8, 8)
    return dict(x=x, lens=lens, want=want, accs=[t['acc'] for t in net.trace], blob=blob, pm=pm, cfg=cfg)


@pytest.mark.parametrize('family', GENERATIONS, ids=GEN_IDS)
def test_quartznet_t500_every_accumulator(eng, oracle_quartznet_t500, family):
    """The production kernels at the production tile count (T = 500 -> 250 frames = 8 / 4 time tiles per utterance, halos
    crossing tile borders, a ragged utterance): EVERY conv accumulator of QuartzNet15x5 against the CPU oracle
    (quant_modules.py:301-305), bit-exact, for both kernel generations and every tile size."""
    o = oracle_quartznet_t500
    fam = dict(family)
    e = _engine_gen(eng, o['blob'], fam.pop('gen'), debug=True, **fam)
    logp, tokens, enc_len = e.forward(torch.from_numpy(o['x']).cuda(), torch.tensor(o['lens']))
    torch.cuda.synchronize()
    couts = _site_dims(o['cfg'])
    labels = e.op_labels()
    if family.get('gen') == 2:
        assert sum(l.startswith('k_sep2<') for l in labels) >= 70, labels      # the new kernel is what runs
        want_tt = '128, 1>' if family.get('tile128') == 1 else ('64, 1>' if family.get('wide_tiles') else '32, 1>')
        assert sum(l.startswith('k_sep2<') and l.endswith(want_tt) for l in labels) >= 50, (want_tt, labels)
    else:
        assert not any(l.startswith('k_sep2<') for l in labels)
    for i, (op, pane) in enumerate(o['pm']['sites']):
        want = o['accs'][i]
        got = e.read_acc(op, pane, couts[i], want.shape[2])
        assert np.array_equal(got, want), f'conv {i} (op {op}, pane {pane}, {labels[op]})'
    want = o['want']
    assert np.array_equal(enc_len.cpu().numpy(), want['enc_len'])
    for b in range(2):
        n = int(want['enc_len'][b])
        assert np.array_equal(tokens.cpu().numpy()[b, :n], want['tokens'][b, :n])
        np.testing.assert_allclose(logp.cpu().numpy()[b, :n], want['log_probs'][b, :n], rtol=1e-4, atol=5e-5)
    e.close()


@pytest.fixture(scope='module', params=[('net_quartznet_w8a8', 8, 8), ('net_quartznet_w6a6', 6, 6)], ids=['w8a8', 'w6a6'])
def oracle_full_size(request, golden_dir):
    """oracle.fakequant_torch.FakeQuantNet (the reference's fake-quant op sequence, pinned by tests/test_oracle_golden.py)
    on BASELINE.json's config 2 / config 3 shape: 32 utterances x 500 frames, ragged lengths."""
    from oracle.fakequant_torch import FakeQuantNet
    name, wbit, abit = request.param
    d, meta = _load(golden_dir, name)
    cfg = topology.quartznet15x5()
    sd = synth.make_state_dict(cfg, meta['seed'])
    B, T = 32, 500
    x = synth.make_features(B, 64, T, 21)
    lens = [T - 11 * (i % 13) - (i % 3) for i in range(B)]      # 365 .. 500: lengths inside every tile of the last 128 frames
    lens[3], lens[17] = 500, 129                               # a full utterance and one that ends one frame into tile 1
    net = FakeQuantNet(topology.conv_plan(cfg), cfg, sd, d['act_min'], d['act_max'], wbit, abit)
    want = net.forward(x, lens)
    blob, pm = pack.pack_model(cfg, sd, d['act_min'], d['act_max'], wbit, abit)
    return dict(x=x, lens=lens, want=want, blob=blob, pm=pm, wbit=wbit)


@pytest.mark.parametrize('opts', [dict(tile=32), dict(tile=128)], ids=['tile32', 'tile128'])
def test_production_engine_full_size_against_oracle(eng, oracle_full_size, opts):
    """The NON-debug engine (production instantiations: pipelined depthwise stage, packed-code masks, no accumulator hooks)
    at full size, w8a8 and w6a6, against the CPU oracle: tokens, encoded lengths, log-probs (rtol 1e-4) and the final
    encoder codes (the decoder's input, qasr_engine_read_tensor) on every valid frame."""
    o = oracle_full_size
    e = eng.Engine(o['blob'], 0, **opts)
    logp, tokens, enc_len = e.forward(torch.from_numpy(o['x']).cuda(), torch.tensor(o['lens']))
    torch.cuda.synchronize()
    labels = e.op_labels()
    assert not any('true' in l for l in labels), labels        # no debug instantiation anywhere
    tt = f", {opts['tile']}, 1>"
    assert sum(l.startswith('k_sep2<') and l.endswith(tt) for l in labels) >= 60, labels
    n_launch = e.num_launches()                                # QuartzNet15x5: 80 launches, layer by layer
    assert 78 <= n_launch <= 82, n_launch
    want = o['want']
    wl = want['enc_len'].numpy()
    assert np.array_equal(enc_len.cpu().numpy(), wl)
    codes = e.read_tensor(o['pm']['dec_in'], 1024)
    tk, lp = tokens.cpu().numpy(), logp.cpu().numpy()
    for b in range(len(wl)):
        n = int(wl[b])
        assert np.array_equal(codes[b, :, :n], want['enc_codes'][b, :, :n].numpy().astype(np.int8)), b
        assert np.array_equal(tk[b, :n], want['tokens'][b, :n].numpy()), b
        np.testing.assert_allclose(lp[b, :n], want['log_probs'][b, :n].numpy(), rtol=1e-4, atol=5e-5)
    e.close()


@pytest.fixture(scope='module')
def oracle_jasper_full_size(golden_dir):
    """FakeQuantNet on Jasper10x5dr w8a8 (BASELINE.json config 4's net) at 4 utterances x 500 frames, ragged lengths:
    full, one frame into the second 128-frame tile, inside the last 32-frame tile, odd."""
    from oracle.fakequant_torch import FakeQuantNet
    d, meta = _load(golden_dir, 'net_jasper_w8a8')
    cfg = topology.jasper10x5dr()
    sd = synth.make_state_dict(cfg, meta['seed'])
    B, T = 4, 500
    x = synth.make_features(B, 64, T, 33)
    lens = [500, 257, 471, 389]                                # after block 0's stride 2: 250, 129, 236, 195
    net = FakeQuantNet(topology.conv_plan(cfg), cfg, sd, d['act_min'], d['act_max'], 8, 8)
    want = net.forward(x, lens)
    blob, pm = pack.pack_model(cfg, sd, d['act_min'], d['act_max'], 8, 8)
    return dict(x=x, lens=lens, want=want, blob=blob, pm=pm)


@pytest.mark.parametrize('tile', [32, 128])
def test_production_engine_jasper_full_size_against_oracle(eng, oracle_jasper_full_size, tile):
    """The NON-debug engine on Jasper10x5dr at its real channel counts (k_dense2's `DBG = false` plain and block-end forms;
    jasper.py:601-630,664-682) against the CPU oracle: final encoder codes, tokens, encoded lengths, log-probs (rtol 1e-4)."""
    o = oracle_jasper_full_size
    e = eng.Engine(o['blob'], 0, tile=tile)
    logp, tokens, enc_len = e.forward(torch.from_numpy(o['x']).cuda(), torch.tensor(o['lens']))
    torch.cuda.synchronize()
    labels = e.op_labels()
    d2 = [l[len('k_dense2<'):-1].split(', ') for l in labels if l.startswith('k_dense2<')]      # [MT, DBG, RES]
    assert len(d2) >= 50 and all(a[1] == 'false' for a in d2), labels     # the 50 dense convs of blocks 1-10 + block 11, no hooks
    assert sum(a[2] == 'true' for a in d2) == 10, labels                  # every block-end (dense residual) layer on the RES form
    want = o['want']
    wl = want['enc_len'].numpy()
    assert np.array_equal(enc_len.cpu().numpy(), wl)
    codes = e.read_tensor(o['pm']['dec_in'], 1024)
    tk, lp = tokens.cpu().numpy(), logp.cpu().numpy()
    for b in range(len(wl)):
        n = int(wl[b])
        assert np.array_equal(codes[b, :, :n], want['enc_codes'][b, :, :n].numpy().astype(np.int8)), b
        assert np.array_equal(tk[b, :n], want['tokens'][b, :n].numpy()), b
        np.testing.assert_allclose(lp[b, :n], want['log_probs'][b, :n].numpy(), rtol=1e-4, atol=5e-5)
    e.close()


@pytest.fixture(scope='module')
def oracle_jasper_t300(golden_dir):
    """OracleNet on one utterance x 300 frames of Jasper10x5dr w8a8 (150 frames after block 0: 5 / 3 / 2 time tiles)."""
    d, meta = _load(golden_dir, 'net_jasper_w8a8')
    cfg = topology.jasper10x5dr()
    sd = synth.make_state_dict(cfg, meta['seed'])
    net = O.OracleNet(topology.conv_plan(cfg), cfg, sd, d['act_min'], d['act_max'], 8, 8)
    x = synth.make_features(1, 64, 300, 4)
    want = net.forward(x, [300])
    blob, pm = pack.pack_model(cfg, sd, d['act_min'], d['act_max'], 8, 8)
    return dict(x=x, want=want, accs=[t['acc'] for t in net.trace], blob=blob, pm=pm, cfg=cfg)


@pytest.mark.parametrize('family', [dict(gen=2), dict(gen=2, wide_tiles=True, tile128=0), dict(gen=2, wide_tiles=True, tile128=1)],
                         ids=['tiles32', 'tiles64', 'tiles128'])
def test_jasper_t300_every_accumulator(eng, oracle_jasper_t300, family):
    """Jasper10x5dr's dense convs (tap-shifted GEMMs over a staged window), dense residual panes and many-consumer
    requants at a production frame count: EVERY conv accumulator (108 convs + decoder) against the CPU oracle, for the
    three tile sizes the engine uses (32: one step in flight; 64 / 128: throughput mode)."""
    o = oracle_jasper_t300
    fam = dict(family)
    e = _engine_gen(eng, o['blob'], fam.pop('gen'), debug=True, **fam)
    logp, tokens, enc_len = e.forward(torch.from_numpy(o['x']).cuda(), torch.tensor([300]))
    torch.cuda.synchronize()
    couts = _site_dims(o['cfg'])
    labels = e.op_labels()
    for i, (op, pane) in enumerate(o['pm']['sites']):
        want = o['accs'][i]
        got = e.read_acc(op, pane, couts[i], want.shape[2])
        assert np.array_equal(got, want), f'conv {i} (op {op}, pane {pane}, {labels[op]})'
    want = o['want']
    n = int(want['enc_len'][0])
    assert np.array_equal(enc_len.cpu().numpy(), want['enc_len'])
    assert np.array_equal(tokens.cpu().numpy()[0, :n], want['tokens'][0, :n])
    np.testing.assert_allclose(logp.cpu().numpy()[0, :n], want['log_probs'][0, :n], rtol=1e-4, atol=5e-5)
    e.close()


def test_jasper_bench_size_properties(eng, oracle_jasper_t300):
    """BASELINE config 4 (Jasper10x5dr w8a8, B = 64, T = 500): size-independent properties at full size - re-running is
    bit-reproducible, the three tile sizes produce identical integers, an utterance's result does not depend on its batch
    neighbours or on batch padding."""
    blob = oracle_jasper_t300['blob']
    B, T = 64, 500
    x = torch.from_numpy(synth.make_features(B, 64, T, 12)).cuda()
    lens = torch.tensor([T - 11 * (i % 7) for i in range(B)])
    e = _engine_gen(eng, blob, 2)
    lp1, tk1, el1 = e.forward(x, lens)
    lp1, tk1, el1 = lp1.cpu().numpy(), tk1.cpu().numpy(), el1.cpu().numpy()
    lp2, tk2, _ = e.forward(x, lens)
    assert np.array_equal(tk1, tk2.cpu().numpy()) and np.array_equal(lp1, lp2.cpu().numpy())
    assert np.array_equal(el1, (lens.numpy() + 1) // 2)
    for t128 in (0, 1):
        ew = _engine_gen(eng, blob, 2, wide_tiles=True, tile128=t128)
        lpw, tkw, _ = ew.forward(x, lens)
        assert np.array_equal(tk1, tkw.cpu().numpy()) and np.array_equal(lp1, lpw.cpu().numpy()), t128
        ew.close()
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1))
    _, tk3, _ = e.forward(x[perm].contiguous(), lens[perm])
    assert np.array_equal(tk3.cpu().numpy(), tk1[perm.numpy()])
    i, L = 9, int(lens[9])
    lp4, tk4, el4 = e.forward(x[i:i + 1, :, :L].contiguous(), lens[i:i + 1])
    n = int(el4[0])
    assert np.array_equal(tk4.cpu().numpy()[0, :n], tk1[i, :n])
    np.testing.assert_array_equal(lp4.cpu().numpy()[0, :n], lp1[i, :n])
    e.close()


# ---------------------------------------------------------------------------------- fused layer, operator level
def _sep_case(rng, B, T, cin, cout, K, x_unsigned, res, n_outs, big=False):
    """Operands of one separable layer with realistic magnitudes; `big` pushes accumulators past 2^22 in a few channels."""
    lens = np.array([T] + list(rng.integers(1, T + 1, B - 1)))
    x = rng.integers(0 if x_unsigned else -128, 256 if x_unsigned else 128, (B, cin, T))
    x = np.where(np.arange(T)[None, None, :] < lens[:, None, None], x, 0)
    wdw = rng.integers(-127, 127, (cin, K))
    m_dw = 127.0 / (np.abs(wdw).sum(1) * (255 if x_unsigned else 128) * rng.uniform(0.2, 0.6, cin))
    wpw = rng.integers(-127, 127, (cout, cin))
    bias = rng.integers(-20000, 20000, cout)
    sb = rng.uniform(1e-6, 3e-6, cout).astype(np.float32)
    if big:                                                    # same-sign rows + a large bias: |acc| well past 2^22
        wpw[:8] = np.abs(wpw[:8])
        wpw[8:16] = -np.abs(wpw[8:16])
        wdw[:] = np.abs(wdw)
        m_dw = m_dw * 4                                        # the depthwise output saturates at +127 almost everywhere
        bias[:8] += 3_000_000
        bias[8:16] -= 3_000_000
    outs = []
    for j in range(n_outs):
        hi = 255 if j % 2 == 0 else 127
        lo = 0 if res is None else -hi - 1
        outs.append(dict(mode=1, lo=lo, hi=hi, M=rng.uniform(2e-4, 2e-3, cout)))
    r = None
    if res:
        rcin = res
        rx = np.where(np.arange(T)[None, None, :] < lens[:, None, None], rng.integers(0, 256, (B, rcin, T)), 0)
        r = dict(x=rx, w=rng.integers(-127, 127, (cout, rcin)), bias=rng.integers(-20000, 20000, cout),
                 m=rng.uniform(2e-4, 2e-3, cout), sb=rng.uniform(1e-6, 3e-6, cout).astype(np.float32),
                 m_main=rng.uniform(2e-4, 2e-3, cout), qlo=-128, qhi=127)
        outs = [dict(mode=2, lo=0, hi=127)] + [dict(mode=0, lo=0, hi=255, m=float(rng.uniform(1.2, 2.0)))][: n_outs - 1]
    return dict(x=x, lens=lens, wdw=wdw, m_dw=m_dw, wpw=wpw, bias=bias, sb=sb, outs=outs, res=r)


SEP_SHAPES = [(33, 256, 256, 0), (39, 256, 256, 256), (51, 256, 512, 0), (51, 512, 512, 256), (51, 512, 512, 512),
              (63, 512, 512, 0), (63, 512, 512, 512), (75, 512, 512, 0), (75, 512, 512, 512)]


@pytest.mark.parametrize('gen,tile', [(2, 32), (2, 64), (2, 128), (1, 32), (1, 64)],
                         ids=['k_sep2_32', 'k_sep2_64', 'k_sep2_128', 'k_sep_32', 'k_sep_64'])
@pytest.mark.parametrize('K,cin,cout,rcin', SEP_SHAPES)
def test_sep_layer_against_oracle(eng, gen, tile, K, cin, cout, rcin):
    """The fused separable-layer kernels at operator level (qasr_sep_layer): every production tap count / channel shape of
    QuartzNet15x5 at T = 250 and 301 (several time tiles, halos across tile borders, ragged lengths incl. 1 frame),
    u8 depthwise input, one res_act case per shape, accumulators beyond 2^22 with QASR_F_EXACT_Z - depthwise accumulator,
    1x1 accumulator, residual accumulator and every consumer's requantised output bit-exact against the numpy oracle."""
    from qasr.pack import F_EXACT_Z, F_MASK_OUT, F_RELU
    for T, big in ((250, False), (301, True)):
        rng = np.random.default_rng(K * 1000 + T + cin + rcin)
        # consumers: a block-end layer feeds the next block's depthwise conv and residual conv (2); a plain layer has one
        # (k_sep2's pipelined epilogue is built for exactly that; k_sep takes any number)
        c = _sep_case(rng, 3, T, cin, cout, K, True, rcin or None, 2 if (rcin or gen == 1) else 1, big=big)
        want = O.sep_layer_ref(c['x'], c['lens'], c['wdw'], c['m_dw'], (-128, 127), c['wpw'], c['bias'], c['outs'], relu=True,
                               mask_out=True, sb=c['sb'], exact_z=big, res=c['res'])
        if big:
            assert np.abs(want['acc']).max() >= 1 << 22           # the float32 round trip is really exercised
        res = None
        if c['res'] is not None:
            res = dict(c['res'], x=torch.from_numpy(c['res']['x'].astype(np.uint8)).cuda())
        got = eng.sep_layer(torch.from_numpy(c['x'].astype(np.uint8)).cuda(), c['lens'], c['wpw'], c['bias'], c['outs'],
                            wdw=c['wdw'], m_dw=c['m_dw'], x_unsigned=True, flags=F_RELU | F_MASK_OUT | (F_EXACT_Z if big else 0),
                            sb=c['sb'], res=res, tile=tile, gen=gen)
        assert got['label'].startswith('k_sep2<' if gen == 2 else 'k_sep<'), got['label']
        assert np.array_equal(got['dw_acc'].cpu().numpy(), want['dw_acc']), (got['label'], T, 'depthwise accumulator')
        assert np.array_equal(got['acc'].cpu().numpy(), want['acc']), (got['label'], T, '1x1 accumulator')
        if rcin:
            assert np.array_equal(got['racc'].cpu().numpy(), want['racc']), (got['label'], T, 'residual accumulator')
        for j, o in enumerate(c['outs']):
            g = got['outs'][j].cpu().numpy()
            g = g.view(np.uint8).astype(np.int64) if o['hi'] > 127 else g.astype(np.int64)
            assert np.array_equal(g, want['outs'][j]), (got['label'], T, f'consumer {j}', int((g != want['outs'][j]).sum()))


@pytest.mark.parametrize('tile', [32, 64, 128])
@pytest.mark.parametrize('K,cin,cout,rcin', SEP_SHAPES)
def test_sep_layer_production_kernels_against_oracle(eng, tile, K, cin, cout, rcin):
    """The PRODUCTION instantiations of k_sep2 (no accumulator hooks: `DBG = false`, i.e. the pipelined depthwise stage that
    requantises group g-1 behind group g's MFMAs, and the length masks on packed codes) at operator level: every consumer's
    output bit-exact against the numpy oracle, with lengths on every edge of the mask logic - full, inside the last 32-frame
    tile, on a 32- / 16- / 4-frame boundary, one frame, and a tile that lies entirely behind the length."""
    from qasr.pack import F_EXACT_Z, F_MASK_OUT, F_RELU
    T = 250
    rng = np.random.default_rng(K * 77 + cin + rcin + tile)
    for big in (False, True):
        c = _sep_case(rng, 8, T, cin, cout, K, True, rcin or None, 2 if rcin else 1, big=big)
        lens = np.array([250, 249, 224, 208, 131, 100, 17, 1])
        c['lens'] = lens
        c['x'] = np.where(np.arange(T)[None, None, :] < lens[:, None, None], c['x'], 0)
        if c['res'] is not None:
            c['res']['x'] = np.where(np.arange(T)[None, None, :] < lens[:, None, None], c['res']['x'], 0)
        want = O.sep_layer_ref(c['x'], c['lens'], c['wdw'], c['m_dw'], (-128, 127), c['wpw'], c['bias'], c['outs'], relu=True,
                               mask_out=True, sb=c['sb'], exact_z=big, res=c['res'])
        res = None
        if c['res'] is not None:
            res = dict(c['res'], x=torch.from_numpy(c['res']['x'].astype(np.uint8)).cuda())
        got = eng.sep_layer(torch.from_numpy(c['x'].astype(np.uint8)).cuda(), c['lens'], c['wpw'], c['bias'], c['outs'],
                            wdw=c['wdw'], m_dw=c['m_dw'], x_unsigned=True, flags=F_RELU | F_MASK_OUT | (F_EXACT_Z if big else 0),
                            sb=c['sb'], res=res, tile=tile, gen=2, hooks=False)
        assert got['label'].startswith('k_sep2<') and 'false' in got['label'], got['label']
        for j, o in enumerate(c['outs']):
            g = got['outs'][j].cpu().numpy()
            g = g.view(np.uint8).astype(np.int64) if o['hi'] > 127 else g.astype(np.int64)
            assert np.array_equal(g, want['outs'][j]), (got['label'], big, f'consumer {j}', int((g != want['outs'][j]).sum()))
            for b_, n_ in enumerate(lens):
                assert not g[b_, :, n_:].any(), (got['label'], 'frames behind the length are not zero', b_)


def test_sep_layer_dilated_and_bare_1x1(eng):
    """The dilated depthwise layer of block 16 (K = 87, dilation 2) on round 1's k_sep AND on k_sep2's dilation-2 form (channels
    staged as even / odd frame rows; debug and production instantiations, 64- and 128-frame tiles, lengths on the mask edges
    incl. an odd one), and a bare 1x1 conv."""
    from qasr.pack import F_MASK_OUT, F_RELU
    rng = np.random.default_rng(87)
    c = _sep_case(rng, 2, 250, 512, 512, 87, True, None, 1)
    want = O.sep_layer_ref(c['x'], c['lens'], c['wdw'], c['m_dw'], (-128, 127), c['wpw'], c['bias'], c['outs'], dilation=2,
                           relu=True, mask_out=True)
    for gen, tiles in ((1, (32, 64)), (2, (32, 64, 128))):
        for tile in tiles:
            got = eng.sep_layer(torch.from_numpy(c['x'].astype(np.uint8)).cuda(), c['lens'], c['wpw'], c['bias'], c['outs'],
                                wdw=c['wdw'], m_dw=c['m_dw'], x_unsigned=True, dilation=2, flags=F_RELU | F_MASK_OUT, tile=tile, gen=gen)
            assert got['label'].startswith('k_sep2<87, 4, 0, 2, true') == (gen == 2), got['label']
            assert np.array_equal(got['dw_acc'].cpu().numpy(), want['dw_acc']), (gen, tile)
            assert np.array_equal(got['acc'].cpu().numpy(), want['acc']), (gen, tile)
            assert np.array_equal(got['outs'][0].cpu().numpy().view(np.uint8).astype(np.int64), want['outs'][0]), (gen, tile)
    c8 = _sep_case(rng, 8, 250, 512, 512, 87, True, None, 1)
    lens = np.array([250, 249, 224, 209, 131, 100, 17, 1])
    c8['lens'] = lens
    c8['x'] = np.where(np.arange(250)[None, None, :] < lens[:, None, None], c8['x'], 0)
    want8 = O.sep_layer_ref(c8['x'], lens, c8['wdw'], c8['m_dw'], (-128, 127), c8['wpw'], c8['bias'], c8['outs'], dilation=2,
                            relu=True, mask_out=True)
    for tile in (64, 128):
        got = eng.sep_layer(torch.from_numpy(c8['x'].astype(np.uint8)).cuda(), lens, c8['wpw'], c8['bias'], c8['outs'],
                            wdw=c8['wdw'], m_dw=c8['m_dw'], x_unsigned=True, dilation=2, flags=F_RELU | F_MASK_OUT, tile=tile, gen=2,
                            hooks=False)
        assert got['label'] == f'k_sep2<87, 4, 0, 2, false, {tile}, 2>', got['label']
        g = got['outs'][0].cpu().numpy().view(np.uint8).astype(np.int64)
        assert np.array_equal(g, want8['outs'][0]), (tile, int((g != want8['outs'][0]).sum()))
    want = O.sep_layer_ref(c['x'], c['lens'], None, None, None, c['wpw'], c['bias'], c['outs'], relu=True, mask_out=True)
    got = eng.sep_layer(torch.from_numpy(c['x'].astype(np.uint8)).cuda(), c['lens'], c['wpw'], c['bias'], c['outs'],
                        x_unsigned=True, flags=F_RELU | F_MASK_OUT)
    assert np.array_equal(got['acc'].cpu().numpy(), want['acc'])
    assert np.array_equal(got['outs'][0].cpu().numpy().view(np.uint8).astype(np.int64), want['outs'][0])
    # block 17 of QuartzNet (512 -> 1024, bare 1x1): k_sep2's K = 0 form, both tile sizes
    wpw = rng.integers(-127, 127, (1024, 512))
    bias = rng.integers(-20000, 20000, 1024)
    outs = [dict(mode=1, lo=0, hi=127, M=rng.uniform(2e-4, 2e-3, 1024))]
    want = O.sep_layer_ref(c['x'], c['lens'], None, None, None, wpw, bias, outs, relu=True, mask_out=False)
    for tile in (32, 64):
        got = eng.sep_layer(torch.from_numpy(c['x'].astype(np.uint8)).cuda(), c['lens'], wpw, bias, outs, x_unsigned=True,
                            flags=F_RELU, tile=tile, gen=2)
        assert got['label'].startswith('k_sep2<0, 4, 0, 4'), got['label']
        assert np.array_equal(got['acc'].cpu().numpy(), want['acc'])
        assert np.array_equal(got['outs'][0].cpu().numpy().astype(np.int64), want['outs'][0])
