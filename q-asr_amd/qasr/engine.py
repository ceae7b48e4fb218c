"""ctypes binding of libqasr_hip.so (include/qasr.h).  PyTorch is used only to own device
memory and streams.  There is no CPU fallback: a missing library or a failing call raises."""
import ctypes as C
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('QASR_LIB', os.path.join(HERE, 'libqasr_hip.so'))   # QASR_LIB: A/B builds in one run

SYMBOLS = ['qasr_blob_check', 'qasr_engine_create', 'qasr_engine_create_ex', 'qasr_engine_default_opts', 'qasr_engine_destroy', 'qasr_engine_forward', 'qasr_engine_forward_audio', 'qasr_engine_out_frames',
           'qasr_engine_num_ops', 'qasr_engine_num_launches', 'qasr_engine_read_acc', 'qasr_engine_read_tensor', 'qasr_engine_last_op_ms',
           'qasr_engine_time_ops', 'qasr_engine_run_op', 'qasr_engine_op_label',
           'qasr_frontend_mel', 'qasr_frontend_plan', 'qasr_frontend_mel_planned', 'qasr_frontend_frames',
           'qasr_frontend_workspace_bytes', 'qasr_pw_conv_acc',
           'qasr_dw_conv_acc', 'qasr_dense_conv_acc', 'qasr_requant', 'qasr_dyn_range', 'qasr_dyn_range_percentile', 'qasr_dyn_residue_codes', 'qasr_dyn_act_params', 'qasr_dyn_requant',
           'qasr_dyn_quant_in', 'qasr_dyn_conv_params', 'qasr_sep_layer', 'qasr_quantile2', 'qasr_quantile_workspace_bytes', 'qasr_debug_prof',
           'qasr_debug_timeline',
           'qasr_last_error', 'qasr_version']

_lib = None


class _SepOut(C.Structure):
    _fields_ = [('ptr', C.c_void_p), ('mtab', C.c_void_p), ('m', C.c_double), ('lo', C.c_int32), ('hi', C.c_int32),
                ('mode', C.c_int32), ('pad_', C.c_int32)]


class SepLayerArgs(C.Structure):
    """qasr_sep_layer_args (include/qasr.h)."""
    _fields_ = ([(n, C.c_int32) for n in ('B', 'T', 'Tp', 'cin', 'cout', 'K', 'dilation', 'tile', 'gen')] +
                [('flags', C.c_uint32), ('x', C.c_void_p)] +
                [(n, C.c_int32) for n in ('x_unsigned', 'dw_lo', 'dw_hi', 'n_outs')] +
                [(n, C.c_void_p) for n in ('wdw', 'wdw2', 'bias_dw', 'm_dw', 'w', 'bias', 'sb', 'm_main', 'lens', 'rx', 'rw',
                                           'rbias', 'rm', 'rsb')] +
                [(n, C.c_int32) for n in ('rcin', 'r_unsigned', 'qlo', 'qhi')] +
                [('outs', _SepOut * 3), ('dw_acc', C.c_void_p), ('acc', C.c_void_p), ('racc', C.c_void_p)])


class EngineOpts(C.Structure):
    """qasr_engine_opts (include/qasr.h): the launch-plan choices of one engine."""
    _fields_ = ([('struct_size', C.c_uint32), ('debug', C.c_uint32)] +
                [(n, C.c_int32) for n in ('tile_frames', 'sep_gen', 'fuse_dw', 'fuse_stem', 'fuse_decoder', 'graph',
                                          'retired_whole_utterance', 'res_tile128', 'dense_tile128', 'retired_legacy_pw', 'retired_persistent', 'fuse_norm')] +
                [('reserved', C.c_int32 * 2)])


class QasrError(RuntimeError):
    pass


def load_library():
    """Loads the HIP extension; raises (loudly) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise QasrError(f'{LIB_PATH} is missing: run `python __graft_entry__.py` (build()) first; '
                        'the quantised inference path has no CPU fallback')
    lib = C.CDLL(LIB_PATH)
    vp, i32, sz = C.c_void_p, C.c_int, C.c_size_t
    lib.qasr_blob_check.argtypes = [C.c_char_p, sz, C.c_char_p, sz]
    lib.qasr_engine_create.argtypes = [vp, sz, i32, i32, C.POINTER(vp)]
    lib.qasr_engine_create_ex.argtypes = [vp, sz, i32, C.POINTER(EngineOpts), C.POINTER(vp)]
    lib.qasr_engine_default_opts.argtypes = [C.POINTER(EngineOpts)]
    lib.qasr_engine_default_opts.restype = None
    lib.qasr_engine_destroy.argtypes = [vp]
    lib.qasr_engine_destroy.restype = None
    lib.qasr_engine_forward.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp, vp]
    lib.qasr_engine_out_frames.argtypes = [vp, i32]
    lib.qasr_engine_num_ops.argtypes = [vp]
    lib.qasr_engine_num_launches.argtypes = [vp]
    lib.qasr_engine_read_acc.argtypes = [vp, i32, i32, vp, sz]
    lib.qasr_engine_read_tensor.argtypes = [vp, i32, vp, sz, C.POINTER(i32), C.POINTER(i32)]
    lib.qasr_engine_last_op_ms.argtypes = [vp, vp, i32]
    lib.qasr_engine_time_ops.argtypes = [vp, vp, i32, vp, i32]
    lib.qasr_engine_run_op.argtypes = [vp, vp, i32]
    lib.qasr_engine_op_label.argtypes = [vp, i32, C.c_char_p, sz]
    lib.qasr_engine_forward_audio.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp, i32, C.c_float, i32, vp, sz, vp, vp, vp, vp, vp]
    lib.qasr_frontend_mel.argtypes = [vp, vp, vp, i32, i32, vp, vp, i32, C.c_float, i32, vp, vp, vp, sz]
    lib.qasr_frontend_mel_planned.argtypes = lib.qasr_frontend_mel.argtypes
    lib.qasr_frontend_plan.argtypes = [vp, vp, i32, vp, sz]
    lib.qasr_frontend_frames.argtypes = [i32, i32]
    lib.qasr_frontend_workspace_bytes.argtypes = [i32, i32, i32]
    lib.qasr_frontend_workspace_bytes.restype = sz
    lib.qasr_pw_conv_acc.argtypes = [vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.qasr_dw_conv_acc.argtypes = [vp, vp, i32, vp, vp] + [i32] * 11 + [vp]
    lib.qasr_dense_conv_acc.argtypes = [vp, vp, i32, vp, vp] + [i32] * 12 + [vp]
    lib.qasr_requant.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.qasr_dyn_range.argtypes = [vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, vp]
    lib.qasr_dyn_residue_codes.argtypes = [vp, vp, i32, vp, sz, vp, vp]
    lib.qasr_dyn_range_percentile.argtypes = [vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, C.c_float, C.c_float, vp, vp, sz, vp]
    lib.qasr_dyn_act_params.argtypes = [vp, vp, i32, i32, vp, i32, vp, i32, vp, vp, vp]
    lib.qasr_dyn_requant.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.qasr_dyn_quant_in.argtypes = [vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp, vp]
    lib.qasr_dyn_conv_params.argtypes = [vp, vp, vp, vp, vp, i32, i32, vp, vp]
    lib.qasr_debug_prof.argtypes = [vp]
    lib.qasr_debug_timeline.argtypes = [vp, sz]
    lib.qasr_sep_layer.argtypes = [vp, C.POINTER(SepLayerArgs), C.c_char_p, sz]
    lib.qasr_quantile2.argtypes = [vp, vp, sz, C.c_float, C.c_float, vp, vp, sz]
    lib.qasr_quantile_workspace_bytes.argtypes = []
    lib.qasr_quantile_workspace_bytes.restype = sz
    lib.qasr_last_error.restype = C.c_char_p
    lib.qasr_version.restype = C.c_char_p
    _lib = lib
    return lib


def _check(rc, what):
    if rc != 0:
        raise QasrError(f'{what} failed ({rc}): {load_library().qasr_last_error().decode()}')


def blob_check(blob: bytes):
    """qasr_blob_check: raises QasrError naming the first malformed field of a packed model (host-only, no GPU needed)."""
    lib = load_library()
    why = C.create_string_buffer(256)
    if lib.qasr_blob_check(bytes(blob), len(blob), why, len(why)) != 0:
        raise QasrError('malformed blob: ' + why.value.decode())


def _stream_ptr(stream=None):
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


def _ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


class Engine:
    """One packed model on one GPU (qasr_engine_*)."""

    def __init__(self, blob: bytes, device=0, debug=False, timing=False, wide_tiles=False,
                 graph=False, tile=None, sep_gen=None, fuse_dw=None, fuse_stem=None, fuse_decoder=None, res_tile128=None,
                 dense_tile128=None, fuse_norm=None, legacy_create=False):
        """Options = qasr_engine_opts (include/qasr.h).  tile: frames per work-group (32 / 64 / 128; `wide_tiles=True` is the
        older spelling of 128); None leaves a choice at the engine's default."""
        lib = load_library()
        if not torch.cuda.is_available():
            raise QasrError('no GPU: the integer engine needs an MI355X (there is no CPU fallback)')
        self.lib = lib
        self.device = torch.device('cuda', device)
        self._blob = blob
        self._h = C.c_void_p()
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        o = EngineOpts()
        lib.qasr_engine_default_opts(C.byref(o))
        assert o.struct_size == C.sizeof(EngineOpts), 'qasr_engine_opts: header and binding disagree'
        o.debug = int(bool(debug)) | (2 if timing else 0)
        o.tile_frames = int(tile) if tile else (128 if wide_tiles else 32)
        o.sep_gen = int(sep_gen or 0)
        for name, v in (('fuse_dw', fuse_dw), ('fuse_stem', fuse_stem), ('fuse_decoder', fuse_decoder),
                        ('res_tile128', res_tile128), ('dense_tile128', dense_tile128), ('fuse_norm', fuse_norm)):
            if v is not None:
                setattr(o, name, int(bool(v)))
        o.graph = int(bool(graph))
        self.opts = o
        if legacy_create:       # rounds 1 / 2 entry point: the option bits of its `debug` argument (include/qasr.h), nothing else
            bits = o.debug | (8 if o.tile_frames == 128 else 0) | (16 if graph else 0)
            _check(lib.qasr_engine_create(C.cast(buf, C.c_void_p), len(blob), device, bits, C.byref(self._h)), 'qasr_engine_create')
        else:
            _check(lib.qasr_engine_create_ex(C.cast(buf, C.c_void_p), len(blob), device, C.byref(o), C.byref(self._h)),
                   'qasr_engine_create_ex')
        self.debug = debug
        self.n_ops = lib.qasr_engine_num_ops(self._h)
        hdr = np.frombuffer(blob[:40], dtype=np.uint32)
        self.feat_in, self.n_classes = int(hdr[4]), int(hdr[5])
        self.B = self.T = None

    def close(self):
        if getattr(self, '_h', None) and self._h.value:
            self.lib.qasr_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def out_frames(self, T):
        return self.lib.qasr_engine_out_frames(self._h, int(T))

    def num_launches(self):
        """kernel launches of one forward of the current plan (after a forward)"""
        return self.lib.qasr_engine_num_launches(self._h)

    def forward(self, feats: torch.Tensor, lens: torch.Tensor, want_logp=True, stream=None, out=None):
        """feats f32 [B, feat_in, T] (cuda, contiguous), lens [B] -> (log_probs [B,T',C], tokens [B,T'], enc_len [B])."""
        assert feats.is_cuda and feats.dtype == torch.float32 and feats.dim() == 3 and feats.shape[1] == self.feat_in
        feats = feats.contiguous()
        lens32 = lens.to(device=feats.device, dtype=torch.int32).contiguous()
        B, _, T = feats.shape
        To = self.out_frames(T)
        if out is not None:                                  # caller-owned (logp or None, tokens, enc_len): stable pointers
            logp, tokens, enc_len = out
        else:
            logp = torch.empty(B, To, self.n_classes, device=feats.device, dtype=torch.float32) if want_logp else None
            tokens = torch.empty(B, To, device=feats.device, dtype=torch.int32)
            enc_len = torch.empty(B, device=feats.device, dtype=torch.int32)
        _check(self.lib.qasr_engine_forward(self._h, _stream_ptr(stream), _ptr(feats), _ptr(lens32), B, T,
                                            _ptr(logp), _ptr(tokens), _ptr(enc_len)), 'qasr_engine_forward')
        self.B, self.T = B, T
        self._keep = (feats, lens32)        # keep inputs alive until the stream has consumed them
        return logp, tokens, enc_len

    def forward_audio(self, audio, audio_lens, fb, window, plan, preemph=0.97, pad_to=16, want_logp=True, stream=None,
                      feats=None, feat_lens=None, out=None):
        """qasr_engine_forward_audio: audio f32 [B, S] (cuda) -> (log_probs, tokens, enc_len) with the mel front-end inside the
        engine's call (one hipGraph launch per batch once the buffer set has been seen twice).  `plan` = frontend_plan(fb);
        `feats` / `feat_lens` / `out` = caller-owned buffers (stable pointers keep the captured graph)."""
        assert audio.is_cuda and audio.dtype == torch.float32 and audio.dim() == 2 and audio.is_contiguous()
        assert audio_lens.is_cuda and audio_lens.dtype == torch.int32 and fb.is_cuda and window.is_cuda
        B, S = audio.shape
        n_mels = fb.shape[0]
        T = self.lib.qasr_frontend_frames(S, pad_to)
        dev = audio.device
        if feats is None:
            feats = torch.empty(B, n_mels, T, device=dev, dtype=torch.float32)
        if feat_lens is None:
            feat_lens = torch.empty(B, device=dev, dtype=torch.int32)
        To = self.out_frames(T)
        if out is not None:
            logp, tokens, enc_len = out
        else:
            logp = torch.empty(B, To, self.n_classes, device=dev, dtype=torch.float32) if want_logp else None
            tokens = torch.empty(B, To, device=dev, dtype=torch.int32)
            enc_len = torch.empty(B, device=dev, dtype=torch.int32)
        _check(self.lib.qasr_engine_forward_audio(self._h, _stream_ptr(stream), _ptr(audio), _ptr(audio_lens), B, S, _ptr(fb),
                                                  _ptr(window), n_mels, C.c_float(preemph), pad_to, _ptr(plan), plan.numel(),
                                                  _ptr(feats), _ptr(feat_lens), _ptr(logp), _ptr(tokens), _ptr(enc_len)),
               'qasr_engine_forward_audio')
        self.B, self.T = B, T
        self._keep = (audio, audio_lens, feats, feat_lens, fb, window, plan)
        return logp, tokens, enc_len

    # ---- parity hooks (debug engines)
    def read_acc(self, op, pane, cout, T_out):
        Tp = (T_out + 63) // 64 * 64
        out = np.empty((self.B, cout, Tp), dtype=np.int32)
        _check(self.lib.qasr_engine_read_acc(self._h, op, pane, out.ctypes.data_as(C.c_void_p), out.size),
               'qasr_engine_read_acc')
        return out[:, :, :T_out]

    def read_tensor(self, tensor, channels, dtype=np.int8):
        T, Tp = C.c_int(), C.c_int()
        _check(self.lib.qasr_engine_read_tensor(self._h, tensor, None, 0, C.byref(T), C.byref(Tp)), 'read_tensor')
        out = np.empty((self.B, channels, Tp.value), dtype=dtype)
        _check(self.lib.qasr_engine_read_tensor(self._h, tensor, out.ctypes.data_as(C.c_void_p), out.nbytes,
                                                C.byref(T), C.byref(Tp)), 'qasr_engine_read_tensor')
        return out[:, :, :T.value]

    def time_ops(self, reps=20, stream=None):
        """Average duration per launch (ms) of every op, each replayed `reps` times between one HIP event pair."""
        ms = np.zeros(self.n_ops, dtype=np.float32)
        _check(self.lib.qasr_engine_time_ops(self._h, _stream_ptr(stream), reps, ms.ctypes.data_as(C.c_void_p),
                                             self.n_ops), 'qasr_engine_time_ops')
        return ms

    def op_labels(self):
        """kernel instantiation every op is routed to (names as rocprofv3 prints them)"""
        out = []
        buf = C.create_string_buffer(96)
        for op in range(self.n_ops):
            _check(self.lib.qasr_engine_op_label(self._h, op, buf, 96), 'qasr_engine_op_label')
            out.append(buf.value.decode())
        return out

    def run_op(self, op, stream=None):
        _check(self.lib.qasr_engine_run_op(self._h, _stream_ptr(stream), int(op)), 'qasr_engine_run_op')

    def last_op_ms(self):
        ms = np.zeros(self.n_ops, dtype=np.float32)
        _check(self.lib.qasr_engine_last_op_ms(self._h, ms.ctypes.data_as(C.c_void_p), self.n_ops), 'last_op_ms')
        return ms


# ---- stand-alone operators -------------------------------------------------------------------
def _rup(x, m):
    return (x + m - 1) // m * m


def pw_conv_acc(x: torch.Tensor, w: torch.Tensor, bias=None, x_unsigned=False):
    """int32 accumulator of a 1x1 conv.  x int8/uint8 [B,cin,T] cuda, w int8 [cout,cin] (host or cuda)."""
    lib = load_library()
    B, cin, T = x.shape
    cout = w.shape[0]
    Tp, cinp, coutp = _rup(T, 64), _rup(cin, 128), _rup(cout, 128)
    dev = x.device
    xp = torch.zeros(B, cin, Tp, dtype=torch.int8, device=dev)
    xp[:, :, :T] = x.view(torch.int8) if x.dtype == torch.uint8 else x
    from .pack import fragment_order
    wp = torch.zeros(coutp, cinp, dtype=torch.int8)
    wp[:cout, :cin] = w.cpu()
    wp = torch.from_numpy(fragment_order(wp.numpy())).to(dev)
    bp = torch.zeros(coutp, dtype=torch.int32, device=dev)
    if bias is not None:
        bp[:cout] = bias.to(dev)
    if x_unsigned:                          # kernels feed u8 as (x - 128): fold the correction like pack.py does
        bp[:cout] += 128 * w.to(dev).to(torch.int32).sum(1)
    acc = torch.zeros(B, cout, Tp, dtype=torch.int32, device=dev)
    _check(lib.qasr_pw_conv_acc(_stream_ptr(), _ptr(xp), int(x_unsigned), _ptr(wp), _ptr(bp), B, cin, cinp, cout, T,
                                Tp, _ptr(acc)), 'qasr_pw_conv_acc')
    return acc[:, :, :T]


def dw_conv_acc(x: torch.Tensor, w: torch.Tensor, stride=1, dilation=1, padding=0):
    """int32 accumulator of a depthwise conv.  x int8 [B,C,T] cuda (signed), w int8 [C,K]."""
    lib = load_library()
    B, Cc, T = x.shape
    K = w.shape[1]
    T_out = (T + 2 * padding - dilation * (K - 1) - 1) // stride + 1
    Tp, Tpo, kp = _rup(T, 64), _rup(T_out, 64), _rup(K, 4)
    dev = x.device
    xp = torch.zeros(B, Cc, Tp, dtype=torch.int8, device=dev)
    xp[:, :, :T] = x
    wp = torch.zeros(Cc, kp, dtype=torch.int8, device=dev)
    wp[:, :K] = w.to(dev)
    acc = torch.zeros(B, Cc, Tpo, dtype=torch.int32, device=dev)
    _check(lib.qasr_dw_conv_acc(_stream_ptr(), _ptr(xp), 0, _ptr(wp), None, B, Cc, K, kp, stride, dilation, padding, T, Tp,
                                T_out, Tpo, _ptr(acc)), 'qasr_dw_conv_acc')
    return acc[:, :, :T_out]


def requant(acc: torch.Tensor, M: torch.Tensor, lo, hi, sb=None, exact_z=False, relu=False):
    """clamp(rint(z*M[c]), lo, hi) for int32 acc [B,C,T] (T a multiple of 64 after padding)."""
    lib = load_library()
    B, Cc, T = acc.shape
    Tp = _rup(T, 64)
    a = torch.zeros(B, Cc, Tp, dtype=torch.int32, device=acc.device)
    a[:, :, :T] = acc
    out = torch.zeros(B, Cc, Tp, dtype=torch.int8, device=acc.device)
    Md = M.to(device=acc.device, dtype=torch.float64).contiguous()
    sbd = None if sb is None else sb.to(device=acc.device, dtype=torch.float32).contiguous()
    _check(lib.qasr_requant(_stream_ptr(), _ptr(a), _ptr(Md), _ptr(sbd), int(exact_z), int(relu), B, Cc, Tp, lo, hi,
                            _ptr(out)), 'qasr_requant')
    return out[:, :, :T]


def sep_layer(x, lens, wpw, bias, outs, wdw=None, m_dw=None, dw_range=(-128, 127), x_unsigned=False, dilation=1,
              flags=0, sb=None, res=None, tile=32, gen=2, hooks=True):
    """One fused separable layer through qasr_sep_layer (the production kernels with caller-made operands).

    x            int8 / uint8 [B, cin, T] (cuda)      lens  valid frames per utterance
    wdw          int8 [cin, K] depthwise taps (None: the layer is a bare 1x1 conv), m_dw f64 [cin], dw_range (lo, hi)
    wpw, bias    int8 [cout, cin], int32 [cout]       sb f32 [cout] conv output scales (QASR_F_EXACT_Z)
    outs         list of dicts(mode, lo, hi, M=f64 [cout] (mode 1) or m=float (mode 0))
    res          None or dict(x [B, rcin, T] (uint8 / int8), w int8 [cout, rcin], bias int32 [cout], m f64 [cout],
                 sb f32 [cout], m_main f64 [cout], qlo, qhi)   -> QASR_F_RESADD
    Biases are the natural ones: the +128 sum(W) correction of u8 inputs is folded here, as pack.py does.
    Returns dict(outs=[int8 [B, cout, T]...], dw_acc, acc, racc (int32, None without hooks), label)."""
    from .pack import F_RESADD, fragment_order
    lib = load_library()
    dev = x.device
    B, cin, T = x.shape
    cout = wpw.shape[0]
    Tp, cinp, coutp = _rup(T, 64), _rup(cin, 128), _rup(cout, 128)
    keep = []

    def dv(t, dtype=None):
        t = torch.as_tensor(t)
        t = (t.to(dtype) if dtype is not None else t).to(dev).contiguous()
        keep.append(t)
        return t

    def padded_x(t):
        p = torch.zeros(t.shape[0], t.shape[1], Tp, dtype=torch.int8, device=dev)
        p[:, :, :T] = t.view(torch.int8) if t.dtype == torch.uint8 else t
        keep.append(p)
        return p

    def frag(w, rows, cols):
        wp = torch.zeros(rows, cols, dtype=torch.int8)
        wp[:w.shape[0], :w.shape[1]] = torch.as_tensor(w).cpu()
        return dv(torch.from_numpy(fragment_order(wp.numpy())))

    def padvec(v, n, dtype, fill=0):
        o = torch.full((n,), fill, dtype=dtype)
        o[:len(v)] = torch.as_tensor(v).to(dtype).cpu()
        return dv(o)

    a = SepLayerArgs()
    a.B, a.T, a.Tp, a.cin, a.cout, a.dilation, a.tile, a.gen = B, T, Tp, cin, cout, dilation, tile, gen
    a.flags = flags | (F_RESADD if res is not None else 0)
    a.x = padded_x(x).data_ptr()
    a.x_unsigned = int(x_unsigned)
    a.lens = dv(lens, torch.int32).data_ptr()
    if wdw is not None:
        wdw = torch.as_tensor(wdw).cpu().to(torch.int8)
        K = wdw.shape[1]
        kp = _rup(K, 4)
        w1 = torch.zeros(cin, kp, dtype=torch.int8)
        w1[:, :K] = wdw
        w2 = torch.zeros(cin, kp + 32, dtype=torch.int8)
        w2[:, 8:8 + K] = wdw
        a.K = K
        a.wdw, a.wdw2 = dv(w1).data_ptr(), dv(torch.cat([w2.view(-1), torch.zeros(64, dtype=torch.int8)])).data_ptr()
        bdw = 128 * wdw.to(torch.int32).sum(1) if x_unsigned else torch.zeros(cin, dtype=torch.int32)
        a.bias_dw = padvec(bdw, cinp, torch.int32).data_ptr()
        a.m_dw = padvec(m_dw, cinp, torch.float64).data_ptr()
        a.dw_lo, a.dw_hi = dw_range
        pw_bias = torch.as_tensor(bias).to(torch.int32).cpu()
    else:
        a.K = 0
        pw_bias = torch.as_tensor(bias).to(torch.int32).cpu()
        if x_unsigned:
            pw_bias = pw_bias + 128 * torch.as_tensor(wpw).cpu().to(torch.int32).sum(1)
    a.w = frag(wpw, coutp, cinp).data_ptr()
    a.bias = padvec(pw_bias, coutp, torch.int32).data_ptr()
    if sb is not None:
        a.sb = padvec(sb, coutp, torch.float32, 1.0).data_ptr()
    if res is not None:
        rcin = res['x'].shape[1]
        a.rx = padded_x(res['x']).data_ptr()
        a.r_unsigned = int(res['x'].dtype == torch.uint8)
        a.rcin = rcin
        a.rw = frag(res['w'], coutp, _rup(rcin, 128)).data_ptr()
        rb = torch.as_tensor(res['bias']).to(torch.int32).cpu()
        if a.r_unsigned:
            rb = rb + 128 * torch.as_tensor(res['w']).cpu().to(torch.int32).sum(1)
        a.rbias = padvec(rb, coutp, torch.int32).data_ptr()
        a.rm = padvec(res['m'], coutp, torch.float64).data_ptr()
        if res.get('sb') is not None:
            a.rsb = padvec(res['sb'], coutp, torch.float32, 1.0).data_ptr()
        a.m_main = padvec(res['m_main'], coutp, torch.float64).data_ptr()
        a.qlo, a.qhi = res['qlo'], res['qhi']
    a.n_outs = len(outs)
    out_t = []
    for j, o in enumerate(outs):
        t = torch.zeros(B, cout, Tp, dtype=torch.int8, device=dev)
        out_t.append(t)
        a.outs[j].ptr = t.data_ptr()
        a.outs[j].mode, a.outs[j].lo, a.outs[j].hi = o['mode'], o['lo'], o['hi']
        if o['mode'] == 1:
            a.outs[j].mtab = padvec(o['M'], coutp, torch.float64).data_ptr()
        elif o['mode'] == 0:
            a.outs[j].m = float(o['m'])
    hk = {}
    if hooks:
        if wdw is not None:
            hk['dw_acc'] = torch.zeros(B, cin, Tp, dtype=torch.int32, device=dev)
            a.dw_acc = hk['dw_acc'].data_ptr()
        hk['acc'] = torch.zeros(B, cout, Tp, dtype=torch.int32, device=dev)
        a.acc = hk['acc'].data_ptr()
        if res is not None:
            hk['racc'] = torch.zeros(B, cout, Tp, dtype=torch.int32, device=dev)
            a.racc = hk['racc'].data_ptr()
    label = C.create_string_buffer(96)
    _check(lib.qasr_sep_layer(_stream_ptr(), C.byref(a), label, 96), 'qasr_sep_layer')
    torch.cuda.synchronize()
    return dict(outs=[t[:, :, :T] for t in out_t], label=label.value.decode(),
                **{k: (hk[k][:, :, :T] if k in hk else None) for k in ('dw_acc', 'acc', 'racc')})


def frontend_plan(fb: torch.Tensor):
    """qasr_frontend_plan: the filterbank-only tables of the front-end (filter runs, packed weights, twiddles) as a
    workspace tensor for frontend_mel(..., plan=...).  Read-only afterwards; one plan serves any number of streams."""
    lib = load_library()
    assert fb.is_cuda and fb.dtype == torch.float32 and fb.is_contiguous() and fb.dim() == 2 and fb.shape[1] == 257
    ws = torch.empty(max(lib.qasr_frontend_workspace_bytes(0, 0, fb.shape[0]), 16), dtype=torch.uint8, device=fb.device)
    _check(lib.qasr_frontend_plan(_stream_ptr(), _ptr(fb), fb.shape[0], _ptr(ws), ws.numel()), 'qasr_frontend_plan')
    ws._qasr_fb = fb                                          # the table is only valid for this filterbank: keep it alive
    return ws


def frontend_mel(audio: torch.Tensor, lens: torch.Tensor, fb: torch.Tensor, window: torch.Tensor, preemph=0.97,
                 pad_to=16, out=None, plan=None):
    """qasr_frontend_mel: audio f32 [B,S] (cuda), lens [B] samples, fb [n_mels,257], window [320]
    -> (features f32 [B,n_mels,T_pad], feature lengths i32 [B]).
    out = caller-owned (features, lengths, workspace); plan = frontend_plan(fb) of the same filterbank: skips the
    per-call table build (qasr_frontend_mel_planned)."""
    lib = load_library()
    assert audio.is_cuda and audio.dtype == torch.float32 and audio.dim() == 2
    B, S = audio.shape
    n_mels = fb.shape[0]
    dev = audio.device
    T_pad = lib.qasr_frontend_frames(S, pad_to)
    if out is not None:
        feats, flens, ws_out = out
    else:
        feats = torch.empty(B, n_mels, T_pad, device=dev, dtype=torch.float32)
        flens = torch.empty(B, device=dev, dtype=torch.int32)
        ws_out = None
    a = audio.contiguous()
    l32 = lens.to(device=dev, dtype=torch.int32).contiguous()
    fbd = fb.to(device=dev, dtype=torch.float32).contiguous()
    wd = window.to(device=dev, dtype=torch.float32).contiguous()
    if plan is not None:
        _check(lib.qasr_frontend_mel_planned(_stream_ptr(), _ptr(a), _ptr(l32), B, S, _ptr(fbd), _ptr(wd), n_mels,
                                             C.c_float(preemph), pad_to, _ptr(feats), _ptr(flens), _ptr(plan), plan.numel()),
               'qasr_frontend_mel_planned')
        return feats, flens
    ws_bytes = lib.qasr_frontend_workspace_bytes(B, S, n_mels)
    ws = ws_out if ws_out is not None else torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
    _check(lib.qasr_frontend_mel(_stream_ptr(), _ptr(a), _ptr(l32), B, S, _ptr(fbd), _ptr(wd), n_mels,
                                 C.c_float(preemph), pad_to, _ptr(feats), _ptr(flens), _ptr(ws), ws.numel()),
           'qasr_frontend_mel')
    return feats, flens


def quantile2(x: torch.Tensor, q_lo: float, q_hi: float):
    """qasr_quantile2: (torch.quantile(x.flatten(), q_lo), torch.quantile(x.flatten(), q_hi)) of a float32 cuda tensor
    as one radix select on device; returns a 2-element float32 cuda tensor (no host synchronisation)."""
    lib = load_library()
    assert x.is_cuda and x.dtype == torch.float32
    flat = x.detach().contiguous().view(-1)
    out = torch.empty(2, device=x.device, dtype=torch.float32)
    ws = torch.empty(lib.qasr_quantile_workspace_bytes(), dtype=torch.uint8, device=x.device)
    _check(lib.qasr_quantile2(_stream_ptr(), _ptr(flat), flat.numel(), C.c_float(q_lo), C.c_float(q_hi), _ptr(out), _ptr(ws),
                              ws.numel()), 'qasr_quantile2')
    return out
