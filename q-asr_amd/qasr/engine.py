"""ctypes binding of libqasr_hip.so (include/qasr.h).  PyTorch is used only to own device
memory and streams.  There is no CPU fallback: a missing library or a failing call raises."""
import ctypes as C
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('QASR_LIB', os.path.join(HERE, 'libqasr_hip.so'))   # QASR_LIB: A/B builds in one run

SYMBOLS = ['qasr_engine_create', 'qasr_engine_destroy', 'qasr_engine_forward', 'qasr_engine_out_frames',
           'qasr_engine_num_ops', 'qasr_engine_read_acc', 'qasr_engine_read_tensor', 'qasr_engine_last_op_ms',
           'qasr_engine_time_ops', 'qasr_engine_run_op', 'qasr_engine_op_label',
           'qasr_frontend_mel', 'qasr_frontend_frames', 'qasr_frontend_workspace_bytes', 'qasr_pw_conv_acc',
           'qasr_dw_conv_acc', 'qasr_requant', 'qasr_quantile2', 'qasr_quantile_workspace_bytes', 'qasr_debug_prof',
           'qasr_last_error', 'qasr_version']

_lib = None


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
    lib.qasr_engine_create.argtypes = [vp, sz, i32, i32, C.POINTER(vp)]
    lib.qasr_engine_destroy.argtypes = [vp]
    lib.qasr_engine_destroy.restype = None
    lib.qasr_engine_forward.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp, vp]
    lib.qasr_engine_out_frames.argtypes = [vp, i32]
    lib.qasr_engine_num_ops.argtypes = [vp]
    lib.qasr_engine_read_acc.argtypes = [vp, i32, i32, vp, sz]
    lib.qasr_engine_read_tensor.argtypes = [vp, i32, vp, sz, C.POINTER(i32), C.POINTER(i32)]
    lib.qasr_engine_last_op_ms.argtypes = [vp, vp, i32]
    lib.qasr_engine_time_ops.argtypes = [vp, vp, i32, vp, i32]
    lib.qasr_engine_run_op.argtypes = [vp, vp, i32]
    lib.qasr_engine_op_label.argtypes = [vp, i32, C.c_char_p, sz]
    lib.qasr_frontend_mel.argtypes = [vp, vp, vp, i32, i32, vp, vp, i32, C.c_float, i32, vp, vp, vp, sz]
    lib.qasr_frontend_frames.argtypes = [i32, i32]
    lib.qasr_frontend_workspace_bytes.argtypes = [i32, i32, i32]
    lib.qasr_frontend_workspace_bytes.restype = sz
    lib.qasr_pw_conv_acc.argtypes = [vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.qasr_dw_conv_acc.argtypes = [vp, vp, i32, vp] + [i32] * 11 + [vp]
    lib.qasr_requant.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.qasr_debug_prof.argtypes = [vp]
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


def _stream_ptr(stream=None):
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


def _ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


class Engine:
    """One packed model on one GPU (qasr_engine_*)."""

    def __init__(self, blob: bytes, device=0, debug=False, timing=False, whole_utterance=False, wide_tiles=False,
                 graph=False):
        lib = load_library()
        if not torch.cuda.is_available():
            raise QasrError('no GPU: the integer engine needs an MI355X (there is no CPU fallback)')
        self.lib = lib
        self.device = torch.device('cuda', device)
        self._blob = blob
        self._h = C.c_void_p()
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        _check(lib.qasr_engine_create(C.cast(buf, C.c_void_p), len(blob), device, int(bool(debug)) | (2 if timing else 0) | (4 if whole_utterance else 0) | (8 if wide_tiles else 0) | (16 if graph else 0),
                                      C.byref(self._h)),
               'qasr_engine_create')
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
    _check(lib.qasr_dw_conv_acc(_stream_ptr(), _ptr(xp), 0, _ptr(wp), B, Cc, K, kp, stride, dilation, padding, T, Tp,
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


def frontend_mel(audio: torch.Tensor, lens: torch.Tensor, fb: torch.Tensor, window: torch.Tensor, preemph=0.97,
                 pad_to=16, out=None):
    """qasr_frontend_mel: audio f32 [B,S] (cuda), lens [B] samples, fb [n_mels,257], window [320]
    -> (features f32 [B,n_mels,T_pad], feature lengths i32 [B])."""
    lib = load_library()
    assert audio.is_cuda and audio.dtype == torch.float32 and audio.dim() == 2
    B, S = audio.shape
    n_mels = fb.shape[0]
    dev = audio.device
    T_pad = lib.qasr_frontend_frames(S, pad_to)
    if out is not None:                                      # caller-owned (features, lengths, workspace)
        feats, flens, ws_out = out
    else:
        feats = torch.empty(B, n_mels, T_pad, device=dev, dtype=torch.float32)
        flens = torch.empty(B, device=dev, dtype=torch.int32)
        ws_out = None
    a = audio.contiguous()
    l32 = lens.to(device=dev, dtype=torch.int32).contiguous()
    fbd = fb.to(device=dev, dtype=torch.float32).contiguous()
    wd = window.to(device=dev, dtype=torch.float32).contiguous()
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
