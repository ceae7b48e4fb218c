"""Where k_mel's time goes: builds qasr_frontend.hip with -DMEL_CUT=<mask> (phases left out: 1 framing, 2 FFT stages,
4 even/odd split + power, 8 mel projection, 16 the whole per-frame loop) and times qasr_frontend_mel on the bench shape (32 x 80000 samples).
Run on the GPU box from the repo root:  python profiles/microbench/mel_cut.py  > gpurun_out/mel_cut.txt"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, 'q-asr_amd', 'csrc')
masks = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 4, 8, 15, 31]
B, S, NM = 32, 80000, 64
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
audio = (torch.randn(B, S, generator=g) * 0.1).to(dev)
lens = torch.full((B,), S, dtype=torch.int32, device=dev)
fb = torch.zeros(NM, 257)
edges = np.linspace(1, 256, NM + 2)
for m in range(NM):                                             # triangular stand-in with QuartzNet-like run lengths
    lo, c, hi = edges[m] ** 1.0, edges[m + 1], edges[m + 2]
    for k in range(257):
        if lo < k < hi:
            fb[m, k] = (k - lo) / (c - lo) if k <= c else (hi - k) / (hi - c)
fb = fb.to(dev)
win = torch.hann_window(320, periodic=False).to(dev)
tmp = tempfile.mkdtemp()
for mask in masks:
    so = os.path.join(tmp, f'fe{mask}.so')
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC',
                           f'-DMEL_CUT={mask}', '-I', os.path.join(ROOT, 'include'), '-o', so,
                           os.path.join(SRC, 'qasr_frontend.hip')])
    lib = C.CDLL(so)
    lib.qasr_frontend_frames.restype = C.c_int
    lib.qasr_frontend_workspace_bytes.restype = C.c_size_t
    T = lib.qasr_frontend_frames(S, 16)
    wsb = lib.qasr_frontend_workspace_bytes(B, S, NM)
    feats = torch.empty(B, NM, T, device=dev)
    fl = torch.empty(B, dtype=torch.int32, device=dev)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream()

    rc = lib.qasr_frontend_plan(C.c_void_p(st.cuda_stream), C.c_void_p(fb.data_ptr()), NM, C.c_void_p(ws.data_ptr()), C.c_size_t(wsb))
    assert rc == 0, rc

    def run():
        rc = lib.qasr_frontend_mel_planned(C.c_void_p(st.cuda_stream), C.c_void_p(audio.data_ptr()), C.c_void_p(lens.data_ptr()), B, S,
                                   C.c_void_p(fb.data_ptr()), C.c_void_p(win.data_ptr()), NM, C.c_float(0.97), 16,
                                   C.c_void_p(feats.data_ptr()), C.c_void_p(fl.data_ptr()), C.c_void_p(ws.data_ptr()),
                                   C.c_size_t(wsb))
        assert rc == 0, rc
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        run()
    b.record()
    torch.cuda.synchronize()
    print(f'MEL_CUT={mask:2d}: {a.elapsed_time(b) / 50 * 1e3:8.1f} us per qasr_frontend_mel call (k_mel + k_norm)', flush=True)
