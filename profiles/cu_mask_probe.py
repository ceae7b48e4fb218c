#!/usr/bin/env python3
"""Where do the work-groups of a k_sep2 launch land when its stream carries a CU mask (hipExtStreamCreateWithCUMask)?
For several mask patterns: the (XCC, SE, CU) slots of the 64 work-groups of `k_sep2<75, 4, 0, 2, false, 128, 1>` and the launch
duration.  Purpose: can four launch chains be given disjoint quarters of the chip (no competition for CUs at kernel boundaries)?"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'q-asr_amd'))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from qasr import engine, pack, synth, topology  # noqa: E402

hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so'))
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipGetErrorString.restype = C.c_char_p
hip.hipGetErrorString.argtypes = [C.c_int]


def masked_stream(bits):
    words = (max(bits) // 32) + 1 if bits else 1
    arr = (C.c_uint32 * words)()
    for b in bits:
        arr[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), words, arr)
    if rc:
        raise RuntimeError(f'hipExtStreamCreateWithCUMask: {hip.hipGetErrorString(rc).decode()}')
    return torch.cuda.ExternalStream(s.value)


d = np.load(os.path.join(ROOT, 'tests/golden/net_quartznet_w8a8.npz'))
meta = json.loads(str(d['meta']))
cfg = topology.quartznet15x5()
blob, pm = pack.pack_model(cfg, synth.make_state_dict(cfg, meta['seed']), d['act_min'], d['act_max'], 8, 8)
e = engine.Engine(blob, 0, tile=128)
x = torch.from_numpy(synth.make_features(32, 64, 512, 1)).cuda()
lens = torch.full((32,), 500)
for _ in range(3):
    e.forward(x, lens)
torch.cuda.synchronize()
lib = engine.load_library()
oi = next(i for i, l in enumerate(e.op_labels()) if l.startswith('k_sep2<75, 4, 0, 2'))
buf = torch.zeros(4 * 4096, dtype=torch.int64, device='cuda')
n_cu = torch.cuda.get_device_properties(0).multi_processor_count
print(f'# {torch.cuda.get_device_name(0)}: {n_cu} CUs; op {oi} = {e.op_labels()[oi]}')
patterns = {
    'no mask (torch stream)': None,
    'bits 0..63': list(range(64)),
    'bits 64..127': list(range(64, 128)),
    'bits 0..255 step 4': list(range(0, 256, 4)),
    'bits 1..255 step 4': list(range(1, 256, 4)),
    'bits b with (b % 8) < 2': [b for b in range(256) if b % 8 < 2],
    'bits b with (b // 32) < 2': [b for b in range(256) if b // 32 < 2],
    'bits 0..127': list(range(128)),
    'bits 0..7': list(range(8)),
    'bits 0..3': list(range(4)),
    'bits 8..15': list(range(8, 16)),
    'bits 0..31': list(range(32)),
    'bits 32..63': list(range(32, 64)),
    'bits 0..15 + 32..47': list(range(16)) + list(range(32, 48)),
}
for name, bits in patterns.items():
    st_ = torch.cuda.Stream() if bits is None else masked_stream(bits)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    with torch.cuda.stream(st_):
        for _ in range(3):
            e.run_op(oi, stream=st_)
        ev[0].record(st_)
        for _ in range(20):
            e.run_op(oi, stream=st_)
        ev[1].record(st_)
    torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) / 20 * 1e3
    buf.zero_()
    lib.qasr_debug_timeline(C.c_void_p(buf.data_ptr()), buf.numel() // 4)
    e.run_op(oi, stream=st_)
    torch.cuda.synchronize()
    lib.qasr_debug_timeline(C.c_void_p(0), 0)
    t = buf.cpu().numpy().reshape(-1, 4)
    t = t[t[:, 1] > 0]
    xcc = (t[:, 2] >> 32) & 15
    cu = ((t[:, 2] & 0xffffffff) >> 8) & 15
    se = ((t[:, 2] & 0xffffffff) >> 13) & 7
    slots = sorted(set(zip(xcc.tolist(), se.tolist(), cu.tolist())))
    per_xcc = {int(k): int((xcc == k).sum()) for k in sorted(set(xcc.tolist()))}
    dur = (t[:, 1] - t[:, 0]) / 100.0
    print(f'{name:28s} launch {us:6.2f} us | {len(t)} WGs on {len(slots)} slots | per XCC {per_xcc} | WG p50 {np.median(dur):5.2f} us, last end {((t[:, 1] - t[:, 0].min()) / 100.0).max():6.2f}')
    secu = sorted(set((a, b) for _, a, b in slots))
    print('    distinct (se, cu) over all XCCs:', len(secu), secu if len(secu) <= 40 else str(secu[:40]) + ' ...')
e.close()
