#!/usr/bin/env python3
"""Per-work-group timeline of one k_sep2 launch (qasr_debug_timeline: every work-group stamps its start and end with
the 100 MHz s_memrealtime counter and its HW_ID / XCC_ID): how much of a launch's duration is work-group execution and
how much is ramp-up, placement and drain.  Usage: python profiles/wg_timeline.py [--tile 32|64|128]"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'q-asr_amd'))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--tile', type=int, default=64)
ap.add_argument('--batch', type=int, default=32)
args = ap.parse_args()
from qasr import engine, pack, synth, topology  # noqa: E402

d = np.load(os.path.join(ROOT, 'tests/golden/net_quartznet_w8a8.npz'))
meta = json.loads(str(d['meta']))
cfg = topology.quartznet15x5()
sd = synth.make_state_dict(cfg, meta['seed'])
blob, pm = pack.pack_model(cfg, sd, d['act_min'], d['act_max'], 8, 8)
e = engine.Engine(blob, 0, tile=args.tile, sep_gen=getattr(args, "gen", 2))
B, T = args.batch, 512
x = torch.from_numpy(synth.make_features(B, 64, T, 1)).cuda()
lens = torch.full((B,), 500)
for _ in range(3):
    e.forward(x, lens)
torch.cuda.synchronize()
lib = engine.load_library()
buf = torch.zeros(4 * 4096, dtype=torch.int64, device='cuda')
labels = e.op_labels()
seen = {}
for oi, lab in enumerate(labels):
    if lab.startswith('k_sep2'):
        seen[lab] = oi
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for lab, oi in seen.items():
    lib.qasr_debug_timeline(C.c_void_p(0), 0)
    for _ in range(3):
        e.run_op(oi)
    ev[0].record()
    for _ in range(20):
        e.run_op(oi)
    ev[1].record()
    torch.cuda.synchronize()
    per_launch = ev[0].elapsed_time(ev[1]) / 20 * 1e3
    buf.zero_()
    lib.qasr_debug_timeline(C.c_void_p(buf.data_ptr()), buf.numel() // 4)
    e.run_op(oi)                                              # previous launch of the same op just ended: warm caches
    e.run_op(oi)
    torch.cuda.synchronize()
    st = buf.cpu().numpy().reshape(-1, 4)
    st = st[st[:, 1] > 0]
    t0 = st[:, 0].min()
    start, end = (st[:, 0] - t0) / 100.0, (st[:, 1] - t0) / 100.0      # us
    dur = end - start
    xcc = (st[:, 2] >> 32) & 15
    cu = ((st[:, 2] & 0xffffffff) >> 8) & 15
    se = ((st[:, 2] & 0xffffffff) >> 13) & 7
    slots = len(set(zip(xcc.tolist(), se.tolist(), cu.tolist())))
    print(f'{lab:36s} {len(st):4d} WGs on {slots:3d} (xcc, se, cu) slots | launch {per_launch:6.2f} us (events, back to back) | '
          f'first start -> last end {end.max():6.2f} | starts spread {start.max():5.2f} (p50 {np.median(start):5.2f}) | '
          f'WG duration min {dur.min():5.2f} p50 {np.median(dur):5.2f} max {dur.max():5.2f} | clock {np.median(st[:, 3] / (dur * 1e3)):.2f} GHz')
lib.qasr_debug_timeline(C.c_void_p(0), 0)
e.close()
