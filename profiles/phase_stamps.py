#!/usr/bin/env python3
"""Phase stamps of one work-group of the fused separable-layer kernels (qasr_debug_prof: s_memtime at the phase
boundaries of work-group (0, 1), thread 0).  Usage: python profiles/phase_stamps.py [--tile 32|64] [--gen 1|2]
Prints, per selected op, the cycle deltas between consecutive stamps: k_sep2 = start | per 256-channel chunk: window
committed, depthwise done | after the post-depthwise barrier(s) | per 256-channel pass: main GEMM, (residual GEMM,) epilogue."""
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
ap.add_argument('--tile', type=int, default=32, choices=[32, 64, 128])
ap.add_argument('--gen', type=int, default=2)
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
buf = torch.zeros(32, dtype=torch.int64, device='cuda')
lib.qasr_debug_prof(C.c_void_p(buf.data_ptr()))
labels = e.op_labels()
seen = {}
for oi, lab in enumerate(labels):
    if lab.startswith('k_sep'):
        seen[lab] = oi                                        # last op of every instantiation
for lab, oi in seen.items():
    for _ in range(3):
        e.run_op(oi)
    torch.cuda.synchronize()
    st = buf.cpu().numpy()
    n = int(st[31])
    deltas = [int(st[i + 1] - st[i]) for i in range(n - 1)]
    print(f'{lab:34s} op {oi:3d} stamps {n:2d} total {int(st[n - 1] - st[0]):6d}  deltas {deltas}')
lib.qasr_debug_prof(C.c_void_p(0))
e.close()
