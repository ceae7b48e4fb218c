"""Per-op launch time and MFMA fraction of Jasper10x5dr bs64 (qasr_engine_time_ops; output: profiles/r03_v1_jasper_ops.txt; the\n"mfma frac" column is x 1000 and counts the main conv only)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'q-asr_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
from qasr import engine, pack, synth, topology
d = np.load(os.path.join(ROOT, 'tests/golden/net_jasper_w8a8.npz'))
meta = json.loads(str(d['meta']))
cfg = topology.jasper10x5dr()
sd = synth.make_state_dict(cfg, meta['seed'])
blob, pm = pack.pack_model(cfg, sd, d['act_min'], d['act_max'], 8, 8)
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 128
e = engine.Engine(blob, 0, tile=tile)
B, T = 64, 512
x = torch.from_numpy(synth.make_features(B, 64, T, 1)).cuda()
lens = torch.full((B,), 500)
for _ in range(2):
    e.forward(x, lens)
torch.cuda.synchronize()
ms = e.time_ops(reps=10)
labels = e.op_labels()
import struct
# op table: kind, flags, in, cin, cout, kernel ...
hdr = np.frombuffer(blob[:40], dtype=np.uint32)
n_ops = int(hdr[3]); op_size = int(hdr[9])
ops_off = struct.unpack_from('<Q', blob, 48)[0]
tot = 0
for oi in range(n_ops):
    kind, flags, tin, cin, cout, kernel, stride, dil, padd, npanes = struct.unpack_from('<IIiIIIIIII', blob, ops_off + oi * op_size)
    if ms[oi] <= 0 or kind not in (2, 3):
        continue
    gop = 2.0 * B * 256 * cin * cout * max(kernel, 1) / 1e9
    print(f'op {oi:3d} {labels[oi]:34s} cin {cin:4d} cout {cout:4d} k {kernel:2d} d {dil} panes {npanes:2d}  {ms[oi]*1e3:8.1f} us  {gop:7.1f} GOP  mfma frac {gop/ (ms[oi]*1e-3) / 5033e3 * 1e3:5.3f}')
    tot += ms[oi]
print('sum', tot)
e.close()
