"""Launch time of every QuartzNet15x5 op at 32- / 64- / 128-frame tiles, one step in flight (output: profiles/r03_v1_tile_times.txt)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'q-asr_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
from qasr import engine, pack, synth, topology
d = np.load(os.path.join(ROOT, 'tests/golden/net_quartznet_w8a8.npz'))
meta = json.loads(str(d['meta']))
cfg = topology.quartznet15x5()
sd = synth.make_state_dict(cfg, meta['seed'])
blob, pm = pack.pack_model(cfg, sd, d['act_min'], d['act_max'], 8, 8)
B, T = 32, 512
x = torch.from_numpy(synth.make_features(B, 64, T, 1)).cuda()
lens = torch.full((B,), 500)
res = {}
for tile in (32, 64, 128):
    e = engine.Engine(blob, 0, tile=tile)
    for _ in range(2):
        e.forward(x, lens)
    torch.cuda.synchronize()
    ms = e.time_ops(reps=20)
    res[tile] = (ms, e.op_labels())
    e.close()
tot = {t: 0.0 for t in res}
best = 0.0
seen = {}
for oi in range(len(res[32][0])):
    m = [res[t][0][oi] for t in (32, 64, 128)]
    if max(m) <= 0:
        continue
    lab = res[32][1][oi]
    key = lab.rsplit(',', 1)[0]
    seen.setdefault(key, []).append(m)
    for t, v in zip((32, 64, 128), m):
        tot[t] += v
    best += min(m)
for key, v in seen.items():
    a = np.mean(np.array(v), axis=0) * 1e3
    print(f'{key:30s} x{len(v):2d}   32: {a[0]:6.2f} us   64: {a[1]:6.2f}   128: {a[2]:6.2f}')
print('serial sums (ms):', {t: round(float(v), 4) for t, v in tot.items()}, 'best per op:', round(float(best), 4))
