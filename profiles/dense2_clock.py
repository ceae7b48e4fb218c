import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # (run from profiles/)
sys.path.insert(0, os.path.join(ROOT, 'q-asr_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
from qasr import engine, pack, synth, topology
d = np.load(os.path.join(ROOT, 'tests/golden/net_jasper_w8a8.npz'))
meta = json.loads(str(d['meta']))
cfg = topology.jasper10x5dr()
sd = synth.make_state_dict(cfg, meta['seed'])
blob, pm = pack.pack_model(cfg, sd, d['act_min'], d['act_max'], 8, 8)
S = 4
engs = [engine.Engine(blob, 0, tile=128) for _ in range(S)]
streams = [torch.cuda.Stream() for _ in range(S)]
B, T = 64, 512
x = torch.from_numpy(synth.make_features(B, 64, T, 1)).cuda()
lens = torch.full((B,), 500)
for e in engs:
    e.forward(x, lens)
torch.cuda.synchronize()
lib = engine.load_library()
buf = torch.zeros(4 * 8192, dtype=torch.int64, device='cuda')
labels = engs[0].op_labels()
for oi in (23, 43, 51, 52):
    for n in (1, 4):
        lib.qasr_debug_timeline(C.c_void_p(0), 0)
        for r in range(12):
            if r == 8:
                torch.cuda.synchronize(); buf.zero_(); lib.qasr_debug_timeline(C.c_void_p(buf.data_ptr()), buf.numel() // 4)
            for k in range(n):
                engs[k].run_op(oi, stream=streams[k])
        torch.cuda.synchronize()
        st = buf.cpu().numpy().reshape(-1, 4)
        st = st[st[:, 1] > 0]
        dur = (st[:, 1] - st[:, 0]) / 100.0
        print(f'op {oi} {labels[oi]} {n} launch(es) in flight: {len(st)} WGs, WG duration p50 {np.median(dur):.1f} us, cycles p50 {np.median(st[:,3]):.0f}, clock {np.median(st[:,3]/(dur*1e3)):.2f} GHz')
lib.qasr_debug_timeline(C.c_void_p(0), 0)
