"""clock of the work-groups of k_sep2<75,...> with 1 and with 4 launches in flight"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'q-asr_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
from qasr import engine, pack, synth, topology
d = np.load(os.path.join(ROOT, 'tests/golden/net_quartznet_w8a8.npz'))
meta = json.loads(str(d['meta']))
cfg = topology.quartznet15x5()
sd = synth.make_state_dict(cfg, meta['seed'])
blob, pm = pack.pack_model(cfg, sd, d['act_min'], d['act_max'], 8, 8)
S = 4
engs = [engine.Engine(blob, 0, tile=128) for _ in range(S)]
streams = [torch.cuda.Stream() for _ in range(S)]
x = torch.from_numpy(synth.make_features(32, 64, 512, 1)).cuda()
lens = torch.full((32,), 500)
for e in engs:
    for _ in range(2):
        e.forward(x, lens)
torch.cuda.synchronize()
lib = engine.load_library()
buf = torch.zeros(4 * 4096, dtype=torch.int64, device='cuda')
labels = engs[0].op_labels()
ops = [i for i, l in enumerate(labels) if l.startswith('k_sep2<75, 4, 0')]
for n in (1, 2, 3, 4):
    lib.qasr_debug_timeline(C.c_void_p(0), 0)
    for r in range(30):
        if r == 20:
            torch.cuda.synchronize(); buf.zero_(); lib.qasr_debug_timeline(C.c_void_p(buf.data_ptr()), buf.numel() // 4)
        for o in ops:
            for k in range(n):
                engs[k].run_op(o, stream=streams[k])
    torch.cuda.synchronize()
    st = buf.cpu().numpy().reshape(-1, 4)
    st = st[st[:, 1] > 0]
    dur = (st[:, 1] - st[:, 0]) / 100.0
    print(f'{n} launches in flight: WG duration p50 {np.median(dur):.2f} us, cycles p50 {np.median(st[:,3]):.0f}, clock {np.median(st[:,3]/(dur*1e3)):.2f} GHz')
lib.qasr_debug_timeline(C.c_void_p(0), 0)
