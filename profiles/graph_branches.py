"""Two (or more) 32-utterance steps as parallel branches of ONE hipGraph per stream: does the runtime run the branches side by
side, so that 4 hardware queues carry 8+ steps in flight and the dispatcher has work-groups to back-fill kernel boundaries
with?  Engines are created without their own graph replay; torch.cuda.CUDAGraph captures `branches` engines' forwards forked
from / joined into the capture stream.  usage: python profiles/graph_branches.py [branches=2] [streams=4] [rounds=25]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'q-asr_amd'))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from qasr import engine, synth  # noqa: E402

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 2
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ROUNDS = int(sys.argv[3]) if len(sys.argv) > 3 else 25
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
blob, meta, fb, window, _, _ = bench.build_model(dev, 'QuartzNet15x5Base-En', 8, 8)
fb, window = fb.to(dev), window.to(dev)
lib = engine.load_library()
B, SAMPLES = 32, 80000
T_pad = lib.qasr_frontend_frames(SAMPLES, 16)
plan = engine.frontend_plan(fb)
alen = torch.full((B,), SAMPLES, dtype=torch.int32, device=dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
side = [[torch.cuda.Stream(device=dev) for _ in range(NB - 1)] for _ in range(S)]
lanes = []
for k in range(S):
    br = []
    for j in range(NB):
        e = engine.Engine(blob, 0, tile=128, graph=False)
        audio = torch.from_numpy(synth.make_audio(B, SAMPLES, seed=100 + 8 * k + j)).to(dev)
        feats = torch.empty(B, 64, T_pad, device=dev)
        flen = torch.empty(B, dtype=torch.int32, device=dev)
        T_out = e.out_frames(T_pad)
        out = (None, torch.empty(B, T_out, dtype=torch.int32, device=dev), torch.empty(B, dtype=torch.int32, device=dev))
        br.append(dict(e=e, audio=audio, feats=feats, flen=flen, out=out))
    lanes.append(br)


def fwd(x, stream):
    x['e'].forward_audio(x['audio'], alen, fb, window, plan, 0.97, 16, want_logp=False, stream=stream, feats=x['feats'],
                         feat_lens=x['flen'], out=x['out'])


ref = []
for k in range(S):                                           # direct launches once (one-time kernel attribute setup), reference tokens
    for x in lanes[k]:
        with torch.cuda.stream(streams[k]):
            fwd(x, streams[k])
        torch.cuda.synchronize()
        ref.append(x['out'][1].clone())
graphs = []
for k in range(S):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=streams[k]):
        for j, x in enumerate(lanes[k]):
            if j == 0:
                continue
            side[k][j - 1].wait_stream(streams[k])           # fork
            with torch.cuda.stream(side[k][j - 1]):
                fwd(x, side[k][j - 1])
        fwd(lanes[k][0], streams[k])
        for j in range(1, NB):
            streams[k].wait_stream(side[k][j - 1])           # join
    graphs.append(g)
torch.cuda.synchronize()


def region(rounds):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(rounds):
        for k in range(S):
            with torch.cuda.stream(streams[k]):
                graphs[k].replay()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


region(3)
dt = region(ROUNDS)
i = 0
for k in range(S):
    for x in lanes[k]:
        assert torch.equal(x['out'][1], ref[i]), (k, i)
        i += 1
steps = ROUNDS * S * NB
print(f'{S} streams x graph of {NB} branch(es): {steps} steps of 32 utterances in {1e3 * dt:.2f} ms = {1e3 * dt / steps:.3f} ms per step')
