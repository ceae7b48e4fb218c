#!/usr/bin/env python3
"""Headline benchmark: RTFx of the quantised hot path (BASELINE.json config 2).

One "step" = one pass of the hot path over one batch resident in HBM:
    synthetic 16 kHz audio [32, 80000] (5 s each -> 500 valid mel frames)
    -> HIP mel front-end -> integer QuartzNet15x5 encoder (w8a8, percentile 99.996 calibration)
    -> CTC decoder -> log-softmax + greedy argmax tokens            (all in libqasr_hip.so)
With --gpus N (launched by torch.distributed.run, one rank per GPU) every rank runs its own 32-utterance
shard (weak scaling); rank 0 calibrates + packs the model and broadcasts the packed int weights over RCCL,
and each step ends with an RCCL gather of the greedy tokens to rank 0.

Prints ONE JSON line on rank 0.  `roofline` is measured live with HIP events on the launch stream (each op
replayed 20x between one event pair) for the kernel instantiation with the largest share of the step (a k_sep<K,...> fused separable layer); `cpu_baseline`
times the reference's fake-quant CPU op sequence (oracle/fakequant_torch.py) on the host cores (N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, 'q-asr_amd'), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

# steps in flight live on separate HIP streams; ROCm maps streams onto 4 hardware queues by default
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_INT8_OPS = 256 * 4 * 2048 * 2.4e9       # 256 CUs x 4 SIMDs x 1024 MAC/clk (v_mfma_i32_32x32x32_i8) x 2.4 GHz
PEAK_HBM = 8.0e12


def pmc_traffic():
    """HBM bytes per dispatch by kernel, from the newest PMC summaries committed under profiles/ (written by
    profiles/collect.sh: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes, KB per dispatch).  gfx950 tallies a
    128-B read request as 64 B, so FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact."""
    import glob
    import re
    out, src = {}, None
    fs = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_FETCH_SIZE.txt')))
    if not fs:
        return out, src
    src = os.path.basename(fs[-1]).replace('_pmc_FETCH_SIZE.txt', '')
    for name, mult in (('FETCH_SIZE', 2.0), ('WRITE_SIZE', 1.0)):
        path = os.path.join(ROOT, 'profiles', f'{src}_pmc_{name}.txt')
        if not os.path.exists(path):
            return {}, None
        for line in open(path):
            m = re.match(r'\s*(?:void )?qasr::(.+?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s*$', line)
            if m:
                out[m.group(1)] = out.get(m.group(1), 0.0) + mult * 1024.0 * float(m.group(4))
    return out, src
MODEL = 'QuartzNet15x5Base-En'
BATCH, SAMPLES, FRAMES = 32, 80000, 500


_T0 = time.time()


def log(msg):
    print(f'[bench {time.time() - _T0:7.1f}s] {msg}', file=sys.stderr, flush=True)


def host_cores():
    """CPU share of this process: min(affinity, cgroup quota); os.cpu_count() reports the whole host."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def algorithmic_work(cfg, B, T_out):
    """Ops (2 per MAC) per forward by kernel class, from the topology (273.4 GOP MFMA-class,
    28.2 GOP depthwise at B=32, T=500->250) and the minimal int8 activation bytes of the depthwise class."""
    from qasr import topology
    mfma = dw = dw_bytes = 0
    for sites in topology.conv_plan(cfg):
        for s in sites:
            macs = B * T_out * s.cout * (s.cin // s.groups) * s.kernel
            if s.role == 'dw':
                dw += 2 * macs
                dw_bytes += B * s.cout * T_out * (2 if s.stride == 1 else 3)
            else:
                mfma += 2 * macs
    mfma += 2 * B * T_out * cfg.blocks[-1].filters * (cfg.num_classes + 1)
    return mfma, dw, dw_bytes


def dominant_kernel_roofline(eng, cfg, meta, ms, B, T):
    """Roofline entry for the kernel instantiation with the largest summed time per step.

    Algorithmic work per launch (2 ops per MAC; a fused dw->pw layer reads the depthwise input once,
    writes the pointwise output once, reads its weights once), T = valid output frames.  The bound is whichever of
    t_MFMA = ops / peak_int8 and t_HBM = bytes / 8 TB/s is larger for that work."""
    import collections
    from qasr import topology
    labels = eng.op_labels()
    plan = [s for ss in topology.conv_plan(cfg) for s in ss]
    main, panes = {}, collections.defaultdict(list)
    for i, (op, pane) in enumerate(meta['sites']):
        if i < len(plan):
            (panes[op].append(plan[i]) if pane >= 0 else main.__setitem__(op, plan[i]))
    groups = collections.defaultdict(list)
    for oi, lab in enumerate(labels):
        if not lab.startswith('('):
            groups[lab].append(oi)
    tot = {lab: sum(ms[o] for o in ops) for lab, ops in groups.items()}
    lab = max(tot, key=tot.get)
    ops_l = groups[lab]
    mfma = vdot = byts = 0
    for oi in ops_l:
        s = main.get(oi)
        if s is None:                                            # decoder / non-conv op
            continue
        cin = s.cin // s.groups
        if s.role == 'dw':
            vdot += 2 * B * T * s.cout * s.kernel
        else:
            mfma += 2 * B * T * s.cout * cin * s.kernel
        byts += B * T * (s.cin + s.cout) + s.cout * cin * s.kernel
        for r in panes.get(oi, []):
            mfma += 2 * B * T * r.cout * r.cin
            byts += B * T * r.cin + r.cout * r.cin
        if oi > 0 and labels[oi - 1].startswith('(fused') and (oi - 1) in main:
            d = main[oi - 1]                                     # depthwise stage inside this launch
            vdot += 2 * B * T * d.cout * d.kernel
            byts += d.cout * d.kernel                            # its input replaces the pw input already counted
    n = len(ops_l)
    traffic, traffic_src = pmc_traffic()
    t = tot[lab] * 1e-3 / n                                      # s per launch
    # the fused depthwise taps run on the matrix cores too (v_mfma_i32_4x4x4), so they count towards the MFMA bound
    t_mfma, t_hbm = (mfma + vdot) / n / PEAK_INT8_OPS, byts / n / PEAK_HBM
    serial = float(sum(ms))
    out = {'kernel': 'qasr::' + lab, 'launches_per_step': n, 'avg_launch_us': 1e6 * t,
           'share_of_step_device_time': tot[lab] / serial,
           'algorithmic_bytes_per_launch': byts / n, 'mfma_ops_per_launch': mfma / n,
           'depthwise_ops_per_launch': vdot / n, 'traffic': traffic.get(lab)}
    if t_hbm >= t_mfma:
        out.update(bound='hbm', achieved=byts / n / t / 1e9, peak=PEAK_HBM / 1e9, unit='GB/s', frac=t_hbm / t)
    else:
        out.update(bound='mfma', achieved=(mfma + vdot) / n / t / 1e12, peak=PEAK_INT8_OPS / 1e12, unit='TFLOP/s', frac=t_mfma / t)
    out['other'] = {
        'mfma_frac': t_mfma / t, 'hbm_frac': t_hbm / t, 'all_ops_ms_per_step_serial': serial,
        'per_kernel_ms_per_step': {k: round(v, 4) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])},
        'timing': 'each op replayed 20x between one HIP event pair on the launch stream, no other work in flight',
        'traffic_source': 'rocprofv3 --pmc FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, separate passes, per '
                          f'dispatch average, bytes (profiles/{traffic_src}_pmc_*.txt); null = not collected'}
    return out


def build_model(device):
    """Random-init QuartzNet15x5 (no network for checkpoints), calibrated exactly like inference.py does."""
    import nemo.quantization.utils.quantize_model as qm
    from nemo.collections.asr.models import EncDecCTCModel
    from qasr import pack, synth
    torch.set_grad_enabled(False)
    m = EncDecCTCModel.from_synthetic(MODEL, seed=0).to(device)
    m.eval()
    m.set_quant_bit(8, mode='weight')
    m.set_quant_bit(8, mode='act')
    qm.set_percentile(m, 99.996)
    m.encoder.bn_folding()
    qm.calibrate(m)
    log('model built; calibrating 2 x [8,64,500] (host PyTorch-ROCm)')
    length = torch.tensor([FRAMES] * 8, device=device)
    for c in synth.make_calibration(2, 8, 64, FRAMES, seed=0):
        e, _, sf = m.encoder(audio_signal=torch.from_numpy(c).to(device), length=length)
        m.decoder(encoder_output=e, encoder_output_scaling_factor=sf)
    qm.evaluate(m)
    log('calibrated; packing')
    inputs = m.export_pack_inputs()
    blob, meta = pack.pack_model(*inputs)
    f = m.preprocessor.featurizer
    return blob, meta, f.fb[0].detach().cpu().contiguous(), f.window.detach().cpu().contiguous(), inputs[2], inputs[3]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--tile', type=int, default=0, choices=[0, 32, 64],
                    help='k_sep frames per work-group: 32 = lowest single-step latency, 64 = less weight/halo traffic '
                         'per frame but half the work-groups; 0 = 64 when more than one step is in flight')
    ap.add_argument('--no-graph', action='store_true', help='enqueue every kernel instead of replaying the captured hipGraph')
    ap.add_argument('--whole-utterance', action='store_true', help='use the k_utt kernels (one work-group per utterance)')
    ap.add_argument('--streams', type=int, default=int(os.environ.get('QASR_BENCH_STREAMS', 4)),
                    help='independent steps in flight per GPU (each on its own HIP stream + engine arena)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.gpus != world and world > 1:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the integer engine has no CPU fallback')
    # QASR_BENCH_BACKEND=gloo rehearses the N>1 control flow on a one-GPU box: every rank uses GPU 0 and the two
    # exchange steps go through host memory.  The measured configuration is always nccl (= RCCL), one GPU per rank.
    backend = os.environ.get('QASR_BENCH_BACKEND', 'nccl')
    if backend != 'nccl':
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    comm_dev = dev if backend == 'nccl' else torch.device('cpu')
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)     # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend)

    from qasr import engine, synth, topology
    engine.load_library()
    cfg = topology.MODELS[MODEL]()

    # rank 0 calibrates + packs; the packed int weights travel to the other ranks over RCCL/xGMI
    from qasr import dist as qdist
    blob = meta = None
    fb, window = torch.zeros(64, 257), torch.zeros(320)
    if rank == 0:
        blob, meta, fb, window, amin, amax = build_model(dev)
    if world > 1:
        blob = qdist.broadcast_bytes(blob, 0, comm_dev)
        fb, window = qdist.broadcast_tensors([fb, window], 0, comm_dev)
    fb, window = fb.to(dev), window.to(dev)
    # throughput mode: consecutive steps are independent batches, so S of them are kept in flight, each on its own
    # HIP stream with its own engine arena (kernels of different steps overlap each other's launch gaps and tails)
    S = max(1, args.streams)
    tile = args.tile or (64 if S > 1 else 32)
    engs = [engine.Engine(blob, local, whole_utterance=args.whole_utterance, wide_tiles=(tile == 64),
                          graph=not args.no_graph) for _ in range(S)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    eng = engs[0]
    log(f'{S} engine(s) ready ({len(blob) / 1e6:.1f} MB blob); warm-up')

    audio = torch.from_numpy(synth.make_audio(BATCH, SAMPLES, seed=100 + rank)).to(dev)
    alen = torch.full((BATCH,), SAMPLES, dtype=torch.int32, device=dev)
    T_out = eng.out_frames(engine.load_library().qasr_frontend_frames(SAMPLES, 16))
    gathered = [torch.empty(BATCH, T_out, dtype=torch.int32, device=comm_dev) for _ in range(world)] if rank == 0 else None

    # every step in flight owns its buffers (features, lengths, tokens): stable pointers let the engine replay its
    # forward as one hipGraph launch instead of ~90 kernel launches
    T_pad = engine.load_library().qasr_frontend_frames(SAMPLES, 16)
    lib_ws = engine.load_library().qasr_frontend_workspace_bytes(BATCH, SAMPLES, 64)
    bufs = [dict(fe=(torch.empty(BATCH, 64, T_pad, device=dev), torch.empty(BATCH, dtype=torch.int32, device=dev),
                     torch.empty(max(lib_ws, 16), dtype=torch.uint8, device=dev)),
                 out=(None, torch.empty(BATCH, T_out, dtype=torch.int32, device=dev),
                      torch.empty(BATCH, dtype=torch.int32, device=dev))) for _ in range(S)]

    def step(i):
        k = i % S
        with torch.cuda.stream(streams[k]):
            feats, flen = engine.frontend_mel(audio, alen, fb, window, 0.97, 16, out=bufs[k]['fe'])
            _, tokens, _ = engs[k].forward(feats, flen, want_logp=False, out=bufs[k]['out'])
            done = torch.cuda.Event()
            done.record(streams[k])
        if world > 1:
            # one communicator: the per-step gathers are issued in step order on the default stream, each behind its
            # step's compute stream; compute of later steps keeps running on the other streams
            torch.cuda.current_stream().wait_event(done)
            tokens.record_stream(torch.cuda.current_stream())
            qdist.gather_tokens(tokens if backend == 'nccl' else tokens.cpu(), 0, gathered)
        return tokens

    # one-time setup, like the model build above: each engine's first forward launches kernel by kernel, the second is
    # captured into its hipGraph; only then do the W warm-up steps and the K timed steps run (all as graph replays)
    for i in range(2 * S):
        step(i)
    torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = []
    for i in range(args.steps):
        tokens = step(i)
        last = (last + [tokens])[-S:]
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    # every step decodes the same batch: the steps that were in flight together must agree bit for bit
    if not all(torch.equal(t_, last[0]) for t_ in last):
        raise SystemExit('bench: concurrent steps produced different tokens')
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax[0])
    audio_s = world * BATCH * SAMPLES / 16000.0 * args.steps
    log(f'timed {args.steps} steps: {1e3 * dt / args.steps:.3f} ms/step (host enqueue {1e3 * t_enq / args.steps:.3f} ms/step)')
    result = {
        'metric': 'RTFx (audio-sec/wall-sec) QuartzNet15x5 int8 bs32', 'value': audio_s / dt, 'unit': 'audio-s/wall-s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'int8 (s8 x s8 -> i32 on MFMA; float32/float64 fixed-point requant)',
        'data': 'synthetic',
        'config': {'workload': 'QuartzNet15x5Base-En w8a8 percentile=99.996, bs=32/GPU, 5 s synthetic 16 kHz audio '
                               '(500 mel frames): HIP mel front-end + integer encoder + CTC decoder + greedy argmax',
                   'global_batch': BATCH * world, 'seq_len': FRAMES, 'weights': 'random-init (qasr.synth, seed 0)',
                   'steps_in_flight': S, 'hip_graph': not args.no_graph, 'kernels': 'k_utt' if args.whole_utterance else f'k_sep, {tile}-frame tiles', 'parallelism': f'utterance-sharded x{world}' + (f', {"RCCL" if backend == "nccl" else backend} blob broadcast + token gather' if world > 1 else ''),
                   'wer': 'not measurable here: no LibriSpeech / checkpoint in the image'},
    }

    if rank == 0:
        # ---- roofline of the dominant kernel, HIP events on the launch stream ---------------------------------
        # every op is replayed 20x back to back between one HIP event pair on the launch stream
        # (qasr_engine_time_ops); buffers hold the real activations of the last timed step
        ms = eng.time_ops(reps=20).astype(np.float64)
        result['roofline'] = dominant_kernel_roofline(eng, cfg, meta, ms, BATCH, FRAMES // 2)
        mfma_ops, dw_ops, _ = algorithmic_work(cfg, BATCH, FRAMES // 2)
        # whole-step view (all launches, steps in flight as timed): SURVEY §8d bytes = every conv reads its int8 input
        # once, writes its output once, weights once (1065.9 + 18.85 MB at B=32, T'=250)
        step_bytes = 1065.9e6 * BATCH / 32 + 18.85e6
        result['roofline']['other'].update(step_mfma_class_gop=mfma_ops / 1e9, step_depthwise_gop=dw_ops / 1e9,
                                           step_mfma_class_top_s=mfma_ops / (dt / args.steps) / 1e12,
                                           step_algorithmic_gb_s=step_bytes / (dt / args.steps) / 1e9,
                                           step_hbm_frac=step_bytes / (dt / args.steps) / PEAK_HBM,
                                           work_groups_per_launch=BATCH * 256 // tile,
                                           note='a 64-frame-tile launch fills 128 of the 256 CUs; the bench keeps two '
                                                'such launches (of different steps) on the chip at a time')

        # ---- CPU baseline: the reference's fake-quant op sequence on this host's cores (N=1 only) --------------
        if world == 1 and not args.no_cpu_baseline:
            from oracle.fakequant_torch import FakeQuantNet
            cores = host_cores()
            torch.set_num_threads(cores)
            log(f'roofline pass done; CPU baseline on {cores} host threads')
            sd = synth.make_state_dict(cfg, 0)
            net = FakeQuantNet(topology.conv_plan(cfg), cfg, sd, amin, amax, 8, 8)
            feats, _ = engine.frontend_mel(audio, alen, fb, window, 0.97, 16)
            x = feats[:, :, :FRAMES].cpu().numpy()
            lens = [FRAMES] * BATCH
            net.forward(x[:4], lens[:4])                         # warm-up on a small slice
            log('cpu baseline warm-up done')
            t1 = time.perf_counter()
            n_fwd = 2
            for _ in range(n_fwd):
                out = net.forward(x, lens)
                log('cpu baseline forward done')
            tc = (time.perf_counter() - t1) / n_fwd
            agree = float((out['tokens'].numpy() == tokens.cpu().numpy()[:, :out['tokens'].shape[1]]).mean())
            result['cpu_baseline'] = {
                'value': BATCH * FRAMES * 0.01 / tc, 'unit': 'audio-s/wall-s', 'cores': torch.get_num_threads(),
                'kind': 'port',
                'sample': f'{n_fwd} forwards of encoder+decoder on the same 32x500-frame feature batch '
                          f'({tc:.2f} s each; front-end excluded); token agreement with the GPU run {agree:.4f}'}
        print(json.dumps(result))
    for e_ in engs:
        e_.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
