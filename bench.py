#!/usr/bin/env python3
"""Headline benchmark: RTFx of the quantised hot path (BASELINE.json config 2).

One "step" = one pass of the hot path over one batch resident in HBM:
    synthetic 16 kHz audio [32, 80000] (5 s each -> 500 valid mel frames)
    -> HIP mel front-end -> integer QuartzNet15x5 encoder (w8a8, percentile 99.996 calibration)
    -> CTC decoder -> log-softmax + greedy argmax tokens            (all in libqasr_hip.so)
`--gpus N`: one rank per GPU, every rank runs its own 32-utterance shard (weak scaling); rank 0 calibrates + packs the
model and broadcasts the packed int weights over RCCL, and each step ends with an RCCL gather of the greedy tokens to
rank 0.  Started without a launcher (`python bench.py --gpus N`, WORLD_SIZE unset) the parent - before anything touches
the GPU - starts the N ranks itself as child processes and relays rank 0's JSON line; started by torch.distributed.run
it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.

Prints ONE JSON line on rank 0.  `roofline` is measured live with HIP events on the launch stream (each op replayed 20x
between one event pair) for the kernel instantiation with the largest share of the step; `cpu_baseline` times the
reference's fake-quant CPU op sequence (oracle/fakequant_torch.py) on the host cores (N=1 only).
`--config w6a6|jasper` runs BASELINE.json's configurations 3 / 4 through the same loop (config 2 is the default and the
headline).  At N = 1 the default run ALSO measures, after the headline region and never inside it, 20 steps each of config 3,
config 4 and config 2 with log-probs written (`other_configs` in the JSON line; `--no-other-configs` skips them)."""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, 'q-asr_amd'), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

# Steps in flight live on separate HIP streams, and those need hardware queues of their own: with the runtime's default
# (4 queues per process) the first four torch streams of this process land pairwise on TWO queues - measured in round 4 by
# dropping this line: the timed region's streams finished at 6.4 / 6.5 / 12.3 / 12.3 ms and the step took 0.617 ms instead of
# 0.39 (gpurun_out -> profiles/r04_v1_no_hw_queues_experiment.json).  With 8 queues the four launch chains run side by side.  More
# chains than 4 do not pay at batch 32 (profiles/r03_v3_queue_experiments.txt: 8 streams / 16 queues: 0.525 ms/step).
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

PEAK_INT8_OPS = 256 * 4 * 2048 * 2.4e9       # 256 CUs x 4 SIMDs x 1024 MAC/clk (v_mfma_i32_32x32x32_i8) x 2.4 GHz
PEAK_HBM = 8.0e12
SAMPLES, FRAMES = 80000, 500

CONFIGS = {
    # name: (model, weight bits, act bits, batch per GPU, default steps in flight, BASELINE.json configuration)
    'quartznet': ('QuartzNet15x5Base-En', 8, 8, 32, 4, 2),
    'w6a6': ('QuartzNet15x5Base-En', 6, 6, 32, 4, 3),
    'jasper': ('Jasper10x5Dr-En', 8, 8, 64, 4, 4),
}

_T0 = time.time()


def log(msg):
    print(f'[bench {time.time() - _T0:7.1f}s] {msg}', file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', choices=sorted(CONFIGS), default='quartznet',
                    help='quartznet = BASELINE.json config 2 (headline); w6a6 = config 3; jasper = config 4')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--tile', type=int, default=0, choices=[0, 32, 64, 128],
                    help='frames per work-group of the separable-layer kernels: 32 = lowest single-step latency; 64 / 128 = '
                         'less weight / halo / tap-row traffic per frame but 1/2, 1/4 of the work-groups (128: the plain '
                         'k_sep2 layers only, the rest stays on 64); 0 = 128 when more than one step is in flight, else 32')
    ap.add_argument('--no-graph', action='store_true', help='enqueue every kernel instead of replaying the captured hipGraph')
    ap.add_argument('--streams', type=int, default=int(os.environ.get('QASR_BENCH_STREAMS', 0)),
                    help='independent steps in flight per GPU (each on its own HIP stream + engine arena); 0 = the config default')
    ap.add_argument('--no-other-configs', action='store_true',
                    help='skip the extra measurements of BASELINE.json configs 3 / 4 and of config 2 with log-probs (N = 1 only)')
    ap.add_argument('--gather', choices=['tokens', 'logits'], default='tokens',
                    help='what the per-step exchange sends to rank 0: int32 greedy tokens [B, T\'] (default; conv_asr.py:275 + '
                         'ctc_models.py:405) or the float32 CTC log-probs [B, T\', 29] (north_star\'s wording; the step then writes them)')
    ap.add_argument('--check-gather', action='store_true',
                    help='after the timed region rank 0 recomputes every other rank\'s last steps locally (the audio seeds are a '
                         'function of rank and stream) and compares them with what the gather delivered')
    ap.add_argument('--dry-run', action='store_true',
                    help='rank plumbing only (launcher, process group, blob broadcast, per-step token gather) on synthetic '
                         'payloads: no model, no engine, no GPU needed (gloo when no GPU is visible)')
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ launcher
def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks as child processes (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set) BEFORE this process touches the GPU, relay rank 0's JSON line, fail if any rank fails.  The parent
    never initialises HIP (torch.cuda.device_count() does not) and never re-executes itself."""
    import torch
    n_vis = torch.cuda.device_count()
    if not args.dry_run and n_vis < args.gpus:
        print(f'bench.py: --gpus {args.gpus} but only {n_vis} GPU(s) visible', file=sys.stderr)
        return 2
    port = _free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's JSON line is drained by a thread while EVERY child is polled: the first rank that dies takes the others
    # with it at once (a survivor would otherwise sit in RCCL initialisation / a collective until its timeout)
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rcs = [None] * len(procs)
    while any(rc is None for rc in rcs):
        for i, p in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = p.poll()
        if any(rc not in (None, 0) for rc in rcs):
            for i, p in enumerate(procs):
                if rcs[i] is None:
                    p.kill()                                 # exactly the children this call started
                    rcs[i] = p.wait()
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    out0 = b''.join(chunks).decode()
    if any(rcs):
        print(f'bench.py: rank exit codes {rcs}', file=sys.stderr)
        return max(1, max(abs(rc) for rc in rcs))
    sys.stdout.write(out0)
    sys.stdout.flush()
    return 0


# ------------------------------------------------------------------------------------------------ roofline helpers
def pmc_traffic():
    """HBM bytes per dispatch by kernel, from the newest PMC summaries committed under profiles/ (written by
    profiles/collect.sh: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes, KB per dispatch).  gfx950 tallies a
    128-B read request as 64 B, so FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact."""
    import glob
    import re
    out, src = {}, None
    def tag(path):                                           # 'r02_v10_...' sorts after 'r02_v9_...': (round, version) as numbers
        m = re.match(r'r(\d+)_v(\d+)', os.path.basename(path))
        return (int(m.group(1)), int(m.group(2))) if m else (-1, -1)
    fs = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_FETCH_SIZE.txt')), key=tag)
    if not fs:
        return out, src
    rnd, ver = tag(fs[-1])
    # every summary of the newest (round, version): one per configuration (r03_v2_*, r03_v2_jasper_*), different kernels
    srcs = [os.path.basename(f).replace('_pmc_FETCH_SIZE.txt', '') for f in fs if tag(f) == (rnd, ver)]
    src = ' + '.join(srcs)
    for one in srcs:
        for name, mult in (('FETCH_SIZE', 2.0), ('WRITE_SIZE', 1.0)):
            path = os.path.join(ROOT, 'profiles', f'{one}_pmc_{name}.txt')
            if not os.path.exists(path):
                continue
            for line in open(path):
                m = re.match(r'\s*(?:void )?qasr::(.+?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s*$', line)
                if m:
                    out[m.group(1)] = out.get(m.group(1), 0.0) + mult * 1024.0 * float(m.group(4))
    return out, src


def pmc_counters(kernel):
    """rocprofv3's own matrix-pipe counters for `kernel`, from the newest committed summary profiles/r*_pmc_MFMA.json
    (written by profiles/collect.sh from separate --pmc passes; not collected in this run): the fraction of the CUs' busy
    cycles in which the MFMA pipe was busy, and int8 MFMA MOPS per launch."""
    import glob
    import re
    def tag(path):
        m = re.match(r'r(\d+)_v(\d+)', os.path.basename(path))
        return (int(m.group(1)), int(m.group(2))) if m else (-1, -1)
    fs = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_MFMA.json')), key=tag)
    if not fs:
        return None
    newest = tag(fs[-1])
    for f in reversed([f for f in fs if tag(f) == newest]):
        try:
            rows = json.load(open(f)).get('kernels', {})
        except (OSError, ValueError):
            continue
        row = rows.get(kernel) or rows.get(kernel.replace('qasr::', ''))
        if row:
            return {'mfma_busy_frac_rocprof': row.get('mfma_busy_frac'), 'rocprof_counters': row,
                    'rocprof_counters_source': f'profiles/{os.path.basename(f)} (separate rocprofv3 --pmc passes of `bench.py --streams 1`, per-dispatch averages)'}
    return None


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
    28.2 GOP depthwise at B=32, T=500->250) and the minimal int8 activation bytes (each conv reads its input once and
    writes its output once, weights once)."""
    from qasr import topology
    mfma = dw = act_bytes = w_bytes = 0
    for sites in topology.conv_plan(cfg):
        for s in sites:
            macs = B * T_out * s.cout * (s.cin // s.groups) * s.kernel
            if s.role == 'dw':
                dw += 2 * macs
            else:
                mfma += 2 * macs
            act_bytes += B * T_out * (s.cin * s.stride + s.cout)
            w_bytes += s.cout * (s.cin // s.groups) * s.kernel
    mfma += 2 * B * T_out * cfg.blocks[-1].filters * (cfg.num_classes + 1)
    return mfma, dw, act_bytes + w_bytes


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
    # frames per work-group: k_sep2<K, NG, NGP, NP, DBG, TT, DIL> / k_sep<K, DIL, EP, DBG, TT>
    tt = int(lab.rstrip('>').split(',')[-2 if lab.startswith('k_sep2<') else -1]) if lab.startswith('k_sep') else 0
    out = {'kernel': 'qasr::' + lab, 'launches_per_step': n, 'avg_launch_us': 1e6 * t, '_ops': ops_l,
           '_wgs': B * (-(-T // tt)) if tt else None,
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
                          f'dispatch average, bytes, read from the committed summary profiles/{traffic_src}_pmc_*.txt '
                          '(not collected in this run); null = that summary has no row for this kernel'}
    return out


def in_flight_timing(lane, roof, ops, reps=20):
    """The dominant kernel the way the timed region runs it: one launch per step in flight, all S of them concurrently
    (S engines, S HIP streams).  A 128-frame-tile launch of a 32-utterance batch has 64 work-groups for 256 CUs, so the
    single-launch figure leaves three quarters of the chip idle by construction; elapsed / launches with S streams
    replaying the kernel is the per-launch cost the chip pays in the timed region."""
    import torch
    S = lane['S']
    for e_, s_ in zip(lane['engs'], lane['streams']):
        for o in ops:
            e_.run_op(o, stream=s_)
    torch.cuda.synchronize()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(S)]
    for k in range(S):
        ev[k][0].record(lane['streams'][k])
    for _ in range(reps):
        for o in ops:
            for k in range(S):
                lane['engs'][k].run_op(o, stream=lane['streams'][k])
    for k in range(S):
        ev[k][1].record(lane['streams'][k])
    torch.cuda.synchronize()
    ms = max(ev[k][0].elapsed_time(ev[k][1]) for k in range(S))
    t = ms * 1e-3 / (S * reps * len(ops))                        # s per launch, all streams' launches counted
    # the clock the chip holds meanwhile: every work-group of the kernel stamps its start / end (100 MHz s_memrealtime) and its
    # shader cycles (s_memtime) into a diagnostic buffer (qasr_debug_timeline; the timed launches above run without it)
    clock = None
    import ctypes as C
    import numpy as np
    lib = lane['engs'][0].lib
    buf = torch.zeros(4 * 16384, dtype=torch.int64, device='cuda')
    try:
        lib.qasr_debug_timeline(C.c_void_p(buf.data_ptr()), buf.numel() // 4)
        for _ in range(3):
            for o in ops:
                for k in range(S):
                    lane['engs'][k].run_op(o, stream=lane['streams'][k])
        torch.cuda.synchronize()
        st = buf.cpu().numpy().reshape(-1, 4)
        st = st[(st[:, 1] > st[:, 0]) & (st[:, 3] > 0)]
        if len(st):
            clock = float(np.median(st[:, 3] / ((st[:, 1] - st[:, 0]) * 10.0)))      # cycles / ns = GHz
    except Exception as exc:                                     # diagnostics only
        log(f'clock probe skipped: {exc}')
    finally:
        # the process-global diagnostic pointer never outlives `buf`: launches still in flight are drained first, then the
        # pointer is cleared on every path (a later graph capture would otherwise be refused, or stamps land in freed memory)
        try:
            torch.cuda.synchronize()
        finally:
            lib.qasr_debug_timeline(C.c_void_p(0), 0)
    extra = {}
    if clock:
        peak_at_clock = PEAK_INT8_OPS * clock / 2.4
        extra = {'sustained_clock_ghz': clock,
                 'mfma_frac_at_sustained_clock': (roof['mfma_ops_per_launch'] + roof['depthwise_ops_per_launch']) / t / peak_at_clock,
                 'clock_note': 'median over the work-groups of the kernel of shader cycles / elapsed 100 MHz ticks with the S launches in '
                               'flight; `peak` (and every *_frac) is quoted at 2.4 GHz, the chip holds less under load'}
    return {**extra, 'launches_in_flight': S, 'effective_launch_us': 1e6 * t,
            'stream_launch_us': 1e3 * ms / (reps * len(ops)),
            'achieved_gb_s': roof['algorithmic_bytes_per_launch'] / t / 1e9,
            'hbm_frac': roof['algorithmic_bytes_per_launch'] / t / PEAK_HBM,
            'mfma_frac': (roof['mfma_ops_per_launch'] + roof['depthwise_ops_per_launch']) / t / PEAK_INT8_OPS,
            'timing': f'{reps} x the kernel\'s launches of one step on each of the {S} HIP streams (one engine each), HIP '
                      'events recorded on those streams; effective = slowest stream\'s elapsed / all launches'}


def cu_masked_stream(k, dev, parts=4):
    """(experiment, QASR_BENCH_CU_MASK=1) A HIP stream whose queue may only use the k-th of `parts` disjoint sets of CUs
    (hipExtStreamCreateWithCUMask).  On MI355X mask bit b is CU (b // 8) of XCC (b % 8) - profiles/r04_cu_mask_probe.txt - so bits
    [64 k, 64 k + 64) are 8 CUs (two per shader engine) on every XCC: a 64-work-group launch of chain k then never competes
    with the other chains' launches for CUs, and still has 8 work-groups per XCC."""
    import ctypes as C
    import torch
    hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so'))
    hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    per = n_cu // parts
    words = (n_cu + 31) // 32
    arr = (C.c_uint32 * words)()
    if os.environ.get('QASR_BENCH_CU_MASK') == 'xcc':          # whole XCCs per chain: XCC = b % 8; chain k owns XCCs 2 k, 2 k + 1
        bits = [b for b in range(n_cu) if (b % 8) * parts // 8 == k % parts]
    else:
        bits = range(per * (k % parts), per * (k % parts + 1))
    for b in bits:
        arr[b // 32] |= 1 << (b % 32)
    h = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(h), words, arr)
    if rc:
        raise SystemExit(f'hipExtStreamCreateWithCUMask failed ({rc})')
    return torch.cuda.ExternalStream(h.value, device=dev)


def build_model(device, model_name, wbit, abit):
    """Random-init model (no network for checkpoints), calibrated exactly like inference.py does."""
    import torch
    import nemo.quantization.utils.quantize_model as qm
    from nemo.collections.asr.models import EncDecCTCModel
    from qasr import pack, synth
    torch.set_grad_enabled(False)
    m = EncDecCTCModel.from_synthetic(model_name, seed=0).to(device)
    m.eval()
    m.set_quant_bit(wbit, mode='weight')
    m.set_quant_bit(abit, mode='act')
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


# ------------------------------------------------------------------------------------------------ one rank
def run_dry(args, rank, world):
    """Rank plumbing without the engine: process group, blob broadcast, per-step token gather, rank census."""
    import torch
    from qasr import dist as qdist
    use_gpu = torch.cuda.is_available() and os.environ.get('QASR_BENCH_BACKEND', 'nccl') == 'nccl'
    dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', 0))) if use_gpu else torch.device('cpu')
    seen = 1
    if world > 1:
        import torch.distributed as dist
        if use_gpu:
            torch.cuda.set_device(dev)
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group('gloo')
        blob = qdist.broadcast_bytes(bytes(range(256)) * 64 if rank == 0 else None, 0, dev)
        assert blob == bytes(range(256)) * 64
        gathered = [torch.empty(4, 8, dtype=torch.int32, device=dev) for _ in range(world)] if rank == 0 else None
        for i in range(args.warmup + args.steps):
            qdist.gather_tokens(torch.full((4, 8), 100 * rank + i, dtype=torch.int32, device=dev), 0, gathered)
            if rank == 0:
                assert [int(g[0, 0]) for g in gathered] == [100 * r + i for r in range(world)]
        ones = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(ones)
        seen = int(ones[0])
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({'metric': 'dry-run (rank plumbing only)', 'value': None, 'n_gpus': world, 'n_ranks_seen': seen,
                          'steps': args.steps, 'warmup': args.warmup, 'dry_run': True,
                          'backend': 'nccl' if use_gpu else 'gloo'}))


class Workload:
    """One BASELINE.json configuration on this rank: the calibrated + packed model (rank 0 builds it, the others receive the
    blob over the process group), the front-end plan, and the step / timed-region loop over S steps in flight."""

    def __init__(self, args, config, env):
        import torch
        from qasr import dist as qdist
        from qasr import engine, topology
        self.args, self.env, self.config = args, env, config
        self.model_name, self.wbit, self.abit, self.batch, self.S_default, self.baseline_cfg = CONFIGS[config]
        self.batch_override = 'QASR_BENCH_BATCH' in os.environ
        if self.batch_override:                                  # (experiment: several batches per launch; the metric label follows it)
            self.batch = int(os.environ['QASR_BENCH_BATCH'])
        rank, dev, comm_dev = env['rank'], env['dev'], env['comm_dev']
        self.lib = engine.load_library()
        self.cfg = topology.MODELS[self.model_name]()
        # rank 0 calibrates + packs; the packed int weights travel to the other ranks over RCCL/xGMI
        self.blob = self.meta = self.amin = self.amax = None
        fb, window = torch.zeros(64, 257), torch.zeros(320)
        if rank == 0:
            self.blob, self.meta, fb, window, self.amin, self.amax = build_model(dev, self.model_name, self.wbit, self.abit)
        self.n_ranks_seen, self.blob_digests_equal = 1, None
        if env['use_dist']:
            import hashlib
            dist = env['dist']
            self.blob = qdist.broadcast_blob(self.blob, 0, comm_dev)     # receiving ranks validate it (qasr_blob_check)
            fb, window = qdist.broadcast_tensors([fb, window], 0, comm_dev)
            ones = torch.ones(1, dtype=torch.int64, device=comm_dev)
            dist.all_reduce(ones)
            self.n_ranks_seen = int(ones[0])
            # every rank's engines are built from the bytes IT holds: compare their digests with rank 0's
            dig = torch.frombuffer(bytearray(hashlib.sha256(self.blob).digest()), dtype=torch.uint8).to(torch.int64).to(comm_dev)
            lo, hi = dig.clone(), dig.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            self.blob_digests_equal = bool(torch.equal(lo, hi))
        self.fb, self.window = fb.to(dev), window.to(dev)
        self.T_pad = self.lib.qasr_frontend_frames(SAMPLES, 16)
        self.lib_ws = self.lib.qasr_frontend_workspace_bytes(self.batch, SAMPLES, 64)
        self.fe_plan = engine.frontend_plan(self.fb)  # filterbank-only tables, built once like a model does (ctc_models._frontend_hip)
        self.alen = torch.full((self.batch,), SAMPLES, dtype=torch.int32, device=dev)
        self.step_trace = [] if os.environ.get('QASR_BENCH_STEP_TRACE') else None   # (diagnostic) when each step finished

    def with_batch(self, batch):
        """The same model and blob with another number of utterances per launch (context measurements only)."""
        import copy
        import torch
        w = copy.copy(self)
        w.batch, w.batch_override = batch, True
        w.lib_ws = self.lib.qasr_frontend_workspace_bytes(batch, SAMPLES, 64)
        w.alen = torch.full((batch,), SAMPLES, dtype=torch.int32, device=self.env['dev'])
        return w

    def audio_seed(self, rank, k):
        return 100 + 16 * rank + k

    def make_lane(self, S, tile, want_logp=False, rank=None):
        """S steps in flight: per step in flight an engine (own arena), a HIP stream, its own audio batch and its own
        feature / length / token (/ log-prob) buffers (stable pointers: the forward replays as one hipGraph launch)."""
        import torch
        from qasr import engine, synth
        args, dev, B = self.args, self.env['dev'], self.batch
        rank = self.env['rank'] if rank is None else rank
        engs = [engine.Engine(self.blob, self.env['local'], tile=tile, graph=not args.no_graph) for _ in range(S)]   # qasr_engine_opts (include/qasr.h)
        # The process keeps ONE set of HIP streams for every lane it builds (headline, one-step-in-flight, other_configs): which
        # hardware queue a stream lands on is decided when it is created, and a later lane on fresh streams can find itself
        # pairwise on shared queues (measured: the log-prob lane at 0.617 instead of 0.394 ms/step, gpurun_out r04_v5 first run)
        pool = self.env.setdefault('streams', [])
        while len(pool) < S:
            pool.append(cu_masked_stream(len(pool), dev) if os.environ.get('QASR_BENCH_CU_MASK') else torch.cuda.Stream(device=dev))
        streams = pool[:S]
        T_out = engs[0].out_frames(self.T_pad)
        audio = [torch.from_numpy(synth.make_audio(B, SAMPLES, seed=self.audio_seed(rank, k))).to(dev) for k in range(S)]
        ncls = engs[0].n_classes
        bufs = [dict(fe=(torch.empty(B, 64, self.T_pad, device=dev), torch.empty(B, dtype=torch.int32, device=dev),
                         torch.empty(max(self.lib_ws, 16), dtype=torch.uint8, device=dev)),
                     out=(torch.empty(B, T_out, ncls, device=dev) if want_logp else None,
                          torch.empty(B, T_out, dtype=torch.int32, device=dev),
                          torch.empty(B, dtype=torch.int32, device=dev))) for _ in range(S)]
        return dict(S=S, engs=engs, streams=streams, audio=audio, bufs=bufs, T_out=T_out, want_logp=want_logp, n_classes=ncls)

    def step(self, lane, i, gathered=None):
        import torch
        from qasr import dist as qdist
        from qasr import engine
        k = i % lane['S']
        b = lane['bufs'][k]
        want_logp = lane['want_logp']
        with torch.cuda.stream(lane['streams'][k]):
            if os.environ.get('QASR_BENCH_SPLIT_FE'):            # (A/B) front-end as its own two launches in front of the graph
                feats, flen = engine.frontend_mel(lane['audio'][k], self.alen, self.fb, self.window, 0.97, 16, out=b['fe'], plan=self.fe_plan)
                logp, tokens, _ = lane['engs'][k].forward(feats, flen, want_logp=want_logp, out=b['out'])
            else:
                # mel front-end + encoder + decoder as one engine call (one hipGraph launch per step)
                logp, tokens, _ = lane['engs'][k].forward_audio(lane['audio'][k], self.alen, self.fb, self.window, self.fe_plan, 0.97, 16,
                                                                want_logp=want_logp, feats=b['fe'][0], feat_lens=b['fe'][1], out=b['out'])
            if self.step_trace is not None:                       # (diagnostic only)
                done = torch.cuda.Event(enable_timing=True)
                done.record(lane['streams'][k])
                self.step_trace.append((i, k, done))
        if self.env['use_dist'] and gathered is not False:
            # Exchange step: the step's result goes to rank 0 through the one communicator, in step order, issued
            # ON THE STEP'S OWN COMPUTE STREAM: same-stream order puts the gather behind the step that produced the tokens
            # and in front of the step that overwrites them (4 steps later), so no event of this file's making is needed;
            # the other 3 streams keep computing while this one exchanges.  (ProcessGroupNCCL runs the collective on its own
            # stream and orders it against the calling stream itself.)  What does NOT work on this runtime - measured,
            # profiles/r03_v3_exchange_experiments.txt: any per-step activity on an extra stream of this process (even a
            # bare event record) or on the legacy default stream halves the concurrency of the 4 compute streams
            # (0.37 -> 0.65-1.15 ms/step), and a hipStreamWaitEvent in front of a hipGraphLaunch does the same.
            payload = logp if self.args.gather == 'logits' else tokens
            with torch.cuda.stream(lane['streams'][k]):
                # (one set of receive buffers per stream in flight: a later step of ANOTHER stream never overwrites them)
                qdist.gather_tokens(payload if self.env['backend'] == 'nccl' else payload.cpu(), 0,
                                    gathered[k] if gathered is not None else None)
        return tokens

    def timed(self, lane, steps, warmup, gathered=None):
        """One-time setup like the model build (each engine's first forward launches kernel by kernel, the second is captured
        into its hipGraph; the serial results of that phase are the reference every later step must reproduce), `warmup`
        untimed steps, then EXACTLY `steps` steps between barrier + synchronize pairs."""
        import torch
        env, step_trace = self.env, self.step_trace
        dist = env['dist'] if (env['use_dist'] and gathered is not False) else None
        S = lane['S']
        ref = []
        for i in range(2 * S):
            t_ = self.step(lane, i, gathered)
            torch.cuda.synchronize()
            if i >= S:
                ref.append(t_.clone())
        for i in range(warmup):
            self.step(lane, i, gathered)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ev0 = torch.cuda.Event(enable_timing=True)
        ev0.record(lane['streams'][0])                           # the region's device-side origin (diagnostic), on a stream of the lane
        t0 = time.perf_counter()
        last = {}
        if step_trace is not None:
            step_trace.clear()
        for i in range(steps):
            last[i % S] = self.step(lane, i, gathered)
        t_enq = time.perf_counter() - t0
        ends = []
        for st in lane['streams']:                               # when each stream's last step finished (diagnostic)
            e_ = torch.cuda.Event(enable_timing=True)
            e_.record(st)
            ends.append(e_)
        torch.cuda.synchronize()
        log('streams finished at ' + ', '.join(f'{ev0.elapsed_time(e_):.2f}' for e_ in ends) + ' ms after the region began')
        if step_trace is not None:
            for k in range(S):
                log(f'  stream {k}: steps finished at ' + ' '.join(f'{ev0.elapsed_time(e_):.2f}' for i_, k_, e_ in step_trace if k_ == k))
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        # every step in flight decodes its own batch: each must reproduce its serial result bit for bit
        for k, t_ in last.items():
            if not torch.equal(t_, ref[k]):
                raise SystemExit(f'bench: stream {k} produced different tokens with other steps in flight')
        if dist is not None and env['rank'] == 0 and gathered is not None:
            torch.cuda.synchronize()
            for k in last:                                       # rank 0's own slot of every stream's receive buffers
                mine = lane['bufs'][k]['out'][0] if self.args.gather == 'logits' else last[k]
                if not torch.equal(gathered[k][0].to(mine.device), mine):
                    raise SystemExit(f'bench: what the gather delivered for rank 0 (stream {k}) differs from the local result')
        return dt, t_enq, last[(steps - 1) % S]

    def check_gather(self, lane, gathered, steps):
        """Rank 0 recomputes the other ranks' steps itself - their audio is a function of (rank, stream) - with a serial
        one-step-in-flight engine and compares with what the last gather of every stream delivered.  Returns the ranks checked."""
        import torch
        S = lane['S']
        checked = []
        for r in range(1, self.env['world']):
            lane_r = self.make_lane(S, 32, want_logp=self.args.gather == 'logits', rank=r)
            for k in range(min(S, steps)):
                tok = self.step(lane_r, k, gathered=False)
                torch.cuda.synchronize()
                want = lane_r['bufs'][k]['out'][0] if self.args.gather == 'logits' else tok
                if not torch.equal(gathered[k][r].to(want.device), want):
                    raise SystemExit(f'bench: rank {r}, stream {k}: gathered result differs from a local recomputation')
            for e_ in lane_r['engs']:
                e_.close()
            checked.append(r)
        return checked


def measure_other_config(args, env, name, steps, warmup, want_logp=False, base=None):
    """20 steps of another BASELINE.json configuration through the same loop (own model, own engines), AFTER the headline
    region: {ms_per_step, rtfx, step_mfma_frac, roofline of its dominant kernel}.  `base`: reuse the headline workload
    (config 2 with log-probs written)."""
    import numpy as np
    import torch
    w = base or Workload(args, name, env)
    S = max(1, args.streams or w.S_default)
    lane = w.make_lane(S, args.tile or (128 if S > 1 else 32), want_logp=want_logp)
    dt, _, _ = w.timed(lane, steps, warmup, gathered=False)
    per = dt / steps
    mfma_ops, _, step_bytes = algorithmic_work(w.cfg, w.batch, FRAMES // 2)
    out = {'baseline_config': w.baseline_cfg, 'workload': f'{w.model_name} w{w.wbit}a{w.abit} bs{w.batch}' + (' + log-probs written' if want_logp else ''),
           'steps': steps, 'warmup': warmup, 'steps_in_flight': S, 'ms_per_step': 1e3 * per,
           'rtfx': w.batch * SAMPLES / 16000.0 / per, 'step_mfma_frac': mfma_ops / per / PEAK_INT8_OPS,
           'step_hbm_frac': step_bytes / per / PEAK_HBM}
    if base is None:
        torch.cuda.synchronize()
        roof = dominant_kernel_roofline(lane['engs'][0], w.cfg, w.meta, lane['engs'][0].time_ops(reps=20).astype(np.float64), w.batch, FRAMES // 2)
        out['blob_mb'] = len(w.blob) / 1e6
        out['roofline'] = {k: roof[k] for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'avg_launch_us', 'launches_per_step',
                                                 'share_of_step_device_time', 'traffic')}
    for e_ in lane['engs']:
        e_.close()
    log(f'other config {name}{" + logp" if want_logp else ""}: {1e3 * per:.3f} ms/step')
    return out


def run(args):
    import numpy as np
    import torch
    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.gpus != world:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    # QASR_BENCH_FORCE_DIST=1: run the exchange steps (blob broadcast, per-step token gather, max-over-ranks) through the
    # process group even at N = 1 - a one-rank RCCL communicator on a one-GPU box executes the same calls the N-GPU run makes
    use_dist = world > 1 or bool(os.environ.get('QASR_BENCH_FORCE_DIST'))
    if use_dist and world == 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
    if args.dry_run:
        return run_dry(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the integer engine has no CPU fallback')
    # QASR_BENCH_BACKEND=gloo rehearses the N>1 control flow on a one-GPU box: every rank uses GPU 0 and the two
    # exchange steps go through host memory.  The measured configuration is always nccl (= RCCL), one GPU per rank.
    backend = os.environ.get('QASR_BENCH_BACKEND', 'nccl')
    if backend != 'nccl':
        local = local % max(1, torch.cuda.device_count())
    elif local >= torch.cuda.device_count():
        raise SystemExit(f'rank {rank}: LOCAL_RANK {local} but only {torch.cuda.device_count()} GPU(s) visible')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    comm_dev = dev if backend == 'nccl' else torch.device('cpu')
    dist = None
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)     # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend)
    env = dict(rank=rank, local=local, world=world, dev=dev, comm_dev=comm_dev, dist=dist, backend=backend, use_dist=use_dist)

    from qasr import engine, synth, topology
    w = Workload(args, args.config, env)
    model_name, wbit, abit, BATCH, baseline_cfg, cfg = w.model_name, w.wbit, w.abit, w.batch, w.baseline_cfg, w.cfg

    # throughput mode: consecutive steps are independent batches, so S of them are kept in flight, each on its own
    # HIP stream with its own engine arena (kernels of different steps overlap each other's launch gaps and tails)
    S = max(1, args.streams or w.S_default)
    tile = args.tile or (128 if S > 1 else 32)
    lane = w.make_lane(S, tile, want_logp=args.gather == 'logits')
    eng = lane['engs'][0]
    T_out = lane['T_out']
    log(f'{S} engine(s) ready ({len(w.blob) / 1e6:.1f} MB blob); warm-up')
    gshape, gdtype = ((BATCH, T_out, lane['n_classes']), torch.float32) if args.gather == 'logits' else ((BATCH, T_out), torch.int32)
    gathered = ([[torch.empty(gshape, dtype=gdtype, device=comm_dev) for _ in range(world)] for _ in range(S)]
                if rank == 0 else None)
    dt, t_enq, tokens = w.timed(lane, args.steps, args.warmup, gathered)
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax[0])
    audio_s = world * BATCH * SAMPLES / 16000.0 * args.steps
    log(f'timed {args.steps} steps: {1e3 * dt / args.steps:.3f} ms/step (host enqueue {1e3 * t_enq / args.steps:.3f} ms/step)')
    metric = ('RTFx (audio-sec/wall-sec) QuartzNet15x5 int8 bs32' if (args.config == 'quartznet' and not w.batch_override)
              else f'RTFx (audio-sec/wall-sec) {model_name} w{wbit}a{abit} bs{BATCH}')
    result = {
        'metric': metric, 'value': audio_s / dt, 'unit': 'audio-s/wall-s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': f'int{wbit} (s8 x s8 -> i32 on MFMA; float64 fixed-point requant)',
        'data': 'synthetic', 'n_ranks_seen': w.n_ranks_seen,
        'config': {'workload': f'BASELINE.json config {baseline_cfg}: {model_name} w{wbit}a{abit} percentile=99.996, bs={BATCH}/GPU, '
                               '5 s synthetic 16 kHz audio (500 mel frames): HIP mel front-end + integer encoder + CTC '
                               'decoder + greedy argmax' + (' [QASR_BENCH_BATCH override: NOT the BASELINE batch]' if w.batch_override else ''),
                   'global_batch': BATCH * world, 'seq_len': FRAMES, 'weights': 'random-init (qasr.synth, seed 0)',
                   'steps_in_flight': S, 'inputs': 'one audio batch per step in flight (different seeds)',
                   'steps_in_flight_note': f'one {BATCH}-utterance launch chain per HIP stream (GPU_MAX_HW_QUEUES=8 so that each has a hardware '
                                           'queue: without it 0.617 ms/step, profiles/r04_v1_no_hw_queues_experiment.json); 4 chains is the measured '
                                           'optimum at batch 32 (profiles/r03_v3_queue_experiments.txt: 8 chains are slower)',
                   'log_probs': ('written in the timed step (the exchange gathers them)' if args.gather == 'logits' else
                                 'not written in the timed step (tokens and encoded lengths are; the reference forward also returns '
                                 'log-probs: 0.9 MB of stores per step - other_configs.quartznet_with_logp times that variant)'),
                   'hip_graph': not args.no_graph,
                   'kernels': ('k_dense2 (128 output channels x 256 / 128 frames per work-group), k_dense, k_sep' if args.config == 'jasper' else
                               f'k_mel, k_stem (normalisation + block 0), k_sep2 ({tile}-frame tiles; block 16: its dilation-2 form), k_dec'),
                   'parallelism': f'utterance-sharded x{world}' + (f', {"RCCL" if backend == "nccl" else backend} blob broadcast + {"token" if args.gather == "tokens" else "logits"} gather' if use_dist else ''),
                   'wer': 'not measurable here: no LibriSpeech / checkpoint in the image'},
    }
    if use_dist:
        result['blob_digests_equal_across_ranks'] = w.blob_digests_equal
        if args.check_gather:
            checked = w.check_gather(lane, gathered, args.steps) if rank == 0 else None
            dist.barrier()
            if rank == 0:
                result['gather_checked_ranks'] = checked

    if rank == 0:
        # ---- roofline of the dominant kernel, HIP events on the launch stream ---------------------------------
        # every op is replayed 20x back to back between one HIP event pair on the launch stream
        # (qasr_engine_time_ops); buffers hold the real activations of the last timed step
        torch.cuda.synchronize()
        ms = eng.time_ops(reps=20).astype(np.float64)
        result['roofline'] = dominant_kernel_roofline(eng, cfg, w.meta, ms, BATCH, FRAMES // 2)
        dom_ops = result['roofline'].pop('_ops')
        if S > 1:
            result['roofline']['other']['in_flight'] = in_flight_timing(lane, result['roofline'], dom_ops)
        mfma_ops, dw_ops, step_bytes = algorithmic_work(cfg, BATCH, FRAMES // 2)
        # whole-step view (all launches, steps in flight as timed): SURVEY §8d bytes = every conv reads its int8 input
        # once, writes its output once, weights once
        result['roofline']['other'].update(step_mfma_class_gop=mfma_ops / 1e9, step_depthwise_gop=dw_ops / 1e9,
                                           step_mfma_class_top_s=mfma_ops / (dt / args.steps) / 1e12,
                                           step_mfma_frac=mfma_ops / (dt / args.steps) / PEAK_INT8_OPS,
                                           step_algorithmic_gb_s=step_bytes / (dt / args.steps) / 1e9,
                                           step_hbm_frac=step_bytes / (dt / args.steps) / PEAK_HBM,
                                           work_groups_per_launch=result['roofline'].pop('_wgs', None))
        counters = pmc_counters(result['roofline']['kernel'])
        if counters:
            result['roofline']['other'].update(counters)
    for e_ in lane['engs']:
        e_.close()

    if world == 1 and rank == 0:
        # ---- one step in flight (32-frame tiles): the latency view of the same workload ------------------------
        if S > 1:
            lane1 = w.make_lane(1, 32)
            dt1, _, _ = w.timed(lane1, args.steps, args.warmup, gathered=False)
            result['single_stream_ms_per_step'] = 1e3 * dt1 / args.steps
            result['single_stream_rtfx'] = BATCH * SAMPLES / 16000.0 * args.steps / dt1
            log(f'one step in flight, 32-frame tiles: {1e3 * dt1 / args.steps:.3f} ms/step')
            # the same layer at the launch geometry of that mode: 256 work-groups, ONE launch fills the chip - the
            # figure `roofline.frac` (a 64-work-group launch alone on a 256-CU chip) cannot show
            torch.cuda.synchronize()
            r1 = dominant_kernel_roofline(lane1['engs'][0], cfg, w.meta, lane1['engs'][0].time_ops(reps=20).astype(np.float64),
                                          BATCH, FRAMES // 2)
            result['roofline']['other']['one_launch_fills_chip'] = {
                'kernel': r1['kernel'], 'work_groups_per_launch': r1['_wgs'], 'avg_launch_us': r1['avg_launch_us'],
                'hbm_frac': r1['other']['hbm_frac'], 'mfma_frac': r1['other']['mfma_frac'],
                'all_ops_ms_per_step_serial': r1['other']['all_ops_ms_per_step_serial'],
                'timing': 'the one-step-in-flight engine (32-frame tiles), each op replayed 20x between one HIP event pair, '
                          'nothing else in flight'}
            for e_ in lane1['engs']:
                e_.close()
        # ---- BASELINE.json configs 3 / 4 and config 2 with log-probs, after the headline region (driver-visible) -
        if args.config == 'quartznet' and not args.no_other_configs and not use_dist and not w.batch_override:
            oc = {'quartznet_with_logp': measure_other_config(args, env, 'quartznet', 20, 5, want_logp=True, base=w)}
            for name in ('w6a6', 'jasper'):
                oc[name] = measure_other_config(args, env, name, 20, 5)
            result['other_configs'] = oc
            # context, NOT BASELINE's configuration 2: the same kernels when ONE launch has work-groups for the whole chip (four
            # 32-utterance batches per launch, one chain) - a quarter of the kernel boundaries per utterance
            w4 = w.with_batch(4 * BATCH)
            lane4 = w4.make_lane(1, 128)
            dt4, _, _ = w4.timed(lane4, 8, 2, gathered=False)
            torch.cuda.synchronize()
            r4 = dominant_kernel_roofline(lane4['engs'][0], cfg, w.meta, lane4['engs'][0].time_ops(reps=20).astype(np.float64), 4 * BATCH, FRAMES // 2)
            result['roofline']['other']['four_batches_per_launch'] = {
                'utterances_per_launch': 4 * BATCH, 'chains_in_flight': 1, 'ms_per_launch_chain': 1e3 * dt4 / 8,
                'ms_per_32_utterances': 1e3 * dt4 / 8 / 4, 'kernel': r4['kernel'], 'work_groups_per_launch': r4['_wgs'],
                'avg_launch_us': r4['avg_launch_us'], 'hbm_frac': r4['other']['hbm_frac'], 'mfma_frac': r4['other']['mfma_frac'],
                'note': 'context only: batch 128 per launch is not BASELINE.json configuration 2 (bs32); it is what the dominant kernel '
                        'and the step cost when a launch fills the chip by itself'}
            for e_ in lane4['engs']:
                e_.close()
            log(f"four batches per launch (context): {result['roofline']['other']['four_batches_per_launch']['ms_per_32_utterances']:.3f} ms per 32 utterances")
        # ---- CPU baseline: the reference's fake-quant op sequence on this host's cores (N=1 only) --------------
        if not args.no_cpu_baseline:
            from oracle.fakequant_torch import FakeQuantNet
            cores = host_cores()
            torch.set_num_threads(cores)
            log(f'roofline pass done; CPU baseline on {cores} host threads')
            sd = synth.make_state_dict(cfg, 0)
            net = FakeQuantNet(topology.conv_plan(cfg), cfg, sd, w.amin, w.amax, wbit, abit)
            audio0 = torch.from_numpy(synth.make_audio(BATCH, SAMPLES, seed=w.audio_seed(rank, (args.steps - 1) % S))).to(dev)
            feats, _ = engine.frontend_mel(audio0, w.alen, w.fb, w.window, 0.97, 16)
            Bs = BATCH                                           # the same 32 x 500-frame batch the GPU timed
            x = feats[:Bs, :, :FRAMES].cpu().numpy()
            lens = [FRAMES] * Bs
            n_warm, n_timed = 3, 10                              # 3 warm-up + 10 timed forwards, median
            if args.config == 'jasper':
                n_warm, n_timed = 1, 3
            for _ in range(n_warm):
                net.forward(x, lens)
            log('cpu baseline warm-up done')
            ts = []
            for _ in range(n_timed):
                t1 = time.perf_counter()
                out = net.forward(x, lens)
                ts.append(time.perf_counter() - t1)
            tc = float(np.median(ts))
            agree = float((out['tokens'].numpy() == tokens.cpu().numpy()[:Bs, :out['tokens'].shape[1]]).mean())
            result['cpu_baseline'] = {
                'value': Bs * FRAMES * 0.01 / tc, 'unit': 'audio-s/wall-s', 'cores': torch.get_num_threads(),
                'kind': 'port',
                'sample': f'{n_warm} warm-up + {n_timed} timed forwards (median {tc:.2f} s) of encoder + decoder on the first '
                          f'{Bs} utterances x {FRAMES} frames of the last timed batch (the whole batch); the mel front-end is NOT included '
                          f'(features come from the GPU front-end); token agreement with the GPU run {agree:.4f}'}
    if rank == 0:
        print(json.dumps(result))
    if use_dist:
        dist.destroy_process_group()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args, argv))
    run(args)


if __name__ == '__main__':
    main()
