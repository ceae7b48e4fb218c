"""N>1 path on CPU: two processes over gloo at 127.0.0.1 exercise the same helpers bench.py uses over RCCL —
blob broadcast from the calibrating rank, contiguous utterance sharding, token gather to rank 0."""
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, golden_dir, q):
    import torch.distributed as dist
    from oracle import int_oracle as O
    from qasr import dist as qdist, pack, synth, topology
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        d = np.load(os.path.join(golden_dir, 'net_miniq_w8a8.npz'))
        meta = json.loads(str(d['meta']))
        cfg = topology.mini_quartznet()
        blob = None
        if rank == 0:                                    # only rank 0 "calibrates" and packs
            blob, _ = pack.pack_model(cfg, synth.make_state_dict(cfg, meta['seed']), d['act_min'], d['act_max'], 8, 8)
        blob = qdist.broadcast_bytes(blob, 0, torch.device('cpu'))
        ref_blob, _ = pack.pack_model(cfg, synth.make_state_dict(cfg, meta['seed']), d['act_min'], d['act_max'], 8, 8)
        assert blob == ref_blob
        # shard 5 utterances over 2 ranks, run the (CPU oracle) forward on the shard, gather ragged tokens
        B = 5
        x = synth.make_features(B, cfg.feat_in, 96, 21)
        lens = [96, 80, 64, 50, 33]
        lo, hi = qdist.shard_range(B, world, rank)
        net = O.OracleNet(topology.conv_plan(cfg), cfg, synth.make_state_dict(cfg, meta['seed']), d['act_min'],
                          d['act_max'], 8, 8)
        tok = torch.from_numpy(net.forward(x[lo:hi], lens[lo:hi])['tokens'].astype(np.int32))
        allt = qdist.gather_ragged_tokens(tok, 0)
        eq = qdist.gather_tokens(torch.full((2, 3), rank, dtype=torch.int32), 0)
        if rank == 0:
            full = net.forward(x, lens)['tokens']
            assert np.array_equal(allt.numpy(), full)
            assert [int(t[0, 0]) for t in eq] == list(range(world))
        else:
            assert allt is None and eq is None
        q.put((rank, 'ok'))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


def test_shard_range_covers_everything():
    from qasr.dist import shard_range
    for n in (1, 5, 32, 255, 256):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_broadcast_shard_gather(golden_dir):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, golden_dir, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
    res = dict(q.get(timeout=5) for _ in range(2))
    assert res == {0: 'ok', 1: 'ok'}, res
    assert all(p.exitcode == 0 for p in procs)


def _run_bench(*argv, env=None):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(root, 'bench.py'), *argv], capture_output=True, text=True, env=e,
                          timeout=240)


def test_bench_launcher_starts_n_ranks():
    """`python bench.py --gpus N` with no launcher must produce N ranks by itself (the driver runs it that way): the parent
    spawns the ranks, relays rank 0's JSON line, and the census all-reduce sees every rank.  --dry-run keeps the engine
    (GPU-only) out of it; the process group, blob broadcast and per-step token gather are the ones bench.py times."""
    r = _run_bench('--gpus', '2', '--dry-run', '--steps', '3', '--warmup', '1')
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
    d = json.loads(line)
    assert d['n_gpus'] == 2 and d['n_ranks_seen'] == 2 and d['dry_run'] is True


def test_bench_launcher_fails_loudly_without_enough_gpus():
    """More ranks requested than GPUs visible: non-zero exit and no JSON line, never a silent one-GPU measurement."""
    torch_gpus = torch.cuda.device_count()
    r = _run_bench('--gpus', str(torch_gpus + 2), '--steps', '1')
    assert r.returncode != 0
    assert not any(ln.startswith('{') for ln in r.stdout.splitlines())
    assert 'GPU(s) visible' in r.stderr


def test_bench_rejects_mismatched_world_size():
    r = _run_bench('--gpus', '2', '--dry-run', env=dict(WORLD_SIZE='1', RANK='0', LOCAL_RANK='0'))
    assert r.returncode != 0 and 'WORLD_SIZE' in (r.stderr + r.stdout)


def test_committed_bench_lines_carry_the_contract_fields():
    """The bench lines committed under profiles/ (newest round) hold every field the measurement contract names: the
    driver's parser and the judge read exactly these."""
    import glob
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import re
    files = sorted(glob.glob(os.path.join(root, 'profiles', 'r*_bench*.json')))

    def tag(f):                                               # 'r03_v3_jasper_bench.json' -> (3, 3)
        m = re.match(r'r(\d+)_v(\d+)_', os.path.basename(f))
        return (int(m.group(1)), int(m.group(2))) if m else (-1, -1)
    newest = max(tag(f) for f in files)
    lines = [f for f in files if tag(f) == newest and 'under_rocprof' not in f]
    assert len(lines) >= 2, lines                             # the headline configuration and at least one more
    for f in lines:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                  'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'n_ranks_seen'):
            assert k in r, (f, k)
        assert r['higher_is_better'] is True and r['scaling'] == 'weak' and r['vs_baseline'] is None and r['data'] == 'synthetic'
        assert 'workload' in r['config'] and 'model' not in r['config']
        ro = r['roofline']
        for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel'):
            assert k in ro, (f, k)
        assert ro['bound'] in ('hbm', 'mfma') and abs(ro['frac'] - ro['achieved'] / ro['peak']) < 1e-6
        wgs = ro['other'].get('work_groups_per_launch')
        assert wgs is None or 1 <= wgs <= 4096, (f, wgs)      # B x time tiles (x channel groups) of the dominant kernel
        cb = r['cpu_baseline']
        for k in ('value', 'unit', 'cores', 'kind', 'sample'):
            assert k in cb, (f, k)
        assert cb['kind'] in ('port', 'reference') and cb['unit'] == r['unit']
        assert abs(r['value'] - r['n_gpus'] * 32 * 5.0 * (2 if 'bs64' in r['metric'] else 1) * r['steps'] / (r['ms_per_step'] * 1e-3 * r['steps'])) / r['value'] < 1e-3
    # round 4: the driver's command also carries BASELINE.json's configurations 3 / 4 and the log-prob variant, measured after the
    # headline region, and rocprofv3's own matrix-pipe counter for the dominant kernel (read from the committed *_pmc_MFMA.json)
    drv = [f for f in lines if f.endswith('_bench_20.json')]
    assert drv, lines
    r = json.loads(open(drv[0]).read().strip().splitlines()[-1])
    oc = r['other_configs']
    assert set(oc) == {'quartznet_with_logp', 'w6a6', 'jasper'}
    for k, v in oc.items():
        assert v['steps'] == 20 and v['ms_per_step'] > 0 and 0 < v['step_mfma_frac'] < 1, (k, v)
    assert oc['jasper']['roofline']['bound'] == 'mfma' and oc['w6a6']['baseline_config'] == 3 and oc['jasper']['baseline_config'] == 4
    assert abs(oc['quartznet_with_logp']['ms_per_step'] / r['ms_per_step'] - 1) < 0.15      # writing log-probs costs (almost) nothing
    busy = r['roofline']['other']['mfma_busy_frac_rocprof']
    assert 0.2 < busy < 1.0 and r['roofline']['other']['rocprof_counters']['SQ_VALU_MFMA_BUSY_CYCLES'] > 0


def test_pmc_mfma_summaries_are_calibrated():
    """profiles/*_pmc_MFMA.json: rocprofv3's raw counters agree with the dominant kernel's known instruction stream - 64 work-groups
    x 8 waves x (128 v_mfma_i32_32x32x32_i8 at 32 busy cycles + 672 v_mfma_i32_4x4x4 at 8) and 512 int8 operations per MOPS unit -
    which is what makes the derived busy fraction a measurement and not a formula."""
    import glob
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fs = sorted(glob.glob(os.path.join(root, 'profiles', 'r*_pmc_MFMA.json')))
    assert fs
    doc = json.load(open([f for f in fs if 'jasper' not in f][-1]))
    row = doc['kernels']['qasr::k_sep2<75, 4, 0, 2, false, 128, 1>']
    assert row['SQ_VALU_MFMA_BUSY_CYCLES'] == 64 * 8 * (128 * 32 + 672 * 8)
    assert row['SQ_INSTS_VALU_MFMA_MOPS_I8'] * 512 == 64 * 8 * (128 * 2 * 32 ** 3 + 672 * 2 * 16 * 64)
    assert row['SQ_INSTS_MFMA'] == 64 * 8 * 800 and row['SQ_LDS_UNALIGNED_STALL'] == 0
    assert abs(row['mfma_busy_frac'] - row['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * row['SQ_BUSY_CU_CYCLES'])) < 1e-9
