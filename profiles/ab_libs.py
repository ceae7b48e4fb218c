#!/usr/bin/env python3
"""A/B of several builds of the native library in ONE gpurun call, interleaved (A B C A B C ...), each run a fresh process:
    python profiles/ab_libs.py out.txt hip tail pf both            # names = q-asr_amd/qasr/libqasr_<name>.so
Per run: bench.py --no-cpu-baseline --no-other-configs at K = 200 and at the driver's K = 20; reported: ms/step with 4 steps in
flight, one step in flight, and the dominant kernel alone (HIP events)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_path, names = sys.argv[1], sys.argv[2:]
extra = os.environ.get('AB_EXTRA', '').split()
rounds = int(os.environ.get('AB_ROUNDS', 2))
rows = {n: [] for n in names}
with open(out_path, 'w') as fh:
    fh.write(f'# bench.py {" ".join(extra)} --no-cpu-baseline --no-other-configs, builds interleaved, {rounds} rounds; columns: K, ms/step (4 in flight), '
             'ms/step (1 in flight, 32-frame tiles), dominant kernel, its launch alone (us)\n')
    for r in range(rounds):
        for n in names:
            lib = os.path.join(ROOT, 'q-asr_amd', 'qasr', f'libqasr_{n}.so')
            for K, W in ((200, 10), (20, 5)):
                p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', str(K), '--warmup', str(W), '--no-cpu-baseline',
                                    '--no-other-configs'] + extra, capture_output=True, text=True, env=dict(os.environ, QASR_LIB=lib))
                if p.returncode:
                    fh.write(f'{n:8s} K={K}: FAILED rc={p.returncode}: {p.stderr[-400:]}\n')
                    fh.flush()
                    continue
                d = json.loads(p.stdout.strip().splitlines()[-1])
                rf = d['roofline']
                line = (f'{n:8s} round {r} K={K:3d}  {d["ms_per_step"]:.4f}  {d.get("single_stream_ms_per_step", float("nan")):.4f}  '
                        f'{rf["kernel"]}  {rf["avg_launch_us"]:.2f}  serial {rf["other"]["all_ops_ms_per_step_serial"]:.4f}')
                rows[n].append((K, d['ms_per_step']))
                fh.write(line + '\n')
                fh.flush()
                print(line, flush=True)
    fh.write('# mean ms/step per build: ' + '; '.join(
        f'{n}: K=200 {sum(v for k, v in rows[n] if k == 200) / max(1, sum(1 for k, v in rows[n] if k == 200)):.4f}, '
        f'K=20 {sum(v for k, v in rows[n] if k == 20) / max(1, sum(1 for k, v in rows[n] if k == 20)):.4f}' for n in names) + '\n')
