#!/bin/bash
# Collects the profiles a round commits (run on the GPU box through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the default bench command      -> gpurun_out/prof_<tag>/
#   2. PMC passes (FETCH_SIZE, WRITE_SIZE; each in its own run, kernel-trace only) -> gpurun_out/pmc_<tag>_*.txt
# usage: bash profiles/collect.sh <tag>
set -o pipefail
TAG=${1:-rXX}
EXTRA=${2:-}                     # e.g. "--config jasper"
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py $EXTRA --steps 50 --warmup 5 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || exit 1
# kernel durations: one step in flight (what bench.py's HIP-event roofline pass measures), then the default 4 in flight
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o p -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --streams 1 --tile 128 $EXTRA > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_rocprof.err || exit 2
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof4_$TAG -o p -- python3 $R/bench.py $EXTRA --steps 50 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_bench_under_rocprof_4inflight.json 2>> $OUT/${TAG}_rocprof.err || exit 2
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_${TAG}_$C -o c -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --streams 1 --tile 128 $EXTRA > $OUT/pmc_${TAG}_$C.log 2>&1 || exit 3
done
python3 - "$TAG" "$OUT" <<'PY'
import collections, csv, glob, os, sys
tag, out = sys.argv[1:3]
# kernel stats filtered to this repo's kernels
for d, suffix in (('prof_', ''), ('prof4_', '_4inflight')):
    for f in glob.glob(os.path.join(out, d + tag, '**', '*kernel_stats.csv'), recursive=True):
        rows = list(csv.reader(open(f)))
        keep = [rows[0]] + [r for r in rows[1:] if 'qasr::' in r[0]]
        csv.writer(open(os.path.join(out, tag + '_kernel_stats' + suffix + '.csv'), 'w')).writerows(keep)
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(out, f'pmc_{tag}_{c}', '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get('Counter_Name') == c and 'qasr::' in row['Kernel_Name']:
                a = agg[row['Kernel_Name'].split('(')[0]]
                a[0] += 1
                a[1] += float(row['Counter_Value'])
    with open(os.path.join(out, f'{tag}_pmc_{c}.txt'), 'w') as fh:
        fh.write(f'# rocprofv3 --kernel-trace --pmc {c}: per kernel, dispatches / sum / average per dispatch (counter units as reported: KB)\n')
        for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            fh.write(f'{k:70s} {n:6d} {v:14.1f} {v / n:12.2f}\n')
PY
# keep the merged directory small (gpurun_out is capped at 64 MiB)
rm -rf $OUT/prof_$TAG $OUT/prof4_$TAG $OUT/pmc_${TAG}_FETCH_SIZE $OUT/pmc_${TAG}_WRITE_SIZE
ls -la $OUT | grep $TAG
