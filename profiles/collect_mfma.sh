#!/bin/bash
# rocprofv3's own matrix-pipe / issue / LDS counters for the kernels of one bench configuration (north_star: "rocprof
# int8-MFMA utilisation"), each counter group in its OWN run with --kernel-trace only (SQ block: 8 slots per pass):
#   A  SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES
#   B  SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU
#   C  SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
#   D  SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL
#   E  GRBM_GUI_ACTIVE
# bench.py runs with --streams 1 --tile <tile>: every launch alone on the chip, like the kernel-stats pass of collect.sh.
# Output: gpurun_out/<tag>_pmc_MFMA.txt (table) and <tag>_pmc_MFMA.json (what bench.py's roofline.other reads once the
# file is committed under profiles/).  A group whose counters this rocprofv3 does not know is reported and skipped.
# usage: bash profiles/collect_mfma.sh <tag> ["--config jasper"] [tile]
set -o pipefail
TAG=${1:-rXX}
EXTRA=${2:-}
TILE=${3:-128}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/${TAG}_counters_available.txt 2>&1 || true
declare -A CG
CG[A]="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
CG[B]="SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU"
CG[C]="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
CG[D]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL"
CG[E]="GRBM_GUI_ACTIVE"
for G in A B C D E; do
  # keep only the counters this installation lists (an unknown name fails the whole pass)
  USE=""
  for C in ${CG[$G]}; do
    if grep -qw "$C" $OUT/${TAG}_counters_available.txt; then USE="$USE $C"; else echo "group $G: counter $C not available" >> $OUT/${TAG}_pmc_MFMA.skipped; fi
  done
  [ -z "$USE" ] && continue
  echo "[collect_mfma] group $G:$USE"
  rocprofv3 --kernel-trace --pmc $USE --output-format csv -d $OUT/pmcm_${TAG}_$G -o c -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs --streams 1 --tile $TILE $EXTRA > $OUT/pmcm_${TAG}_$G.log 2>&1 \
    || echo "group $G failed (see pmcm_${TAG}_$G.log)" >> $OUT/${TAG}_pmc_MFMA.skipped
done
GRAFT_REPO_ROOT=$R python3 - "$TAG" "$OUT" "$EXTRA" "$TILE" <<'PY'
import collections, csv, glob, json, os, sys
tag, out, extra, tile = sys.argv[1:5]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))     # kernel -> counter -> [n, sum]
dur = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(os.path.join(out, f'pmcm_{tag}_*', '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'qasr::' not in k:
            continue
        a = agg[k][row['Counter_Name']]
        a[0] += 1
        a[1] += float(row['Counter_Value'])
for f in glob.glob(os.path.join(out, f'pmcm_{tag}_A', '**', '*kernel_trace.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'qasr::' in k:
            d = dur[k]
            d[0] += 1
            d[1] += (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) * 1e-3
kern = {}
for k, cs in agg.items():
    avg = {c: v[1] / v[0] for c, v in cs.items() if v[0]}
    row = {'dispatches': max(v[0] for v in cs.values()), **{c: round(x, 1) for c, x in avg.items()}}
    if dur[k][0]:
        row['avg_us_under_pmc'] = round(dur[k][1] / dur[k][0], 2)
    kern[k] = row
doc = {'tag': tag, 'command': f'bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs --streams 1 --tile {tile} {extra}'.strip(),
       'note': '', 'kernels': kern}
path = os.path.join(out, f'{tag}_pmc_MFMA.json')
json.dump(doc, open(path, 'w'), indent=1)
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', os.getcwd()), 'profiles'))
import pmc_mfma_derive as D
doc = D.derive(doc)
json.dump(doc, open(path, 'w'), indent=1)
open(path[:-5] + '.txt', 'w').write(D.table(doc))
PY
rm -rf $OUT/pmcm_${TAG}_A $OUT/pmcm_${TAG}_B $OUT/pmcm_${TAG}_C $OUT/pmcm_${TAG}_D $OUT/pmcm_${TAG}_E
ls -la $OUT | grep ${TAG}_
