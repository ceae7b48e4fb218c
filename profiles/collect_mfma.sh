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
python3 - "$TAG" "$OUT" "$EXTRA" "$TILE" <<'PY'
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
N_CU, N_SIMD = 256, 4
kern = {}
for k, cs in agg.items():
    avg = {c: v[1] / v[0] for c, v in cs.items() if v[0]}
    row = {'dispatches': max(v[0] for v in cs.values()), **{c: round(x, 1) for c, x in avg.items()}}
    if dur[k][0]:
        row['avg_us_under_pmc'] = round(dur[k][1] / dur[k][0], 2)
    mb = avg.get('SQ_VALU_MFMA_BUSY_CYCLES')
    if mb is not None:
        # busy cycles of the matrix pipes summed over SIMDs: against (a) the CUs that had waves (SQ_BUSY_CU_CYCLES counts, per
        # the gfx94x MfmaUtil formula's units, one per CU and cycle) and (b) the whole chip for the dispatch (GRBM_GUI_ACTIVE is
        # the sum over the 8 XCDs: / 8 = elapsed shader cycles)
        if avg.get('SQ_BUSY_CU_CYCLES'):
            row['mfma_busy_frac_of_busy_cus'] = mb / (avg['SQ_BUSY_CU_CYCLES'] * N_SIMD)
        if avg.get('GRBM_GUI_ACTIVE'):
            row['mfma_busy_frac'] = mb / (avg['GRBM_GUI_ACTIVE'] / 8.0 * N_CU * N_SIMD)
            row['elapsed_shader_cycles'] = avg['GRBM_GUI_ACTIVE'] / 8.0
    kern[k] = row
doc = {'tag': tag, 'command': f'bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs --streams 1 --tile {tile} {extra}'.strip(),
       'note': 'per-dispatch averages; each counter group collected in its own rocprofv3 --kernel-trace --pmc run; mfma_busy_frac = '
               'SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs) (the gfx94x MfmaUtil formula; ROCm 7.2 has no gfx950 '
               'derived-metric section); *_of_busy_cus divides by SQ_BUSY_CU_CYCLES x 4 instead (only CUs that held waves)',
       'kernels': kern}
json.dump(doc, open(os.path.join(out, f'{tag}_pmc_MFMA.json'), 'w'), indent=1)
cols = ['SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CU_CYCLES', 'GRBM_GUI_ACTIVE', 'SQ_INSTS_VALU_MFMA_MOPS_I8', 'SQ_VALU_MFMA_COEXEC_CYCLES',
        'SQ_WAVE_CYCLES', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE']
with open(os.path.join(out, f'{tag}_pmc_MFMA.txt'), 'w') as fh:
    fh.write('# ' + doc['command'] + '\n# ' + doc['note'] + '\n')
    fh.write(f'{"kernel":60s} {"n":>5s} {"us":>8s} {"mfma_busy":>9s} {"of_busy_cu":>10s} ' + ' '.join(f'{c[3:] if c.startswith("SQ_") else c:>26s}' for c in cols) + '\n')
    for k, r in sorted(kern.items(), key=lambda kv: -kv[1].get('SQ_VALU_MFMA_BUSY_CYCLES', 0) * kv[1]['dispatches']):
        fh.write(f'{k.replace("qasr::", ""):60s} {r["dispatches"]:5d} {r.get("avg_us_under_pmc", 0):8.2f} {r.get("mfma_busy_frac", float("nan")):9.3f} '
                 f'{r.get("mfma_busy_frac_of_busy_cus", float("nan")):10.3f} ' + ' '.join(f'{r.get(c, float("nan")):26.1f}' for c in cols) + '\n')
PY
rm -rf $OUT/pmcm_${TAG}_A $OUT/pmcm_${TAG}_B $OUT/pmcm_${TAG}_C $OUT/pmcm_${TAG}_D $OUT/pmcm_${TAG}_E
ls -la $OUT | grep ${TAG}_
