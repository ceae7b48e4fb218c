#!/usr/bin/env python3
"""Derived matrix-pipe figures from the raw per-dispatch counter averages collect_mfma.sh gathers (one JSON per tag).
Re-runnable offline: python profiles/pmc_mfma_derive.py profiles/r04_v1_pmc_MFMA.json

What the raw counters mean on gfx950, calibrated on k_sep2<75, 4, 0, 2, false, 128, 1> whose instruction counts are known
(64 work-groups x 8 waves x (128 v_mfma_i32_32x32x32_i8 + 672 v_mfma_i32_4x4x4_16B_i8)):
  SQ_VALU_MFMA_BUSY_CYCLES   = sum over waves of 32 cycles per 32x32x32 and 8 per 4x4x4 MFMA   (4 849 664: exact)
  SQ_INSTS_VALU_MFMA_MOPS_I8 = int8 MFMA operations / 512                                      (9 764 864: exact)
  SQ_BUSY_CU_CYCLES          = sum over CUs of the cycles a CU holds waves (35.7 k per work-group's CU)
  SQ_BUSY_CYCLES             = sum over the 32 shader engines of their busy cycles (/ 32 = the dispatch in shader cycles)
  GRBM_GUI_ACTIVE / 8        reads 40 % high on these 10-30 us dispatches (MI355X_MICROARCH.md: "reads high below 0.3 ms"): not used
Derived per kernel:
  mfma_busy_frac            = MFMA_BUSY / (BUSY_CU_CYCLES x 4 SIMDs): share of the cycles in which the CUs that hold the launch's
                              work-groups have their matrix pipes busy - rocprof's MfmaUtil restricted to the occupied CUs
  mfma_busy_frac_chip_alone = MFMA_BUSY / (SQ_BUSY_CYCLES / 32 x 256 CUs x 4): the same over the WHOLE chip for a launch that runs
                              alone (a 64-work-group launch occupies a quarter of it)
  int8_gop, int8_top_s_under_pmc, int8_frac_of_peak_alone: counted int8 MFMA operations per dispatch, per second of the
                              dispatch's duration under the profiler, and / 5.03 POP/s (2.4 GHz nominal)"""
import json
import sys

N_CU, N_SIMD, N_SE, PEAK = 256, 4, 32, 256 * 4 * 2048 * 2.4e9


def derive(doc):
    for k, r in doc['kernels'].items():
        for f in ('mfma_busy_frac', 'mfma_busy_frac_of_busy_cus', 'mfma_busy_frac_chip_alone', 'elapsed_shader_cycles', 'int8_gop',
                  'int8_top_s_under_pmc', 'int8_frac_of_peak_alone', 'valu_insts_per_mfma_inst', 'lds_bank_conflict_frac'):
            r.pop(f, None)
        mb, cu, sq = r.get('SQ_VALU_MFMA_BUSY_CYCLES'), r.get('SQ_BUSY_CU_CYCLES'), r.get('SQ_BUSY_CYCLES')
        if mb is not None and cu:
            r['mfma_busy_frac'] = mb / (cu * N_SIMD)
        if mb is not None and sq:
            r['elapsed_shader_cycles'] = sq / N_SE
            r['mfma_busy_frac_chip_alone'] = mb / (sq / N_SE * N_CU * N_SIMD)
        mops = r.get('SQ_INSTS_VALU_MFMA_MOPS_I8')
        if mops is not None:
            r['int8_gop'] = mops * 512 / 1e9
            if r.get('avg_us_under_pmc'):
                r['int8_top_s_under_pmc'] = mops * 512 / (r['avg_us_under_pmc'] * 1e-6) / 1e12
                r['int8_frac_of_peak_alone'] = mops * 512 / (r['avg_us_under_pmc'] * 1e-6) / PEAK
        if r.get('SQ_INSTS_MFMA'):
            r['valu_insts_per_mfma_inst'] = (r.get('SQ_INSTS_VALU', 0) - r['SQ_INSTS_MFMA']) / r['SQ_INSTS_MFMA']
        if r.get('SQ_LDS_IDX_ACTIVE'):
            r['lds_bank_conflict_frac'] = r.get('SQ_LDS_BANK_CONFLICT', 0) / r['SQ_LDS_IDX_ACTIVE']
    doc['note'] = ('per-dispatch averages, each counter group collected in its own rocprofv3 --kernel-trace --pmc run of the command '
                   'above; derived fields: profiles/pmc_mfma_derive.py (docstring: what each raw counter counts on gfx950)')
    return doc


def table(doc):
    cols = [('mfma_busy_frac', 9, '.3f'), ('mfma_busy_frac_chip_alone', 10, '.3f'), ('int8_gop', 8, '.2f'), ('int8_frac_of_peak_alone', 9, '.3f'),
            ('SQ_VALU_MFMA_BUSY_CYCLES', 12, '.0f'), ('SQ_BUSY_CU_CYCLES', 12, '.0f'), ('SQ_VALU_MFMA_COEXEC_CYCLES', 12, '.0f'),
            ('SQ_WAVE_CYCLES', 12, '.0f'), ('SQ_WAIT_INST_ANY', 12, '.0f'), ('SQ_WAIT_ANY', 12, '.0f'), ('SQ_ACTIVE_INST_ANY', 12, '.0f'),
            ('valu_insts_per_mfma_inst', 8, '.2f'), ('lds_bank_conflict_frac', 8, '.3f')]
    short = {'mfma_busy_frac': 'mfma_busy', 'mfma_busy_frac_chip_alone': 'chip_alone', 'int8_frac_of_peak_alone': 'int8/peak',
             'valu_insts_per_mfma_inst': 'valu/mfma', 'lds_bank_conflict_frac': 'lds_confl'}
    out = ['# ' + doc.get('command', ''), '# ' + doc['note'],
           f'{"kernel":46s} {"n":>5s} {"us":>7s} ' + ' '.join(f'{short.get(c, c.replace("SQ_", "").replace("_CYCLES", ""))[:w]:>{w}s}' for c, w, _ in cols)]
    rows = sorted(doc['kernels'].items(), key=lambda kv: -kv[1].get('SQ_VALU_MFMA_BUSY_CYCLES', 0) * kv[1]['dispatches'])
    for k, r in rows:
        out.append(f'{k.replace("qasr::", "")[:46]:46s} {r["dispatches"]:5d} {r.get("avg_us_under_pmc", 0):7.2f} ' +
                   ' '.join(f'{r.get(c, float("nan")):{w}{f}}' for c, w, f in cols))
    return '\n'.join(out) + '\n'


if __name__ == '__main__':
    path = sys.argv[1]
    doc = derive(json.load(open(path)))
    json.dump(doc, open(path, 'w'), indent=1)
    open(path[:-5] + '.txt', 'w').write(table(doc))
    print(table(doc))
