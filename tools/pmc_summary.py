#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV passes of one conv kernel (tools/pmc_one.sh) into one markdown row set.
Usage: python tools/pmc_summary.py gpurun_out/pc_<layer>_<kind>   (prefix; passes _a/_b/_c[/_d/_e] are read if present)"""
import csv, glob, os, re, sys

csv.field_size_limit(1 << 30)


def read_pass(d):
    """-> {kernel short name: {counter: mean value per dispatch}, '_dur': mean duration us}"""
    out = {}
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'igemm' not in k and 'wgrad' not in k and 'pw_gemm' not in k:
                continue
            if 'reduce' in k or 'pack' in k:
                continue
            m = re.search(r'(igemm_kd_kernel|igemm_s2k_kernel|igemm_s2d_kernel|igemm_box_kernel|igemm_s2_kernel|pw_gemm_kernel|pw_wgrad_kernel|igemm_pc_kernel|igemm_kernel|wgrad_pc_kernel|wgrad_kernel|igemm_ws_kernel|wgrad\w*_kernel)<([^>]*)>', k)
            name = (m.group(1) + '<' + m.group(2) + '>') if m else k[:60]
            e = out.setdefault(name, {})
            e.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
            e.setdefault('_dur', []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
            e['_regs'] = (r['VGPR_Count'], r['Accum_VGPR_Count'], r['SGPR_Count'], r['LDS_Block_Size'], r['Grid_Size'], r['Workgroup_Size'])
    return {k: {c: (sum(v) / len(v) if isinstance(v, list) else v) for c, v in e.items()} for k, e in out.items()}


def main():
    pre = sys.argv[1]
    merged = {}
    for suf in 'abcdefg':
        d = f'{pre}_{suf}'
        if not os.path.isdir(d):
            continue
        for k, e in read_pass(d).items():
            merged.setdefault(k, {}).update({c: v for c, v in e.items() if c != '_dur' or '_dur' not in merged.get(k, {})})
    for k, e in merged.items():
        print(f'### {k}')
        vg, ag, sg, lds, grid, wg = e['_regs']
        print(f'- grid {grid} threads / wg {wg}, VGPR {vg} + AGPR {ag}, SGPR {sg}, LDS {lds} B; duration {e["_dur"]:.1f} us (profiled clock)')
        class _Z(float):                                     # a counter that read 0 (LDS-free kernels): ratios print as nan instead of raising
            def __rtruediv__(self, o):
                return float('nan') if self == 0 else float(o) / float(self)
        g = lambda c: _Z(e.get(c, float('nan')))
        wc = g('SQ_WAVE_CYCLES')
        mf = g('SQ_VALU_MFMA_BUSY_CYCLES')
        busy = g('SQ_BUSY_CYCLES')
        waves = g('SQ_WAVES')
        rows = [
            ('SQ_WAVES', f'{waves:.0f}'),
            ('SQ_BUSY_CYCLES (sum over SEs)', f'{busy:.3e}'),
            ('SQ_WAVE_CYCLES (quad-cycles)', f'{wc:.3e}'),
            ('ACTIVE_INST_ANY / WAVE_CYCLES', f'{100 * g("SQ_ACTIVE_INST_ANY") / wc:.1f} %'),
            ('WAIT_ANY / WAVE_CYCLES', f'{100 * g("SQ_WAIT_ANY") / wc:.1f} %'),
            ('WAIT_INST_ANY / WAVE_CYCLES', f'{100 * g("SQ_WAIT_INST_ANY") / wc:.1f} %'),
            ('MFMA instr (MFMA_BUSY/32) total', f'{mf / 32:.3e}'),
            ('MFMA-busy: MFMA_BUSY_CYCLES / (1024 SIMDs x duration x clk)', 'see below'),
            ('VALU (non-MFMA incl.) per MFMA', f'{g("SQ_INSTS_VALU") / (mf / 32):.2f}'),
            ('SALU per MFMA', f'{g("SQ_INSTS_SALU") / (mf / 32):.2f}'),
            ('LDS instr per MFMA', f'{g("SQ_INSTS_LDS") / (mf / 32):.2f}'),
            ('VMEM rd / wr per MFMA', f'{g("SQ_INSTS_VMEM_RD") / (mf / 32):.3f} / {g("SQ_INSTS_VMEM_WR") / (mf / 32):.3f}'),
            ('LDS_BANK_CONFLICT / LDS_IDX_ACTIVE', f'{100 * g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"):.1f} %'),
            ('LDS_IDX_ACTIVE cycles per MFMA (per CU: x4 SIMDs share one LDS)', f'{g("SQ_LDS_IDX_ACTIVE") / (mf / 32):.2f}'),
            ('WAIT_INST_LDS / WAVE_CYCLES', f'{100 * g("SQ_WAIT_INST_LDS") / wc:.1f} %'),
            ('ACTIVE_INST_{VALU,LDS,VMEM,SCA,MISC} / WAVE_CYCLES', ' / '.join(f'{100 * g("SQ_ACTIVE_INST_" + x) / wc:.1f}' for x in ('VALU', 'LDS', 'VMEM', 'SCA', 'MISC')) + ' %'),
            ('INST_LEVEL_VMEM / INST_LEVEL_LDS (avg in flight per wave-cycle)', f'{g("SQ_INST_LEVEL_VMEM") / wc:.2f} / {g("SQ_INST_LEVEL_LDS") / wc:.2f}'),
            ('FETCH_SIZE x2 / WRITE_SIZE (MB)', f'{2 * g("FETCH_SIZE") / 1024:.1f} / {g("WRITE_SIZE") / 1024:.1f}'),
            ('GRBM_GUI_ACTIVE', f'{g("GRBM_GUI_ACTIVE"):.3e}'),
        ]
        print('| counter | value |\n|---|---|')
        for a, b in rows:
            if 'nan' not in b and 'see below' not in b:
                print(f'| {a} | {b} |')
        # MFMA utilisation: busy cycles summed over all SIMDs / (SIMD count x kernel cycles); kernel cycles from GRBM_GUI_ACTIVE if present
        if 'GRBM_GUI_ACTIVE' in e:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs: kernel cycles = GRBM_GUI_ACTIVE / 8
            print(f'| MFMA busy = MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) | {100 * mf * 8 / (1024 * g("GRBM_GUI_ACTIVE")):.1f} % |')
        print()


if __name__ == '__main__':
    main()
