"""MFMA-utilisation table per conv kernel from rocprofv3 --pmc counter_collection.csv files (tools/pmc_mfma.sh).
Counters are summed per dispatch by rocprofv3; every row below is the mean over the dispatches of one kernel.
  SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe cycles (64 per v_mfma_f32_32x32x2_f32, 32 per v_mfma_f32_32x32x16_bf16:
  MI355X_MICROARCH.md; checked: = 64 x SQ_INSTS_MFMA for the fp32 kernels) summed over all SIMDs of the chip.
  GRBM_GUI_ACTIVE is summed over the 8 XCDs (calibrated: / 8 = dispatch duration x ~1.9 GHz), so
      MFMA pipe busy (chip) = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)
  = the fraction of the chip's matrix-pipe cycles the dispatch keeps busy (counter passes serialise the kernels, so this
  is each kernel ALONE; a kernel launched on 128 of 256 CUs cannot exceed 0.5).
  MFMA busy per resident wave = MFMA_BUSY / (4 x SQ_WAVE_CYCLES) (SQ_WAVE_CYCLES counts quad-cycles): 1 wave per SIMD ->
  the matrix-pipe utilisation of an occupied SIMD."""
import csv, sys, re, collections
agg = collections.OrderedDict()
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).replace('void ', '')
        n = re.sub(r'[<(].*', '', n)
        if 'conv3x3' not in n and 'wgrad' not in n:
            continue
        d = agg.setdefault(n, collections.OrderedDict())
        d.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
        d.setdefault('_wgs', []).append(float(r.get('Grid_Size', 0) or 0) / max(float(r.get('Workgroup_Size', 1) or 1), 1))
print('# MFMA / issue counters per conv kernel (rocprofv3 --pmc inside the training step; mean per dispatch)\n')
cols = ['SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_INSTS_MFMA', 'SQ_INSTS_VALU', 'GRBM_GUI_ACTIVE',
        'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_INSTS_LDS',
        'SQ_LDS_BANK_CONFLICT']
print('| kernel | dispatches | ' + ' | '.join(cols) + ' | MFMA pipe busy (chip) | MFMA busy per resident wave | VALU insts per MFMA |')
print('|---|---|' + '---|' * (len(cols) + 3))
for n, d in agg.items():
    m = {c: (sum(d[c]) / len(d[c]) if c in d else None) for c in cols}
    cnt = max(len(v) for k, v in d.items() if not k.startswith('_'))
    busy = (m['SQ_VALU_MFMA_BUSY_CYCLES'] * 8.0 / (m['GRBM_GUI_ACTIVE'] * 1024)) if m['SQ_VALU_MFMA_BUSY_CYCLES'] and m['GRBM_GUI_ACTIVE'] else None
    perw = (m['SQ_VALU_MFMA_BUSY_CYCLES'] / (4.0 * m['SQ_WAVE_CYCLES'])) if m['SQ_VALU_MFMA_BUSY_CYCLES'] and m['SQ_WAVE_CYCLES'] else None
    vpm = (m['SQ_INSTS_VALU'] / m['SQ_INSTS_MFMA']) if m['SQ_INSTS_MFMA'] else None
    print('| `%s` | %d | ' % (n, cnt) + ' | '.join('%.4g' % m[c] if m[c] is not None else '-' for c in cols) +
          ' | %s | %s | %s |' % ('%.3f' % busy if busy is not None else '-', '%.3f' % perw if perw is not None else '-',
                                 '%.2f' % vpm if vpm is not None else '-'))
