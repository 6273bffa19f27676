"""MFMA-utilisation table per conv kernel from rocprofv3 --pmc counter_collection.csv files (tools/pmc_mfma.sh).
Counters are summed per dispatch by rocprofv3; every row below is the mean over the dispatches of one kernel.
  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES-equivalent SIMD cycles): SQ_VALU_MFMA_BUSY_CYCLES counts cycles
  (64 per v_mfma_f32_32x32x2_f32, 32 per v_mfma_f32_32x32x16_bf16: MI355X_MICROARCH.md) summed over SIMDs;
  GRBM_GUI_ACTIVE = shader-clock cycles of the dispatch -> busy fraction = MFMA_BUSY / (GRBM_GUI_ACTIVE * 4 SIMDs * CUs used)."""
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
print('| kernel | dispatches | ' + ' | '.join(cols) + ' | MFMA busy / (GUI_ACTIVE x 4 x 256) | VALU insts per MFMA |')
print('|---|---|' + '---|' * (len(cols) + 2))
for n, d in agg.items():
    m = {c: (sum(d[c]) / len(d[c]) if c in d else None) for c in cols}
    cnt = max(len(v) for k, v in d.items() if not k.startswith('_'))
    busy = (m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['GRBM_GUI_ACTIVE'] * 4 * 256)) if m['SQ_VALU_MFMA_BUSY_CYCLES'] and m['GRBM_GUI_ACTIVE'] else None
    vpm = (m['SQ_INSTS_VALU'] / m['SQ_INSTS_MFMA']) if m['SQ_INSTS_MFMA'] else None
    print('| `%s` | %d | ' % (n, cnt) + ' | '.join('%.4g' % m[c] if m[c] is not None else '-' for c in cols) +
          ' | %s | %s |' % ('%.3f' % busy if busy is not None else '-', '%.2f' % vpm if vpm is not None else '-'))
