#!/bin/bash
# wgrad4: which LDS access class carries the bank conflicts?  time + SQ LDS counters per probe build, isolated kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
rocprofv3 -L 2>/dev/null | grep -i "LDS" | head -30 > $O/r4k_lds_counters.txt
out=$O/r4k_g4_probes.txt; : > $out
for lib in "" g4_noput g4_novstore g4_nozstore g4_nostores g4_novread g4_nofrag; do
  if [ -n "$lib" ]; then export AIDE_HIP_LIB=$R/abtest/lib_$lib.so; else unset AIDE_HIP_LIB; fi
  for L in "128 64 256" "512 256 64"; do
    python tools/ab_one.py wgrad4 $L 2>/dev/null | tail -1 >> $out
  done
  tag=r4k_${lib:-base}
  bash tools/pmc.sh $tag SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_MFMA -- python $R/tools/ab_one.py wgrad4 128 64 256 > /dev/null 2>&1
  python - "$tag" >> $out <<'PY'
import csv, glob, os, sys, collections
tag = sys.argv[1]
f = glob.glob(os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out', 'pmc_' + tag, '**', '*counter_collection.csv'), recursive=True)
acc = collections.defaultdict(lambda: [0.0, 0])
for p in f:
    for r in csv.DictReader(open(p)):
        if 'wgrad4_kernel' in r['Kernel_Name']:
            a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
print('   PMC %-12s ' % tag + '  '.join('%s %.3e' % (k, v[0] / max(v[1], 1)) for k, v in sorted(acc.items())))
PY
done
cat $out; cat $O/r4k_lds_counters.txt | head -20
