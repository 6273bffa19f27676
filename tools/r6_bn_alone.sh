# usage (GPU box): tools/r6_bn_alone.sh -> gpurun_out/bn_alone.txt: every BatchNorm kernel alone on the FuseUNet level shapes (rocprofv3 kernel
# durations: the Python loop of tools/bench_bn.py is host-bound below ~13 us), shipped library vs the two-pass kernels everywhere
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
: > $O/bn_alone.txt
for cfg in "c2 fp32" "c5 bf16"; do for v in "" bn_2pass; do
  if [ -n "$v" ]; then export AIDE_HIP_LIB=$R/abtest/lib_$v.so; else unset AIDE_HIP_LIB; fi
  rm -rf $O/bnp; rocprofv3 --kernel-trace --stats --output-format csv -d $O/bnp -o bn -- python $R/tools/bench_bn.py $cfg 20 > /dev/null 2>&1
  echo "== $cfg ${v:-shipped (one pass where the size rule admits it)}: kernel, calls, average us" >> $O/bn_alone.txt
  python - "$(find $O/bnp -name '*kernel_stats.csv' | head -1)" >> $O/bn_alone.txt <<'PY'
import csv, sys
for r in list(csv.reader(open(sys.argv[1])))[1:]:
    n = r[0].replace('void (anonymous namespace)::', '')
    if n.startswith('bn_'):
        print('   %-70s %5s %9.1f' % (n[:70], r[1], float(r[3]) / 1e3))
PY
done; done
cat $O/bn_alone.txt
