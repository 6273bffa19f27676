cd $GRAFT_REPO_ROOT
for w in c5 c2; do for v in "" upold; do
  if [ -n "$v" ]; then export AIDE_HIP_LIB=$GRAFT_REPO_ROOT/abtest/lib_$v.so; else unset AIDE_HIP_LIB; fi
  python bench.py --workload $w --no-cpu-baseline --traffic none --allow-probes 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); s=j['streaming']; print('$w', '${v:-new}', j['value'], {k:(s[k]['ms_per_step'], s[k]['effective_tb_s']) for k in s})"
done; done
