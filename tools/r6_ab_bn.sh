cd $GRAFT_REPO_ROOT
for w in c5 c2 c4; do
for r in 1 2; do
for v in "" bn_2pass bn_lim8m bn_lim17m bn_lim34m; do
  if [ -n "$v" ]; then export AIDE_HIP_LIB=$GRAFT_REPO_ROOT/abtest/lib_$v.so; else unset AIDE_HIP_LIB; fi
  line=$(python bench.py --workload $w --no-cpu-baseline --traffic none --allow-probes 2>/dev/null | tail -1)
  echo "$w r$r [${v:-onepass}] $(echo "$line" | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])")"
done; done; done
