cd $GRAFT_REPO_ROOT
for w in c2 c4; do
for r in 1 2 3; do
for v in "" g4_t112 g4_t144 g4_t160; do
  if [ -n "$v" ]; then export AIDE_HIP_LIB=$GRAFT_REPO_ROOT/abtest/lib_$v.so; else unset AIDE_HIP_LIB; fi
  line=$(python bench.py --workload $w --no-cpu-baseline --traffic none --allow-probes --no-kernel-events 2>/dev/null | tail -1)
  echo "$w r$r [${v:-t128}] $(echo "$line" | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])")"
done; done; done
