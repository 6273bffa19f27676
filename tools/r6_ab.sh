# usage (GPU box): tools/r6_ab.sh "<workloads>" <rounds> <probe> [<probe> ...]   ("" = the shipped library)
cd $GRAFT_REPO_ROOT
ws=$1; rounds=$2; shift 2
for w in $ws; do for r in $(seq 1 $rounds); do for v in "" "$@"; do
  if [ -n "$v" ]; then export AIDE_HIP_LIB=$GRAFT_REPO_ROOT/abtest/lib_$v.so; else unset AIDE_HIP_LIB; fi
  line=$(python bench.py --workload $w --no-cpu-baseline --traffic none --allow-probes --no-kernel-events 2>/dev/null | tail -1)
  echo "$w r$r [${v:-shipped}] $(echo "$line" | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])")"
done; done; done
