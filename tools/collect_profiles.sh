#!/bin/bash
# usage (here, after gpurun merged gpurun_out/): tools/collect_profiles.sh <tag>  -> copies the judged summaries into profiles/
tag=${1:-r03}; O=gpurun_out; P=profiles
for w in c2 c3 c4 c5; do
  grep '^{' $O/${tag}_bench_$w.json | tail -1 > $P/${tag}_bench_$w.json
  grep '^{' $O/${tag}_bench_${w}_under_rocprof.json | tail -1 > $P/${tag}_bench_${w}_under_rocprof.json
  f=$(find $O/${tag}_prof_$w -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $P/${tag}_bench_${w}_kernel_stats.csv
done
for w in c2 c5; do cp $O/${tag}_pmc_mfma_$w.md $P/ 2>/dev/null; cp $O/${tag}_hbm_traffic_$w.md $O/${tag}_traffic_$w.json $P/ 2>/dev/null; done
for w in c2 c5; do [ -f $O/${tag}_timeline_$w.txt ] && cp $O/${tag}_timeline_$w.txt $P/; done
grep -v amdgpu.ids $O/${tag}_phase_c2.txt > $P/${tag}_phase_c2.txt
grep -v amdgpu.ids $O/${tag}_host_c2.txt > $P/${tag}_host_c2.txt
[ -f $O/${tag}_gputests.log ] && grep -E "passed|failed" $O/${tag}_gputests.log | tail -1 > $P/${tag}_gputests_summary.txt
ls $P | grep "^${tag}_"
