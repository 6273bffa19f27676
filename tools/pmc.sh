#!/bin/bash
# usage: tools/pmc.sh <tag> <counters...> -- <command...>   (run on the GPU box)
tag=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -o $tag -- "$@" > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
