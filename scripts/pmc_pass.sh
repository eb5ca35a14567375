#!/bin/bash
# usage: scripts/pmc_pass.sh <tag> <counters...> -- <command...>   (one rocprofv3 --pmc pass, csv into gpurun_out/pmc_<tag>)
tag=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -o p -- "$@" > gpurun_out/pmc_$tag.log 2>&1
python scripts/pmc_summary.py gpurun_out/pmc_$tag/p_counter_collection.csv
