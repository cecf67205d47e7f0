#!/bin/bash
# tools/prof.sh TAG SCRIPT [ARGS...] -- rocprofv3 --kernel-trace --stats over one python script (a probe or bench.py);
# leaves gpurun_out/prof_TAG/ (csv) and gpurun_out/prof_TAG.log, prints the top kernels.
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
rm -rf gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o run -- python3 "$@" > gpurun_out/prof_$tag.log 2>&1 || { tail -5 gpurun_out/prof_$tag.log; exit 1; }
grep -v "^\[" gpurun_out/prof_$tag.log | tail -3
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -14 "$f" | cut -c1-170
