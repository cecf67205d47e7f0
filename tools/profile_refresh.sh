#!/bin/bash
# tools/profile_refresh.sh TAG -- re-runs the probe profiles and the full bench of tools/profile_round.sh (not the headline
# kernel trace and PMC passes) after a change that does not touch the headline kernel.
tag=${1:-r02}
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
for pair in "real tools/real_probe.py 3000" "fq tools/fastq_probe.py 10e6" "l3 tools/l3_probe.py 512e6"; do
  set -- $pair; name=$1; shift
  rm -rf gpurun_out/${tag}_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_$name -o run -- python3 "$@" > gpurun_out/${tag}_$name.log 2>&1 || exit 1
done
python3 bench.py > gpurun_out/${tag}_bench_full.log 2>&1
tail -1 gpurun_out/${tag}_bench_full.log | cut -c1-200
