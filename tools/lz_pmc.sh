#!/bin/bash
# tools/lz_pmc.sh TAG [LEVELS] -- the LZ-dense probes (level-3 DNA, FASTQ-like) under rocprofv3: one kernel-trace --stats
# pass and separate FETCH_SIZE / WRITE_SIZE passes each (counters never combined with other trace domains).  Leaves
# gpurun_out/TAG_{l3,fq}_{stats,fetch,write}*; tools/lz_pmc_collect.py TAG folds the passes into gpurun_out/TAG_lz_pmc.json.
tag=${1:-rXX}
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
run() {   # run NAME "rocprof flags" SCRIPT ARGS...
  name=$1; flags=$2; shift 2
  rm -rf gpurun_out/${tag}_$name
  rocprofv3 $flags --kernel-trace --output-format csv -d gpurun_out/${tag}_$name -o run -- python3 "$@" > gpurun_out/${tag}_$name.log 2>&1 || { tail -5 gpurun_out/${tag}_$name.log; exit 1; }
  grep "^level" gpurun_out/${tag}_$name.log | cut -c1-300
}
export NAFGPU_PROBE_ONE_DECODE=1
run l3_stats "--stats" tools/l3_probe.py 512e6 3 || exit 1
run l3_fetch "--pmc FETCH_SIZE" tools/l3_probe.py 512e6 3 || exit 1
run l3_write "--pmc WRITE_SIZE" tools/l3_probe.py 512e6 3 || exit 1
run fq_stats "--stats" tools/fastq_probe.py 10e6 1 || exit 1
run fq_fetch "--pmc FETCH_SIZE" tools/fastq_probe.py 10e6 1 || exit 1
run fq_write "--pmc WRITE_SIZE" tools/fastq_probe.py 10e6 1 || exit 1
python3 tools/lz_pmc_collect.py $tag
