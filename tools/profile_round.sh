#!/bin/bash
# tools/profile_round.sh TAG -- the round's evidence in one go (run on the GPU box): rocprofv3 --kernel-trace --stats of the
# headline bench command, separate --pmc passes for its dominant kernel (HBM bytes, SQ, LDS; never combined with trace
# domains other than the kernel trace), the masked configuration, and the probes for archives as found in the wild.
# Leaves everything under gpurun_out/TAG_*; tools/profile_collect.py TAG then writes the summaries into profiles/.
tag=${1:-rXX}
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
stats() {   # stats NAME SCRIPT ARGS...
  name=$1; shift
  rm -rf gpurun_out/${tag}_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_$name -o run -- python3 "$@" > gpurun_out/${tag}_$name.log 2>&1 || { tail -5 gpurun_out/${tag}_$name.log; exit 1; }
  grep -v "^\[\|^E2\|^W2\|^I2" gpurun_out/${tag}_$name.log | tail -2 | cut -c1-400
}
pmc() {     # pmc NAME "COUNTERS"
  name=$1; ctrs=$2
  rm -rf gpurun_out/${tag}_pmc_$name
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc_$name -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-verify --real-copies 0 --small-real-copies 0 --fastq-reads 0 --l3-bases 0 --no-iterator --no-masked-leg > gpurun_out/${tag}_pmc_$name.log 2>&1 || { tail -5 gpurun_out/${tag}_pmc_$name.log; exit 1; }
}
stats headline bench.py --steps 5 --warmup 2 --no-cpu --real-copies 0 --small-real-copies 0 --fastq-reads 0 --l3-bases 0 --no-iterator --no-masked-leg || exit 1
pmc fetch "FETCH_SIZE" || exit 1
pmc write "WRITE_SIZE" || exit 1
pmc sq "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" || exit 1
pmc lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" || exit 1
stats mask bench.py --mask --steps 5 --warmup 2 --no-cpu --real-copies 0 --small-real-copies 0 --fastq-reads 0 --l3-bases 0 --no-iterator --no-masked-leg || exit 1
stats real tools/real_probe.py 3000 || exit 1
stats fq tools/fastq_probe.py 10e6 || exit 1
stats l3 tools/l3_probe.py 512e6 || exit 1
python3 bench.py > gpurun_out/${tag}_bench_full.log 2>&1
tail -1 gpurun_out/${tag}_bench_full.log | cut -c1-600
