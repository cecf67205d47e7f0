#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
rm -rf gpurun_out/prof_v4
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_v4 -o run -- python3 bench.py --steps 5 --warmup 2 --no-cpu > gpurun_out/prof_v4.log 2>&1 || exit 1
tools/pmc.sh v4_fetch "FETCH_SIZE" || exit 1
tools/pmc.sh v4_write "WRITE_SIZE" || exit 1
tools/pmc.sh v4_sq "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" || exit 1
tools/pmc.sh v4_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" || exit 1
python3 bench.py > gpurun_out/bench_v4_full.log 2>&1
tail -1 gpurun_out/bench_v4_full.log
