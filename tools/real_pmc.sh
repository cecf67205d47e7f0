#!/bin/bash
# tools/real_pmc.sh TAG -- the rocprofv3 --pmc passes (kernel trace only, one counter group per pass) over the real-genome
# workload of bench.py's path.real_genome (tools/real_probe.py 3000), per-launch averages of the k_huf_decode classes;
# tools/real_pmc_collect.py TAG writes profiles/TAG_real_pmc_summary.json from them.
tag=${1:-rXX}
cd "${GRAFT_REPO_ROOT:-.}"
bash tools/pmc_probe.sh ${tag}_real_sq "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" k_huf_decode tools/real_probe.py 3000 || exit 1
bash tools/pmc_probe.sh ${tag}_real_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" k_huf_decode tools/real_probe.py 3000 || exit 1
bash tools/pmc_probe.sh ${tag}_real_fetch "FETCH_SIZE" k_huf_decode tools/real_probe.py 3000 || exit 1
bash tools/pmc_probe.sh ${tag}_real_write "WRITE_SIZE" k_huf_decode tools/real_probe.py 3000 || exit 1
