#!/bin/bash
# tools/k2_waves_probe.sh -- the FSE state chains (k_seq_states_lds / _lds16) spread over one, two and three waves per SIMD
cd "${GRAFT_REPO_ROOT:-.}"
export NAFGPU_PROBE_HOOKS=1
for w in 1 2 3 1 2; do
  echo "== NAFGPU_K2_WAVES=$w"
  NAFGPU_K2_WAVES=$w python3 tools/l3_probe.py 512e6 3 2>&1 | grep "^level" | cut -c1-200
  NAFGPU_K2_WAVES=$w python3 tools/fastq_probe.py 10e6 1 2>&1 | grep "^level" | cut -c1-230
done
