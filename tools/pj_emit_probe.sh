#!/bin/bash
# tools/pj_emit_probe.sh -- the sweeps writing the output themselves against the separate k_pj_emit pass (NAFGPU_PJ_EMIT=1)
cd "${GRAFT_REPO_ROOT:-.}"
export NAFGPU_PROBE_HOOKS=1
for e in 0 1 0 1; do
  echo "== NAFGPU_PJ_EMIT=$e (1: separate pass)"
  NAFGPU_PJ_EMIT=$e python3 tools/l3_probe.py 512e6 3 2>&1 | grep "^level" | cut -c1-200
  NAFGPU_PJ_EMIT=$e python3 tools/fastq_probe.py 10e6 2>&1 | grep "^level" | cut -c1-230
done
