#!/bin/bash
# tools/pj_xcd_probe.sh -- the sweeps' tiles dealt in launch order (NAFGPU_PJ_XCD=0) and following the XCDs (1)
cd "${GRAFT_REPO_ROOT:-.}"
export NAFGPU_PROBE_HOOKS=1
for e in 0 1 0 1; do
  echo "== NAFGPU_PJ_XCD=$e"
  NAFGPU_PJ_XCD=$e python3 tools/l3_probe.py 512e6 3 2>&1 | grep "^level" | cut -c1-200
  NAFGPU_PJ_XCD=$e python3 tools/fastq_probe.py 10e6 2>&1 | grep "^level" | cut -c1-230
done
