#!/bin/bash
# tools/pj_hops_probe.sh -- look-ups per element and pass (NAFGPU_PJ_HOPS), with and without the first sweep listing what it leaves
cd "${GRAFT_REPO_ROOT:-.}"
export NAFGPU_PROBE_HOOKS=1
for cfg in ${PJ_CFGS:-"1:-" "2:-" "3:-" "4:-" "3:1" "1:-"}; do
  h=${cfg%%:*}; f=${cfg##*:}
  echo "== NAFGPU_PJ_HOPS=$h NAFGPU_PJ_FIRST_LIST=$f"
  if [ "$f" = "-" ]; then unset NAFGPU_PJ_FIRST_LIST; else export NAFGPU_PJ_FIRST_LIST=$f; fi
  NAFGPU_PJ_HOPS=$h python3 tools/l3_probe.py 512e6 3 2>&1 | grep "^level" | cut -c1-200
  NAFGPU_PJ_HOPS=$h python3 tools/fastq_probe.py 10e6 2>&1 | grep "^level" | cut -c1-230
done
