#!/bin/bash
# tools/ab_k1.sh OLD_LIB [N] -- the headline archive decoded by the product library and by OLD_LIB in the same process,
# N fresh processes (K1's time differs from allocation to allocation: several processes, both libraries in each).
cd "${GRAFT_REPO_ROOT:-.}"
for i in $(seq 1 "${2:-3}"); do
  NAFGPU_PROBE_LIBS=$1 python3 tools/synth_probe.py 40e9 0 2>&1 | grep synthetic | cut -c1-200
done
