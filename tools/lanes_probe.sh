#!/bin/bash
# tools/lanes_probe.sh -- K1 task width (streams per wave) on archives with fewer streams than the chip has lanes: the
# real-genome archive at 3.1 Gbases (47 k streams) and at 0.1 Gbases, NAFGPU_TASK_LANES = 64 / 48 / 32 / 16.
cd "${GRAFT_REPO_ROOT:-.}"
for copies in 565 20; do
  for lanes in 64 48 32 16; do
    echo "copies $copies lanes $lanes"
    NAFGPU_TASK_LANES=$lanes python3 tools/real_probe.py $copies 2>&1 | grep "real-genome" | cut -c1-200
  done
done
