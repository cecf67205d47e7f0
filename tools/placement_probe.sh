#!/bin/bash
# tools/placement_probe.sh -- K1's two placement cases: four decoders one after the other in ONE process,
# twice, with plain hipMalloc and with hipDeviceMallocContiguous for the large buffers.
cd "${GRAFT_REPO_ROOT:-.}"
L=nafcodec_amd/libnafgpu.so
for mode in 0 1 0 1; do
  echo "NAFGPU_ALLOC_CONTIGUOUS=$mode"
  NAFGPU_ALLOC_CONTIGUOUS=$mode NAFGPU_PROBE_LIBS=$L,$L,$L python3 tools/synth_probe.py 40e9 0 2>&1 | grep synthetic | cut -c1-170
done
