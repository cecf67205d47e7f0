#!/bin/bash
# tools/placement_probe.sh -- K1's placement cases: four decoders one after the other in ONE process, twice
# with everything from hipMalloc (NAFGPU_ALLOC_PLAIN=1, the library up to round 3) and twice with the large buffers as address
# ranges backed by hipMemCreate chunks (the library now).
cd "${GRAFT_REPO_ROOT:-.}"
L=nafcodec_amd/libnafgpu.so
for mode in 1 0 1 0; do
  echo "NAFGPU_ALLOC_PLAIN=$mode"
  NAFGPU_ALLOC_PLAIN=$mode NAFGPU_PROBE_LIBS=$L,$L,$L python3 tools/synth_probe.py 40e9 0 2>&1 | grep synthetic | cut -c1-170
done
