#!/bin/bash
# tools/ablate.sh MASK [EXTRA_HIPCC_FLAGS] -- builds gpurun_tmp/libnafgpu_x<MASK>.so: the product sources with the timing
# ablation mask of kernels.hip compiled in (results are wrong by construction).  Experiment libraries are loaded by the
# probes through NAFGPU_PROBE_LIB; the product (nafcodec_amd/libnafgpu.so) never has ablations compiled in.
mask=$1; shift
cd "$(dirname "$0")/../nafcodec_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DNAFGPU_ABLATE=$mask "$@" -shared -o ../../gpurun_tmp/libnafgpu_x$mask.so container.cpp zplan.cpp engine.cpp api.cpp synth.cpp kernels.hip -lpthread
