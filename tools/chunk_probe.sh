cd "${GRAFT_REPO_ROOT:-.}"
L=nafcodec_amd/libnafgpu.so
for c in 1024 64 2 1024 64 2; do
  echo "NAFGPU_VMM_CHUNK_MIB=$c"
  NAFGPU_VMM_CHUNK_MIB=$c NAFGPU_PROBE_LIBS=$L,$L,$L,$L,$L python3 tools/synth_probe.py 40e9 0 2>&1 | grep synthetic | cut -c60-170
done
