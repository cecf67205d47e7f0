#!/bin/bash
# tools/small_api_trace.sh [FIXTURE] -- where a small archive's open-to-close cycle goes: nafcodec_amd/iter_bench in repeat mode (40 cycles)
# under rocprofv3 --hip-trace --kernel-trace --stats; the HIP API calls and kernels by total time, per cycle.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
fx=${1:-NZ_AAEN01000029.naf}
rm -rf gpurun_out/small_api
nafcodec_amd/iter_bench tests/golden/$fx 0 40
rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d gpurun_out/small_api -o run -- nafcodec_amd/iter_bench tests/golden/$fx 0 40 > gpurun_out/small_api.log 2>&1 || { tail -5 gpurun_out/small_api.log; exit 1; }
tail -1 gpurun_out/small_api.log
python3 - <<'PY'
import csv
for f, what in (("gpurun_out/small_api/run_hip_api_stats.csv", "HIP API"), ("gpurun_out/small_api/run_kernel_stats.csv", "kernels")):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -int(r["TotalDurationNs"]))
    print("--", what, "(per cycle of 40: calls, us)")
    for r in rows[:16]:
        print("  %-44s %6.1f calls %8.1f us" % (r["Name"][:44], int(r["Calls"]) / 40, int(r["TotalDurationNs"]) / 40e3))
PY
