#!/bin/bash
# tools/kstats.sh TAG SCRIPT [ARGS...] -- rocprofv3 --kernel-trace --stats of a python script on the GPU box; prints the
# kernels by total time (name, calls, average us, percent) and leaves the CSVs under gpurun_out/TAG.
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
rm -rf gpurun_out/$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -o run -- python3 "$@" > gpurun_out/$tag.log 2>&1 || { tail -5 gpurun_out/$tag.log; exit 1; }
grep -v "^\[\|^E2\|^W2\|^I2" gpurun_out/$tag.log | tail -2 | cut -c1-300
python3 - "$tag" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/%s/**/*kernel_stats.csv" % sys.argv[1], recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:18]:
    n = r["Name"].replace("void nafgpu::(anonymous namespace)::", "").replace("nafgpu::(anonymous namespace)::", "")
    print("%-46s calls %5s avg %10.1f us  %5.1f %%" % (n.split("(")[0][:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
