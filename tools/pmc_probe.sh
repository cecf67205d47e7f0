#!/bin/bash
# tools/pmc_probe.sh TAG "COUNTER COUNTER ..." KERNEL_SUBSTRING SCRIPT [ARGS...] -- one rocprofv3 --pmc pass (counters in
# their own pass: kernel trace only, as the GPU pool requires) over a python script; prints the per-launch average of
# every counter for the kernels whose name contains KERNEL_SUBSTRING.
tag=$1; ctrs=$2; kern=$3; shift 3
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
out=gpurun_out/pmc_$tag
rm -rf "$out"
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$out" -o run -- python3 "$@" > gpurun_out/pmc_$tag.log 2>&1 || { tail -5 gpurun_out/pmc_$tag.log; exit 1; }
python3 - "$out" "$tag" "$kern" <<'PY'
import csv, glob, sys, collections, json
out, tag, kern = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            name = r["Kernel_Name"].replace("void nafgpu::(anonymous namespace)::", "").replace("nafgpu::(anonymous namespace)::", "")
            acc[name.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: {"avg": round(sum(v) / len(v)), "max": round(max(v)), "launches": len(v)} for c, v in sorted(cs.items())} for k, cs in acc.items()}
print(tag, json.dumps(res))
json.dump(res, open("gpurun_out/pmc_%s.json" % tag, "w"), indent=1)
PY
