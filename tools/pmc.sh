#!/bin/bash
# tools/pmc.sh TAG "COUNTER COUNTER ..." [ENV=VAL ...] -- one rocprofv3 --pmc pass over a short bench.py run;
# prints the per-launch average of every counter for k_huf_decode.  Counters are collected in their own
# pass (no trace domains besides the kernel trace), as the GPU pool requires.  The bench's other legs are switched off: the pass
# covers the headline kernel only and no child process is started under the profiler.
tag=$1; ctrs=$2; shift 2
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
out=gpurun_out/pmc_$tag
rm -rf "$out"
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$out" -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-verify --real-copies 0 --small-real-copies 0 --fastq-reads 0 --l3-bases 0 --no-iterator --no-masked-leg > gpurun_out/pmc_$tag.log 2>&1
python3 - "$out" "$tag" <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_huf_decode" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(tag, {k: round(sum(v) / len(v)) for k, v in sorted(acc.items())}, "launches", {k: len(v) for k, v in acc.items()})
PY
