#!/bin/bash
# tools/k2_alone_probe.sh -- the FASTQ-like probe under rocprofv3 --kernel-trace, as built and with K2 of the Quality section in its
# own place (NAFGPU_NO_K2_AHEAD: not beside the Sequence section's K1), and with its own section's literal-buffer K1 classes after it
# as well (NAFGPU_NO_EARLY_K1): how long k_seq_states_lds16 takes with the chip to itself.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
export NAFGPU_PROBE_HOOKS=1
for mode in ahead alone pure; do
  rm -rf gpurun_out/k2_$mode
  if [ $mode = alone ]; then export NAFGPU_NO_K2_AHEAD=1; fi
  if [ $mode = pure ]; then export NAFGPU_NO_K2_AHEAD=1 NAFGPU_NO_EARLY_K1=1; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/k2_$mode -o run -- python3 tools/fastq_probe.py 10e6 1 > gpurun_out/k2_$mode.log 2>&1 || { tail -5 gpurun_out/k2_$mode.log; exit 1; }
  echo "== $mode"; grep "^level" gpurun_out/k2_$mode.log | cut -c1-200
  python3 - gpurun_out/k2_$mode/run_kernel_trace.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if "k_seq_states_lds16" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[max(0, last - 6):last + 12]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    if e - s > 0.1:
        print("%8.2f %8.2f %7.2f  q%s wg %6d  %s" % (s, e, e - s, r["Queue_Id"], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), r["Kernel_Name"].replace("nafgpu::(anonymous namespace)::", "")[:44]))
PY
done
