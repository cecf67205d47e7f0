#!/bin/bash
# tools/split_probe.sh -- Huffman streams in parts (plan.h: HufStream::sub): the reference's fixtures and the real-genome
# archive at several sizes, never split (NAFGPU_HUF_SPLIT=0) and as the library decides by itself
cd "${GRAFT_REPO_ROOT:-.}"
export NAFGPU_PROBE_HOOKS=1
echo "== fixtures, never split"; NAFGPU_HUF_SPLIT=0 python3 tools/small_probe.py 2>&1 | grep -v "first call" | cut -c1-250
echo "== fixtures, as the library decides"; python3 tools/small_probe.py 2>&1 | grep -v "first call" | cut -c1-250
for c in ${SPLIT_COPIES:-1 8 30 100 250 565 1200}; do
  echo "== real-genome archive x $c: never split, then as the library decides"
  NAFGPU_HUF_SPLIT=0 python3 tools/real_probe.py $c 2>&1 | grep "^real-genome" | cut -c1-260
  python3 tools/real_probe.py $c 2>&1 | grep "^real-genome\|parts}" | cut -c1-330 | sed 's/\[nafgpu\] section plan: tile 0 of 1, //' | grep -v "0 streams"
done
