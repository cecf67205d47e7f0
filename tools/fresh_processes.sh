#!/bin/bash
# tools/fresh_processes.sh [N] -- the headline archive decoded by N fresh processes one after the other (one decoder each,
# best of four decodes): K1's time per process, for the question whether the allocation still decides it.
cd "${GRAFT_REPO_ROOT:-.}"
for i in $(seq 1 "${1:-8}"); do
  python3 tools/synth_probe.py 40e9 0 2>&1 | grep synthetic | cut -c1-170
done
