#!/bin/bash
# usage: ab.sh tag [ENV=VAL ...]  -> gpurun_out/ab_<tag>.log (ms_huf only summary)
tag=$1; shift
env "$@" python bench.py --no-cpu --no-verify --steps 5 --warmup 2 > gpurun_out/ab_$tag.log 2>&1
python - <<PY
import json
for l in open("gpurun_out/ab_$tag.log"):
    if l.startswith("{"):
        d=json.loads(l); print("$tag", "ms_huf", d["path"]["ms_huf"], "ms_step", d["ms_per_step"])
PY
