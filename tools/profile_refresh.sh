cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
rm -rf gpurun_out/r02_real gpurun_out/r02_l3
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_real -o run -- python3 tools/real_probe.py 3000 > gpurun_out/r02_real.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_l3 -o run -- python3 tools/l3_probe.py 512e6 > gpurun_out/r02_l3.log 2>&1 || exit 1
python3 bench.py > gpurun_out/r02_bench_full.log 2>&1
tail -1 gpurun_out/r02_bench_full.log | cut -c1-200
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -2
