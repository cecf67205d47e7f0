"""bench.py's multi-rank control flow (one shard per rank, barrier, all-gather of the shard counts,
max-over-ranks timing, one JSON line on rank 0) rehearsed on CPU: two gloo ranks driving the
CPU-harness build of the library.  The numbers are meaningless; the contract fields are checked."""
import json
import os
import socket
import subprocess
import sys

import pytest

import zstd_ref
from conftest import ROOT


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [1, 2])
def test_bench_contract(world):
    csrc = os.path.join(ROOT, "nafcodec_amd", "csrc")
    subprocess.check_call(["make", "-s", "-C", csrc, "emu"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    env = dict(os.environ, NAFGPU_LIB=os.path.join(ROOT, "tests", "emu", "_build", "libnafgpu_emu.so"),
               OMP_NUM_THREADS="1")
    args = ["--gpus", str(world), "--steps", "2", "--warmup", "1", "--bases", "600001", "--cpu-sample-bases", "200000"]
    if world == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "bench.py")] + args
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # exactly one JSON line, from rank 0
    j = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in j, key
    assert ("cpu_baseline" in j) == (world == 1)   # the CPU baseline is reported at N=1 only
    assert (j["n_gpus"], j["steps"], j["warmup"], j["unit"], j["scaling"], j["dtype"]) == (world, 2, 1, "Gbases/s", "weak", "u8")
    assert j["vs_baseline"] is None and j["higher_is_better"] is True and "workload" in j["config"]
    assert set(j["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and j["roofline"]["bound"] == "hbm"
    if world == 1:
        assert set(j["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and j["cpu_baseline"]["kind"] == "port"
    assert "passed" in j["config"]["workload"]  # the full-size checksum check ran and held on every rank
