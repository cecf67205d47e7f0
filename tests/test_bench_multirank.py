"""bench.py's multi-rank control flow rehearsed on CPU: two gloo ranks driving the CPU-harness build of the
library write ONE archive together (each its share of the blocks, into /dev/shm), open it with
shard_rank / shard_count, decode their block ranges, exchange {bases, packed bytes, first record, carry} with
one all-gather per step, add up the shards' checksums, take the max-over-ranks time; rank 0 prints one JSON
line.  The numbers are meaningless (and withheld: --rehearsal-lib); the contract fields are checked."""
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
    env = dict(os.environ, OMP_NUM_THREADS="1")
    args = ["--gpus", str(world), "--steps", "2", "--warmup", "1", "--bases", "600001", "--cpu-sample-bases", "200000",
            "--real-copies-per-gpu", "1", "--rehearsal-lib", os.path.join(ROOT, "tests", "emu", "_build", "libnafgpu_emu.so")]
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
    # a rehearsal on the CPU harness never prints a measurement
    assert j["value"] is None and j["roofline"] is None and j["metric"].startswith("REHEARSAL")
    if world == 1:
        assert set(j["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
        assert j["config"]["oracle_checked_bases"] > 0
    else:
        assert "ONE archive" in j["config"]["sharding"]      # configs[4]: block ranges of one archive, not one archive per rank
        assert j["backend"].startswith("gloo") and j["ranks"] == world and len(j["devices"]) == world   # the run describes itself
        if zstd_ref.available():                             # the archive WITH LZ sequences went through the shard protocol on the same ranks
            real = j["path"]["real_genome"]
            assert real["n_gpus"] == world and real["bases"] == 2 * 5488676 and "shard protocol" in real["workload"]
    assert "passed" in j["config"]["workload"]  # the full-size checksum check ran and held on every rank


def test_headline_line_survives_a_sharded_leg_that_does_not_finish():
    """The second workload of the multi-rank run has a time limit of its own: when it runs out every rank leaves, and rank 0
    still prints the one line -- with the headline in it and the failure named where the leg's numbers would be (which step
    of the protocol, which rank) -- but the run's status is NOT zero: a stall on N GPUs must not look like a finished run."""
    if not zstd_ref.available():
        pytest.skip("libzstd not loadable")
    csrc = os.path.join(ROOT, "nafcodec_amd", "csrc")
    subprocess.check_call(["make", "-s", "-C", csrc, "emu"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--bases", "300001", "--real-copies-per-gpu", "1", "--sharded-leg-limit", "0",
           "--rehearsal-lib", os.path.join(ROOT, "tests", "emu", "_build", "libnafgpu_emu.so")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert p.returncode != 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and "passed" in j["config"]["workload"]
    assert "not finished" in j["path"]["real_genome"]["error"] and "rank 0 at step" in j["path"]["real_genome"]["where"]
    assert "sharded real-genome leg" in p.stderr


def test_real_genome_leg_runs_and_checks_itself():
    """bench.py's second workload (path.real_genome): the reference's NZ_AAEN01000029 fixture tiled and recompressed by
    libzstd, decoded by the library it is given (here the CPU harness), compared with the tiled fixture."""
    import zstd_ref
    if not zstd_ref.available():
        pytest.skip("libzstd not loadable")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "nafcodec_amd", "csrc"), "emu"],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    sys.path.insert(0, ROOT)
    import bench
    from nafcodec_amd import _ffi
    leg = bench.real_genome_leg(_ffi.Library(os.path.join(ROOT, "tests", "emu", "_build", "libnafgpu_emu.so")), 0, 1)
    assert leg["bases"] == 5488676 and "output checksum equals the tiled fixture" in leg["workload"]
    assert set(leg["roofline"]) >= {"bound", "achieved", "peak", "frac", "algorithmic_bytes_per_step"}


@pytest.mark.gpu
def test_three_ranks_share_the_one_gpu():
    """The sharded bench flow with the REAL library: three processes (gloo for the exchange) write one 6-Gbase archive in parts,
    each opens it with its shard_rank, decodes its block range on GPU 0, and the summed shard checksums must equal the
    writers'.  (RCCL itself needs one GPU per rank: the driver's multi-GPU run is where that is exercised.)"""
    world = 3
    env = dict(os.environ, OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
           "--bases", "2e9", "--no-cpu", "--rehearsal-one-gpu", "--real-copies-per-gpu", "200"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["metric"].startswith("REHEARSAL") and j["value"] is None and j["n_gpus"] == world
    assert "ONE archive" in j["config"]["sharding"] and "passed" in j["config"]["workload"]
    # ... and the real-genome archive (a few LZ sequences per block) through the shard protocol on the same three ranks
    real = j["path"]["real_genome"]
    assert real["n_gpus"] == world and real["bases"] == 600 * 5488676 and "shard protocol" in real["workload"]


RCCL_PROBE = r"""
import os, sys, ctypes
sys.path.insert(0, %r)
import torch, torch.distributed as dist                      # (before libnafgpu.so: see bench.py)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)        # "nccl" is RCCL on ROCm
from nafcodec_amd import _ffi
from nafcodec_amd.sharding import gather_placement
lib = _ffi.default()
arc = lib.synth(50_000_000, seed=3)
opts = _ffi.Opts(); lib.c.nafgpu_opts_default(ctypes.byref(opts)); opts.device = 0
h, err = ctypes.c_void_p(), _ffi.Error()
assert lib.c.nafgpu_open_bytes(ctypes.cast(arc.bytes, ctypes.c_char_p), arc.n, ctypes.byref(opts), ctypes.byref(h), ctypes.byref(err)) == 0
assert lib.c.nafgpu_upload(h) == 0
res = _ffi.DeviceResult()
scratch = (torch.zeros(4, dtype=torch.int64, device="cuda"), torch.zeros(4, dtype=torch.int64, device="cuda"))
for _ in range(3):
    assert lib.c.nafgpu_decode_all_device(h, ctypes.byref(res)) == 0
    pl = gather_placement(dist, torch, res.n_bases, res.packed_bytes, res.first_record, res.carry, res.n_records, "cuda", scratch)
dist.barrier(); torch.cuda.synchronize()
t = torch.tensor([1.5], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
out = ctypes.c_uint64(); lib.c.nafgpu_hash64_device(h, res.d_sequence, res.n_bases, ctypes.byref(out))
assert pl.total_bases == 50_000_000 and out.value == arc.seq_hash and float(t.item()) == 1.5
print("rccl probe ok", flush=True)
dist.destroy_process_group()
"""


@pytest.mark.gpu
def test_rccl_collectives_and_the_library_share_a_process():
    """What a rank of `bench.py --gpus N` does, with a world of one: RCCL's all-gather / all-reduce / barrier on device tensors
    around decodes of the HIP library in the same process (PyTorch's HIP runtime and the library's must be one)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    p = subprocess.run([sys.executable, "-c", RCCL_PROBE % ROOT], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0 and "rccl probe ok" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]
