"""The N>1 exchange step (nafcodec_amd/sharding.py) on CPU: two gloo ranks, no GPU."""
import os
import socket
import subprocess
import sys

from conftest import ROOT

WORKER = r"""
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from nafcodec_amd.sharding import gather_placement
dist.init_process_group("gloo")
r, w = dist.get_rank(), dist.get_world_size()
# per rank: bases, packed bytes, first record that starts in the shard, carry (shard begins inside a record)
table = [(1000003, 500002, 0, 0), (2000000, 1000000, 17, 1), (7, 4, 22, 0)][:w]
total_records = 23
for _ in range(3):
    p = gather_placement(dist, torch, *table[r], total_records, device="cpu")
assert p.base_offset == sum(t[0] for t in table[:r]) and p.packed_offset == sum(t[1] for t in table[:r])
assert p.first_record == table[r][2] and p.first_records == [t[2] for t in table]
assert p.n_owned_records == ((table[r + 1][2] if r + 1 < w else total_records) - table[r][2])
assert (p.total_bases, p.total_packed, p.total_records) == (sum(t[0] for t in table), sum(t[1] for t in table), total_records)
assert p.carries == [t[3] for t in table] and (p.rank, p.world) == (r, w)
dist.barrier()
dist.destroy_process_group()
print("rank", r, "ok")
"""


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_offset_gather_two_and_three_ranks(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    for n in (2, 3):
        p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
                            "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script)],
                           capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, OMP_NUM_THREADS="1"))
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
        assert p.stdout.count("ok") == n
