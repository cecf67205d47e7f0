"""The N>1 exchange steps (nafcodec_amd/sharding.py) on CPU: gloo ranks (2, 3 and 8), no GPU."""
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


PROTOCOL_WORKER = r"""
import io, os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import torch, torch.distributed as dist
import cases
from nafcodec_amd import _ffi
from nafcodec_amd.decoder import Decoder
from nafcodec_amd.sharding import decode_sharded
dist.init_process_group("gloo")
r, w = dist.get_rank(), dist.get_world_size()
emu = _ffi.Library(os.path.join(%r, "tests", "emu", "_build", "libnafgpu_emu.so"))
for name, blob, want_seq, want_qual, lens in cases.lz_shard_archives(1):
    if name not in os.environ.get("NAFGPU_TEST_ARCHIVES", "real_genome_l1,fastq_like_l1").split(","):
        continue
    dec = Decoder(io.BytesIO(blob), shard_rank=r, shard_count=w, shard_protocol=True, _lib=emu)
    for _ in range(2):                                     # (a second decode: the protocol starts over)
        res = decode_sharded(dec, dist, torch, "cpu")
    assert res.sharded == 1
    got = dec.copy_to_host(res.d_sequence, res.n_bases)
    assert got == want_seq[res.base_offset:res.base_offset + res.n_bases], (name, r, "sequence")
    if want_qual is not None:
        assert dec.copy_to_host(res.d_quality, res.n_quality) == want_qual[res.quality_offset:res.quality_offset + res.n_quality], (name, r)
    # the shards tile the sections: sizes added up over the ranks
    t = torch.tensor([res.n_bases, res.n_quality], dtype=torch.int64)
    dist.all_reduce(t)
    assert int(t[0]) == len(want_seq) and int(t[1]) == (len(want_qual) if want_qual is not None else 0), name
    # ... in rank order
    offs = torch.zeros(w, dtype=torch.int64)
    offs[r] = res.base_offset
    dist.all_reduce(offs)
    assert list(offs) == sorted(offs) and int(offs[0]) == 0
    dec.close()
    if name == "real_genome_l1" and w <= 3:
        # damage inside ONE rank's range (a stretch of a Huffman stream zeroed): that rank finds out while it decodes, the
        # others learn of it through the protocol -- nobody waits for a window that never comes, every rank raises
        bad = bytearray(blob)
        at = len(bad) * 6 // 10
        bad[at:at + 200] = bytes(200)
        dec = Decoder(io.BytesIO(bytes(bad)), shard_rank=r, shard_count=w, shard_protocol=True, _lib=emu)
        try:
            decode_sharded(dec, dist, torch, "cpu")
            raise SystemExit("rank %%d: the damaged archive decoded without an error" %% r)
        except OSError:
            pass
        dec.close()
dist.barrier()
dist.destroy_process_group()
print("rank", r, "ok")
"""


def test_shard_protocol_over_gloo_two_and_three_ranks(tmp_path):
    """The shard protocol's exchange (one all-gather of 64 bytes, then the windows point to point) between real
    processes: gloo ranks driving the CPU harness decode archives WITH LZ sequences -- the statistics of a real genome,
    FASTQ-like reads (Sequence and Quality both swept) -- and every rank's share must be its part of the whole."""
    import zstd_ref
    import pytest
    if not zstd_ref.available():
        pytest.skip("libzstd not loadable (the archives are written with it)")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "nafcodec_amd", "csrc"), "emu"],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    script = tmp_path / "protocol_worker.py"
    script.write_text(PROTOCOL_WORKER % (ROOT, ROOT, ROOT))
    for n in (2, 3, 8):
        # (eight ranks: the world of configs[4]; the FASTQ-like archive only -- both of its sections are swept, the ranks in
        # the middle both receive and send a window, posted as one group)
        env = dict(os.environ, OMP_NUM_THREADS="1")
        if n == 8:
            env["NAFGPU_TEST_ARCHIVES"] = "fastq_like_l1"
        p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
                            "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script)],
                           capture_output=True, text=True, timeout=900, env=env)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
        assert p.stdout.count("ok") == n
