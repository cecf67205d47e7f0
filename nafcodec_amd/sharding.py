"""Multi-GPU placement of block-range shards.

Each rank decodes one contiguous range of the sequence section's zstd blocks into its own HBM
(no data-path collective: blocks are independent once the host walk has resolved their tables).
The only exchange is ONE all-gather of a 32-byte struct per rank

    {bases, packed_bytes, first_record, carry}

with the meaning include/nafgpu.h gives the fields of nafgpu_device_result:
  bases         nucleotides (or text bytes) this shard holds
  packed_bytes  decoded 4-bit bytes of the shard (0 for protein / text archives)
  first_record  global index of the first record that STARTS inside the shard
  carry         1 if the shard begins inside a record: its first bases are the tail of record
                first_record - 1, which the previous rank owns
after which every rank knows the global base offset of every shard and which records each rank
owns (rank r owns records [first_record[r], first_record[r+1])).  On MI355X the collective is
RCCL over xGMI (`backend="nccl"`); the CPU tests run the same code over gloo.
"""
from dataclasses import dataclass

FIELDS = 4  # bases, packed_bytes, first_record, carry


@dataclass
class ShardPlacement:
    rank: int
    world: int
    base_offset: int      # global index of this shard's first base (= sum of the bases of the ranks before)
    packed_offset: int    # global index of its first packed byte
    first_record: int     # first record that starts in this shard
    n_owned_records: int  # records that start in this shard
    total_bases: int
    total_packed: int
    total_records: int
    carries: list         # per rank: 1 if that shard begins inside a record (nafgpu_device_result.carry)
    first_records: list   # per rank


def gather_placement(dist, torch, bases, packed_bytes, first_record, carry, total_records, device, scratch=None):
    """One all_gather_into_tensor of 4 x int64 per rank -> ShardPlacement for this rank."""
    rank, world = dist.get_rank(), dist.get_world_size()
    if scratch is None:
        scratch = (torch.zeros(FIELDS, dtype=torch.int64, device=device),
                   torch.zeros(FIELDS * world, dtype=torch.int64, device=device))
    mine, everyone = scratch
    mine.copy_(torch.tensor([bases, packed_bytes, first_record, carry], dtype=torch.int64))
    dist.all_gather_into_tensor(everyone, mine)
    g = everyone.view(world, FIELDS).cpu()
    firsts = [int(x) for x in g[:, 2]]
    nxt = firsts[rank + 1] if rank + 1 < world else int(total_records)
    return ShardPlacement(rank=rank, world=world,
                          base_offset=int(g[:rank, 0].sum()), packed_offset=int(g[:rank, 1].sum()),
                          first_record=firsts[rank], n_owned_records=max(0, nxt - firsts[rank]),
                          total_bases=int(g[:, 0].sum()), total_packed=int(g[:, 1].sum()),
                          total_records=int(total_records),
                          carries=[int(x) for x in g[:, 3]], first_records=firsts)
