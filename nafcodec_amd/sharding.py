"""Multi-GPU placement of block-range shards.

Each rank decodes one contiguous range of the sequence section's zstd blocks into its own HBM
(no data-path collective: Huffman-literal blocks are independent).  The only exchange is ONE
all-gather of a 32-byte struct per rank -- {bases, packed_bytes, records, carry} -- after which
every rank knows the global base offset and the global index of its first record.  On MI355X
the collective is RCCL over xGMI (`backend="nccl"`); the CPU tests run the same code over gloo.
"""
from dataclasses import dataclass

FIELDS = 4  # bases, packed_bytes, records, carry


@dataclass
class ShardPlacement:
    rank: int
    world: int
    base_offset: int      # global index of this shard's first base
    packed_offset: int    # global index of its first packed byte
    record_offset: int    # global index of its first record
    total_bases: int
    total_packed: int
    total_records: int
    carries: list         # per rank: 1 if the shard ends on an odd nibble (next shard starts mid-byte)


def gather_placement(dist, torch, bases, packed_bytes, records, carry, device, scratch=None):
    """One all_gather_into_tensor of 4 x int64 per rank -> ShardPlacement for this rank."""
    rank, world = dist.get_rank(), dist.get_world_size()
    if scratch is None:
        scratch = (torch.zeros(FIELDS, dtype=torch.int64, device=device),
                   torch.zeros(FIELDS * world, dtype=torch.int64, device=device))
    mine, everyone = scratch
    mine.copy_(torch.tensor([bases, packed_bytes, records, carry], dtype=torch.int64))
    dist.all_gather_into_tensor(everyone, mine)
    g = everyone.view(world, FIELDS).cpu()
    return ShardPlacement(rank=rank, world=world,
                          base_offset=int(g[:rank, 0].sum()), packed_offset=int(g[:rank, 1].sum()),
                          record_offset=int(g[:rank, 2].sum()), total_bases=int(g[:, 0].sum()),
                          total_packed=int(g[:, 1].sum()), total_records=int(g[:, 2].sum()),
                          carries=[int(x) for x in g[:, 3]])
