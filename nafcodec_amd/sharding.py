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


# ---------------------------------------------------------------------------------------------------------------------
# The shard protocol: sections WITH LZ sequences over several GPUs (include/nafgpu.h: nafgpu_shard_*).
#
# A block range of such a section needs two things from the ranges in front of it: the repeat offsets they leave behind
# and the last window of their output.  Both arrive without a rank waiting for more than it must:
#   1. every rank entropy-decodes its range (Huffman literals, FSE sequences) and puts together its block sizes and its
#      repeat-offset map;
#   2. ONE all-gather of 64 bytes per rank (RCCL over xGMI): decoded sizes and maps -- every rank now knows where its
#      range begins, which repeat offsets it inherits and how large the window in front of it is (and, behind step 3, a
#      1-element all-reduce of an error flag, so that a rank that failed leaves nobody waiting for its window);
#   3. every rank places its literals and resolves every match that does not reach -- directly or through other
#      matches -- into that window;
#   4. the windows travel down the line, point to point (ncclSend / ncclRecv, at most window_size bytes each): a rank
#      whose last window is final already (nothing pending touches it: the usual case) posts its send and its receive as
#      one group, the others receive, finish, send; the serial part is the handful of matches that waited.
SUMMARY_BYTES = 64

# Where the protocol stands in this process: {"step": ..., "section": ..., "rank": ...}.  A caller that puts a time limit around
# decode_sharded (bench.py) names the step that did not return.
PROGRESS = {"step": None, "section": None, "rank": None}


def _at(step, section=None):
    PROGRESS["step"], PROGRESS["section"] = step, section


def decode_sharded(dec, dist, torch, device):
    """One sharded decode of `dec` (a Decoder opened with shard_rank / shard_count / shard_protocol=True) over the
    process group: returns the nafgpu_device_result of this rank's share.  `device`: "cuda" (RCCL) or "cpu" (gloo).
    A rank that fails (a corrupt block in its range, say) still takes part in every exchange -- nobody is left waiting in
    a receive -- and every rank raises: the ranks agree on one flag (a 1-element all-reduce) after the placement step, and
    skip the window exchange together when any of them has failed by then, and on one more at the end.
    The windows of a section travel as ONE group of point-to-point operations per rank (`batch_isend_irecv`: on RCCL a
    single grouped launch): a rank whose tail is final posts its send and its receive together; a rank whose tail waits
    for the window in front posts the receive, finishes, and then posts the send."""
    rank, world = dist.get_rank(), dist.get_world_size()
    PROGRESS["rank"] = rank
    error = None
    _at("shard_begin")
    try:
        summary = dec.shard_begin()
    except Exception as e:                                 # noqa: BLE001 -- carried past the collectives, raised below
        error = e
        summary = bytes(56) + b"\x01\x01" + bytes(6)        # nafgpu_shard_summary with failed[0] = failed[1] = 1
    mine = torch.frombuffer(bytearray(summary), dtype=torch.uint8).to(device)
    everyone = torch.empty(SUMMARY_BYTES * world, dtype=torch.uint8, device=device)
    _at("all_gather(summaries)")
    dist.all_gather_into_tensor(everyone, mine)
    halo = [(0, 0, True), (0, 0, True)]
    if error is None:
        try:
            _at("shard_place")
            dec.shard_place(everyone.cpu().numpy().tobytes())
            _at("shard_halo")
            halo = [dec.shard_halo(0), dec.shard_halo(1)]  # (synchronises: what the placement step found wrong is known here)
        except Exception as e:                             # noqa: BLE001
            error = e
    flag = torch.tensor([0 if error is None else 1], dtype=torch.int32, device=device)
    _at("all_reduce(error flag after placement)")
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if int(flag.item()) != 0:
        _at("failed")
        if error is not None:
            raise error
        raise OSError(0, "zstd: another rank could not decode its part of the archive")

    def exchange(ops, what, section):                      # one group: every operation of the list is in flight at once
        _at(what, section)
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        if device != "cpu":
            torch.cuda.current_stream().synchronize()

    for section in (0, 1):
        recv_n, send_n, ready = halo[section]
        sends = bool(send_n) and rank + 1 < world
        recvs = bool(recv_n) and rank > 0
        sbuf = torch.empty(send_n, dtype=torch.uint8, device=device) if sends else None
        rbuf = torch.empty(recv_n, dtype=torch.uint8, device=device) if recvs else None
        ops = []
        if sends and ready:                                 # the tail is final: it leaves together with the receive
            _at("shard_export_tail", section)
            dec.shard_export_tail(section, sbuf.data_ptr(), send_n)
            ops.append(dist.P2POp(dist.isend, sbuf, rank + 1))
        if recvs:
            ops.append(dist.P2POp(dist.irecv, rbuf, rank - 1))
        if ops:
            exchange(ops, "window exchange (%s)" % ("send + recv" if len(ops) == 2 else "send" if sends and ready else "recv"), section)
        if recvs:
            try:
                _at("shard_import_halo", section)
                dec.shard_import_halo(section, rbuf.data_ptr(), recv_n)
            except Exception as e:                         # noqa: BLE001 -- the ranks behind still get their window (of a failed decode)
                error = error or e
        if sends and not ready:
            try:
                _at("shard_export_tail", section)
                dec.shard_export_tail(section, sbuf.data_ptr(), send_n)
            except Exception as e:                         # noqa: BLE001
                error = error or e
            exchange([dist.P2POp(dist.isend, sbuf, rank + 1)], "window exchange (send after finishing)", section)
    res = None
    if error is None:
        try:
            _at("shard_finish")
            res = dec.shard_finish()                       # (synchronises: a corrupt block in this rank's range is known here at the latest)
        except Exception as e:                             # noqa: BLE001
            error = e
    flag = torch.tensor([0 if error is None else 1], dtype=torch.int32, device=device)
    _at("all_reduce(error flag at the end)")
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)            # one archive, one verdict: every rank raises when any of them failed
    _at("done" if error is None and int(flag.item()) == 0 else "failed")
    if error is not None:
        raise error
    if int(flag.item()) != 0:
        raise OSError(0, "zstd: another rank could not decode its part of the archive")
    return res


def decode_sharded_local(decoders):
    """The same protocol with every rank in THIS process (tests, and several shards on one GPU): `decoders` are the
    ranks' Decoders in rank order; windows are handed over through host buffers.  Returns their results."""
    import ctypes
    world = len(decoders)
    everyone = b"".join(d.shard_begin() for d in decoders)
    for d in decoders:
        d.shard_place(everyone)
    for section in (0, 1):
        info = [d.shard_halo(section) for d in decoders]
        tails = {}

        def export(r):
            tails[r] = ctypes.create_string_buffer(info[r][1])
            decoders[r].shard_export_tail(section, ctypes.cast(tails[r], ctypes.c_void_p), info[r][1])

        for r in range(world - 1):                          # tails that are final leave before anything arrives, as over RCCL
            if info[r][1] and info[r][2]:
                export(r)
        for r, d in enumerate(decoders):
            recv_n, send_n, _ready = info[r]
            if recv_n and r > 0:
                assert len(tails[r - 1]) == recv_n, (r, recv_n, len(tails[r - 1]))
                d.shard_import_halo(section, ctypes.cast(tails[r - 1], ctypes.c_void_p), recv_n)
            if send_n and r + 1 < world and r not in tails:
                export(r)
    return [d.shard_finish() for d in decoders]
