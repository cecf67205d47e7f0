"""Test helper: writes NAF archives (any section mix, any zstd level) with the system libzstd.

Container layout follows what the reference reads (nafcodec/src/decoder/mod.rs:169-256,
parser.rs) and writes (encoder/mod.rs:334-384, writer.rs:30-90); mask encoding is the inverse
of MaskReader (reader.rs:198-231).  Not product code."""
import struct

import zstd_ref

NUC = "-TGKCYSBAWRDMHVN"
CODE = {c: i for i, c in enumerate(NUC)}
CODE["U"] = 1


def varint(v: int) -> bytes:
    out = [v & 0x7F]
    v >>= 7
    while v:
        out.append(0x80 | (v & 0x7F))
        v >>= 7
    return bytes(reversed(out))


def pack_nucleotides(seq: str) -> bytes:
    """two per byte, first in the low nibble; one pad nibble if odd (writer.rs:21-28,82)"""
    codes = [CODE[c] for c in seq.upper()]
    if len(codes) & 1:
        codes.append(0)
    return bytes(codes[i] | (codes[i + 1] << 4) for i in range(0, len(codes), 2))


def length_words(lengths) -> bytes:
    """u32-LE words; 0xFFFFFFFF continues (encoder/mod.rs:37-44)"""
    out = bytearray()
    for n in lengths:
        while n >= 0xFFFFFFFF:
            out += struct.pack("<I", 0xFFFFFFFF)
            n -= 0xFFFFFFFF
        out += struct.pack("<I", n)
    return bytes(out)


def mask_bytes(runs) -> bytes:
    """runs alternate unmasked / masked starting unmasked; run = 255*k + r -> k x FF then r"""
    out = bytearray()
    for n in runs:
        while n >= 255:
            out.append(0xFF)
            n -= 255
        out.append(n)
    return bytes(out)


def runs_from_case(seq: str):
    """mask runs that reproduce the case pattern of `seq` under a spec-correct decoder"""
    runs, cur, masked = [], 0, False
    for c in seq:
        low = c.islower()
        if low != masked:
            runs.append(cur)
            cur, masked = 0, low
        cur += 1
    runs.append(cur)
    return runs


def write_naf(records, *, sequence_type="dna", version=None, level=1, separator=" ", line_length=60,
              ids=True, comments=True, lengths=True, sequence=True, quality=False, mask_runs=None, title=None,
              number_of_sequences=None, raw_sections=None, zstd_kwargs=None, extended=False):
    """records: iterable of dicts with id/comment/sequence/quality (strings).
    mask_runs: explicit list of run lengths (else no Mask section).
    raw_sections: {name: bytes} replaces the *decoded* payload of a section before compression."""
    records = list(records)
    st = {"dna": 0, "rna": 1, "protein": 2, "text": 3}[sequence_type]
    if version is None:
        version = 1 if st == 0 else 2
    zk = dict(zstd_kwargs or {})
    flags = 0
    sections = []
    raw_sections = raw_sections or {}

    def add(bit, name, data, original_size=None):
        nonlocal flags
        flags |= bit
        data = raw_sections.get(name, data)
        payload = zstd_ref.compress_magicless(data, level, True, **zk)
        sections.append((len(data) if original_size is None else original_size, payload))

    seqs = [r.get("sequence") or "" for r in records]
    if ids:
        add(0x20, "ids", b"".join((r.get("id") or "").encode() + b"\0" for r in records))
    if comments:
        add(0x10, "comments", b"".join((r.get("comment") or "").encode() + b"\0" for r in records))
    if lengths:
        add(0x08, "lengths", length_words(len(s) for s in seqs))
    if mask_runs is not None:
        add(0x04, "mask", mask_bytes(mask_runs))
    if sequence:
        joined = "".join(seqs)
        if st <= 1:
            add(0x02, "sequence", pack_nucleotides(joined), original_size=len(joined))
        else:
            add(0x02, "sequence", joined.encode())
    if quality:
        add(0x01, "quality", "".join(r.get("quality") or "" for r in records).encode())
    if title is not None:
        flags |= 0x40
    if extended:
        flags |= 0x80
    out = bytearray(b"\x01\xF9\xEC")
    out.append(version)
    if version == 2:
        out.append(st)
    out.append(flags)
    out += separator.encode()
    out += varint(line_length)
    out += varint(len(records) if number_of_sequences is None else number_of_sequences)
    if title is not None:
        t = title if isinstance(title, bytes) else title.encode()
        out += varint(len(t)) + t
    for orig, payload in sections:
        out += varint(orig) + varint(len(payload)) + payload
    return bytes(out)
