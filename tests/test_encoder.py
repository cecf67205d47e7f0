"""Encoder: nafgpu_encoder_* / nafcodec_amd.Encoder against what the reference's own encoder tests
expect (nafcodec/tests/encoder.rs, nafcodec-py/nafcodec/tests/test_encoder.py, encoder/mod.rs:391-460): every archive is
read back by the oracle, its sections by the system libzstd (the decompressor the reference links), its container bytes
compared with tests/naf_writer.py (the layout of encoder/mod.rs:325-384).  Host code: no GPU needed."""
import io
import itertools

import numpy as np
import pytest

import cases
import naf_writer as nw
import zstd_ref
import nafcodec_amd
from nafcodec_amd import Encoder, Record
from oracle import oracle

R1 = dict(id="r1", comment="record 1", sequence="NGCTCTTAAACCTGCTA", quality="#8CCCGGGGGGGGGGGG", length=17)
R2 = dict(id="r2", comment="record 2", sequence="NTAATAAGCAATGACGGCAGC", quality="#8AACCFF<FFGGFGE@@@@@", length=21)   # encoder.rs:12-31


def encode(records, sequence_type="dna", **fields):
    buf = io.BytesIO()
    with Encoder(buf, sequence_type, **fields) as enc:      # (compression_level 0 = the default level: blocks with LZ sequences)
        for r in records:
            enc.write(Record(**r))
    return buf.getvalue()


def decoded(blob):
    return [(r.id, r.comment, r.sequence, r.quality, r.length) for r in oracle.Decoder(blob)]


@pytest.mark.parametrize("fields", [c for n in range(1, 5) for c in itertools.combinations(("id", "comment", "sequence", "quality"), n)])
def test_every_field_selection_reads_back(fields):
    """encoder.rs encode_id / encode_id_sequence / ... : only the enabled fields come back; `length` iff sequence or quality."""
    blob = encode([R1, R2], **{f: True for f in fields})
    d = oracle.Decoder(blob)
    flags = d.header.flags
    assert bool(flags & 0x20) == ("id" in fields) and bool(flags & 0x10) == ("comment" in fields)
    assert bool(flags & 0x02) == ("sequence" in fields) and bool(flags & 0x01) == ("quality" in fields)
    assert bool(flags & 0x08) == ("sequence" in fields or "quality" in fields) and not flags & 0x04   # never a mask (mod.rs:225)
    got = decoded(blob)
    assert len(got) == 2
    for g, r in zip(got, (R1, R2)):
        want = tuple(r[f] if f in fields else None for f in ("id", "comment", "sequence", "quality"))
        assert g[:4] == want
        assert g[4] == (r["length"] if ("sequence" in fields or "quality" in fields) else None)


def test_python_surface_of_the_reference():
    """test_encoder.py:21-84 of the reference's Python package."""
    with pytest.raises(ValueError):
        Encoder(io.BytesIO(), sequence_type="dna", sequence=True).write(Record(sequence="hello world?!"))
    with pytest.raises(ValueError):
        Encoder(io.BytesIO(), sequence_type="dna", sequence=True).write(Record())
    with pytest.raises(ValueError):
        Encoder(io.BytesIO(), sequence_type="dna", sequence=True).write(Record(id="r1"))
    blob = encode([dict(id="r1", sequence="ATTATTAGACAGAGC"), dict(id="r2", sequence="CTATTG"), dict(id="r3", sequence="TTAGTNNNNN")],
                  id=True, sequence=True)
    assert decoded(blob) == [("r1", None, "ATTATTAGACAGAGC", None, 15), ("r2", None, "CTATTG", None, 6), ("r3", None, "TTAGTNNNNN", None, 10)]
    blob = encode([dict(id="r1", sequence="AUUAU", quality="GGGGG"), dict(id="r2", sequence="CUAUU", quality="#8A@C"),
                   dict(id="r3", sequence="UUAGU", quality="CCGGG")], "rna", id=True, sequence=True, quality=True)
    assert decoded(blob) == [("r1", None, "AUUAU", "GGGGG", 5), ("r2", None, "CUAUU", "#8A@C", 5), ("r3", None, "UUAGU", "CCGGG", 5)]
    with pytest.raises(ValueError):
        Encoder(io.BytesIO(), sequence_type="genome")
    enc = Encoder(io.BytesIO(), id=True)
    enc.close()
    with pytest.raises(RuntimeError):
        enc.write(Record(id="x"))                       # "operation on closed encoder." (lib.rs:584)
    with nafcodec_amd.open(io.BytesIO(), "w", sequence_type="protein", id=True, sequence=True) as w:   # encoder/mod.rs:415-438
        w.write(Record(id="r1", comment="record 1", sequence="MYYK"))
        w.write(Record(id="r2", comment="record 2", sequence="MTTE"))


def test_checks_of_push():
    """encoder/mod.rs:236-317: lengths must agree, T is DNA and U is RNA, lower case is not a nucleotide; a refused record
    leaves the archive as it was."""
    buf = io.BytesIO()
    enc = Encoder(buf, "dna", id=True, sequence=True, quality=True)
    enc.write(Record(id="a", sequence="ACGT", quality="IIII"))
    for bad in (Record(id="b", sequence="ACGT", quality="III"), Record(id="b", sequence="ACGT", quality="IIII", length=5),
                Record(id="b", sequence="ACGU", quality="IIII"), Record(id="b", sequence="acgt", quality="IIII"),
                Record(id="b", sequence="ACGT"), Record(sequence="ACGT", quality="IIII")):
        with pytest.raises(ValueError):
            enc.write(bad)
    enc.write(Record(id="c", sequence="TTA", quality="#II", length=3))
    enc.close()
    assert decoded(buf.getvalue()) == [("a", None, "ACGT", "IIII", 4), ("c", None, "TTA", "#II", 3)]


@pytest.mark.skipif(not zstd_ref.available(), reason="libzstd not loadable")
def test_levels_with_and_without_lz_sequences():
    """compression_level 1-2: entropy-coded literals only; 0 (default) and >= 3: greedy matches, sequences coded with the
    predefined FSE tables.  Repetitive records must come out much smaller with matches -- and identical either way."""
    rng = np.random.default_rng(3)
    motif = cases.rand_dna(rng, 5000)
    recs = []
    for i in range(12):
        seq = motif * (3 + i % 5) + cases.rand_dna(rng, 1000 * i) + "N" * (700 * (i % 3))
        recs.append(dict(id="read%d" % i, comment="the same comment over and over %d" % (i % 7), sequence=seq,
                         quality=("IIIIIHHHGG#" * (len(seq) // 11 + 1))[:len(seq)]))
    want = [(r["id"], r["comment"], r["sequence"], r["quality"], len(r["sequence"])) for r in recs]
    sizes = {}
    for level in (1, 3, 19, 0):
        blob = encode(recs, id=True, comment=True, sequence=True, quality=True, compression_level=level)
        assert decoded(blob) == want, level
        d = oracle.Decoder(blob)
        for sec in range(6):
            try:
                data, orig, comp, off = d.section(sec)
            except Exception:
                continue
            assert zstd_ref.decompress_magicless(blob[off:off + comp], len(data) + 8) == data, (level, sec)
        sizes[level] = len(blob)
    assert sizes[3] == sizes[19] == sizes[0] and sizes[3] * 4 < sizes[1], sizes


@pytest.mark.skipif(not zstd_ref.available(), reason="libzstd not loadable")
def test_large_archives_sections_and_container():
    rng = np.random.default_rng(11)
    for st, alphabet, iupac in (("dna", "ACGT", 0.02), ("rna", "ACGU", 0.0), ("protein", "ACDEFGHIKLMNPQRSTVWY", 0.0), ("text", "abc xyz,.", 0.0)):
        lens = [0, 1, 2, 151, 0, 70001, 300000, 3, 999]
        recs = []
        for i, n in enumerate(lens):
            seq = cases.rand_dna(rng, n, alphabet, iupac)
            recs.append(dict(id="rec%d" % i, comment="comment number %d é" % i if i % 3 else "", sequence=seq,
                             quality="".join(rng.choice(list("#8CGGGGGG<AFFJJ"), n))))
        blob = encode(recs, st, id=True, comment=True, sequence=True, quality=True)
        got = decoded(blob)
        assert got == [(r["id"], r["comment"], r["sequence"], r["quality"], len(r["sequence"])) for r in recs], st
        # the same bytes in front of the sections as the layout of encoder/mod.rs:325-347 gives, and every section a frame
        # libzstd reads back to the section's content
        ref = nw.write_naf(recs, sequence_type=st, quality=True)
        hdr = 6 if st == "dna" else 7
        assert blob[:hdr] == ref[:hdr] and blob[hdr:hdr + 2] == ref[hdr:hdr + 2]     # magic, version, [type], flags, separator; line length, count
        d = oracle.Decoder(blob)
        for sec in range(6):
            try:
                data, orig, comp, off = d.section(sec)
            except Exception:
                continue
            if comp:
                assert zstd_ref.decompress_magicless(blob[off:off + comp], len(data) + 8) == data, (st, sec)
        assert len(blob) < 0.75 * sum(len(r["sequence"]) * (1.5 if st in ("dna", "rna") else 2) for r in recs) + 4096   # it does compress
