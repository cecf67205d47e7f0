"""Pins the CPU oracle (oracle/*.c) against every fixture and known answer the reference's own
tests hold for the decode path.

Reference tests restated here (file:line under /root/reference):
  nafcodec/src/decoder/parser.rs:141-152          header known answer
  nafcodec/src/encoder/mod.rs:391-413             varint known answers
  nafcodec/src/decoder/mod.rs:463-516             empty input, LuxC count, mask units, skip_sequence
  nafcodec/tests/decoder/{dna,fastq,protein}.rs   fixture-driven decode checks
  nafcodec-py/nafcodec/tests/test_decoder.py      the same through the Python API
"""
import hashlib

import pytest

from oracle import oracle
from conftest import golden_bytes

# first 16 hex of SHA-256, SURVEY.md Appendix C: (file, seq, ids, comments, qual)
HASHES = {
    "LuxC": ("afaab61a2c11abde", "b3dd0e7c601e2e0d", "3242b4419071aff8", "1f85d2778a50347c", None),
    "masked": ("0a687ff2a0c2379e", "c921ec989ea0cd2c", "a2a8f0b444cee650", "6e340b9cffb37a98", None),
    "phix": ("b6fa311712e0ca54", "31adb5c8cf3806ec", "69ce9145062e09ef", "50db0b0dd86ddccb", "1ed7cb3cdcc2bf22"),
    "CP040672": ("956d7eba207f0c15", "c3bc2d8e85b84292", "14c54b8aa660cea4", "95cf0a798c54bc80", None),
    "NZ_AAEN01000029": ("295d12b363969491", "84242bd01d97b877", "0eca7fb986e0525d", "b0521bdfd231c83a", None),
}


def h16(b):
    return hashlib.sha256(b).hexdigest()[:16]


@pytest.mark.parametrize("name", sorted(HASHES))
def test_golden_hashes(name):
    data = golden_bytes(name + ".naf")
    fh, sh, ih, ch, qh = HASHES[name]
    assert h16(data) == fh
    recs = list(oracle.Decoder(data, raw=True))
    assert h16(b"".join(r.sequence for r in recs)) == sh
    assert h16(b"\0".join(r.id for r in recs)) == ih
    assert h16(b"\0".join(r.comment for r in recs)) == ch
    if qh:
        assert h16(b"".join(r.quality for r in recs)) == qh


def _names(recs, sep):
    return [r.id + (sep + r.comment if r.comment else "") for r in recs]


def test_luxc_equals_source_fasta():
    d = oracle.Decoder(golden_bytes("LuxC.naf"))
    recs = list(d)
    names, seqs = [], []
    for line in golden_bytes("LuxC.faa").decode().splitlines():
        if line.startswith(">"):
            names.append(line[1:])
            seqs.append("")
        else:
            seqs[-1] += line
    assert _names(recs, d.name_separator) == names
    assert [r.sequence for r in recs] == seqs


def test_masked_equals_source_fasta():
    d = oracle.Decoder(golden_bytes("masked.naf"))
    recs = list(d)
    names, seqs = [], []
    for line in golden_bytes("masked.fna").decode().splitlines():
        if line.startswith(">"):
            names.append(line[1:])
            seqs.append("")
        else:
            seqs[-1] += line
    assert _names(recs, d.name_separator) == names
    assert [r.sequence for r in recs] == seqs  # case-sensitive


def test_phix_equals_source_fastq():
    d = oracle.Decoder(golden_bytes("phix.naf"))
    recs = list(d)
    lines = golden_bytes("phix.fastq").decode().splitlines()
    assert len(lines) == 4 * len(recs)
    for i, r in enumerate(recs):
        assert lines[4 * i] == "@" + r.id + (d.name_separator + r.comment if r.comment else "")
        assert lines[4 * i + 1] == r.sequence
        assert lines[4 * i + 3] == r.quality


# ---- parser.rs / encoder varint known answers ----------------------------------------------

def test_header_known_answer():  # parser.rs:141-152
    h, used = oracle.parse_header(bytes([0x01, 0xF9, 0xEC, 0x01, 0x3E, 0x20, 0x3C, 0x20]))
    assert chr(h.name_separator) == " " and h.line_length == 60 and h.number_of_sequences == 32
    assert used == 8 and h.format_version == 1 and h.sequence_type == 0 and h.flags == 0x3E


@pytest.mark.parametrize("value,enc", [  # encoder/mod.rs:391-413
    (0, "00"), (127, "7f"), (128, "8100"), (129, "8101"),
    (34359738367, "ffffffff7f"), (34359738368, "818080808000")])
def test_varint_known_answers(value, enc):
    assert oracle.variable_u64(bytes.fromhex(enc)) == (value, len(enc) // 2)


def test_varint_incomplete_and_overflow():
    with pytest.raises(oracle.OracleError) as e:
        oracle.variable_u64(b"\x81\x80")
    assert e.value.kind == oracle.E_IO_EOF
    # parser.rs:38 guards only the addition; `limb * basis` wraps in a release build,
    # which makes TooLarge unreachable: 70 significant bits come back truncated to 64.
    assert oracle.variable_u64(b"\xff" * 9 + b"\x7f") == (2**64 - 1, 10)
    assert oracle.variable_u64(b"\xff" * 12 + b"\x7f") == (2**64 - 1, 13)


def test_header_errors():
    with pytest.raises(oracle.OracleError) as e:  # mod.rs:470-476 error_empty
        oracle.Decoder(b"")
    assert e.value.kind == oracle.E_IO_EOF
    with pytest.raises(oracle.OracleError) as e:
        oracle.Decoder(b"\x01\xF9\xED\x01\x3E\x20\x3C\x20")
    assert (e.value.kind, e.value.nom_code) == (oracle.E_NOM, oracle.NOM_VERIFY)
    with pytest.raises(oracle.OracleError) as e:
        oracle.Decoder(b"\x01\xF9\xEC\x03\x3E\x20\x3C\x20")
    assert (e.value.kind, e.value.nom_code) == (oracle.E_NOM, oracle.NOM_MAPRES)
    with pytest.raises(oracle.OracleError) as e:
        oracle.Decoder(b"\x01\xF9\xEC\x02\x04\x3E\x20\x3C\x20")
    assert (e.value.kind, e.value.nom_code) == (oracle.E_NOM, oracle.NOM_MAPRES)
    with pytest.raises(oracle.OracleError) as e:
        oracle.Decoder(b"\x01\xF9\xEC\x01\x3E\x1F\x3C\x20")
    assert (e.value.kind, e.value.nom_code) == (oracle.E_NOM, oracle.NOM_VERIFY)


# ---- decoder/mod.rs inline tests ------------------------------------------------------------

def test_luxc_count():  # mod.rs:479-483
    assert len(list(oracle.Decoder(golden_bytes("LuxC.naf")))) == 12


def test_mask_units():  # mod.rs:486-504 + Appendix C
    d = oracle.Decoder(golden_bytes("masked.naf"))
    units = d.mask_units()
    assert units[:5] == [(False, 657), (True, 19), (False, 635), (True, 39), (False, 725)]
    assert [n for _, n in units] == [657, 19, 635, 39, 725, 96, 99, 13, 174, 14, 879]
    d = oracle.Decoder(golden_bytes("phix.naf"))
    assert [n for _, n in d.mask_units()] == [301, 7, 12, 4, 12112]


def test_skip_sequence():  # mod.rs:507-515
    for r in oracle.Decoder(golden_bytes("LuxC.naf"), sequence=False):
        assert r.sequence is None and r.length is not None


# ---- tests/decoder/dna.rs -------------------------------------------------------------------

def test_dna_genome():  # dna.rs:9-34
    d = oracle.Decoder(golden_bytes("NZ_AAEN01000029.naf"))
    assert (d.name_separator, d.number_of_sequences, d.line_length, d.sequence_type) == (" ", 30, 80, "dna")
    r1 = next(d)
    assert r1.id == "NZ_AAEN01000029.1"
    assert r1.comment == ("Bacillus anthracis str. CNEVA-9066 map unlocalized plasmid pXO1 cont2250, "
                          "whole genome shotgun sequence")
    assert len(r1.sequence) == 182777
    assert [r1.sequence.count(c) for c in "ACGT"] == [62115, 28747, 30763, 61152]
    r2 = next(d)
    assert r2.id == "NZ_AAEN01000030.3"
    assert r2.comment.startswith("Bacillus anthracis str. CNEVA-9066 map unlocalized plasmid pXO2 cont2251")
    assert len(list(d)) == 28


def test_dna_mask():  # dna.rs:36-63
    d = oracle.Decoder(golden_bytes("masked.naf"))
    assert (d.name_separator, d.number_of_sequences, d.line_length, d.sequence_type) == (" ", 2, 50, "dna")
    r1, r2 = next(d), next(d)
    assert r1.id == "test1" and r2.id == "test2"
    s = r1.sequence
    assert s[:657].isupper() and s[657:676].islower() and s[676:1311].isupper() and s[1311:1350].islower()
    s = r2.sequence
    assert s[:525].isupper() and s[525:621].islower() and s[621:720].isupper() and s[720:733].islower()
    with pytest.raises(StopIteration):
        next(d)


def test_dna_force_nomask():  # dna.rs:65-88
    recs = list(oracle.Decoder(golden_bytes("masked.naf"), mask=False))
    assert [r.id for r in recs] == ["test1", "test2"]
    assert all(r.sequence.isupper() for r in recs)


# ---- tests/decoder/fastq.rs -----------------------------------------------------------------

def test_fastq_header_and_decode():  # fastq.rs:16-53
    d = oracle.Decoder(golden_bytes("phix.naf"))
    assert d.number_of_sequences == 42 and d.sequence_type == "dna" and d.name_separator == " "
    for bit in (0x01, 0x02, 0x20, 0x10):
        assert d.header.flags & bit
    r1 = next(d)
    assert r1.id == "SRR1377138.1"
    assert r1.comment == "a comment that should not be included in the SAM output"
    assert r1.sequence.startswith("NGCTCTTAAACCTGCTATTGAGGCTTGTGGCATTTC")
    assert r1.quality.startswith("#8CCCGGGGGGGGGGGGGGGGGGGGGGGGGG")
    r2 = next(d)
    assert r2.id == "SRR1377138.2" and r2.comment == "some lowercase nucleotides"
    assert len(list(d)) == 40


@pytest.mark.parametrize("field", ["id", "sequence", "comment", "quality"])  # fastq.rs:55-118
def test_fastq_field_off(field):
    d = oracle.Decoder(golden_bytes("phix.naf"), **{field: False})
    recs = list(d)
    assert len(recs) == 42
    for r in recs:
        assert getattr(r, field) is None
        for other in {"id", "sequence", "comment", "quality"} - {field}:
            assert getattr(r, other) is not None
        assert r.length is not None


# ---- tests/decoder/protein.rs + python tests -------------------------------------------------

def test_protein():  # protein.rs:4-22, test_decoder.py:75-85
    d = oracle.Decoder(golden_bytes("LuxC.naf"))
    assert (d.name_separator, d.number_of_sequences, d.line_length, d.sequence_type) == (" ", 12, 60, "protein")
    assert d.format_version == "v2"
    recs = list(d)
    assert len(recs[0].sequence) == 488
    assert recs[0].id == "sp|P19841|LUXC_PHOPO" and recs[0].sequence[:25] == "MCNAEFKGDCMIKKIPMIIGGAERD"
    assert recs[5].id == "sp|P29236|LUXC2_PHOLE" and recs[5].sequence[:25] == "MIKKIPMIIGGVVQNTSGYGMRELT"
    assert all(r.quality is None for r in recs)


def test_python_len_countdown():  # test_decoder.py:40-47
    d = oracle.Decoder(golden_bytes("phix.naf"))
    assert len(d) == 42
    next(d)
    assert len(d) == 41
    assert len(list(d)) == 41 and len(d) == 0


def test_python_fastq_optional():  # test_decoder.py:24-37
    for r in oracle.Decoder(golden_bytes("phix.naf"), id=False, sequence=False, comment=False):
        assert r.id is None and r.sequence is None and r.comment is None and r.quality is not None


def test_python_dna_cp040672():  # test_decoder.py:62-72
    d = oracle.Decoder(golden_bytes("CP040672.naf"))
    recs = list(d)
    assert len(recs) == 100 and d.sequence_type == "dna"
    assert recs[0].id == "lcl|NZ_CP040672.1_cds_WP_044801954.1_1"
    assert [recs[0].sequence.count(c) for c in "ACGT"] == [181, 200, 210, 240]
    assert recs[0].quality is None


def test_first_lengths():  # Appendix C
    want = {"LuxC": [488, 477, 479, 480, 478], "CP040672": [831, 1161, 987, 525, 2007],
            "NZ_AAEN01000029": [182777, 95646, 1087, 265902, 145793], "phix": [301] * 5}
    for name, lens in want.items():
        d = oracle.Decoder(golden_bytes(name + ".naf"), sequence=False, quality=False)
        assert [next(d).length for _ in range(5)] == lens


def test_section_facts():  # Appendix C table: original -> compressed sizes, zstd internals
    d = oracle.Decoder(golden_bytes("NZ_AAEN01000029.naf"))
    data, orig, comp, off = d.section(4)
    assert (orig, comp, len(data)) == (5488676, 1330710, 2744338)
    payload = golden_bytes("NZ_AAEN01000029.naf")[off:off + comp]
    out, st = oracle.zstd_decode(payload, len(data), stats=True)
    assert out == data
    assert (st.blocks, st.blocks_compressed, st.lit_huf, st.lit_treeless) == (21, 21, 15, 6)
    assert (st.sequences, st.lit_bytes, st.window) == (46, 2738152, 512 << 10)
    d = oracle.Decoder(golden_bytes("phix.naf"))
    data, orig, comp, off = d.section(5)
    out, st = oracle.zstd_decode(golden_bytes("phix.naf")[off:off + comp], orig, stats=True)
    assert (orig, comp, st.sequences, st.lit_bytes) == (12436, 2605, 501, 2954)
    assert st.seq_mode_count[2] == st.seq_mode_count[6] == st.seq_mode_count[10] == 1  # all FSE-described


@pytest.mark.parametrize("name", ["LuxC", "masked", "phix", "CP040672", "NZ_AAEN01000029"])
def test_reference_shaped_pipeline_equals_the_checker(name):
    """oracle/ref_shape.c (what bench.py's cpu_baseline leg times: streaming libzstd through 4 KiB buffers, per-nibble
    push, one heap string per field and record) decodes every fixture to the same records as the checker."""
    from oracle import oracle
    if not oracle.ref_shape_available():
        pytest.skip("libzstd.so.1 not loadable")
    blob = golden_bytes(name + ".naf")
    a, b = oracle.ref_shape_drain(blob), oracle.Decoder(blob).drain()
    for field, _ in oracle.DrainResult._fields_:
        assert getattr(a, field) == getattr(b, field), field
    assert oracle.ref_shape_drain_parallel(blob, 3) == 3 * b.n_bases
