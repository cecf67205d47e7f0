"""GPU parity tests (run with -m gpu on an MI355X): the HIP path behind the C-ABI
(include/nafgpu.h via nafcodec_amd) against the CPU oracle on the same inputs.

Bar: bit-exact (integer / byte work only on this path)."""
import ctypes
import io
import os
import sys

import numpy as np
import pytest

import cases
import zstd_ref
from conftest import ROOT, golden_bytes
from oracle import oracle

pytestmark = pytest.mark.gpu

FIXTURES = ["LuxC", "masked", "phix", "CP040672", "NZ_AAEN01000029"]
FIELDS = ("id", "comment", "sequence", "quality", "length")


@pytest.fixture(scope="module")
def lib():
    from nafcodec_amd import _ffi
    L = _ffi.default()           # raises if libnafgpu.so is missing: no CPU fallback exists
    name, hbm, cus = L.device_info(0)
    assert "gfx950" in name, name
    return L


def same_records(got, want):
    assert len(got) == len(want)
    for a, b in zip(got, want):
        for f in FIELDS:
            assert getattr(a, f) == getattr(b, f), (f, a.id)


@pytest.mark.parametrize("name", FIXTURES)
def test_fixture_records_match_oracle(lib, name):
    import nafcodec_amd
    data = golden_bytes(name + ".naf")
    got = list(nafcodec_amd.Decoder(io.BytesIO(data)))
    same_records(got, list(oracle.Decoder(data)))


@pytest.mark.parametrize("off", ["id", "comment", "sequence", "quality", "mask"])
def test_fixture_field_selection(lib, off):
    import nafcodec_amd
    for name in ("phix", "masked"):
        data = golden_bytes(name + ".naf")
        got = list(nafcodec_amd.Decoder(io.BytesIO(data), **{off: False}))
        same_records(got, list(oracle.Decoder(data, **{off: False})))


def test_open_path_and_len(lib):
    import os
    import nafcodec_amd
    from conftest import GOLDEN
    d = nafcodec_amd.open(os.path.join(GOLDEN, "phix.naf"))
    assert (len(d), d.sequence_type, d.format_version, d.line_length, d.name_separator) == (42, "dna", "v1", 301, " ")
    next(d)
    assert len(d) == 41
    assert len(list(d)) == 41 and len(d) == 0
    with pytest.raises(FileNotFoundError):
        nafcodec_amd.Decoder("")
    with pytest.raises(IsADirectoryError):
        nafcodec_amd.Decoder(GOLDEN)


def packed_dna(rng, n, alphabet=(1, 2, 4, 8)):
    codes = np.array(alphabet, dtype=np.uint8)
    return (codes[rng.integers(0, len(codes), n)] | (codes[rng.integers(0, len(codes), n)] << 4)).astype(np.uint8).tobytes()


@pytest.mark.skipif(not zstd_ref.available(), reason="libzstd not loadable")
@pytest.mark.parametrize("level", [1, 3, 19])
@pytest.mark.parametrize("n", [1, 7, 1000, 70000, 1 << 20])
def test_zstd_section_libzstd_written(lib, level, n):
    """L0 on its own: sections written by the real libzstd (Huffman, treeless, FSE sequences, repeats)."""
    rng = np.random.default_rng(n * 31 + level)
    data = packed_dna(rng, n)
    for streaming in (True, False):
        payload = zstd_ref.compress_magicless(data, level, streaming)
        assert oracle.zstd_decode(payload, len(data)) == data
        assert lib.zstd_decompress(payload, len(data)) == data


@pytest.mark.skipif(not zstd_ref.available(), reason="libzstd not loadable")
def test_zstd_text_sections(lib):
    rng = np.random.default_rng(5)
    q = rng.choice(np.frombuffer(b"#8CGGGGGGGG<AFFJJ", dtype=np.uint8), 400000).tobytes()
    ids = b"".join(b"@SRR1377138.%d some read\0" % i for i in range(20000))
    rep = packed_dna(rng, 4000) * 50 + bytes(5000) + b"\x11" * 300000 + rng.integers(0, 256, 200000, dtype=np.uint8).tobytes()
    for data in (q, ids, rep):
        for level, kw in ((1, {}), (3, {}), (3, {"flush_every": 97}), (9, {"flush_every": 4001}), (19, {"checksum": True})):
            payload = zstd_ref.compress_magicless(data, level, True, **kw)
            assert lib.zstd_decompress(payload, len(data)) == data


@pytest.mark.skipif(not zstd_ref.available(), reason="libzstd not loadable")
def test_zstd_multi_frame_many_blocks(lib):
    """Frames back to back, > 64 blocks each: repeat offsets composed across chunks, reset per frame."""
    import cases
    for name, payload, data in cases.zstd_payload_cases(scale=2):
        assert oracle.zstd_decode(payload, len(data)) == data, name
        assert lib.zstd_decompress(payload, len(data)) == data, name
        if name == "multi_frame_checksums":                 # a wrong Content_Checksum: Io(InvalidData), as libzstd refuses the frame
            with pytest.raises(Exception, match="checksum"):
                lib.zstd_decompress(payload[:-1] + bytes([payload[-1] ^ 0x80]), len(data))


@pytest.mark.skipif(not zstd_ref.available(), reason="libzstd not loadable")
def test_zstd_corrupt_is_an_error_not_a_fault(lib):
    from nafcodec_amd import NafError
    rng = np.random.default_rng(11)
    data = packed_dna(rng, 200000)
    payload = bytearray(zstd_ref.compress_magicless(data, 3, True))
    for cut in (1, 2, 5, len(payload) // 2, len(payload) - 1):
        with pytest.raises(NafError):
            lib.zstd_decompress(bytes(payload[:cut]), len(data))
    for k in range(24):
        p = bytearray(payload)
        p[int(rng.integers(0, len(p)))] ^= 1 << int(rng.integers(0, 8))
        try:
            out = lib.zstd_decompress(bytes(p), len(data))
        except NafError:
            continue
        # a flip may land in bits no decoder reads; if it decodes, it must agree with the oracle
        assert out == oracle.zstd_decode(bytes(p), len(data))


@pytest.mark.parametrize("n_bases,mask,iupac", [(1000, False, 0), (300001, True, 0), (3000001, True, 5),
                                                (1500000, False, 200), (40_000_001, True, 0)])
def test_synthetic_archive(lib, n_bases, mask, iupac):
    """Synthetic config-2/4 style archives: bulk device decode, checksum-of-checksums against the
    writer's expected value, and (small sizes) per-record equality with the oracle."""
    import nafcodec_amd
    arc = lib.synth(n_bases, seed=n_bases, with_mask=mask, iupac_permille=iupac)
    try:
        blob = ctypes.string_at(arc.bytes, arc.n)
        dec = nafcodec_amd.Decoder(io.BytesIO(blob))
        res = dec.decode_all_device()
        assert (res.n_bases, res.n_records) == (arc.n_bases, arc.n_records)
        assert dec.hash_device(res.d_sequence, res.n_bases) == arc.seq_hash
        assert dec.hash_device(res.d_record_end, 8 * res.n_records) == arc.offsets_hash
        if n_bases <= 3_000_001:
            same_records(list(nafcodec_amd.Decoder(io.BytesIO(blob))), list(oracle.Decoder(blob)))
    finally:
        lib.c.nafgpu_synth_free(ctypes.byref(arc))


def test_large_buffers_from_plain_hipmalloc_and_from_mapped_chunks(lib, monkeypatch):
    """DevBuf takes buffers of 32 MiB and more from the virtual-memory API (an address range backed by hipMemCreate chunks, engine.cpp);
    NAFGPU_ALLOC_PLAIN=1 keeps everything on hipMalloc, NAFGPU_VMM_CHUNK_MIB changes the chunk size: the same archive decodes to
    the same checksums on all of them, twice per decoder (the second call reuses the buffers), and the buffers go away with the
    decoder (a loop of decoders would run out of device memory otherwise)."""
    import nafcodec_amd
    lib.c.nafgpu_test_hooks(1)
    arc = lib.synth(400_000_001, seed=77, with_mask=True, iupac_permille=0)     # 100 MB of input, 400 MB of output
    try:
        blob = ctypes.string_at(arc.bytes, arc.n)
        for env in ({"NAFGPU_ALLOC_PLAIN": "1"}, {}, {"NAFGPU_VMM_CHUNK_MIB": "2"}, {"NAFGPU_VMM_CHUNK_MIB": "64"}):
            for k in ("NAFGPU_ALLOC_PLAIN", "NAFGPU_VMM_CHUNK_MIB"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            for _ in range(3):
                dec = nafcodec_amd.Decoder(io.BytesIO(blob))
                for _ in range(2):
                    res = dec.decode_all_device()
                    assert (res.n_bases, res.n_records) == (arc.n_bases, arc.n_records), env
                    assert dec.hash_device(res.d_sequence, res.n_bases) == arc.seq_hash, env
                    assert dec.hash_device(res.d_record_end, 8 * res.n_records) == arc.offsets_hash, env
                dec.close()
    finally:
        lib.c.nafgpu_synth_free(ctypes.byref(arc))


def test_real_genome_archive_with_a_few_plain_tasks_beside_many_segmented_ones(lib, capfd, monkeypatch):
    """The reference fixture's sequence tiled 40 times and recompressed by libzstd (bench.py's path.real_genome in small): 838 blocks
    with a Huffman tree each, most with a few LZ sequences -- about fifty K1 tasks whose streams are cut into segments by their
    blocks' sequences and a task or two without.  pack_tasks lets those join the segment-aware class (zplan.cpp); the output must
    still be the tiled fixture, and the plan must show no class of the dictionary format that writes the output without segments."""
    if not zstd_ref.available():
        pytest.skip("libzstd not loadable")
    sys.path.insert(0, ROOT)
    import bench
    lib.c.nafgpu_test_hooks(1)
    monkeypatch.setenv("NAFGPU_DEBUG_PLAN", "1")
    leg = bench.real_genome_leg(lib, 0, 40)
    assert leg["bases"] == 40 * 5488676 and "output checksum equals the tiled fixture" in leg["workload"]
    plans = [l for l in capfd.readouterr().err.splitlines() if "task classes" in l and "out+seg" in l]
    assert plans, "no plan with a segment-aware class was printed"
    assert all("tbl 2, out, " not in l for l in plans), plans[-1]


def test_many_small_archives_one_after_the_other(lib):
    """A directory's worth of small archives in one process: every fixture opened, read to the end through the iterator and closed
    fifty times over -- decoders hand their streams on through the pool, small tiles upload as one packed buffer, sections of a few
    sequences take the short match stage (engine.cpp, kernels.hip) -- and every pass yields the oracle's records."""
    import nafcodec_amd
    names = ["LuxC.naf", "phix.naf", "masked.naf", "NZ_AAEN01000029.naf"]
    rec = lambda r: tuple(getattr(r, f) for f in FIELDS)
    want = {n: [rec(r) for r in oracle.Decoder(golden_bytes(n))] for n in names}
    for rep in range(50):
        for n in names:
            if n.startswith("NZ_") and rep % 10:
                continue                                   # (5.5 Mbases of Python strings: every tenth pass)
            dec = nafcodec_amd.Decoder(io.BytesIO(golden_bytes(n)))
            got = [rec(r) for r in dec]
            dec.close()
            assert got == want[n], (n, rep)


def test_gigabase_archive_against_the_oracle(lib):
    """Full-size parity pinned on the ORACLE, not on the writer: a 1.2 Gbase synthetic archive (with a Mask
    section) is drained by the CPU oracle in C, which accumulates the position-keyed checksum of the
    concatenated masked bases and of the u64 record-end table; the GPU decode must reproduce both."""
    import nafcodec_amd
    arc = lib.synth(1_200_000_003, seed=77, with_mask=True, iupac_permille=3)
    try:
        blob = ctypes.string_at(arc.bytes, arc.n)
        want = oracle.Decoder(blob).drain()
        dec = nafcodec_amd.Decoder(io.BytesIO(blob))
        res = dec.decode_all_device()
        assert (res.n_bases, res.n_records) == (want.n_bases, want.n_records) == (arc.n_bases, arc.n_records)
        assert dec.hash_device(res.d_sequence, res.n_bases) == want.seq_hash == arc.seq_hash
        assert dec.hash_device(res.d_record_end, 8 * res.n_records) == want.ends_hash == arc.offsets_hash
    finally:
        lib.c.nafgpu_synth_free(ctypes.byref(arc))


@pytest.mark.parametrize("parts", [2, 16])
def test_streams_in_parts(lib, parts, monkeypatch):
    """Huffman streams cut into parts (plan.h: HufStream::sub; sections with fewer streams than half the chip's lanes get
    them by themselves -- every fixture here does), forced on every shared case at 4x the harness sizes: all three table
    formats, segment-aware streams and the literal buffer, escapes, parts without symbols; corrupted archives; the
    reference's genome fixture as three block ranges through the shard protocol."""
    monkeypatch.setenv("NAFGPU_HUF_SPLIT", str(parts))
    lib.c.nafgpu_test_hooks(1)
    try:
        bad = [name for name, blob, opts in cases.build_cases(scale=4) if cases.run_product(blob, opts) != cases.run_oracle(blob, opts)]
        for name in FIXTURES:
            blob = golden_bytes(name + ".naf")
            if cases.run_product(blob, {}) != cases.run_oracle(blob, {}):
                bad.append(name)
        bad += cases.fuzz_disagreements(cases.fuzz_cases(seed=13, n=60))
        assert not bad
        if parts == 2:
            cases.check_lz_sharding(None, 1, worlds=(3,), names=("real_genome_l1",))
    finally:
        lib.c.nafgpu_test_hooks(0)


def test_next_batch_equals_next(lib):
    """nafgpu_next_batch against nafgpu_next through the C-ABI on the GPU: fixtures, the shared cases at 4x the harness
    sizes (malformed archives included), batch sizes 1 .. 4096."""
    for name in FIXTURES:
        assert cases.check_next_batch(golden_bytes(name + ".naf")) > 0
    for name, blob, opts in cases.build_cases(scale=4):
        cases.check_next_batch(blob, opts, caps=(3, 4096))


def test_both_ends_of_the_full_size_archive_against_the_oracle(lib):
    """The bench's own archive (configs[1]: 40 Gbases, 10 GB) at FULL size.  `bench.py` checks its decode against the
    writer's checksums -- product code vouching for product code; here the first and the last 256 Mi bases of the decode
    are pinned on the oracle (cases.check_archive_ends)."""
    cases.check_archive_ends(lib, 40_000_000_000, 0x4E4146, 1024)


def test_hash_sees_compensating_and_swapped_bytes(lib):
    """The checksum mixes every 8-byte word with its position before adding: errors that a linear sum
    would let cancel (+1 here, -1 there; two words swapped) change it."""
    import nafcodec_amd
    rng = np.random.default_rng(3)
    a = bytearray(rng.integers(0, 256, 70001, dtype=np.uint8).tobytes())
    h0 = lib.c.nafgpu_hash64_host(bytes(a), len(a))
    b = bytearray(a); b[100] = (b[100] + 1) & 255; b[101] = (b[101] - 1) & 255
    c = bytearray(a); c[8:16], c[4096:4104] = a[4096:4104], a[8:16]
    d = bytearray(a); d[-1] ^= 1
    assert len({h0, lib.c.nafgpu_hash64_host(bytes(b), len(b)), lib.c.nafgpu_hash64_host(bytes(c), len(c)),
                lib.c.nafgpu_hash64_host(bytes(d), len(d))}) == 4
    # device == host, at every alignment and tail length
    blob = golden_bytes("phix.naf")
    dec = nafcodec_amd.Decoder(io.BytesIO(blob))
    res = dec.decode_all_device()
    seq = dec.copy_to_host(res.d_sequence, res.n_bases)
    for off, n in ((0, len(seq)), (1, 1000), (3, 8191), (8, 4096), (5, 7), (0, 0)):
        assert dec.hash_device(res.d_sequence + off, n) == lib.c.nafgpu_hash64_host(seq[off:off + n], n), (off, n)


def test_shared_cases_match_oracle(lib):
    """tests/cases.py at 4x the CPU-harness sizes: levels 1/3/19, RNA, protein, FASTQ, per-record
    flush, every mask shape incl. the record-end rule, field selection, malformed archives."""
    import cases
    bad = []
    for name, blob, opts in cases.build_cases(scale=4):
        if cases.run_product(blob, opts) != cases.run_oracle(blob, opts):
            bad.append(name)
    assert not bad


def test_lz_stages_dense_and_sparse(lib, monkeypatch):
    """Both ways of finishing LZ matches on the same inputs (NAFGPU_LZ_MODE forces one), at 4x the CPU-harness
    sizes: element sweeps (pointer jumping) and list passes + one-workgroup stage + frame-order walk."""
    import cases
    import nafcodec_amd
    todo = [c for c in cases.build_cases(scale=4) if c[0] in ("text_dense_chains", "dna_dense_chains", "dna_homopolymer", "dna_l3",
                                                              "dna_l3_big", "text_quality", "dna_repeats_l1", "fastq_flush_per_record")]
    lib.c.nafgpu_test_hooks(1)
    for mode in ("dense", "sparse"):
        monkeypatch.setenv("NAFGPU_LZ_MODE", mode)
        for name, blob, opts in todo:
            if "dense_chains" in name:
                res = nafcodec_amd.Decoder(io.BytesIO(blob)).decode_all_device()
                assert res.lz_residue_matches > 0, (name, mode)
            assert cases.run_product(blob, opts) == cases.run_oracle(blob, opts), (name, mode)
        for name, payload, data in cases.zstd_payload_cases(scale=2):
            assert lib.zstd_decompress(payload, len(data)) == data, (name, mode)
    monkeypatch.setenv("NAFGPU_LZ_MODE", "dense")
    for strips in ("1", "0"):                               # the sweeps strip-wise and tile-wise (the library picks by the share of literals)
        monkeypatch.setenv("NAFGPU_PJ_STRIPS", strips)
        for name, blob, opts in todo:
            assert cases.run_product(blob, opts) == cases.run_oracle(blob, opts), (name, "strips", strips)
        for name, payload, data in cases.zstd_payload_cases(scale=2):
            assert lib.zstd_decompress(payload, len(data)) == data, (name, "strips", strips)
    monkeypatch.delenv("NAFGPU_PJ_STRIPS")
    monkeypatch.setenv("NAFGPU_PJ_MAX_DIST", "16")          # distances cannot grow: the frame-order walk finishes
    for name, blob, opts in todo[:3]:
        assert cases.run_product(blob, opts) == cases.run_oracle(blob, opts), (name, "limit")


def test_sequence_chains_out_of_lds_and_out_of_l2(lib, monkeypatch):
    """The three bodies of k_seq_states (NAFGPU_K2_LDS forces one: 4-byte cells in LDS, 2-byte cells in LDS, out of L2) on every archive case with LZ sequences and on the fuzz set."""
    import cases
    lib.c.nafgpu_test_hooks(1)
    todo = [c for c in cases.build_cases(scale=4) if c[0].startswith(("dna_l", "text_", "fastq_", "dna_repeat", "dna_dense", "dna_homo", "protein",
                                                                      "rna", "checksum", "truncated", "bitflip"))]
    for force in ("1", "2", "0"):
        monkeypatch.setenv("NAFGPU_K2_LDS", force)
        for name, blob, opts in todo:
            assert cases.run_product(blob, opts) == cases.run_oracle(blob, opts), (name, force)
        for name, payload, data in cases.zstd_payload_cases(scale=2):
            assert lib.zstd_decompress(payload, len(data)) == data, (name, force)
        assert cases.fuzz_disagreements(cases.fuzz_cases(seed=5, n=60)) == []


def test_error_timing_against_the_streaming_reference(lib, monkeypatch):
    """Malformed inside a section: the reference (oracle/ref_shape.c: streaming, as mod.rs:356-399) has handed out k records
    when it meets the damage; this library fails at the first record that needs the section -- or, decoded in tiles, the
    tile that holds the damage -- and what it handed out before are the reference's first records."""
    import cases
    import naf_writer as nw
    import numpy as np
    from oracle import oracle
    if not oracle.ref_shape_available():
        pytest.skip("libzstd not loadable")
    for name, blob, opts in cases.build_cases(scale=2):
        if name in ("truncated_mid", "truncated_tail", "checksum_wrong", "bitflip_sequence"):
            k, rc, j, err = cases.error_timing(blob, opts)
            assert name == "bitflip_sequence" or (rc != 0 and j == 0), (name, k, rc, j, err)
    rng = np.random.default_rng(8)
    good = bytearray(nw.write_naf(cases.make_records(rng, [20000] * 300, iupac=0.01), level=1))
    at = len(good) * 3 // 4
    good[at:at + 300] = bytes(300)
    lib.c.nafgpu_test_hooks(1)
    k, rc, j_whole, _ = cases.error_timing(bytes(good), {})
    monkeypatch.setenv("NAFGPU_TILE_KIB", "1024")
    k2, rc2, j_tiled, _ = cases.error_timing(bytes(good), {}, eager=False, slack=1)
    assert rc != 0 and (k, rc) == (k2, rc2) and j_whole == 0 and k - 27 <= j_tiled <= k + 1, (k, j_whole, j_tiled)


def test_corrupted_archives_terminate_and_never_disagree_silently(lib):
    """The same corrupted inputs the CPU harness runs under AddressSanitizer: on the GPU they must
    come back as errors (or as the same records the oracle gives), never as a fault or a hang."""
    import cases
    assert cases.fuzz_disagreements(cases.fuzz_cases(seed=7, n=70)) == []
    assert cases.fuzz_disagreements(cases.fuzz_cases(seed=11, n=120)) == []


def test_block_range_sharding(lib):
    """configs[4] at small scale: one archive, every rank decodes a contiguous zstd-block range."""
    import cases
    cases.check_sharding(None, 40_000_001, True, worlds=(2, 8))
    cases.check_sharding(None, 3_000_001, False, worlds=(3,))


def test_shard_protocol_on_sections_with_lz_sequences(lib):
    """One archive WITH LZ sequences over 2 / 3 / 8 block ranges (the shard protocol, every rank in this process): the
    statistics of a real genome, level-3 DNA (one frame, three frames), FASTQ-like reads at levels 1 and 3, dense chains;
    the route the library picks and both routes forced."""
    import cases
    cases.check_lz_sharding(None, 8, worlds=(2, 3, 8))
    cases.check_lz_sharding(None, 4, worlds=(3,), force_modes=("dense", "sparse"))


GOLDEN_TEXT = [("LuxC", "LuxC.faa"), ("masked", "masked.fna"), ("phix", "phix.fastq")]


@pytest.mark.parametrize("name,text", GOLDEN_TEXT)
def test_text_output_equals_the_reference_fixture_texts(lib, name, text):
    """nafgpu_format_device pinned on the source texts the reference's fixtures were made from."""
    from nafcodec_amd.decoder import Decoder
    want = golden_bytes(text)
    if not want.endswith(b"\n"):
        want += b"\n"                                   # masked.fna lacks the final newline
    got = Decoder(io.BytesIO(golden_bytes(name + ".naf")), **({})).to_text()
    assert got == want


def test_text_output_matches_oracle_records(lib):
    from nafcodec_amd.decoder import Decoder
    for name, blob in cases.text_cases(scale=4):
        assert Decoder(io.BytesIO(blob), **({})).to_text() == cases.oracle_text(blob), name
    # field selection: no comments in the names, no quality -> FASTA of a FASTQ archive
    name, blob = cases.text_cases(scale=1)[-1]
    assert Decoder(io.BytesIO(blob), comment=False, quality=False, **({})).to_text() == \
        cases.oracle_text(blob, {"comment": False, "quality": False})


def test_device_string_tables_and_utf8_flags(lib):
    """CStringReader on the device (offsets past each NUL) and the UTF-8 verdict per text section."""
    import numpy as np
    import naf_writer as nw
    from nafcodec_amd.decoder import Decoder
    from oracle import oracle
    blob = golden_bytes("phix.naf")
    d = Decoder(io.BytesIO(blob), **({}))
    res = d.decode_all_device()
    recs = list(oracle.Decoder(blob))
    assert (res.n_ids, res.n_comments, res.utf8_invalid) == (len(recs), len(recs), 0)
    ends = np.frombuffer(d.copy_to_host(res.d_id_end, 8 * res.n_ids), dtype=np.uint64)
    assert list(ends) == list(np.cumsum([len(r.id.encode()) + 1 for r in recs]))
    ends = np.frombuffer(d.copy_to_host(res.d_comment_end, 8 * res.n_comments), dtype=np.uint64)
    assert list(ends) == list(np.cumsum([len(r.comment.encode()) + 1 for r in recs]))
    recs = [{"id": "a", "comment": "x", "sequence": "ACGT"}, {"id": "b", "comment": "y", "sequence": "AC"}]
    for section, bit, payload in (("ids", 0, b"a\xff\x00b\x00"), ("comments", 1, b"x\x00\xe0\x80\x80\x00"),
                                  ("ids", 0, b"a\x00b\xc3\x00"), ("comments", 1, b"\xed\xa0\x80\x00y\x00")):
        bad = nw.write_naf(recs, raw_sections={section: payload})
        assert Decoder(io.BytesIO(bad), **({})).decode_all_device().utf8_invalid == 1 << bit, (section, payload)
    good = nw.write_naf(recs, raw_sections={"comments": "é\u20ac\U0001F600\x00y\x00".encode()})
    assert Decoder(io.BytesIO(good), **({})).decode_all_device().utf8_invalid == 0
    # sections longer than the 16-byte chunks of k_utf8_check: multi-byte characters across chunk borders, one byte
    # overwritten at every position around them -- the verdict must be Python's
    rng = np.random.default_rng(9)
    text = "".join(rng.choice(list("abcdefghij é\u20ac\U0001F600z"), 150))
    base = b"x\x00" + text.encode() + b"\x00"
    for pos in list(range(10, 70)) + [len(base) - 2]:
        for byte in (0x80, 0xC3, 0xE2, 0xF0, 0xFF, 0x41):
            payload = bytearray(base)
            payload[pos] = byte
            try:
                bytes(payload).decode("utf-8")
                want = 0
            except UnicodeDecodeError:
                want = 2
            if payload.count(0) != 2:
                continue
            arc = nw.write_naf(recs, raw_sections={"comments": bytes(payload)})
            assert Decoder(io.BytesIO(arc)).decode_all_device().utf8_invalid == want, (pos, hex(byte))


def test_encoder_output_decodes_on_the_gpu(lib):
    """Archives written by nafcodec_amd.Encoder (nafgpu_encoder_*: SURVEY 8f-1) come back unchanged through the HIP path:
    every sequence type, records across many zstd blocks, single-symbol sections (RLE blocks), an odd total of nucleotides."""
    import nafcodec_amd
    rng = np.random.default_rng(23)
    for st, alphabet, iupac in (("dna", "ACGT", 0.02), ("rna", "ACGU", 0.01), ("protein", "ACDEFGHIKLMNPQRSTVWY", 0.0), ("text", "abc xyz,.", 0.0)):
        recs = []
        for i, n in enumerate([0, 1, 2, 151, 0, 700001, 3000003, 3, 999]):
            recs.append(nafcodec_amd.Record(id="rec%d" % i, comment="comment %d é" % i if i % 3 else "",
                                            sequence=cases.rand_dna(rng, n, alphabet, iupac), quality="I" * n if i % 2 else
                                            "".join(rng.choice(list("#8CGGGGGG<AFFJJ"), n))))
        for level in ((1, 3) if st == "dna" else (0,)):       # 1: literals only; 0 / 3: blocks with LZ sequences (predefined FSE tables)
            buf = io.BytesIO()
            with nafcodec_amd.Encoder(buf, st, id=True, comment=True, sequence=True, quality=True, compression_level=level) as enc:
                for r in recs:
                    enc.write(r)
            got = list(nafcodec_amd.Decoder(io.BytesIO(buf.getvalue())))
            assert len(got) == len(recs)
            for a, b in zip(got, recs):
                assert (a.id, a.comment, a.sequence, a.quality, a.length) == (b.id, b.comment, b.sequence, b.quality, len(b.sequence)), (st, level, b.id)


@pytest.mark.gpu
def test_decoders_in_several_threads_at_once(lib):
    """Decoders are independent objects: four threads each open, read and close the reference's fixtures and a few generated
    archives over and over (ctypes releases the GIL inside every call), sharing the library's process-wide pools -- streams,
    pinned windows, small device buffers, the staging threads of large uploads.  Every run must give the oracle's records."""
    import threading
    todo = [(n, golden_bytes(n + ".naf"), {}) for n in ("NZ_AAEN01000029", "phix", "LuxC", "masked")]
    todo += [c for c in cases.build_cases(1) if c[0] in ("dna_l3_big", "text_quality", "dna_multi_tree_compact", "fastq_flush_per_record")]
    want = {name: cases.run_oracle(blob, opts) for name, blob, opts in todo}
    bad = []

    def work(k):
        try:
            for rep in range(6):
                for name, blob, opts in todo[k % 2::2] if rep % 2 else todo:
                    if cases.run_product(blob, opts, lib) != want[name]:
                        bad.append((k, rep, name))
        except Exception as e:                              # noqa: BLE001
            bad.append((k, repr(e)))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not bad, bad[:5]


@pytest.mark.gpu
def test_kept_device_memory_is_given_back_on_request(lib):
    """Closed decoders leave their large mapped ranges and small buffers to the next decoder of the process
    (engine.cpp: range_pool, SmallCache); nafgpu_trim_device_memory gives them back to the driver, and a decoder after that
    works as one before it."""
    hip = ctypes.CDLL("libamdhip64.so")                     # (the runtime the library itself is linked to: hipMemGetInfo)

    def free_bytes():
        free, total = ctypes.c_size_t(), ctypes.c_size_t()
        assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        return free.value

    arc = lib.synth(600_000_011, seed=21)
    try:
        blob = ctypes.string_at(arc.bytes, arc.n)

        def once():
            from nafcodec_amd.decoder import Decoder
            with Decoder(io.BytesIO(blob)) as d:
                res = d.decode_all_device()
                assert d.hash_device(res.d_sequence, res.n_bases) == arc.seq_hash

        once()
        free_kept = free_bytes()
        assert lib.c.nafgpu_trim_device_memory(0) == 0
        free_trimmed = free_bytes()
        assert free_trimmed >= free_kept, (free_kept, free_trimmed)                   # (hipMemGetInfo does not count hipMemCreate chunks: the small buffers show)
        once()                                                                        # ... and a decoder after the trim is as good as one before
        once()
    finally:
        lib.c.nafgpu_synth_free(ctypes.byref(arc))
