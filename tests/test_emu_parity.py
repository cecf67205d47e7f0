"""CPU-harness tests: the SAME host + device sources as libnafgpu.so, compiled with g++ against
tests/emu/hip/hip_runtime.h (one fiber per work-item, switched at __syncthreads()) and compared
with the CPU oracle case by case.  This is test infrastructure for logic and memory-safety
(AddressSanitizer / UBSan are not available on the GPU pool); the parity tests proper are the
`-m gpu` ones, which run the hipcc build on an MI355X."""
import io
import os
import subprocess
import sys

import pytest

import cases
import zstd_ref
from conftest import ROOT, golden_bytes

EMU_DIR = os.path.join(ROOT, "tests", "emu", "_build")
CSRC = os.path.join(ROOT, "nafcodec_amd", "csrc")

pytestmark = pytest.mark.skipif(not zstd_ref.available(), reason="libzstd not loadable (cases are written with it)")


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-s", "-C", CSRC, "emu"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    from nafcodec_amd import _ffi
    return _ffi.Library(os.path.join(EMU_DIR, "libnafgpu_emu.so"))


@pytest.fixture(scope="module")
def all_cases():
    return cases.build_cases(scale=1)


def test_cases_match_oracle(emu, all_cases):
    bad = []
    for name, blob, opts in all_cases:
        if cases.run_product(blob, opts, emu) != cases.run_oracle(blob, opts):
            bad.append(name)
    assert not bad


@pytest.mark.parametrize("parts", [2, 16])
def test_streams_in_parts(emu, all_cases, parts, monkeypatch):
    """Huffman streams cut into parts (plan.h: HufStream::sub -- what sections with few streams get, so that a chip's lanes
    have something to do): forced on EVERY case here, whatever its size -- one-tree (baked), compact and dictionary tables,
    blocks with sequences (segment-aware streams, literal buffer), escapes, checksummed frames, streams of a few symbols
    whose parts are empty -- and on corrupted archives, where a part's guessed decode may run into anything."""
    monkeypatch.setenv("NAFGPU_HUF_SPLIT", str(parts))
    emu.c.nafgpu_test_hooks(1)
    try:
        bad = [name for name, blob, opts in all_cases if cases.run_product(blob, opts, emu) != cases.run_oracle(blob, opts)]
        for name in ("phix", "masked", "CP040672", "NZ_AAEN01000029"):
            blob = golden_bytes(name + ".naf")
            if cases.run_product(blob, {}, emu) != cases.run_oracle(blob, {}):
                bad.append(name)
        if parts == 2:
            bad += cases.fuzz_disagreements(cases.fuzz_cases(seed=13, n=60), emu)
        assert not bad
        if parts == 2:                                      # ... and in tiles, and as block ranges of several ranks
            monkeypatch.setenv("NAFGPU_TILE_KIB", "256")
            blob = golden_bytes("NZ_AAEN01000029.naf")
            assert cases.run_product(blob, {}, emu) == cases.run_oracle(blob, {})
            monkeypatch.delenv("NAFGPU_TILE_KIB")
            cases.check_sharding(emu, 1_500_000, True, worlds=(3,))
            cases.check_lz_sharding(emu, 1, worlds=(3,), names=("real_genome_l1",))
    finally:
        emu.c.nafgpu_test_hooks(0)


def test_zstd_multi_frame_many_blocks(emu):
    from oracle import oracle
    for name, payload, data in cases.zstd_payload_cases(scale=1):
        assert oracle.zstd_decode(payload, len(data)) == data, name
        assert emu.zstd_decompress(payload, len(data)) == data, name
        if name == "multi_frame_checksums":                 # a wrong checksum in the last frame: refused by both
            bad = payload[:-1] + bytes([payload[-1] ^ 0x80])
            with pytest.raises(Exception, match="checksum"):
                emu.zstd_decompress(bad, len(data))
            with pytest.raises(Exception):
                oracle.zstd_decode(bad, len(data))


@pytest.mark.parametrize("name", ["LuxC", "masked", "phix", "CP040672", "NZ_AAEN01000029"])
def test_fixtures_match_oracle(emu, name):
    blob = golden_bytes(name + ".naf")
    assert cases.run_product(blob, {}, emu) == cases.run_oracle(blob, {})


def test_lz_stages_dense_and_sparse(emu, all_cases, monkeypatch):
    """Both ways of finishing LZ matches on the same inputs (NAFGPU_LZ_MODE forces one): the element sweeps of
    dense sections (pointer jumping), and the list passes + one-workgroup stage + frame-order walk of sparse
    ones.  Dense chains must leave the launched passes a residue (else this stopped covering the stages behind them)."""
    import io
    from nafcodec_amd.decoder import Decoder
    from oracle import oracle
    emu.c.nafgpu_test_hooks(1)
    for mode in ("dense", "sparse"):
        monkeypatch.setenv("NAFGPU_LZ_MODE", mode)
        for name, blob, opts in all_cases:
            if name in ("text_dense_chains", "dna_dense_chains", "dna_homopolymer", "dna_l3", "text_quality", "dna_repeats_l1"):
                if "dense_chains" in name:
                    res = Decoder(io.BytesIO(blob), _lib=emu).decode_all_device()
                    assert res.lz_residue_matches > 0, (name, mode)
                assert cases.run_product(blob, opts, emu) == cases.run_oracle(blob, opts), (name, mode)
        for name, payload, data in cases.zstd_payload_cases(scale=1):
            if name in ("multi_frame_l3_flush97", "equal_length_words_l1"):    # (all of them, larger, in the GPU test of the same name)
                assert emu.zstd_decompress(payload, len(data)) == data, (name, mode)
    # the sweeps of dense sections tile-wise and strip-wise (NAFGPU_PJ_STRIPS forces either; the library picks by the share of literals)
    monkeypatch.setenv("NAFGPU_LZ_MODE", "dense")
    for strips in ("1", "0"):
        monkeypatch.setenv("NAFGPU_PJ_STRIPS", strips)
        for name, blob, opts in all_cases:
            if name in ("checksum_text_l3", "text_quality", "dna_l3"):
                assert cases.run_product(blob, opts, emu) == cases.run_oracle(blob, opts), (name, "strips", strips)


def test_sequence_chains_out_of_lds_and_out_of_l2(emu, all_cases, monkeypatch):
    """k_seq_states has two bodies -- tables and bitstream in LDS (sections whose blocks are all resident at once), or read
    through L2 -- chosen by size; NAFGPU_K2_LDS forces either on the same inputs: blocks of thousands of sequences (the
    bitstream ring is topped up many times), one-sequence blocks, RLE and predefined tables, corrupt streams."""
    emu.c.nafgpu_test_hooks(1)
    names = ("dna_l3_big", "text_quality", "fastq_flush_per_record", "truncated_mid", "bitflip_sequence", "checksum_text_l3")
    for force in ("1", "2", "0"):
        monkeypatch.setenv("NAFGPU_K2_LDS", force)
        for name, blob, opts in all_cases:
            if name in names:
                assert cases.run_product(blob, opts, emu) == cases.run_oracle(blob, opts), (name, force)
        for name, payload, data in cases.zstd_payload_cases(scale=1):
            if name in ("equal_length_words_l3",):
                assert emu.zstd_decompress(payload, len(data)) == data, (name, force)


def test_pointer_jumping_distance_limit_falls_back_to_frame_order(emu, all_cases, monkeypatch):
    """D holds 32-bit distances: a chain longer than that cannot be jumped to its end.  With the limit lowered
    to a few elements (NAFGPU_PJ_MAX_DIST) distances stop growing while elements still point at pending ones;
    what the sweeps then leave is finished in frame order -- and the bytes stay exact."""
    emu.c.nafgpu_test_hooks(1)
    monkeypatch.setenv("NAFGPU_LZ_MODE", "dense")
    for limit in ("8", "64"):
        monkeypatch.setenv("NAFGPU_PJ_MAX_DIST", limit)
        for name, blob, opts in all_cases:
            if name in ("text_dense_chains", "dna_dense_chains", "dna_homopolymer"):
                assert cases.run_product(blob, opts, emu) == cases.run_oracle(blob, opts), (name, limit)


def test_error_timing_against_the_streaming_reference(emu, all_cases, monkeypatch):
    """Malformed in the middle of a section: the reference has streamed k records by the time it meets the bad block;
    this library decodes a section when the first record needs it, so its error comes at record j <= k (DESIGN section 8:
    the documented deviation) and what it handed out before are the reference's first j records.  With the section cut
    into tiles the error comes with the tile that holds it: records of the tiles in front are handed out."""
    from oracle import oracle
    if not oracle.ref_shape_available():
        pytest.skip("libzstd not loadable")
    todo = {name: (blob, opts) for name, blob, opts in all_cases if name in ("truncated_mid", "bitflip_sequence", "checksum_wrong", "truncated_tail")}
    seen = {}
    for name, (blob, opts) in todo.items():
        seen[name] = cases.error_timing(blob, opts, emu)
        # (a flipped bit inside Huffman-coded literals decodes to other symbols without any error: both pipelines then agree on them)
        if name != "bitflip_sequence":
            assert seen[name][1] != 0 and seen[name][3] in ("io:invalid", "io:eof"), (name, seen[name])
    # (checksum_wrong: everything decodes, the frame's last four bytes disagree)
    assert seen["checksum_wrong"][2] == 0 and seen["checksum_wrong"][3] == "io:invalid"
    # a larger archive, corrupt three quarters in, decoded in tiles: the tiles in front of the damage hand out their records
    import naf_writer as nw
    import numpy as np
    rng = np.random.default_rng(8)
    recs = cases.make_records(rng, [20000] * 60, iupac=0.01)
    good = bytearray(nw.write_naf(recs, level=1))
    at = len(good) * 3 // 4
    good[at:at + 300] = bytes(300)                          # (a zeroed stretch: the Huffman streams it crosses no longer end where they must)
    emu.c.nafgpu_test_hooks(1)
    k, rc, j_whole, _ = cases.error_timing(bytes(good), {}, emu)
    monkeypatch.setenv("NAFGPU_TILE_KIB", "256")
    k2, rc2, j_tiled, _ = cases.error_timing(bytes(good), {}, emu, eager=False, slack=1)
    assert rc != 0 and (k, rc) == (k2, rc2) and j_whole == 0 and k - 7 <= j_tiled <= k + 1, (k, j_whole, j_tiled)


def test_block_range_sharding(emu):
    cases.check_sharding(emu, 3_000_001, True)
    cases.check_sharding(emu, 1_500_000, False, worlds=(2, 8))


def test_next_batch_equals_next(emu, all_cases, monkeypatch):
    """nafgpu_next_batch hands out exactly what as many calls of nafgpu_next do -- fixtures, every shared case (malformed
    archives included: the error comes with the call that reaches it, the records in front of it are delivered, the iterator
    goes on), and an archive whose output is held a tile at a time (batches end where the host window has to move)."""
    from conftest import golden_bytes
    for name in ("LuxC", "masked", "phix"):
        assert cases.check_next_batch(golden_bytes(name + ".naf"), lib=emu) > 0
    for name, blob, opts in all_cases:                      # (every case, malformed archives included; the *_big ones on the GPU only)
        if not name.endswith("_big"):
            cases.check_next_batch(blob, opts, lib=emu, caps=(3,))
    monkeypatch.setenv("NAFGPU_TILE_KIB", "256")
    cases.check_next_batch(golden_bytes("NZ_AAEN01000029.naf"), lib=emu, caps=(7, 4096))


def test_archive_ends_against_the_oracle(emu):
    """cases.check_archive_ends at a size the harness decodes in seconds (the GPU suite runs it on the 40-Gbase archive)."""
    cases.check_archive_ends(emu, 40_000_001, 0x4E4146, 70)


def test_shard_protocol_calls_out_of_order_are_refused(emu):
    """The protocol is begin -> place -> halo / export / import -> finish: anything else is NAFGPU_E_INVALID_ARG, never a
    decode over buffers that were not prepared."""
    import ctypes
    import io
    from nafcodec_amd import _ffi
    from nafcodec_amd.decoder import Decoder
    from nafcodec_amd.sharding import decode_sharded_local
    name, blob, want_seq, _q, _l = next(x for x in cases.lz_shard_archives(1) if x[0] == "real_genome_l1")
    decs = [Decoder(io.BytesIO(blob), shard_rank=r, shard_count=2, shard_protocol=True, _lib=emu) for r in range(2)]
    d = decs[1]
    res, n, m, ready = _ffi.DeviceResult(), ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_int()
    c = emu.c
    assert c.nafgpu_shard_finish(d._h, ctypes.byref(res)) == _ffi.E_INVALID_ARG            # nothing begun
    assert c.nafgpu_shard_halo(d._h, 0, ctypes.byref(n), ctypes.byref(m), ctypes.byref(ready)) == _ffi.E_INVALID_ARG
    everyone = b"".join(x.shard_begin() for x in decs)
    assert c.nafgpu_shard_finish(d._h, ctypes.byref(res)) == _ffi.E_INVALID_ARG            # begun, not placed
    assert c.nafgpu_shard_import_halo(d._h, 0, everyone, 0) == _ffi.E_INVALID_ARG
    # ... and the decoders are still good for a whole, ordered run
    out = decode_sharded_local(decs)
    got = b"".join(x.copy_to_host(r.d_sequence, r.n_bases) for x, r in zip(decs, out))
    assert got == want_seq
    assert c.nafgpu_shard_finish(d._h, ctypes.byref(res)) == _ffi.E_INVALID_ARG            # finished: a second finish needs a new begin
    for x in decs:
        x.close()


def test_shard_protocol_on_sections_with_lz_sequences(emu):
    """SURVEY 8e for archives as found in the wild: sections WITH LZ sequences over 2 / 3 / 8 block ranges through the
    shard protocol (nafgpu_shard_*) -- real-genome statistics, level-3 DNA in one and in three frames, FASTQ-like reads
    (Sequence and Quality both sharded), dense chains in two frames; both match routes where the archive is small."""
    cases.check_lz_sharding(emu, 1, worlds=(2, 3, 8), names=("real_genome_l1",))
    cases.check_lz_sharding(emu, 1, worlds=(3, 8), names=("random_dna_l3_frames", "text_dense_chains_frames"))
    cases.check_lz_sharding(emu, 1, worlds=(3,), names=("fastq_like_l1",))            # (all of them, larger, in the GPU test of the same name)
    cases.check_lz_sharding(emu, 1, worlds=(2,), names=("random_dna_l3_frames",), force_modes=("dense", "sparse"))


def test_synthetic_writer_roundtrip(emu):
    """nafgpu_synth_write output: valid for libzstd, decodes to the writer's own checksums."""
    import ctypes
    import io
    from nafcodec_amd.decoder import Decoder
    from oracle import oracle
    # (with a mask: no LZ sequences in these archives, so the sequence's writers apply it -- dictionary tables for the
    #  two trees of 300001 bases, compact ones when IUPAC codes bring more than 64 byte values)
    for n, mask, iupac in [(1000, False, 0), (300001, True, 0), (1500000, True, 7), (700001, True, 400)]:
        arc = emu.synth(n, seed=n, with_mask=mask, iupac_permille=iupac)
        try:
            blob = ctypes.string_at(arc.bytes, arc.n)
            d = oracle.Decoder(blob)
            data, orig, comp, off = d.section(4)
            assert zstd_ref.decompress_magicless(blob[off:off + comp], len(data) + 8) == data
            want = cases.run_oracle(blob, {})
            seq = "".join(r[2] for r in want[0]).encode()
            assert emu.c.nafgpu_hash64_host(seq, len(seq)) == arc.seq_hash
            dec = Decoder(io.BytesIO(blob), _lib=emu)
            res = dec.decode_all_device()
            assert (res.n_bases, res.n_records) == (arc.n_bases, arc.n_records)
            assert dec.hash_device(res.d_sequence, res.n_bases) == arc.seq_hash
            assert dec.hash_device(res.d_record_end, 8 * res.n_records) == arc.offsets_hash
            assert cases.run_product(blob, {}, emu) == want
        finally:
            emu.c.nafgpu_synth_free(ctypes.byref(arc))


def test_under_address_sanitizer():
    """A subset of the cases with every kernel running under ASan + UBSan."""
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not os.path.exists(asan):
        pytest.skip("libasan not available")
    subprocess.check_call(["make", "-s", "-C", CSRC, "emu-asan"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    script = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import cases
from conftest import golden_bytes
from nafcodec_amd import _ffi
lib = _ffi.Library(%r)
heavy = ("dna_skewed_blocks_dict_seg", "dna_multi_tree_compact", "text_multi_tree_dict", "checksum_dna_blocks", "checksum_wrong",
         "checksum_text_l3", "fastq_flush_per_record", "text_repeat_offsets_l1", "text_repeat_offsets_l9", "dna_repeat_offsets",
         "mask_compact_tables", "mask_raw_rle_blocks")    # (run without the sanitizer by the other tests)
part, parts = int(sys.argv[1]), int(sys.argv[2])         # (the cases are dealt over `parts` processes running side by side)
todo = [c for c in cases.build_cases(1) if not c[0].endswith("_big") and c[0] not in heavy]
todo += [(n, golden_bytes(n + ".naf"), {}) for n in ("phix", "masked", "CP040672")]
todo = todo[part::parts]
bad = [n for n, blob, opts in todo if cases.run_product(blob, opts, lib) != cases.run_oracle(blob, opts)]
bad += cases.fuzz_disagreements(cases.fuzz_cases(seed=7, n=40)[part::parts], lib)      # corrupted archives: no OOB, no silent garbage
import os
os.environ["NAFGPU_HUF_SPLIT"] = "4"                     # ... and with every Huffman stream cut into parts (plan.h: HufStream::sub)
lib.c.nafgpu_test_hooks(1)
bad += ["parts:" + n for n, blob, opts in todo[-2:] + todo[:6] if cases.run_product(blob, opts, lib) != cases.run_oracle(blob, opts)]
bad += ["parts:" + n for n in cases.fuzz_disagreements(cases.fuzz_cases(seed=7, n=40)[part::parts], lib)]
del os.environ["NAFGPU_HUF_SPLIT"]
lib.c.nafgpu_test_hooks(0)
import io
from nafcodec_amd.decoder import Decoder
bad += ["text:" + n for n, blob in cases.text_cases(1)[part::parts] if Decoder(io.BytesIO(blob), _lib=lib).to_text() != cases.oracle_text(blob)]
if part == 1:                                            # tiles of a dense-LZ section into the WHOLE output: the last tile's upper bound lies behind it
    os.environ["NAFGPU_TILE_KIB"] = "300"
    lib.c.nafgpu_test_hooks(1)
    for n, blob, opts in [c for c in cases.build_cases(1) if c[0] == "dna_l3_big"]:
        d = Decoder(io.BytesIO(blob), _lib=lib)
        res = d.decode_all_device()
        if d.copy_to_host(res.d_sequence, res.n_bases) != "".join(r[2] or "" for r in cases.run_oracle(blob, opts)[0]).encode():
            bad.append("tiles, bulk:" + n)
    del os.environ["NAFGPU_TILE_KIB"]
    lib.c.nafgpu_test_hooks(0)
print("BAD", bad)
sys.exit(1 if bad else 0)
""" % (ROOT, os.path.join(ROOT, "tests"), os.path.join(EMU_DIR, "libnafgpu_emu_asan.so"))
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:allocator_may_return_null=1")
    parts = 4
    procs = [subprocess.Popen([sys.executable, "-c", script, str(k), str(parts)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for k in range(parts)]
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, out[-2000:] + err[-4000:]


GOLDEN_TEXT = [("LuxC", "LuxC.faa"), ("masked", "masked.fna"), ("phix", "phix.fastq")]


@pytest.mark.parametrize("name,text", GOLDEN_TEXT)
def test_text_output_equals_the_reference_fixture_texts(emu, name, text):
    """nafgpu_format_device pinned on the source texts the reference's fixtures were made from."""
    from nafcodec_amd.decoder import Decoder
    want = golden_bytes(text)
    if not want.endswith(b"\n"):
        want += b"\n"                                   # masked.fna lacks the final newline
    got = Decoder(io.BytesIO(golden_bytes(name + ".naf")), **({'_lib': emu})).to_text()
    assert got == want


def test_text_output_matches_oracle_records(emu):
    from nafcodec_amd.decoder import Decoder
    for name, blob in cases.text_cases(scale=1):
        assert Decoder(io.BytesIO(blob), **({'_lib': emu})).to_text() == cases.oracle_text(blob), name
    # field selection: no comments in the names, no quality -> FASTA of a FASTQ archive
    name, blob = cases.text_cases(scale=1)[-1]
    assert Decoder(io.BytesIO(blob), comment=False, quality=False, **({'_lib': emu})).to_text() == \
        cases.oracle_text(blob, {"comment": False, "quality": False})


def test_device_string_tables_and_utf8_flags(emu):
    """CStringReader on the device (offsets past each NUL) and the UTF-8 verdict per text section."""
    import numpy as np
    import naf_writer as nw
    from nafcodec_amd.decoder import Decoder
    from oracle import oracle
    blob = golden_bytes("phix.naf")
    d = Decoder(io.BytesIO(blob), **({'_lib': emu}))
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
        assert Decoder(io.BytesIO(bad), **({'_lib': emu})).decode_all_device().utf8_invalid == 1 << bit, (section, payload)
    good = nw.write_naf(recs, raw_sections={"comments": "é\u20ac\U0001F600\x00y\x00".encode()})
    assert Decoder(io.BytesIO(good), **({'_lib': emu})).decode_all_device().utf8_invalid == 0
    # sections longer than the 16-byte chunks of k_utf8_check: multi-byte characters across chunk borders, one byte
    # overwritten at every position around them -- the verdict must be Python's
    rng = np.random.default_rng(9)
    text = "".join(rng.choice(list("abcdefghij é\u20ac\U0001F600z"), 150))
    base = b"x\x00" + text.encode() + b"\x00"
    for pos in list(range(12, 52)) + [len(base) - 2]:
        for byte in (0x80, 0xC3, 0xE2, 0xF0, 0xFF, 0x41):
            payload = bytearray(base)
            payload[pos] = byte
            try:
                bytes(payload).decode("utf-8")
                want = 0
            except UnicodeDecodeError:
                want = 2
            if payload.count(0) != 2:
                continue                                      # (keep two records)
            arc = nw.write_naf(recs, raw_sections={"comments": bytes(payload)})
            assert Decoder(io.BytesIO(arc), **({'_lib': emu})).decode_all_device().utf8_invalid == want, (pos, hex(byte))
