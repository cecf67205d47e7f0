"""Tiled decode (the reference streams any size through 4 KiB buffers, mod.rs:223,356-399): with a
forced small tile budget (NAFGPU_TILE_KIB) the sequence and quality sections are decoded in block-range tiles --
compressed bytes, task lists and scratch of ONE tile resident at a time; on the iterator path (nafgpu_next) the
output too, with the LZ window and the repeat offsets carried from tile to tile.  Both paths must stay bit-exact."""
import io
import os
import subprocess

import pytest

import cases
import zstd_ref
from conftest import ROOT, enable_hooks, golden_bytes

EMU_DIR = os.path.join(ROOT, "tests", "emu", "_build")
CSRC = os.path.join(ROOT, "nafcodec_amd", "csrc")
NAMES = ("dna_l3_big", "dna_dense_chains", "text_dense_chains", "dna_skewed_blocks_dict_seg", "text_quality", "dna_homopolymer",
         "fastq_flush_per_record", "mask_run_gt_65535", "dna_repeats_l1", "protein_l3", "dna_multi_tree_compact",
         "checksum_dna_blocks", "checksum_text_l3", "checksum_wrong")


def check(lib, scale, tile_kib, monkeypatch, names=NAMES, fixtures=("NZ_AAEN01000029", "phix")):
    from nafcodec_amd.decoder import Decoder
    kw = {} if lib is None else {"_lib": lib}
    todo = [c for c in cases.build_cases(scale) if c[0] in names]
    todo += [(n, golden_bytes(n + ".naf"), {}) for n in fixtures]
    want = {name: cases.run_oracle(blob, opts) for name, blob, opts in todo}
    enable_hooks(lib)
    monkeypatch.setenv("NAFGPU_TILE_KIB", str(tile_kib))
    monkeypatch.setenv("NAFGPU_WINDOW_KIB", "64")         # several read-back windows per tile, each sent ahead of the one being handed out (api.cpp: HostWindow)
    monkeypatch.setenv("NAFGPU_SRC_PREFETCH_MIN", "1024")  # tiles this small send their compressed bytes ahead too (engine.cpp: start_source_upload)
    for name, blob, opts in todo:
        # the record iterator: output held a tile at a time
        assert cases.run_product(blob, opts, lib) == want[name], (name, "iterator")
        # the bulk path: tiles of compressed bytes and scratch, whole output
        d = Decoder(io.BytesIO(blob), **kw)
        if want[name][1] is not None:                      # (a malformed one: the bulk path reports it too)
            with pytest.raises(OSError):
                d.decode_all_device()
            continue
        res = d.decode_all_device()
        got = d.copy_to_host(res.d_sequence, res.n_bases)
        exp = "".join(r[2] or "" for r in want[name][0]).encode()
        assert got[:len(exp)] == exp, (name, "bulk")
        # every call re-runs the kernels (bench.py's warm-up and timed steps): the tile state starts over
        res = d.decode_all_device()
        assert d.copy_to_host(res.d_sequence, res.n_bases)[:len(exp)] == exp, (name, "bulk, second call")
        if res.n_quality:
            assert d.copy_to_host(res.d_quality, res.n_quality)[:len(exp)] == "".join(r[3] or "" for r in want[name][0]).encode(), name
        # and the text formatted from it (needs the whole output: re-prepared if the iterator ran first)
        assert d.to_text() == cases.oracle_text(blob, opts), (name, "text")


@pytest.mark.skipif(not zstd_ref.available(), reason="libzstd not loadable (cases are written with it)")
def test_tiles_on_the_cpu_harness(monkeypatch):
    subprocess.check_call(["make", "-s", "-C", CSRC, "emu"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    from nafcodec_amd import _ffi
    # (a subset: the CPU harness runs one fibre per work-item)
    check(_ffi.Library(os.path.join(EMU_DIR, "libnafgpu_emu.so")), 1, 300, monkeypatch,
          names=("dna_l3_big", "text_dense_chains", "dna_skewed_blocks_dict_seg", "mask_run_gt_65535",
                 "checksum_dna_blocks", "checksum_wrong"), fixtures=("phix",))
    # block ranges of one archive, each decoded in tiles (decode_all_device per range, and the shard protocol)
    cases.check_sharding(_ffi.Library(os.path.join(EMU_DIR, "libnafgpu_emu.so")), 2_000_001, True, worlds=(2,))


@pytest.mark.gpu
def test_tiles_on_the_gpu(monkeypatch):
    check(None, 4, 700, monkeypatch)


@pytest.mark.gpu
def test_tiled_synthetic_archive_checksums(monkeypatch):
    """A 400 Mbase synthetic archive decoded in 64 MiB tiles: checksums of the bulk output against the writer's, the
    iterator's records against the bulk output."""
    import ctypes
    from nafcodec_amd import _ffi
    from nafcodec_amd.decoder import Decoder
    lib = _ffi.default()
    enable_hooks(lib)
    monkeypatch.setenv("NAFGPU_TILE_KIB", str(64 << 10))
    arc = lib.synth(400_000_003, seed=5, with_mask=True, iupac_permille=2)
    try:
        blob = ctypes.string_at(arc.bytes, arc.n)
        d = Decoder(io.BytesIO(blob))
        res = d.decode_all_device()
        assert (res.n_bases, res.n_records) == (arc.n_bases, arc.n_records)
        assert d.hash_device(res.d_sequence, res.n_bases) == arc.seq_hash
        whole = d.copy_to_host(res.d_sequence, res.n_bases)
        pos = 0
        for r in Decoder(io.BytesIO(blob)):
            s = r.sequence.encode()
            assert whole[pos:pos + len(s)] == s
            pos += len(s)
        assert pos == arc.n_bases
    finally:
        lib.c.nafgpu_synth_free(ctypes.byref(arc))


@pytest.mark.gpu
def test_the_iterator_takes_a_large_section_tile_by_tile():
    """No option and no hook: from 4 GiB of decoded sequence on, the record iterator holds the output a tile at a time and sends
    the compressed bytes of the next tile ahead (api.cpp: tile_blocks_for, engine.cpp: start_source_upload).  The records of a
    4.4-Gbase archive, laid end to end, must hash to what the writer says."""
    import ctypes
    import numpy as np
    from nafcodec_amd import _ffi
    from nafcodec_amd.decoder import Decoder
    lib = _ffi.default()
    n = 4_400_000_123
    arc = lib.synth(n, seed=11)
    path = "/dev/shm/nafgpu_iter_tiles_%d.naf" % os.getpid()
    try:
        with open(path, "wb") as f:
            f.write((ctypes.c_char * arc.n).from_address(arc.bytes))
        whole = np.empty(n, dtype=np.uint8)
        pos = n_rec = 0
        with Decoder(path) as d:
            while True:
                batch = d.read_batch(256)
                if not batch:
                    break
                for r in batch:
                    s = r.sequence.encode()
                    whole[pos:pos + len(s)] = np.frombuffer(s, dtype=np.uint8)
                    pos += len(s)
                n_rec += len(batch)
        assert (pos, n_rec) == (arc.n_bases, arc.n_records)
        assert lib.c.nafgpu_hash64_host(whole.ctypes.data_as(ctypes.c_char_p), n) == arc.seq_hash
    finally:
        lib.c.nafgpu_synth_free(ctypes.byref(arc))
        if os.path.exists(path):
            os.unlink(path)


@pytest.mark.gpu
def test_switching_between_the_iterator_and_the_whole_output_while_bytes_travel_ahead():
    """One decoder over a 4.4-Gbase archive: a few records through the iterator (tiles, the next tile's compressed bytes and the
    next window on their way), then the whole output at once (the device side is prepared again: what travels is waited for
    first), then the iterator again from where it stood, then close with a tile still travelling."""
    import ctypes
    import numpy as np
    from nafcodec_amd import _ffi
    from nafcodec_amd.decoder import Decoder
    lib = _ffi.default()
    n = 4_400_000_123
    arc = lib.synth(n, seed=12)
    path = "/dev/shm/nafgpu_iter_switch_%d.naf" % os.getpid()
    try:
        with open(path, "wb") as f:
            f.write((ctypes.c_char * arc.n).from_address(arc.bytes))
        d = Decoder(path)
        first = d.read_batch(40)
        assert len(first) == 40
        res = d.decode_all_device()
        assert (res.n_bases, res.n_records) == (arc.n_bases, arc.n_records)
        assert d.hash_device(res.d_sequence, res.n_bases) == arc.seq_hash
        pos = sum(len(r.sequence) for r in first)
        assert d.copy_to_host(res.d_sequence, pos) == "".join(r.sequence for r in first).encode()
        more = d.read_batch(40)                            # the iterator goes on where it stood
        assert len(more) > 0
        m = sum(len(r.sequence) for r in more)
        assert d.copy_to_host(res.d_sequence + pos, m) == "".join(r.sequence for r in more).encode()
        d.close()
        d = Decoder(path)                                  # ... and a decoder closed while the second tile's bytes are on their way
        assert d.read().sequence == first[0].sequence
        d.close()
    finally:
        lib.c.nafgpu_synth_free(ctypes.byref(arc))
        if os.path.exists(path):
            os.unlink(path)
