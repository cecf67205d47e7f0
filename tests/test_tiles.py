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
