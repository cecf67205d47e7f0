#!/usr/bin/env python3
"""bench.py -- NAF decode throughput on MI355X (BASELINE.json metric: decoded Gbases/s).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--bases B]

One "step" = one pass of the hot path over one synthetic archive already resident in HBM:
per-block Huffman literal decode of the sequence section, record-length scan, 4-bit -> IUPAC
unpack, (mask).  Workload at N=1 = BASELINE.json configs[1]: a synthetic 10 GB DNA-only
.naf (Length + Sequence sections, ~40e9 bases, zstd-level-1 shape: 128 KiB Huffman-literal
blocks).  For N>1 every rank decodes its own 10 GB shard of the N x 10 GB archive (weak
scaling, block-range sharding) and the ranks exchange {bases, packed bytes, records, carry}
with one RCCL all-gather per step to place their records in the global offset table.

Prints ONE JSON line on rank 0 (see the field list in the task contract) with two extra
objects: "roofline" for the dominant kernel (k_huf_decode) and "cpu_baseline" (the CPU oracle
timed on a bounded sample of the same workload, one host thread like the reference).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
DEFAULT_BASES = 40_000_000_000  # ~10 GB archive at ~0.25 B/base


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--bases", type=float, default=float(os.environ.get("NAF_BENCH_BASES", DEFAULT_BASES)),
                    help="nucleotides per GPU (default 40e9 = the 10 GB archive of configs[1])")
    ap.add_argument("--mask", action="store_true", help="configs[3]: add a Mask section")
    ap.add_argument("--iupac", type=int, default=0,
                    help="per-mille of non-ACGT codes (N, R, Y ...): every block then carries its own deep Huffman tree")
    ap.add_argument("--cpu-sample-bases", type=float, default=0,
                    help="size of the CPU-baseline sample (0 = auto, about 15 s of CPU work)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--real-copies", type=int, default=3000,
                    help="N=1 only: also time an archive shaped like a real genome (the reference's NZ_AAEN01000029 fixture tiled "
                         "this many times, libzstd level 1) and report it as path.real_genome; 0 skips it")
    ap.add_argument("--small-real-copies", type=int, default=565,
                    help="N=1 only: the same at the size of a human genome (x 565 = 3.1 Gbases), reported as path.real_genome_3g; 0 skips it")
    ap.add_argument("--l3-bases", type=float, default=1.024e9,
                    help="level-3 DNA leg (path.l3_dna): bases of iid ACGT written by libzstd at level 3; 0 = skip")
    ap.add_argument("--fastq-reads", type=float, default=10e6,
                    help="N=1 only: a FASTQ-shaped archive of this many 151-base reads (bulk decode + the record iterator), path.fastq_like; 0 skips it")
    ap.add_argument("--no-masked-leg", action="store_true", help="N=1 only: skip path.masked (configs[3]: the headline archive with a Mask section)")
    ap.add_argument("--no-iterator", action="store_true", help="skip path.iterator (the headline archive through nafgpu_next)")
    ap.add_argument("--real-copies-per-gpu", type=int, default=1000,
                    help="N>1: the real-genome archive (WITH LZ sequences) holds this many tiles per GPU, ONE archive decoded through the "
                         "shard protocol and reported as path.real_genome; 0 skips it")
    ap.add_argument("--sharded-leg-limit", type=int, default=180,
                    help="N>1: seconds the path.real_genome leg may take before the line is printed without it")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--rehearsal-one-gpu", action="store_true",
                    help="tests only: several ranks (gloo) share GPU 0 -- the sharded flow with the real library on a one-GPU box; "
                         "the line printed carries no metric and no value")
    ap.add_argument("--rehearsal-lib", default="",
                    help="tests only: run the control flow against another build of the library (the CPU harness); "
                         "the line printed then carries no metric and no value")
    return ap.parse_args()


def cpu_baseline(lib, n_bases_target, mask, device):
    """The CPU path timed beside the GPU one, on a bounded sample of the same workload, one thread like the
    reference (decoder/mod.rs:285-296 is single-threaded by construction):
      * the reference pipeline IN ITS OWN SHAPE (oracle/ref_shape.c: streaming libzstd -- the library the reference links --
        through 4 KiB buffers, mod.rs:223; per-nibble push, reader.rs:131-136; one heap string per field and record) when
        libzstd.so.1 can be loaded on this box; else the scalar oracle (oracle/*.c);
      * plus "all_cores": as many independent copies of that pipeline as the host has cores, each on its own copy of the
        sample -- a generous upper bound for what the host could do with one archive per core.
    `kind` stays "port": both are this repository's restatements, the Rust crate cannot be built here.
    The sample also pins parity: the CHECKER (oracle/naf_oracle.c) drains the same archive and its bases and record
    table (position-keyed checksums accumulated in C) must equal what the HIP path produces from the same bytes.
    Returns (cpu_baseline JSON object, bases checked against the oracle)."""
    from oracle import oracle
    shaped = oracle.ref_shape_available()
    drain = (lambda blob, h: oracle.ref_shape_drain(blob, want_hash=h)) if shaped else (lambda blob, h: oracle.Decoder(blob).drain(want_hash=h))
    # calibrate on a small sample, then size the real one for ~10 s of single-thread work
    probe_bases = 8_000_000
    arc = lib.synth(probe_bases, seed=0x4E4146, with_mask=mask)
    blob = ctypes.string_at(arc.bytes, arc.n)
    lib.c.nafgpu_synth_free(ctypes.byref(arc))
    t0 = time.perf_counter()
    n = drain(blob, False).n_bases
    rate = n / (time.perf_counter() - t0)
    sample = int(n_bases_target) if n_bases_target else int(min(max(rate * 4.0, probe_bases), 4e9))
    arc = lib.synth(sample, seed=0x4E4146, with_mask=mask)
    try:
        blob = ctypes.string_at(arc.bytes, arc.n)
        best = None
        for _ in range(5):
            t0 = time.perf_counter()
            r = drain(blob, False)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        n = r.n_bases
        all_cores = None
        cores = os.cpu_count() or 1
        if shaped and cores > 1:
            threads = min(cores, 64)
            t0 = time.perf_counter()
            total = oracle.ref_shape_drain_parallel(blob, threads)
            dt = time.perf_counter() - t0
            if total:
                all_cores = {"value": round(total / dt / 1e9, 3), "unit": "Gbases/s", "threads": threads,
                             "note": "%d independent copies of the one-thread pipeline, one archive each (upper bound)" % threads}
        want = oracle.Decoder(blob).drain(want_hash=True)        # the checker
        # the same archive through the HIP path
        opts = _ffi_mod().Opts()
        lib.c.nafgpu_opts_default(ctypes.byref(opts))
        opts.device = device
        h, err, res = ctypes.c_void_p(), _ffi_mod().Error(), _ffi_mod().DeviceResult()
        if lib.c.nafgpu_open_bytes(blob, len(blob), ctypes.byref(opts), ctypes.byref(h), ctypes.byref(err)) != 0:
            raise RuntimeError("open failed: %s" % err.message.decode())
        try:
            if lib.c.nafgpu_decode_all_device(h, ctypes.byref(res)) != 0:
                lib.c.nafgpu_last_error(h, ctypes.byref(err))
                raise RuntimeError("decode failed: %s" % err.message.decode())
            hs, he = ctypes.c_uint64(), ctypes.c_uint64()
            lib.c.nafgpu_hash64_device(h, res.d_sequence, res.n_bases, ctypes.byref(hs))
            lib.c.nafgpu_hash64_device(h, res.d_record_end, 8 * res.n_records, ctypes.byref(he))
            if (res.n_bases, res.n_records, hs.value, he.value) != (want.n_bases, want.n_records, want.seq_hash, want.ends_hash):
                raise RuntimeError("GPU decode of the cpu_baseline sample differs from the oracle's output")
        finally:
            lib.c.nafgpu_close(h)
    finally:
        lib.c.nafgpu_synth_free(ctypes.byref(arc))
    out = {"value": round(n / best / 1e9, 4), "unit": "Gbases/s", "cores": 1, "kind": "port",
           "sample": "%d bases (%.1f MB archive) of the same synthetic workload; %s; best of 5; host has %d cores"
                     % (n, len(blob) / 1e6,
                        "reference pipeline shape: streaming libzstd with 4 KiB buffers + per-nibble push + per-record strings (oracle/ref_shape.c)"
                        if shaped else "CPU oracle (oracle/*.c: scalar zstd + reader.rs restatement); libzstd.so.1 not loadable here",
                        cores)}
    if all_cores:
        out["all_cores"] = all_cores
    return out, want.n_bases


def _decode_bulk(lib, device, blob):
    ffi = _ffi_mod()
    opts = ffi.Opts()
    lib.c.nafgpu_opts_default(ctypes.byref(opts))
    opts.device = device
    h, err, res = ctypes.c_void_p(), ffi.Error(), ffi.DeviceResult()
    if lib.c.nafgpu_open_bytes(blob, len(blob), ctypes.byref(opts), ctypes.byref(h), ctypes.byref(err)) != 0:
        raise RuntimeError("open failed: %s" % err.message.decode())
    if lib.c.nafgpu_decode_all_device(h, ctypes.byref(res)) != 0:
        lib.c.nafgpu_last_error(h, ctypes.byref(err))
        raise RuntimeError("decode failed: %s" % err.message.decode())
    return h, res


def real_genome_period(lib, device):
    """The packed bytes (numpy uint8) of the reference's NZ_AAEN01000029 fixture, decoded by the HIP path itself and
    packed again -- one period of the real-genome workload -- and its IUPAC characters."""
    import numpy as np
    here = os.path.dirname(os.path.abspath(__file__))
    blob = open(os.path.join(here, "tests", "golden", "NZ_AAEN01000029.naf"), "rb").read()
    h, res = _decode_bulk(lib, device, blob)
    seq = ctypes.create_string_buffer(int(res.n_bases))
    lib.c.nafgpu_copy_to_host(h, res.d_sequence, int(res.n_bases), ctypes.cast(seq, ctypes.c_void_p))
    lib.c.nafgpu_close(h)
    chars = np.frombuffer(seq.raw[:int(res.n_bases) & ~1], dtype=np.uint8)
    chars = chars & 0xDF | (chars == 0x2D) * 0x2D                      # upper case ('-' stays): the tiled archive carries no mask
    code = np.zeros(256, dtype=np.uint8)
    for i, c in enumerate(b"-TGKCYSBAWRDMHVN"):
        code[c] = i
    nib = code[chars]
    one = (nib[0::2] | (nib[1::2] << 4)).astype(np.uint8)
    period = np.frombuffer(b"-TGKCYSBAWRDMHVN", dtype=np.uint8)[np.stack([one & 15, one >> 4], axis=1).reshape(-1)]
    return one, period


def real_genome_archive(one, copies):
    """-> (archive bytes, compressed sequence payload length, bases): `one` tiled `copies` times, written by the system
    libzstd at level 1 in streaming mode (what `ennaf` does): one Huffman tree per 128 KiB block, a few LZ sequences."""
    import numpy as np
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "tests"))
    import zstd_ref                                                   # ctypes binding of libzstd.so.1 (test helper, no oracle)
    packed = np.tile(one, copies).tobytes()
    n_bases = 2 * len(packed)

    def varint(v):
        out = [v & 0x7F]
        v >>= 7
        while v:
            out.append(0x80 | (v & 0x7F))
            v >>= 7
        return bytes(reversed(out))

    payload = zstd_ref.compress_magicless(packed, 1, True)
    del packed
    lens = b"".join((0xFFFFFFFF).to_bytes(4, "little") for _ in range(n_bases // 0xFFFFFFFF)) + (n_bases % 0xFFFFFFFF).to_bytes(4, "little")
    lenp = zstd_ref.compress_magicless(lens, 1, True)
    arc = (bytes([1, 0xF9, 0xEC, 1, 0x0A, 0x20]) + varint(60) + varint(1) + varint(len(lens)) + varint(len(lenp)) + lenp +
           varint(n_bases) + varint(len(payload)) + payload)
    return arc, len(payload), n_bases


def real_genome_leg(lib, device, copies):
    """Second measured workload (reported beside the headline, never as `value`): an archive with the block structure
    `ennaf` gives real genomes -- the sequence of tests/golden/NZ_AAEN01000029.naf (the reference's own fixture: 5.5 Mbases,
    IUPAC codes besides ACGT) tiled `copies` times and compressed by the system libzstd at level 1 in streaming mode: one
    Huffman tree per 128 KiB block, a few LZ sequences per block.  The fixture is decoded by the HIP path itself; the
    expected output of the tiled archive is the tiled output of the fixture (checksum compared)."""
    import numpy as np
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "tests"))
    import zstd_ref
    if not zstd_ref.available():
        return None
    one, period = real_genome_period(lib, device)
    arc, payload_len, n_bases = real_genome_archive(one, copies)
    want = np.tile(period, copies).tobytes()
    want_hash = lib.c.nafgpu_hash64_host(want, len(want))
    del want
    h, res = _decode_bulk(lib, device, arc)
    try:
        best = None
        for _ in range(4):
            if lib.c.nafgpu_decode_all_device(h, ctypes.byref(res)) != 0:
                raise RuntimeError("decode failed")
            if best is None or res.ms_total < best[0]:
                best = (res.ms_total, res.ms_huf, res.ms_seq_lz, res.ms_other)
        hs = ctypes.c_uint64()
        lib.c.nafgpu_hash64_device(h, res.d_sequence, res.n_bases, ctypes.byref(hs))
        if int(res.n_bases) != n_bases or hs.value != want_hash:
            raise RuntimeError("real-genome leg: GPU output differs from the tiled fixture")
        alg = payload_len + n_bases                                    # compressed bytes in, characters out
        gbs = alg / (best[0] * 1e-3) / 1e9
        return {"workload": "sequence of the reference fixture NZ_AAEN01000029 x %d, libzstd level 1 streaming (%d zstd blocks, one Huffman "
                            "tree each, %d Huffman streams); output checksum equals the tiled fixture's" % (copies, res.n_zstd_blocks, res.n_huf_streams),
                "bases": n_bases, "ms_per_step": round(best[0], 3), "value": round(n_bases / best[0] / 1e6, 1), "unit": "Gbases/s",
                "ms_huf": round(best[1], 3), "ms_seq_lz": round(best[2], 3), "ms_other": round(best[3], 3),
                "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_step": int(alg)}}
    finally:
        lib.c.nafgpu_close(h)


def iterator_leg(path, device, batch=0):
    """The drop-in API itself (Decoder::from_path + Iterator::next, mod.rs:304-306, 356-399): nafcodec_amd/iter_bench -- plain C++
    on the C-ABI, what a Rust / C++ shim does -- opens the archive at `path` and calls nafgpu_next until the end (`batch` > 0:
    nafgpu_next_batch, that many records a call); every record's sequence / quality comes to the host through the decoder's
    pinned window.  Seconds from open to the last record; `records_per_s_after_first` leaves out the first call (which decodes
    every section on the GPU -- of a section of 4 GiB or more its first 2 GiB tile, the later ones as the records reach them)."""
    import subprocess
    tool = os.path.join(ROOT, "nafcodec_amd", "iter_bench")
    if not os.path.exists(tool):
        return None
    p = subprocess.run([tool, path, str(device), "1", str(batch)], capture_output=True, text=True, timeout=900)
    if p.returncode != 0:
        raise RuntimeError("iter_bench failed: %s" % p.stderr[-500:])
    j = json.loads(p.stdout.strip().splitlines()[-1])
    it = max(j["iterate_s"], 1e-9)
    rest = max(j["iterate_s"] - j["first_next_s"], 1e-9)
    return {"records": j["records"], "bases": j["bases"], "open_s": round(j["open_s"], 4), "first_next_s": round(j["first_next_s"], 4),
            "iterate_s": round(j["iterate_s"], 4), "records_per_s": round(j["records"] / it), "Gbases_per_s": round(j["bases"] / it / 1e9, 3),
            "records_per_s_after_first": round(j["records"] / rest), "calls": j["calls"], "batch": batch,
            "end_to_end_Gbases_s": round(j["bases"] / max(j["total_s"], 1e-9) / 1e9, 3),
            "note": "first next() to last through %s (the first call decodes every section on the GPU, a section of 4 GiB or more tile by "
                    "2 GiB tile as the records reach them); end_to_end from nafgpu_open_path: host walk + H2D + decode + every record's bytes "
                    "D2H through the 64 MiB pinned window"
                    % ("nafgpu_next_batch, %d records a call" % batch if batch else "nafgpu_next")}


def masked_leg(lib, device, n_bases):
    """configs[3]: the headline archive WITH a Mask section (the same bases, 11 M masked runs over a sixth of them): K1 as in the
    headline, then the run table scan and k_mask_apply's second trip over the lines that hold masked bases.  Two untimed decodes,
    five timed; the output checked against the writer's checksum of the masked bases."""
    ffi = _ffi_mod()
    arc = lib.synth(n_bases, seed=0x4E4146, with_mask=True)
    h, err, res = ctypes.c_void_p(), ffi.Error(), ffi.DeviceResult()
    try:
        opts = ffi.Opts()
        lib.c.nafgpu_opts_default(ctypes.byref(opts))
        opts.device = device
        if lib.c.nafgpu_open_bytes(ctypes.cast(arc.bytes, ctypes.c_char_p), arc.n, ctypes.byref(opts), ctypes.byref(h), ctypes.byref(err)) != 0:
            raise RuntimeError("masked leg: open failed: %s" % err.message.decode())
        tot, huf, oth = [], [], []
        for k in range(7):
            if lib.c.nafgpu_decode_all_device(h, ctypes.byref(res)) != 0:
                lib.c.nafgpu_last_error(h, ctypes.byref(err))
                raise RuntimeError("masked leg: decode failed: %s" % err.message.decode())
            if k >= 2:
                tot.append(res.ms_total)
                huf.append(res.ms_huf)
                oth.append(res.ms_other + res.ms_seq_lz)
        out = ctypes.c_uint64()
        lib.c.nafgpu_hash64_device(h, res.d_sequence, res.n_bases, ctypes.byref(out))
        if out.value != arc.seq_hash or res.n_bases != arc.n_bases:
            raise RuntimeError("masked leg: decoded bases differ from the writer's checksum")
        ms = sum(tot) / len(tot)
        return {"workload": "synthetic %.1f GB DNA .naf (seq+mask+len), %d bases; masked output equals the writer's checksum" % (arc.n / 1e9, res.n_bases),
                "bases": int(res.n_bases), "ms_per_step": round(ms, 3), "value": round(res.n_bases / ms / 1e6, 1), "unit": "Gbases/s",
                "ms_huf": round(sum(huf) / len(huf), 3), "ms_mask_and_scans": round(sum(oth) / len(oth), 3)}
    finally:
        if h:
            lib.c.nafgpu_close(h)
        lib.c.nafgpu_synth_free(ctypes.byref(arc))


def fixtures_leg(device):
    """The reference's own small archives (tests/golden = /root/reference/data: configs[0] LuxC.naf, configs[2] phix.naf, and the
    5.5-Mbase NZ_AAEN01000029.naf) through the drop-in API in a warm process: open, every record, close -- the best of ten cycles in
    ms per archive (nafcodec_amd/iter_bench).  Files this small measure fixed costs, not kernels."""
    import subprocess
    tool = os.path.join(ROOT, "nafcodec_amd", "iter_bench")
    out = {}
    for name in ("LuxC.naf", "phix.naf", "NZ_AAEN01000029.naf"):
        path = os.path.join(ROOT, "tests", "golden", name)
        if not (os.path.exists(tool) and os.path.exists(path)):
            continue
        p = subprocess.run([tool, path, str(device), "11"], capture_output=True, text=True, timeout=300)
        if p.returncode != 0:
            out[name] = {"error": p.stderr[-200:]}
            continue
        j = json.loads(p.stdout.strip().splitlines()[-1])
        out[name] = {"records": j["records"], "bases": j["bases"], "ms_per_archive": round(j["best_cycle_ms"], 3),
                     "first_in_process_ms": round(j["first_cycle_ms"], 1)}
    return out or None


def l3_dna_leg(lib, device, n_bases):
    """Level-3 DNA: `n_bases` of iid ACGT written by the system libzstd at level 3 -- about half the bytes are matches at
    random distances in the window, the rest Huffman literals (what the reference's Encoder writes by default is this
    shape, encoder/mod.rs:81,137): the sequence-chain / pointer-jumping stages of the path and none of the headline's.
    Best of three decodes; output checksum against what was written."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import zstd_ref
    if not zstd_ref.available():
        return None
    rng = np.random.default_rng(1)
    n_packed = n_bases // 2
    n_bases = 2 * n_packed
    codes = np.array([1, 2, 4, 8], dtype=np.uint8)
    packed = (codes[rng.integers(0, 4, n_packed)] | (codes[rng.integers(0, 4, n_packed)] << 4)).astype(np.uint8)
    payload = zstd_ref.compress_magicless(packed.tobytes(), 3, True)

    def varint(v):
        out = [v & 0x7F]
        v >>= 7
        while v:
            out.append(0x80 | (v & 0x7F))
            v >>= 7
        return bytes(reversed(out))

    lens = n_bases.to_bytes(4, "little")
    len_frame = bytes([0x20, 4, (4 << 3) | 1, 0, 0]) + lens                    # single-segment frame, one raw block
    blob = (bytes([1, 0xF9, 0xEC, 1, 0x0A, 0x20]) + varint(60) + varint(1) + varint(4) + varint(len(len_frame)) + len_frame +
            varint(n_bases) + varint(len(payload)) + payload)
    lut = np.frombuffer(b"-TGKCYSBAWRDMHVN", dtype=np.uint8)
    want = np.empty(n_bases, dtype=np.uint8)
    want[0::2] = lut[packed & 15]
    want[1::2] = lut[packed >> 4]
    want_hash = lib.c.nafgpu_hash64_host(want.tobytes(), n_bases)
    del want, packed
    h, res = _decode_bulk(lib, device, blob)
    try:
        best = None
        for _ in range(3):
            if lib.c.nafgpu_decode_all_device(h, ctypes.byref(res)) != 0:
                raise RuntimeError("decode failed")
            if best is None or res.ms_total < best[0]:
                best = (res.ms_total, res.ms_huf, res.ms_seq_lz, res.ms_other)
        hs = ctypes.c_uint64()
        lib.c.nafgpu_hash64_device(h, res.d_sequence, res.n_bases, ctypes.byref(hs))
        if hs.value != want_hash or res.n_bases != n_bases:
            raise RuntimeError("level-3 DNA leg: decoded bases differ from what was written")
    finally:
        lib.c.nafgpu_close(h)
    alg = len(payload) + n_bases
    return {"workload": "%d bases of iid ACGT, libzstd level 3 (%.3f B/base: half the bytes are matches at random distances, every block has "
                        "thousands of LZ sequences); output checksum equals what was written" % (n_bases, len(payload) / n_bases),
            "bases": n_bases, "ms_per_step": round(best[0], 3), "value": round(n_bases / best[0] / 1e6, 1), "unit": "Gbases/s",
            "ms_huf": round(best[1], 3), "ms_seq_lz": round(best[2], 3),
            "roofline": {"bound": "hbm", "achieved": round(alg / (best[0] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(alg / (best[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_step": int(alg)}}


def fastq_like_leg(lib, device, n_reads, level=1):
    """Third workload, FASTQ-shaped (configs[2] scaled up as SURVEY 8d allows): n_reads x 151 bases, iid ACGT + iid quality
    strings from a skewed 32-entry alphabet, Length + Sequence + Quality sections written by the system libzstd at `level` --
    every section holds LZ sequences (dense routes).  Bulk decode timed on the device, then the same archive through the
    iterator.  Checked: quality and sequence checksums against what was written."""
    import numpy as np
    import shutil
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import zstd_ref
    if not zstd_ref.available():
        return None
    rng = np.random.default_rng(2)
    L = 151
    n_bases = n_reads * L
    codes = np.array([1, 2, 4, 8], dtype=np.uint8)
    nib = codes[rng.integers(0, 4, n_bases + (n_bases & 1))]
    packed = (nib[0::2] | (nib[1::2] << 4)).astype(np.uint8)
    qalpha = np.frombuffer(b"#8CGGGGGGGGGG<AFFFJJJJJJJJJJJJJJ", dtype=np.uint8)
    qual = qalpha[rng.integers(0, len(qalpha), n_bases)].tobytes()
    lens = np.full(n_reads, L, dtype="<u4").tobytes()

    def varint(v):
        out = [v & 0x7F]
        v >>= 7
        while v:
            out.append(0x80 | (v & 0x7F))
            v >>= 7
        return bytes(reversed(out))

    blob = bytearray([1, 0xF9, 0xEC, 1, 0x0B, 0x20]) + varint(L) + varint(n_reads)
    comp = 0
    for orig, data in ((len(lens), lens), (n_bases, packed.tobytes()), (len(qual), qual)):
        payload = zstd_ref.compress_magicless(data, level, True)
        comp += len(payload)
        blob += varint(orig) + varint(len(payload)) + payload
    lut = np.frombuffer(b"-TGKCYSBAWRDMHVN", dtype=np.uint8)
    want = np.empty(2 * len(packed), dtype=np.uint8)
    want[0::2] = lut[packed & 15]
    want[1::2] = lut[packed >> 4]
    want_seq = lib.c.nafgpu_hash64_host(want.tobytes()[:n_bases], n_bases)
    want_qual = lib.c.nafgpu_hash64_host(qual, len(qual))
    del want, nib, packed
    blob = bytes(blob)
    h, res = _decode_bulk(lib, device, blob)
    try:
        best = None
        for _ in range(3):
            if lib.c.nafgpu_decode_all_device(h, ctypes.byref(res)) != 0:
                raise RuntimeError("decode failed")
            if best is None or res.ms_total < best[0]:
                best = (res.ms_total, res.ms_huf, res.ms_seq_lz, res.ms_other)
        hs, hq = ctypes.c_uint64(), ctypes.c_uint64()
        lib.c.nafgpu_hash64_device(h, res.d_sequence, res.n_bases, ctypes.byref(hs))
        lib.c.nafgpu_hash64_device(h, res.d_quality, res.n_quality, ctypes.byref(hq))
        if (int(res.n_bases), int(res.n_quality), hs.value, hq.value) != (n_bases, len(qual), want_seq, want_qual):
            raise RuntimeError("FASTQ-like leg: GPU output differs from what was written")
    finally:
        lib.c.nafgpu_close(h)
    alg = comp + n_bases + len(qual) + 4 * n_reads                     # compressed in, bases + qualities + record table out
    out = {"workload": "%d reads x %d: iid ACGT + iid qualities (32-entry skewed alphabet), libzstd level %d, Length + Sequence + Quality "
                       "sections all with LZ sequences; output checksums equal what was written" % (n_reads, L, level),
           "bases": n_bases, "archive_bytes": len(blob), "ms_per_step": round(best[0], 3), "value": round(n_bases / best[0] / 1e6, 1), "unit": "Gbases/s",
           "ms_huf": round(best[1], 3), "ms_seq_lz": round(best[2], 3),
           "roofline": {"bound": "hbm", "achieved": round(alg / (best[0] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(alg / (best[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_step": int(alg)}}
    path = os.path.join("/dev/shm" if shutil.disk_usage("/dev/shm").free > 2 * len(blob) else tempfile.gettempdir(), "nafgpu_bench_fq_%d.naf" % os.getpid())
    try:
        with open(path, "wb") as f:
            f.write(blob)
        del blob
        out["iterator"] = iterator_leg(path, device)
        out["iterator_batch"] = iterator_leg(path, device, batch=4096)     # the same records through nafgpu_next_batch
    finally:
        if os.path.exists(path):
            os.unlink(path)
    return out


def real_genome_sharded_leg(lib, dist, torch, tdev, device, rank, world, copies_per_gpu, steps=3, warmup=1):
    """The real-genome workload over `world` ranks (SURVEY 8e for sections WITH LZ sequences): rank 0 writes ONE archive of
    copies_per_gpu x world tiles into shared memory, every rank maps it and decodes its block range through the shard
    protocol (nafcodec_amd.sharding.decode_sharded: one all-gather of 64 bytes, then the LZ windows point to point).
    Every rank checks its share against the tiled fixture (position-keyed checksum of exactly its range)."""
    import numpy as np
    import shutil
    import tempfile
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "tests"))
    import zstd_ref
    from nafcodec_amd.decoder import Decoder
    from nafcodec_amd.sharding import decode_sharded
    have = torch.tensor([1 if zstd_ref.available() else 0], dtype=torch.int64, device=tdev)
    dist.all_reduce(have, op=dist.ReduceOp.MIN)
    if int(have.item()) == 0:
        return None
    one, period = real_genome_period(lib, device)
    copies = copies_per_gpu * world
    path = os.path.join("/dev/shm" if shutil.disk_usage("/dev/shm").free > 3 * copies * len(one) else tempfile.gettempdir(),
                        "nafgpu_bench_real_%s.naf" % os.environ.get("MASTER_PORT", "0"))
    meta = torch.zeros(2, dtype=torch.int64, device=tdev)
    if rank == 0:
        arc, payload_len, n_bases = real_genome_archive(one, copies)
        with open(path, "wb") as f:
            f.write(arc)
        del arc
        meta = torch.tensor([payload_len, n_bases], dtype=torch.int64, device=tdev)
    dist.broadcast(meta, src=0)
    payload_len, n_bases = int(meta[0]), int(meta[1])
    dec = Decoder(path, device=device, shard_rank=rank, shard_count=world, shard_protocol=True, _lib=lib)
    dist.barrier()
    if rank == 0:
        os.unlink(path)                                     # (mapped by every rank: the pages stay)
    try:
        def sync():
            dist.barrier()
            if tdev == "cuda":
                torch.cuda.synchronize()
            lib.c.nafgpu_device_synchronize(device)
        for _ in range(warmup):
            res = decode_sharded(dec, dist, torch, tdev)
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            res = decode_sharded(dec, dist, torch, tdev)
        sync()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # this rank's share against the tiled fixture: bases [base_offset, base_offset + n) of the period repeated
        P, off, n = len(period), int(res.base_offset) % len(period), int(res.n_bases)
        want = np.tile(period, (off + n + P - 1) // P)[off:off + n].tobytes() if n else b""
        ok = res.sharded == 1 and (n == 0 or res.base_offset % 4096 == 0)
        ok = ok and dec.hash_device(res.d_sequence, n, int(res.base_offset) // 4096) == lib.c.nafgpu_hash64_host_at(want, n, int(res.base_offset) // 4096)
        del want
        tot = torch.tensor([n, 0 if ok else 1, int(res.seq_compressed_bytes)], dtype=torch.int64, device=tdev)
        dist.all_reduce(tot)
        if int(tot[1]) != 0 or int(tot[0]) != n_bases:
            raise RuntimeError("real-genome leg, rank %d: the shards do not add up to the tiled fixture" % rank)
        ms = 1e3 * elapsed / steps
        alg = payload_len + n_bases
        return {"workload": "sequence of the reference fixture NZ_AAEN01000029 x %d (%d per GPU), libzstd level 1 streaming, ONE archive, %d block ranges "
                            "through the shard protocol (all-gather of 64 B + LZ windows point to point); every rank's share equals the tiled fixture's"
                            % (copies, copies_per_gpu, world),
                "n_gpus": world, "bases": n_bases, "steps": steps, "ms_per_step": round(ms, 3), "value": round(n_bases / ms / 1e6, 1), "unit": "Gbases/s",
                "algorithmic_GBps": round(alg / (ms * 1e-3) / 1e9, 1), "compressed_bytes_read_by_all": int(tot[2])}
    finally:
        dec.close()


def _ffi_mod():
    from nafcodec_amd import _ffi
    return _ffi


def committed_traffic(n_bases):
    """(bytes, file): HBM-side bytes per k_huf_decode launch from the committed rocprofv3 PMC passes of this same
    command (profiles/*_pmc_summary.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs, values in bytes;
    see the note in that file about the gfx950 FETCH_SIZE correction), and the file they were read from -- the bench
    run itself collects no counters (they need the profiler).  (None, None) for other workloads."""
    best = (None, None)
    pdir = os.path.join(ROOT, "profiles")
    if not os.path.isdir(pdir):
        return best
    for name in sorted(os.listdir(pdir)):
        if name.endswith("_pmc_summary.json"):
            try:
                with open(os.path.join(pdir, name)) as f:
                    j = json.load(f)
            except (OSError, ValueError):
                continue
            # the summary names the workload it was taken on (older ones: in the command text, the default size)
            same = j.get("n_bases") == n_bases or ("n_bases" not in j and "40e9 bases" in j.get("command", "") and n_bases == DEFAULT_BASES)
            if same:
                best = (int(j["FETCH_SIZE_bytes"] + j["WRITE_SIZE_bytes"]), "profiles/" + name)
    return best


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world

    dist = torch = None
    if world > 1:
        # PyTorch first: it brings its own copy of the HIP runtime, and a process that loads libnafgpu.so (linked against
        # the system's) before it ends up with two runtimes, the second of which finds no device ("no ROCm-capable device is
        # detected" -- seen with three ranks sharing one GPU, tests/test_bench_multirank.py).  Loaded first, torch's copy
        # serves both.
        import torch
        import torch.distributed as dist
    from nafcodec_amd import _ffi
    # raises if libnafgpu.so is missing: no CPU fallback.  Nothing in the environment may redirect the
    # measured path: the product library is the in-tree nafcodec_amd/libnafgpu.so, without debug switches.
    for var in ("NAFGPU_LIB", "NAFGPU_PJ_MAX_DIST", "NAFGPU_LZ_MODE", "NAFGPU_TILE_KIB", "NAFGPU_K2_LANES", "NAFGPU_PROBE_LIBS",
                "NAFGPU_K2_LDS", "NAFGPU_PJ_STRIPS", "NAFGPU_ALLOC_PLAIN", "NAFGPU_VMM_CHUNK_MIB", "NAFGPU_TASK_LANES", "NAFGPU_DICT_SLOTS",
                "NAFGPU_HUF_SPLIT", "NAFGPU_PJ_HOPS", "NAFGPU_DEBUG_PLAN", "NAFGPU_PROBE_ONE_DECODE"):
        if os.environ.get(var):
            raise SystemExit("bench.py: %s is set; refusing to print a headline from a redirected or ablated build" % var)
    lib = _ffi.Library(args.rehearsal_lib) if args.rehearsal_lib else _ffi.default()
    tdev = "cpu"
    if world > 1:
        if torch.cuda.is_available() and not args.rehearsal_one_gpu:
            torch.cuda.set_device(local_rank)
            tdev = "cuda"
            dist.init_process_group("nccl", rank=rank, world_size=world)   # "nccl" is RCCL on ROCm
        else:       # CPU rehearsal of the multi-rank control flow (tests/test_bench_multirank.py)
            dist.init_process_group("gloo", rank=rank, world_size=world)
    device = local_rank if tdev == "cuda" or world == 1 else 0

    n_bases = int(args.bases)                  # per GPU
    opts = _ffi.Opts()
    lib.c.nafgpu_opts_default(ctypes.byref(opts))
    opts.device = device
    h, err = ctypes.c_void_p(), _ffi.Error()
    shared_path = None
    t0 = time.perf_counter()
    if world == 1:
        arc = lib.synth(n_bases, seed=0x4E4146, with_mask=args.mask, iupac_permille=args.iupac)
        t_gen = time.perf_counter() - t0
        expect_seq_hash, archive_bytes = arc.seq_hash, arc.n
        rc = lib.c.nafgpu_open_bytes(ctypes.cast(arc.bytes, ctypes.c_char_p), arc.n, ctypes.byref(opts),
                                     ctypes.byref(h), ctypes.byref(err))
    else:
        # configs[4]: ONE archive of world x n_bases, block-sharded.  The ranks write it together -- each its share of
        # the sequence section's zstd blocks (deterministic per block), into one file in /dev/shm -- and then each
        # opens that file (mapped, nothing copied on the host) with shard_rank / shard_count: it walks the whole
        # block directory but uploads only the compressed bytes of its own block range.
        total_bases = n_bases * world
        arc = lib.synth(total_bases, seed=0x4E4146, with_mask=args.mask, iupac_permille=args.iupac, part_rank=rank, part_count=world)
        sizes = torch.zeros(world, dtype=torch.int64, device=tdev)
        mine = torch.tensor([arc.n], dtype=torch.int64, device=tdev)
        dist.all_gather_into_tensor(sizes, mine)
        sizes = [int(x) for x in sizes.cpu()]
        head = lib.synth_head(total_bases, sum(sizes), seed=0x4E4146, with_mask=args.mask, iupac_permille=args.iupac)
        archive_bytes = head.n + sum(sizes)
        # /dev/shm when it has room for the archive (rank 0 looks, everybody follows), else the temporary directory
        import shutil
        import tempfile
        where = torch.tensor([1 if rank == 0 and shutil.disk_usage("/dev/shm").free > 1.05 * archive_bytes else 0],
                             dtype=torch.int64, device=tdev)
        dist.broadcast(where, src=0)
        shared_path = os.path.join("/dev/shm" if int(where.item()) else tempfile.gettempdir(),
                                   "nafgpu_bench_%s.naf" % os.environ.get("MASTER_PORT", "0"))
        if rank == 0:
            with open(shared_path, "wb") as f:
                f.truncate(archive_bytes)
                f.write((ctypes.c_char * head.n).from_address(head.bytes))
        dist.barrier()
        with open(shared_path, "r+b") as f:
            f.seek(head.n + sum(sizes[:rank]))
            f.write((ctypes.c_char * arc.n).from_address(arc.bytes))
        expect_seq_hash = arc.seq_hash               # of this rank's share of the blocks; the shares add up
        lib.c.nafgpu_synth_free(ctypes.byref(head))
        n_rec_all, hash_off_all = arc.n_records, arc.offsets_hash
        lib.c.nafgpu_synth_free(ctypes.byref(arc))   # the part is in the file now
        dist.barrier()
        t_gen = time.perf_counter() - t0
        opts.shard_rank, opts.shard_count = rank, world
        rc = lib.c.nafgpu_open_path(shared_path.encode(), ctypes.byref(opts), ctypes.byref(h), ctypes.byref(err))
        # every rank has the file mapped now (or has failed): the name can go -- the pages stay for as long as the
        # mappings do, and nothing is left behind in /dev/shm (which is memory) if a later step raises
        flag = torch.tensor([1 if rc == 0 else 0], dtype=torch.int64, device=tdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if rank == 0 and os.path.exists(shared_path):
            os.unlink(shared_path)
        if int(flag.item()) == 0 and rc == 0:
            raise RuntimeError("another rank could not open the shared archive")
    if rc != 0:
        raise RuntimeError("open failed: %s" % err.message.decode())
    t0 = time.perf_counter()
    rc = lib.c.nafgpu_upload(h)                # host walk + H2D: this rank's compressed bytes resident in HBM
    if rc != 0:
        lib.c.nafgpu_last_error(h, ctypes.byref(err))
        raise RuntimeError("upload failed: %s" % err.message.decode())
    t_upload = time.perf_counter() - t0

    res = _ffi.DeviceResult()
    scratch = placement = None
    if world > 1:
        from nafcodec_amd.sharding import gather_placement
        scratch = (torch.zeros(4, dtype=torch.int64, device=tdev), torch.zeros(4 * world, dtype=torch.int64, device=tdev))

    def step():
        rc = lib.c.nafgpu_decode_all_device(h, ctypes.byref(res))
        if rc != 0:
            lib.c.nafgpu_last_error(h, ctypes.byref(err))
            raise RuntimeError("decode failed: %s" % err.message.decode())
        if world > 1:
            # the one exchange step of the sharded path (RCCL all-gather of 32 B per rank):
            # per-rank counts -> global base / record offsets of this shard
            nonlocal placement
            placement = gather_placement(dist, torch, res.n_bases, res.packed_bytes, res.first_record, res.carry,
                                         res.n_records, tdev, scratch)

    def sync():
        if world > 1:
            dist.barrier()
            if tdev == "cuda":
                torch.cuda.synchronize()
        lib.c.nafgpu_device_synchronize(device)

    for _ in range(args.warmup):
        step()
    sync()
    huf_ms, unpack_ms, other_ms, total_ms = [], [], [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        huf_ms.append(res.ms_huf)
        unpack_ms.append(res.ms_unpack)
        other_ms.append(res.ms_other + res.ms_seq_lz)
        total_ms.append(res.ms_total)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        total_bases = placement.total_bases
    else:
        total_bases = int(res.n_bases)

    ok = True
    if not args.no_verify:                          # bit-exactness at full size: position-keyed checksums (hash64.h)
        out = ctypes.c_uint64()
        if world == 1:
            lib.c.nafgpu_hash64_device(h, res.d_sequence, res.n_bases, ctypes.byref(out))
            ok = out.value == expect_seq_hash and res.n_bases == arc.n_bases and res.n_records == arc.n_records
            lib.c.nafgpu_hash64_device(h, res.d_record_end, 8 * res.n_records, ctypes.byref(out))
            ok = ok and out.value == arc.offsets_hash
        else:
            # every shard starts on a 4 KiB chunk of the base stream: the shards' checksums add up to the archive's, and so
            # do the checksums the writers computed for their shares of the blocks (two 32-bit halves: no 64-bit overflow)
            ok = res.sharded == 1 and res.base_offset % 4096 == 0 and res.base_offset == placement.base_offset
            lib.c.nafgpu_hash64_device_at(h, res.d_sequence, res.n_bases, res.base_offset // 4096, ctypes.byref(out))
            halves = torch.tensor([out.value & 0xFFFFFFFF, out.value >> 32, expect_seq_hash & 0xFFFFFFFF, expect_seq_hash >> 32,
                                   0 if ok else 1], dtype=torch.int64, device=tdev)
            dist.all_reduce(halves)
            got = (int(halves[0]) + (int(halves[1]) << 32)) & ((1 << 64) - 1)
            want = (int(halves[2]) + (int(halves[3]) << 32)) & ((1 << 64) - 1)
            ok = int(halves[4]) == 0 and got == want and placement.total_bases == n_bases * world and res.n_records == n_rec_all
            lib.c.nafgpu_hash64_device(h, res.d_record_end, 8 * res.n_records, ctypes.byref(out))   # every rank holds the record table
            ok = ok and out.value == hash_off_all
        if not ok:
            raise RuntimeError("rank %d: decoded bases / offsets differ from the writer's checksums" % rank)

    line = None
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = total_bases * args.steps / elapsed / 1e9
        k1 = sum(huf_ms) / len(huf_ms) / max(res.n_huf_launches, 1)      # ms per k_huf_decode launch
        # algorithmic bytes of ONE k_huf_decode launch (fused form): compressed sequence bytes in +
        # one ASCII byte per base out (the 4-bit intermediate never reaches HBM)
        k1_bytes = res.seq_compressed_bytes + 2 * res.packed_bytes
        achieved = k1_bytes / (k1 * 1e-3) / 1e9 if k1 > 0 else 0.0
        # whole path, SURVEY 8(d): compressed in + 1 ASCII byte per base out + 4 B per record
        path_bytes = res.seq_compressed_bytes + res.n_bases + 4 * res.n_records
        dev_ms = sum(total_ms) / len(total_ms)
        line = {
            "metric": "decoded Gbases/s (+ GB/s off HBM) at 1/2/4/8 MI355X vs CPU ref",
            "value": round(value, 3), "unit": "Gbases/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "synthetic %.1f GB DNA-only .naf (seq+len%s%s), %d bases, %d records, "
                                   "%d zstd blocks / %d Huffman streams in all, zstd-level-1 shape (L1), full-size check against the WRITER's checksums of bases and record table %s"
                                   % (archive_bytes / 1e9, "+mask" if args.mask else "", ", %d permille IUPAC" % args.iupac if args.iupac else "",
                                      total_bases, res.n_records,
                                      res.n_zstd_blocks, res.n_huf_streams,
                                      "passed" if (ok and not args.no_verify) else "skipped"),
                       "sharding": ("ONE archive, %d contiguous zstd-block ranges (one per GPU, %.1f GB of compressed bytes uploaded per GPU), "
                                    "one all-gather of {bases, packed bytes, first record, carry} per step" % (world, res.seq_compressed_bytes / 1e9))
                                   if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": "k_huf_decode", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": committed_traffic(n_bases)[0], "traffic_source": committed_traffic(n_bases)[1],
                         "ms_per_launch": round(k1, 3),
                         "algorithmic_bytes_per_launch": int(k1_bytes)},
            "path": {"device_ms_per_step": round(dev_ms, 3),
                     "algorithmic_GBps": round(path_bytes / (dev_ms * 1e-3) / 1e9, 1) if dev_ms else None,
                     "frac_of_hbm_peak": round(path_bytes / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if dev_ms else None,
                     "ms_huf": round(sum(huf_ms) / len(huf_ms), 3), "ms_unpack": round(sum(unpack_ms) / len(unpack_ms), 3),
                     "ms_other": round(sum(other_ms) / len(other_ms), 3),
                     "host_plan_ms": round(res.ms_host_plan, 1), "h2d_ms": round(res.ms_h2d, 1),
                     "synth_s": round(t_gen, 1), "upload_s": round(t_upload, 2)},
        }
        # first decode of a fresh archive, everything included: host walk + H2D (PCIe) + the device decode
        e2e_s = (res.ms_host_plan + res.ms_h2d + dev_ms) * 1e-3
        line["path"]["end_to_end_Gbases_s"] = round(total_bases / world / e2e_s / 1e9, 1) if e2e_s > 0 else None
        if not args.mask and not args.no_masked_leg and not args.iupac and world == 1 and not args.rehearsal_lib:
            line["path"]["masked"] = masked_leg(lib, device, n_bases)
        if args.real_copies and world == 1 and not args.rehearsal_lib:
            line["path"]["real_genome"] = real_genome_leg(lib, device, args.real_copies)
        if args.small_real_copies and world == 1 and not args.rehearsal_lib:
            line["path"]["real_genome_3g"] = real_genome_leg(lib, device, args.small_real_copies)
        if args.fastq_reads and world == 1 and not args.rehearsal_lib:
            line["path"]["fastq_like"] = fastq_like_leg(lib, device, int(args.fastq_reads))
        if args.l3_bases and world == 1 and not args.rehearsal_lib:
            line["path"]["l3_dna"] = l3_dna_leg(lib, device, int(args.l3_bases))
        if not args.no_iterator and world == 1 and not args.rehearsal_lib:
            line["path"]["fixtures"] = fixtures_leg(device)
            # the headline archive through the drop-in API: Decoder::from_path + Iterator::next (a file, as the reference reads one)
            import shutil
            import tempfile
            ipath = os.path.join("/dev/shm" if shutil.disk_usage("/dev/shm").free > 2 * arc.n else tempfile.gettempdir(), "nafgpu_bench_%d.naf" % os.getpid())
            try:
                with open(ipath, "wb") as f:
                    f.write((ctypes.c_char * arc.n).from_address(arc.bytes))
                line["path"]["iterator"] = iterator_leg(ipath, device)
            finally:
                if os.path.exists(ipath):
                    os.unlink(ipath)
        if not args.no_cpu and world == 1:       # reported baseline, rank 0 at N=1 only
            line["cpu_baseline"], checked = cpu_baseline(lib, args.cpu_sample_bases, args.mask, device)
            line["config"]["oracle_checked_bases"] = checked
            line["config"]["workload"] += "; GPU decode of a %d-base archive of the same generator equals the CPU oracle's output" % checked
        if args.rehearsal_one_gpu:
            line.update({"metric": "REHEARSAL: %d ranks on one GPU -- not a measurement" % world, "value": None, "roofline": None})
        if args.rehearsal_lib:                   # control-flow rehearsal on the CPU harness: never a measurement
            line.update({"metric": "REHEARSAL on %s -- not a measurement" % os.path.basename(args.rehearsal_lib),
                         "value": None, "roofline": None})

    def cleanup_shared():
        if rank == 0 and world > 1 and shared_path and os.path.exists(shared_path):
            os.unlink(shared_path)

    if world > 1:
        # what ran, in the run's own words: the backend and the world size the process group reports, every rank's device
        names = [None] * world
        dist.all_gather_object(names, "rank %d: %s" % (rank, (torch.cuda.get_device_name(local_rank) + " cuda:%d" % local_rank) if tdev == "cuda"
                                                       else "%s device %d (exchange on the CPU)" % (lib.device_info(device)[0] if not args.rehearsal_lib else "CPU harness", device)))
        if rank == 0:
            line["backend"] = "%s%s" % (dist.get_backend(), " (RCCL over xGMI)" if tdev == "cuda" else "")
            line["rccl_ranks" if tdev == "cuda" else "ranks"] = dist.get_world_size()
            line["devices"] = names
    if world > 1 and args.real_copies_per_gpu:
        # sections WITH LZ sequences over the same ranks (every rank takes part).  The headline above is already measured: a
        # rank that fails or stalls in this leg must not take the line with it, so the leg runs under a per-rank time limit,
        # after which rank 0 prints the line with the failure named in it and every rank leaves.
        import threading

        from nafcodec_amd import sharding

        def leave(reason):
            # the headline is printed (it was measured), and the process then ends with a status that says the run did
            # NOT complete: a stall or a failure of the N-GPU leg must never reach the driver as rc 0
            where = "rank %d at step %r, section %r" % (rank, sharding.PROGRESS["step"], sharding.PROGRESS["section"])
            sys.stderr.write("bench.py: sharded real-genome leg: %s (%s)\n" % (reason, where))
            sys.stderr.flush()
            if rank == 0:
                line["path"]["real_genome"] = {"error": reason, "where": where}
                print(json.dumps(line), flush=True)
            cleanup_shared()
            os._exit(3)

        limit = threading.Timer(args.sharded_leg_limit, leave, ["sharded real-genome leg not finished after %d s" % args.sharded_leg_limit])
        limit.daemon = True
        limit.start()
        try:
            real_sharded = real_genome_sharded_leg(lib, dist, torch, tdev, device, rank, world, args.real_copies_per_gpu)
            dist.barrier()
        except Exception as e:                         # the other ranks leave when their own limit runs out
            limit.cancel()
            leave("rank %d: %s: %s" % (rank, type(e).__name__, e))
        limit.cancel()
        if rank == 0:
            line["path"]["real_genome"] = real_sharded
    if rank == 0:
        print(json.dumps(line), flush=True)

    lib.c.nafgpu_close(h)
    if world == 1:
        lib.c.nafgpu_synth_free(ctypes.byref(arc))
    if world > 1:
        dist.barrier()
        cleanup_shared()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
