import sys, time, io, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import zstd_ref, naf_writer as nw
import nafcodec_amd
from nafcodec_amd import _ffi
if os.environ.get("NAFGPU_PROBE_HOOKS"): _ffi.default().c.nafgpu_test_hooks(1)   # (experiments: the NAFGPU_* switches are read)
rng = np.random.default_rng(2)
n_reads = int(float(sys.argv[1]) if len(sys.argv) > 1 else 2e6)
L = 151
n_bases = n_reads * L
codes = np.array([1, 2, 4, 8], dtype=np.uint8)
nib = codes[rng.integers(0, 4, n_bases + (n_bases & 1))]
packed = (nib[0::2] | (nib[1::2] << 4)).astype(np.uint8).tobytes()
qalpha = np.frombuffer(b"#8CGGGGGGGGGG<AFFFJJJJJJJJJJJJJJ", dtype=np.uint8)
qual = qalpha[rng.integers(0, len(qalpha), n_bases)].tobytes()
lens = np.full(n_reads, L, dtype="<u4").tobytes()
for level in ([int(sys.argv[2])] if len(sys.argv) > 2 else [1, 3]):
    t = time.time()
    secs = [(0x08, len(lens), zstd_ref.compress_magicless(lens, level, True)),
            (0x02, n_bases, zstd_ref.compress_magicless(packed, level, True)),
            (0x01, len(qual), zstd_ref.compress_magicless(qual, level, True))]
    tc = time.time() - t
    blob = bytearray([1, 0xF9, 0xEC, 1, 0x0B, 0x20]) + nw.varint(L) + nw.varint(n_reads)
    for _, orig, payload in secs:
        blob += nw.varint(orig) + nw.varint(len(payload)) + payload
    # NAFGPU_PROBE_LIBS: comma-separated experiment builds timed on the same archive after the product
    for path in [None] + [x for x in os.environ.get("NAFGPU_PROBE_LIBS", "").split(",") if x]:
        lib = _ffi.default() if path is None else _ffi.Library(os.path.join(R, path))
        dec = nafcodec_amd.Decoder(io.BytesIO(bytes(blob)), _lib=lib)
        res = dec.decode_all_device()
        if not os.environ.get("NAFGPU_PROBE_ONE_DECODE"): res = dec.decode_all_device()
        okq = dec.hash_device(res.d_quality, res.n_quality) == lib.c.nafgpu_hash64_host(qual, len(qual))
        print("level", level, "product" if path is None else path, "reads", n_reads, "bases", n_bases, "archive MB %.1f" % (len(blob) / 1e6),
              "compress s %.1f" % tc, "qual ok", okq, "records", res.n_records,
              "ms total %.2f huf %.2f seq_lz %.2f other %.2f" % (res.ms_total, res.ms_huf, res.ms_seq_lz, res.ms_other),
              "Gbases/s %.1f" % (n_bases / res.ms_total / 1e6), flush=True)
        dec.close()
