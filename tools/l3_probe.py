import sys, time, ctypes, io, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
import zstd_ref, naf_writer as nw
import nafcodec_amd
from nafcodec_amd import _ffi
if os.environ.get("NAFGPU_PROBE_HOOKS"): _ffi.default().c.nafgpu_test_hooks(1)   # (experiments: the NAFGPU_* switches are read)
rng = np.random.default_rng(1)
n_packed = int(float(sys.argv[1]) if len(sys.argv) > 1 else 64e6)
for level in ([int(sys.argv[2])] if len(sys.argv) > 2 else [1, 3]):
    codes = np.array([1, 2, 4, 8], dtype=np.uint8)
    packed = (codes[rng.integers(0, 4, n_packed)] | (codes[rng.integers(0, 4, n_packed)] << 4)).astype(np.uint8).tobytes()
    t = time.time(); payload = zstd_ref.compress_magicless(packed, level, True); tc = time.time() - t
    n_bases = 2 * n_packed
    lens = nw.length_words([n_bases])
    lenp = zstd_ref.compress_magicless(lens, 1, True)
    blob = bytes([1, 0xF9, 0xEC, 1, 0x0A, 0x20]) + nw.varint(60) + nw.varint(1) + nw.varint(len(lens)) + nw.varint(len(lenp)) + lenp + nw.varint(n_bases) + nw.varint(len(payload)) + payload
    lut = np.frombuffer(b"-TGKCYSBAWRDMHVN", dtype=np.uint8)
    a = np.frombuffer(packed, dtype=np.uint8)
    want = np.empty(n_bases, dtype=np.uint8); want[0::2] = lut[a & 15]; want[1::2] = lut[a >> 4]
    want_hash = _ffi.default().c.nafgpu_hash64_host(want.tobytes(), n_bases)
    # NAFGPU_PROBE_LIBS: comma-separated experiment builds timed on the same archive after the product
    for path in [None] + [x for x in os.environ.get("NAFGPU_PROBE_LIBS", "").split(",") if x]:
        lib = _ffi.default() if path is None else _ffi.Library(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), path))
        dec = nafcodec_amd.Decoder(io.BytesIO(blob), _lib=lib)
        res = dec.decode_all_device()
        if not os.environ.get("NAFGPU_PROBE_ONE_DECODE"): res = dec.decode_all_device()
        ok = dec.hash_device(res.d_sequence, res.n_bases) == want_hash
        print("level", level, "product" if path is None else path, "bases", n_bases, "compress s %.1f" % tc, "B/base %.4f" % (len(payload) / n_bases), "ok", ok,
              "ms total %.2f huf %.2f seq_lz %.2f other %.2f" % (res.ms_total, res.ms_huf, res.ms_seq_lz, res.ms_other),
              "Gbases/s %.1f" % (n_bases / res.ms_total / 1e6), flush=True)
        dec.close()
