"""tools/real_probe.py [copies] -- an archive with the symbol statistics of a real genome: the sequence of
tests/golden/NZ_AAEN01000029.naf (5.5 Mbases, IUPAC codes K R W Y besides ACGT), tiled `copies` times and
compressed by libzstd level 1 in streaming mode (what ennaf does): one Huffman tree per 128 KiB block, a
handful of LZ sequences.  Decodes it on the GPU, checks the checksum against the host expansion."""
import io, os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import zstd_ref, naf_writer as nw
import nafcodec_amd
from nafcodec_amd import _ffi
from oracle import oracle
if os.environ.get("NAFGPU_PROBE_HOOKS"): _ffi.default().c.nafgpu_test_hooks(1)   # (experiments: the NAFGPU_* switches are read)

copies = int(sys.argv[1]) if len(sys.argv) > 1 else 400
blob = open(os.path.join(R, "tests", "golden", "NZ_AAEN01000029.naf"), "rb").read()
seq = "".join(r.sequence.upper() for r in oracle.Decoder(blob)).encode()
lut = b"-TGKCYSBAWRDMHVN"
code = np.zeros(256, dtype=np.uint8)
for i, c in enumerate(lut):
    code[c] = i
nib = code[np.frombuffer(seq, dtype=np.uint8)]
if len(nib) & 1:
    nib = nib[:-1]
one = (nib[0::2] | (nib[1::2] << 4)).astype(np.uint8)
packed = np.tile(one, copies).tobytes()
n_bases = 2 * len(packed)
for level in (1,):
    t = time.time(); payload = zstd_ref.compress_magicless(packed, level, True); tc = time.time() - t
    lens = nw.length_words([n_bases])
    lenp = zstd_ref.compress_magicless(lens, 1, True)
    arc = bytes([1, 0xF9, 0xEC, 1, 0x0A, 0x20]) + nw.varint(60) + nw.varint(1) + nw.varint(len(lens)) + nw.varint(len(lenp)) + lenp \
        + nw.varint(n_bases) + nw.varint(len(payload)) + payload
    a = np.frombuffer(packed, dtype=np.uint8)
    l8 = np.frombuffer(lut, dtype=np.uint8)
    want = np.empty(n_bases, dtype=np.uint8); want[0::2] = l8[a & 15]; want[1::2] = l8[a >> 4]
    want_hash = _ffi.default().c.nafgpu_hash64_host(want.tobytes(), n_bases)
    del want
    # NAFGPU_PROBE_LIBS: comma-separated experiment builds (tools/ablate.sh) timed on the same archive after the product
    libs = [None] + [x for x in os.environ.get("NAFGPU_PROBE_LIBS", "").split(",") if x]
    for path in libs:
        L = _ffi.default() if path is None else _ffi.Library(os.path.join(R, path))
        L.c.nafgpu_test_hooks(1)
        if path is None: os.environ["NAFGPU_DEBUG_PLAN"] = "1"
        dec = nafcodec_amd.Decoder(io.BytesIO(arc), _lib=L)
        res = dec.decode_all_device()
        os.environ.pop("NAFGPU_DEBUG_PLAN", None)
        best = None
        for _ in range(3):
            res = dec.decode_all_device()
            if best is None or res.ms_total < best.ms_total:
                best = type(res).from_buffer_copy(res)
        res = best
        ok = dec.hash_device(res.d_sequence, res.n_bases) == want_hash
        print("real-genome statistics,", "product" if path is None else path, "level", level, "bases", n_bases, "compress s %.1f" % tc,
              "B/base %.4f" % (len(payload) / n_bases), "ok", ok,
              "ms total %.2f huf %.2f seq_lz %.2f other %.2f" % (res.ms_total, res.ms_huf, res.ms_seq_lz, res.ms_other),
              "Gbases/s %.1f" % (n_bases / res.ms_total / 1e6), "blocks", res.n_zstd_blocks, "streams", res.n_huf_streams, flush=True)
        dec.close()
