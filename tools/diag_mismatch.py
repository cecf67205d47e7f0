"""tools/diag_mismatch.py N_BASES IUPAC [MASK] -- decode a synthetic archive on the GPU, compare every base with the CPU
oracle's records, print where the first differences are (block, stream, offset inside the stream)."""
import ctypes, io, os, sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import nafcodec_amd
from nafcodec_amd import _ffi
from oracle import oracle
n, ip = int(float(sys.argv[1])), int(sys.argv[2])
mask = len(sys.argv) > 3 and sys.argv[3] == "1"
lib = _ffi.default()
arc = lib.synth(n, seed=77, with_mask=mask, iupac_permille=ip)
blob = ctypes.string_at(arc.bytes, arc.n)
os.environ["NAFGPU_DEBUG_PLAN"] = "1"
dec = nafcodec_amd.Decoder(io.BytesIO(blob))
res = dec.decode_all_device()
got = np.frombuffer(dec.copy_to_host(res.d_sequence, res.n_bases), dtype=np.uint8)
want = np.frombuffer(b"".join(r.sequence for r in oracle.Decoder(blob, raw=True)), dtype=np.uint8)
bad = np.nonzero(got != want)[0]
print("bases", n, "iupac", ip, "mismatches", len(bad))
if len(bad):
    # runs of mismatching positions
    cuts = np.nonzero(np.diff(bad) > 1)[0]
    starts = np.r_[bad[0], bad[cuts + 1]][:12]
    ends = np.r_[bad[cuts], bad[-1]][:12]
    for a, b in zip(starts, ends):
        blk, off = divmod(int(a), 262144)
        print("  run [%d, %d] len %d: block %d, base offset in block %d (stream %d, +%d), got %r want %r" %
              (a, b, b - a + 1, blk, off, off // 65536, off % 65536, bytes(got[a:a + 12]), bytes(want[a:a + 12])))
    print("  number of runs", len(cuts) + 1, "blocks hit", sorted(set(int(x) // 262144 for x in bad))[:20])
