"""tools/synth_probe.py BASES IUPAC_PERMILLE [MASK] -- the synthetic archive of bench.py decoded by the product library and by
the experiment builds named in NAFGPU_PROBE_LIBS (comma separated, tools/ablate.sh), timings from the library's HIP events."""
import ctypes, io, os, sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import nafcodec_amd
from nafcodec_amd import _ffi
n, ip = int(float(sys.argv[1])), int(sys.argv[2])
mask = len(sys.argv) > 3 and sys.argv[3] == "1"
base = _ffi.default()
base.c.nafgpu_test_hooks(1)
arc = base.synth(n, seed=0x4E4146, with_mask=mask, iupac_permille=ip)
# (the archive goes through a file in /dev/shm and nafgpu_open_path: ctypes cannot hand a 10 GB buffer to a file-like)
path_arc = "/dev/shm/nafgpu_probe_%d.naf" % os.getpid()
with open(path_arc, "wb") as f:
    view = (ctypes.c_char * arc.n).from_address(arc.bytes)
    f.write(view)
import atexit
atexit.register(lambda: os.path.exists(path_arc) and os.unlink(path_arc))
for path in [None] + [x for x in os.environ.get("NAFGPU_PROBE_LIBS", "").split(",") if x]:
    L = base if path is None else _ffi.Library(os.path.join(R, path))
    dec = nafcodec_amd.Decoder(path_arc, _lib=L)
    best = None
    for _ in range(4):
        res = dec.decode_all_device()
        if best is None or res.ms_total < best.ms_total:
            best = type(res).from_buffer_copy(res)
    ok = dec.hash_device(best.d_sequence, best.n_bases) == arc.seq_hash
    print("synthetic", n, "bases iupac", ip, "mask", int(mask), "product" if path is None else path, "ok", ok,
          "ms total %.2f huf %.2f seq_lz %.2f other %.2f" % (best.ms_total, best.ms_huf, best.ms_seq_lz, best.ms_other),
          "Gbases/s %.1f" % (n / best.ms_total / 1e6), "out at 0x%x" % best.d_sequence, flush=True)
    dec.close()
