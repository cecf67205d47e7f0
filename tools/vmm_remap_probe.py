"""tools/vmm_remap_probe.py -- the sequence that showed re-mapped address ranges to be unsafe on this stack ("The
last hours of round 4", (4)): one decoder takes a 4.4-Gbase archive through the tiled iterator and is closed; a second one reads 40
records of another archive (tiles again) and then asks for the whole output -- its 2 GiB tile buffer goes, buffers of 4.4 GB and
1.1 GB come.  With NAFGPU_VMM_NO_POOL=1 (released ranges unmapped and their addresses freed, as the build did until late in round 4)
NAFGPU_DEBUG_VERIFY_UPLOAD reports about half of the 4 KiB pages of the uploaded source bytes holding other data and the decode fails;
as built (released ranges stay mapped in a pool) every byte is in place.  Usage: python tools/vmm_remap_probe.py [no_pool]"""
import ctypes, os, sys
R = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, R)
from nafcodec_amd import _ffi
from nafcodec_amd.decoder import Decoder
lib = _ffi.default()
lib.c.nafgpu_test_hooks(1)
if len(sys.argv) > 1 and sys.argv[1] == "no_pool":
    os.environ["NAFGPU_VMM_NO_POOL"] = "1"
os.environ["NAFGPU_DEBUG_VERIFY_UPLOAD"] = "1"; os.environ["NAFGPU_DEBUG_TIMES"] = "1"
paths = []
for seed in (11, 12):
    arc = lib.synth(4_400_000_123, seed=seed)
    path = "/dev/shm/nafgpu_switch_%d_%d.naf" % (os.getpid(), seed)
    with open(path, "wb") as f:
        f.write((ctypes.c_char * arc.n).from_address(arc.bytes))
    lib.c.nafgpu_synth_free(ctypes.byref(arc))
    paths.append(path)
try:
    print("== first decoder: the whole archive through the iterator", flush=True)
    with Decoder(paths[0]) as d:
        n = 0
        while True:
            b = d.read_batch(256)
            if not b: break
            n += len(b)
    print("records", n, flush=True)
    print("== second decoder: 40 records, then the whole output", flush=True)
    d = Decoder(paths[1])
    first = d.read_batch(40)
    try:
        res = d.decode_all_device()
        print("bulk ok", res.n_bases, flush=True)
    except Exception as e:
        print("bulk FAILED", repr(e)[:160], flush=True)
    d.close()
finally:
    for p in paths: os.unlink(p)
