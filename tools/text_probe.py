"""tools/text_probe.py [n_bases] -- FASTA text of a synthetic DNA archive built on the device
(nafgpu_format_device): size, time, GB/s of text written; checks the text's checksum against a host
formatting of the same records for small sizes."""
import ctypes, io, os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import nafcodec_amd
from nafcodec_amd import _ffi
lib = _ffi.default()
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 40_000_000_000
arc = lib.synth(n, seed=0x4E4146, with_mask=len(sys.argv) > 2)
opts = _ffi.Opts()
lib.c.nafgpu_opts_default(ctypes.byref(opts))
h, err = ctypes.c_void_p(), _ffi.Error()
assert lib.c.nafgpu_open_bytes(ctypes.cast(arc.bytes, ctypes.c_char_p), arc.n, ctypes.byref(opts), ctypes.byref(h), ctypes.byref(err)) == 0
res = _ffi.DeviceResult()
assert lib.c.nafgpu_decode_all_device(h, ctypes.byref(res)) == 0
txt = _ffi.TextResult()
for it in range(3):
    t0 = time.perf_counter()
    rc = lib.c.nafgpu_format_device(h, ctypes.byref(txt))
    assert rc == 0, rc
    dt = time.perf_counter() - t0
    print("bases %d records %d text %.3f GB  kernels %.2f ms = %.0f GB/s of text (%.0f GB/s moved), call %.1f ms"
          % (res.n_bases, txt.n_records, txt.n_text / 1e9, txt.ms, txt.n_text / txt.ms / 1e6, 2 * txt.n_text / txt.ms / 1e6, dt * 1e3), flush=True)
