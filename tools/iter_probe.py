"""tools/iter_probe.py [BASES] -- the synthetic archive written to /dev/shm, then nafcodec_amd/iter_bench over it: single calls,
batches, and with the page population switched off (NAFGPU_NO_STAGING behind the test hooks)."""
import ctypes, os, subprocess, sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from nafcodec_amd import _ffi
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 40_000_000_000
lib = _ffi.default()
arc = lib.synth(n, seed=0x4E4146)
path = "/dev/shm/nafgpu_iter_probe_%d.naf" % os.getpid()
with open(path, "wb") as f:
    f.write((ctypes.c_char * arc.n).from_address(arc.bytes))
lib.c.nafgpu_synth_free(ctypes.byref(arc))
tool = os.path.join(R, "nafcodec_amd", "iter_bench")
try:
    for label, env, batch in (("single calls", {}, 0), ("single calls", {}, 0), ("batches of 4096", {}, 4096)):
        p = subprocess.run([tool, path, "0", "1", str(batch), "1"], capture_output=True, text=True, env=dict(os.environ, **env))
        print(label, p.stdout.strip()[-330:], p.stderr[-200:], flush=True)
finally:
    os.unlink(path)
