"""tools/iter_tile_probe.py [BASES] -- the synthetic archive in /dev/shm through nafcodec_amd/iter_bench with the decoded
output held whole (NAFGPU_ITER_TILE_MIB=0) and in tiles of several sizes: what the first call and the whole run cost."""
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
    for k in range(8):
        env = dict(os.environ)
        if k & 1:
            env["NAFGPU_STAGE_SDMA"] = "1"
        p = subprocess.run([tool, path, "0", "1", "0", "1"], capture_output=True, text=True, env=env)
        print("staged chunks fetched by", "the copy engines" if k & 1 else "a kernel        ", p.stdout.strip()[-330:], flush=True)
finally:
    os.unlink(path)
