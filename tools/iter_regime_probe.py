"""tools/iter_regime_probe.py -- the iterator's D2H rate from child process to child process (the first child of a
parent reads back at ~52 GB/s, later ones at ~30): iter_bench several times over the same 10 GB archive, plain and with the
runtime's copy engines switched off (HSA_ENABLE_SDMA=0), and the bare D2H probe in between."""
import ctypes, os, subprocess, sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from nafcodec_amd import _ffi
lib = _ffi.default()
arc = lib.synth(40_000_000_000, seed=0x4E4146)
path = "/dev/shm/nafgpu_iter_probe_%d.naf" % os.getpid()
with open(path, "wb") as f:
    f.write((ctypes.c_char * arc.n).from_address(arc.bytes))
lib.c.nafgpu_synth_free(ctypes.byref(arc))
tool = os.path.join(R, "nafcodec_amd", "iter_bench")
def run(label, env, tile_mib=0):
    p = subprocess.run([tool, path, "0", "1", "0", "1", str(tile_mib)], capture_output=True, text=True, env=dict(os.environ, **env))
    import json
    try:
        j = json.loads(p.stdout.strip().splitlines()[-1])
        print("%-28s first %.3f s, the rest %.3f s = %.1f GB/s" % (label, j["first_next_s"], j["iterate_s"] - j["first_next_s"], j["bases"] / (j["iterate_s"] - j["first_next_s"]) / 1e9), flush=True)
    except Exception:
        print(label, p.stdout[-200:], p.stderr[-300:], flush=True)
try:
    for i in range(3):
        run("plain #%d" % (i + 1), {})
    for i in range(2):
        run("HSA_ENABLE_SDMA=0 #%d" % (i + 1), {"HSA_ENABLE_SDMA": "0"})
    print(subprocess.run([os.path.join(R, "gpurun_tmp", "d2h_probe"), "vmm"], capture_output=True, text=True).stdout, flush=True)
    for i in range(2):
        run("plain again #%d" % (i + 1), {})
    for i in range(2):
        run("plain hipMemcpy upload #%d" % (i + 1), {"NAFGPU_NO_STAGING": "1"})
    for i in range(2):
        run("staged upload #%d" % (i + 1), {})
finally:
    os.unlink(path)
