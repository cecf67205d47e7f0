"""tools/iter_ahead_probe.py -- bench.py's FASTQ-like archive (10 M reads) and the 10 GB DNA archive through nafcodec_amd/iter_bench with
the next read-back window sent ahead (api.cpp: HostWindow) and without (NAFGPU_NO_AHEAD behind the test hooks), in turns."""
import ctypes, os, subprocess, sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import bench
from nafcodec_amd import _ffi
tool = os.path.join(R, "nafcodec_amd", "iter_bench")


def ab(path, device, batch=0):
    for k in range(6):
        env = dict(os.environ)
        if k & 1:
            env["NAFGPU_NO_AHEAD"] = "1"
        p = subprocess.run([tool, path, "0", "1", str(batch), "1"], capture_output=True, text=True, env=env)
        print("batch", batch, "plain windows" if k & 1 else "window ahead ", p.stdout.strip()[-230:], flush=True)
    return {}


bench.iterator_leg = ab
lib = _ffi.default()
bench.fastq_like_leg(lib, 0, 10_000_000)
arc = lib.synth(40_000_000_000, seed=0x4E4146)
path = "/dev/shm/nafgpu_iter_probe_%d.naf" % os.getpid()
with open(path, "wb") as f:
    f.write((ctypes.c_char * arc.n).from_address(arc.bytes))
lib.c.nafgpu_synth_free(ctypes.byref(arc))
try:
    ab(path, 0)
finally:
    os.unlink(path)
