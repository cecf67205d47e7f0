"""tools/iter_ab.py BASES [OTHER_LIB] -- the synthetic archive of bench.py written to /dev/shm and read through the record iterator
(nafcodec_amd/iter_bench: nafgpu_open_path + nafgpu_next to the end) by the product library and, for an A/B on the same box, by
OTHER_LIB (a copy of iter_bench beside a copy of that library under /tmp)."""
import ctypes, os, shutil, subprocess, sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from nafcodec_amd import _ffi
n = int(float(sys.argv[1]))
base = _ffi.default()
arc = base.synth(n, seed=0x4E4146, with_mask=False, iupac_permille=0)
path = "/dev/shm/nafgpu_iter_ab_%d.naf" % os.getpid()
with open(path, "wb") as f:
    f.write((ctypes.c_char * arc.n).from_address(arc.bytes))
base.c.nafgpu_synth_free(ctypes.byref(arc))
try:
    tools = [("product", os.path.join(R, "nafcodec_amd", "iter_bench"))]
    if len(sys.argv) > 2:
        d = "/tmp/iter_ab_other"
        os.makedirs(d, exist_ok=True)
        shutil.copy(os.path.join(R, "nafcodec_amd", "iter_bench"), d)
        shutil.copy(os.path.join(R, sys.argv[2]), os.path.join(d, "libnafgpu.so"))
        tools.append((sys.argv[2], os.path.join(d, "iter_bench")))
    for rep in range(2):
        for name, tool in tools:
            p = subprocess.run([tool, path, "0"], capture_output=True, text=True, timeout=900)
            print(name, p.stdout.strip() or p.stderr[-300:], flush=True)
finally:
    os.unlink(path)
