"""tools/iter_numa.py BASES -- the record iterator (nafcodec_amd/iter_bench) on the synthetic archive, run with its threads confined to
the CPUs of NUMA node 0, then node 1, then unconfined: does the host side of the PCIe copies depend on where the process sits?"""
import ctypes, os, subprocess, sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from nafcodec_amd import _ffi
n = int(float(sys.argv[1]))
base = _ffi.default()
arc = base.synth(n, seed=0x4E4146, with_mask=False, iupac_permille=0)
path = "/dev/shm/nafgpu_iter_numa_%d.naf" % os.getpid()
with open(path, "wb") as f:
    f.write((ctypes.c_char * arc.n).from_address(arc.bytes))
base.c.nafgpu_synth_free(ctypes.byref(arc))
def cpus(node):
    return open("/sys/devices/system/node/node%d/cpulist" % node).read().strip()
try:
    tool = os.path.join(R, "nafcodec_amd", "iter_bench")
    for rep in range(2):
        for name, pre in (("node0", ["taskset", "-c", cpus(0)]), ("node1", ["taskset", "-c", cpus(1)]), ("free", [])):
            p = subprocess.run(pre + [tool, path, "0"], capture_output=True, text=True, timeout=900)
            print(name, p.stdout.strip() or p.stderr[-300:], flush=True)
finally:
    os.unlink(path)
