"""tools/shard_probe.py [scale] -- the shard protocol (nafgpu_shard_*) with every rank in this process, on the GPU:
archives whose sections hold LZ sequences, 2 / 3 / 8 block ranges, both match routes; tests/cases.py has the checks."""
import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import cases
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for name in [a[0] for a in cases.lz_shard_archives(1)]:
    for modes in ((None,), ("dense", "sparse")):
        t = time.time()
        cases.check_lz_sharding(None, scale, worlds=(2, 3, 8), names=(name,), force_modes=modes)
        print(name, modes, "ok %.1f s" % (time.time() - t), flush=True)
