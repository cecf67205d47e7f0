"""tools/small_probe.py -- the reference's own fixtures (a few Mbases at most) opened, decoded and closed over and over in one
process: wall time per archive, and where it goes (open + host walk, upload, device decode, reading every record back)."""
import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import nafcodec_amd
from nafcodec_amd import _ffi
if os.environ.get("NAFGPU_PROBE_HOOKS"): _ffi.default().c.nafgpu_test_hooks(1)   # (experiments: the NAFGPU_* switches are read)
for name in ("NZ_AAEN01000029.naf", "masked.naf", "phix.naf", "LuxC.naf"):
    path = os.path.join(R, "tests", "golden", name)
    if not os.path.exists(path):
        continue
    rows = []
    for rep in range(12):
        t0 = time.perf_counter()
        dec = nafcodec_amd.Decoder(path)
        t1 = time.perf_counter()
        res = dec.decode_all_device()
        t2 = time.perf_counter()
        dec.close()
        t3 = time.perf_counter()
        dec = nafcodec_amd.Decoder(path)
        n = sum(len(r.sequence or "") for r in dec)
        dec.close()
        t4 = time.perf_counter()
        rows.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, res.ms_total, res.ms_host_plan, res.ms_h2d, int(res.n_bases), n))
    best = min(rows[2:], key=lambda r: r[0] + r[1] + r[2])
    print("%-22s bases %9d | open %.2f ms  decode_all_device %.2f ms (device %.2f, host plan %.2f, h2d %.2f)  close %.2f ms | open + every record "
          "through the iterator + close %.2f ms (%d characters)" % (name, best[7], 1e3 * best[0], 1e3 * best[1], best[4], best[5], best[6],
                                                                   1e3 * best[2], 1e3 * best[3], best[8]), flush=True)
    print("   first call of the process: open %.1f ms decode %.1f ms" % (1e3 * rows[0][0], 1e3 * rows[0][1]), flush=True)
