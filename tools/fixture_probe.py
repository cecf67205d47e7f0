import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, R)
import nafcodec_amd
path = os.path.join(R, "tests", "golden", sys.argv[1])
for rep in range(10):
    dec = nafcodec_amd.Decoder(path)
    res = dec.decode_all_device()
    dec.close()
print("device ms", res.ms_total)
