import sys, os, io
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, zstd_ref, naf_writer as nw, nafcodec_amd
from nafcodec_amd import _ffi
lib = _ffi.default(); lib.c.nafgpu_test_hooks(1); os.environ["NAFGPU_DEBUG_PLAN"] = "1"
rng = np.random.default_rng(2)
n_reads = 300000; L = 151; n_bases = n_reads * L
qalpha = np.frombuffer(b"#8CGGGGGGGGGG<AFFFJJJJJJJJJJJJJJ", dtype=np.uint8)
qual = qalpha[rng.integers(0, len(qalpha), n_bases)].tobytes()
for level in (1, 3):
    blob = nw.write_naf([{"id": "q", "sequence": qual.decode()}], sequence_type="text", level=level)
    print("qual level", level, len(blob), flush=True)
    d = nafcodec_amd.Decoder(io.BytesIO(blob)); d.decode_all_device(); d.close()
seq = "".join(rng.choice(list("ACGT"), 20_000_000))
for level in (1, 3):
    blob = nw.write_naf([{"id": "q", "sequence": seq}], level=level)
    print("dna level", level, len(blob), flush=True)
    d = nafcodec_amd.Decoder(io.BytesIO(blob)); d.decode_all_device(); d.close()
