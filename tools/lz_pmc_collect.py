"""tools/lz_pmc_collect.py TAG -- folds the passes of tools/lz_pmc.sh into gpurun_out/TAG_lz_pmc.json: per probe and
kernel the launches, total time (kernel trace) and FETCH_SIZE x 2 + WRITE_SIZE bytes (rocprofv3 reports both in KiB;
x 2 = the guide's gfx950 correction for 128-byte requests), and traffic / algorithmic for the probe as a whole.
The probes run ONE decode of ONE level under the profiler (NAFGPU_PROBE_ONE_DECODE): level-3 DNA, level-1 FASTQ-like."""
import collections, csv, glob, json, sys

tag = sys.argv[1]
# algorithmic bytes of one decode: compressed bytes read + decoded bytes written
ALGO = {"l3": 1.024e9 * 1.26, "fq": 10e6 * 151 * (1 + 1) + 0.93e9}


def short(n):
    return n.replace("void nafgpu::(anonymous namespace)::", "").replace("nafgpu::(anonymous namespace)::", "").split("(")[0][:56]


def counters(name):
    acc = collections.defaultdict(float)
    n = collections.defaultdict(int)
    for f in glob.glob("gpurun_out/%s_%s/**/*counter_collection.csv" % (tag, name), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])] += float(r["Counter_Value"])
            n[short(r["Kernel_Name"])] += 1
    return acc, n


def times(name):
    acc = collections.defaultdict(float)
    for f in glob.glob("gpurun_out/%s_%s/**/*kernel_trace.csv" % (tag, name), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6
    return acc


out = {}
for p in ("l3", "fq"):
    fe, n = counters(p + "_fetch")
    wr, _ = counters(p + "_write")
    ms = times(p + "_stats")
    keys = sorted(set(fe) | set(wr), key=lambda k: -(fe.get(k, 0.0) * 2 + wr.get(k, 0.0)))
    rows = {k: {"launches": n[k], "ms": round(ms.get(k, 0.0), 3), "fetch_GB": round(fe.get(k, 0.0) * 2 * 1024 / 1e9, 3),
                "write_GB": round(wr.get(k, 0.0) * 1024 / 1e9, 3)} for k in keys}
    tot = sum(r["fetch_GB"] + r["write_GB"] for r in rows.values())
    out[p] = {"note": "one decode (no warm-up decode) of the probe's archive; hash / checksum kernels of the probe itself included",
              "kernels": rows, "traffic_GB": round(tot, 2), "algorithmic_GB": round(ALGO[p] / 1e9, 2),
              "traffic_over_algorithmic": round(tot / (ALGO[p] / 1e9), 2)}
json.dump(out, open("gpurun_out/%s_lz_pmc.json" % tag, "w"), indent=1)
for p in out:
    print(p, "traffic %.1f GB, algorithmic %.2f GB, ratio %.1f" % (out[p]["traffic_GB"], out[p]["algorithmic_GB"], out[p]["traffic_over_algorithmic"]))
    for k, r in list(out[p]["kernels"].items())[:16]:
        print("  %-56s n %4d  ms %8.2f  fetch %7.2f  write %7.2f" % (k, r["launches"], r["ms"], r["fetch_GB"], r["write_GB"]))
