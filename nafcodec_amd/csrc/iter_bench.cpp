// iter_bench.cpp -- times the API the reference exposes: Decoder::from_path + Iterator::next
// (nafcodec/src/decoder/mod.rs:304-306, 356-399, 444-451) through the C-ABI, the way a Rust / C++ / Python shim drives
// it: open by path, then nafgpu_next until NAFGPU_END, touching every field it is handed (length sums), first next() to
// last (with a fourth argument N: through nafgpu_next_batch, N records a call).  Prints one JSON object.  Built by `make tools` into nafcodec_amd/iter_bench; bench.py runs it (path.iterator).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/nafgpu.h"

static double now_s() {
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
}

// iter_bench ARCHIVE.naf DEVICE REPEAT: open, every record, close -- REPEAT times in this process; the best cycle in ms (small
// archives: what a caller that walks a directory of them pays per file once the process is warm).
static int repeat_mode(const char *path, int device, int repeat) {
    double best = 1e30, first = 0;
    unsigned long long n = 0, bases = 0;
    for (int rep = 0; rep < repeat; rep++) {
        nafgpu_opts opts;
        nafgpu_opts_default(&opts);
        opts.device = device;
        nafgpu_decoder *dec = nullptr;
        nafgpu_error err;
        const double t0 = now_s();
        if (nafgpu_open_path(path, &opts, &dec, &err) != NAFGPU_OK) {
            std::fprintf(stderr, "open failed: %s\n", err.message);
            return 1;
        }
        nafgpu_record rec;
        n = bases = 0;
        for (;;) {
            const int rc = nafgpu_next(dec, &rec);
            if (rc == NAFGPU_END) break;
            if (rc != NAFGPU_OK) {
                nafgpu_last_error(dec, &err);
                std::fprintf(stderr, "next failed at record %llu: %s\n", n, err.message);
                return 1;
            }
            n++;
            bases += rec.sequence.len;
        }
        nafgpu_close(dec);
        const double dt = now_s() - t0;
        if (rep == 0) first = dt;
        if (rep > 0 && dt < best) best = dt;
    }
    std::printf("{\"records\": %llu, \"bases\": %llu, \"cycles\": %d, \"best_cycle_ms\": %.4f, \"first_cycle_ms\": %.3f}\n", n, bases, repeat,
                1e3 * best, 1e3 * first);
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: iter_bench ARCHIVE.naf [device [repeat [batch [hooks [tile_mib]]]]]\n");
        return 2;
    }
    if (argc > 3 && std::atoi(argv[3]) > 1) return repeat_mode(argv[1], std::atoi(argv[2]), std::atoi(argv[3]));
    if (argc > 5 && std::atoi(argv[5])) nafgpu_test_hooks(1);        // (experiments: the library reads its NAFGPU_* switches)
    nafgpu_opts opts;
    nafgpu_opts_default(&opts);
    opts.device = argc > 2 ? std::atoi(argv[2]) : 0;
    if (argc > 6) opts.tile_mib = std::atoi(argv[6]);                // decode in tiles of this many MiB of output (nafgpu_opts.tile_mib)
    nafgpu_decoder *dec = nullptr;
    nafgpu_error err;
    const double t_open = now_s();
    if (nafgpu_open_path(argv[1], &opts, &dec, &err) != NAFGPU_OK) {
        std::fprintf(stderr, "open failed: %s\n", err.message);
        return 1;
    }
    // `batch` > 0: records come through nafgpu_next_batch, `batch` at a time (what a binding does that hands out millions of reads)
    const unsigned long long batch = argc > 4 ? std::strtoull(argv[4], nullptr, 10) : 0;
    std::vector<nafgpu_record> recs(batch ? batch : 1);
    unsigned long long n = 0, bases = 0, qual = 0, names = 0, xsum = 0, calls = 0;
    const double t0 = now_s();
    double t_first = 0;
    for (;;) {
        uint64_t got = 1;
        const int rc = batch ? nafgpu_next_batch(dec, recs.data(), batch, &got) : nafgpu_next(dec, recs.data());
        if (calls++ == 0) t_first = now_s();
        if (rc == NAFGPU_END) break;
        if (rc != NAFGPU_OK) {
            nafgpu_last_error(dec, &err);
            std::fprintf(stderr, "next failed at record %llu: %s\n", n, err.message);
            return 1;
        }
        for (uint64_t k = 0; k < got; k++) {
            const nafgpu_record &rec = recs[k];
            n++;
            bases += rec.sequence.len;
            qual += rec.quality.len;
            names += rec.id.len + rec.comment.len;
            // (one byte of every field is read: the views must be on the host)
            if (rec.sequence.len) xsum += rec.sequence.ptr[0] + rec.sequence.ptr[rec.sequence.len - 1];
            if (rec.quality.len) xsum += rec.quality.ptr[rec.quality.len - 1];
        }
    }
    const double t1 = now_s();
    nafgpu_close(dec);
    std::printf("{\"records\": %llu, \"bases\": %llu, \"quality_bytes\": %llu, \"name_bytes\": %llu, \"open_s\": %.6f, "
                "\"first_next_s\": %.6f, \"iterate_s\": %.6f, \"total_s\": %.6f, \"xsum\": %llu, \"batch\": %llu, \"calls\": %llu}\n",
                n, bases, qual, names, t0 - t_open, t_first - t0, t1 - t0, t1 - t_open, xsum, batch, calls);
    return 0;
}
