// engine.cpp -- see engine.h
#include "engine.h"

#include <cstdio>
#include <cstdlib>

#include <atomic>
#include <algorithm>
#include <vector>
#include <thread>
#include <map>
#include <mutex>
#include <chrono>
#include <cstring>

namespace nafgpu {

static std::atomic<bool> g_test_hooks{false};
void set_test_hooks(bool on) { g_test_hooks.store(on); }
const char *hook_env(const char *name) { return g_test_hooks.load() ? std::getenv(name) : nullptr; }

namespace {
inline bool hip_ok(hipError_t e) { return e == hipSuccess; }

Failure dev_fail(const char *what, hipError_t e) {
    return Failure::make(NAFGPU_E_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}

const char *status_text(uint32_t code) {
    switch (code) {
    case kStHufBadEnd: return "zstd: Huffman literal stream is corrupt";
    case kStSeqBadEnd: return "zstd: sequence bitstream is corrupt";
    case kStSeqLiterals: return "zstd: sequences use more literals than the block holds";
    case kStBadOffset: return "zstd: match offset reaches before the frame start";
    case kStSizeMismatch: return "zstd: decoded size differs from the size recorded in the archive";
    case kStRunsOverflow: return "run table overflow";
    case kStInternal: return "internal error: a decode task exceeds its address window";
    case kStChecksum: return "zstd: frame checksum mismatch";
    default: return "zstd: device decoder reported an error";
    }
}

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}
}  // namespace

// ------------------------------------------------------------------ DevBuf
bool DevBuf::alloc_items(uint64_t count, uint64_t item_bytes, uint64_t extra_bytes) {
    // sizes derived from untrusted header fields: refuse anything that does not fit 63 bits instead of wrapping
    if (item_bytes && count > ((1ull << 62) - extra_bytes) / item_bytes) return false;
    return alloc(static_cast<size_t>(count * item_bytes + extra_bytes));
}

#ifndef NAFGPU_EMU
namespace {
// Large buffers come from the virtual-memory API: an address range backed by hipMemCreate chunks of up to 1 GiB.  Why: memory
// from one large hipMalloc writes at 4.9-6.7 TB/s depending on the allocation (the same virtual address after a hipFree can
// land on either side; tools/frontbench4.hip, profiles/r03_frontbench4.log -- a plain streaming fill shows it as well as
// K1's 610 k write fronts), which is what made K1 take 10.7-11.9 ms on the same archive.  Chunked backing gave 6.7 TB/s in
// nine allocations out of nine, whatever the chunk size (2 MiB, 64 MiB, 1 GiB).
constexpr size_t kVmmMinBytes = size_t(32) << 20, kVmmChunk = size_t(1) << 30;
constexpr int kSmallDevices = 16;                          // devices the process-wide pools below know of

bool vmm_usable(int dev, size_t *gran) {
    int ok = 0;
    if (!hip_ok(hipDeviceGetAttribute(&ok, hipDeviceAttributeVirtualMemoryManagementSupported, dev)) || !ok) return false;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    return hip_ok(hipMemGetAllocationGranularity(gran, &prop, hipMemAllocationGranularityRecommended)) && *gran &&
           kVmmChunk % *gran == 0;
}
}  // namespace

// Mapped ranges outlive the buffers they were.  Unmapping a range and mapping memory at the same addresses a moment later is
// (a) slow -- the address ranges and chunks of a 50 GB decoder take 10 ms in one process and 0.3-0.8 s in the next -- and
// (b) NOT SAFE on this stack: a decoder that went from tiles to the whole output (its 2 GiB tile buffer unmapped and freed,
// buffers of 4.4 GB and 1.1 GB reserved and mapped in the same call, the smaller one at the addresses just freed) found
// about every other 4 KiB page of its freshly uploaded source bytes holding something else -- zeros where the chunks were
// new, old bytes where they were reused (NAFGPU_DEBUG_VERIFY_UPLOAD; plain hipMalloc: fine; unmapping chunk by chunk and
// keeping the chunks for reuse: no better).  Translations of the old mapping seem to outlive it.  So a released range stays
// as it is -- reserved, mapped, its chunks in place -- and waits here for the next buffer of its size (sizes are multiples of
// kVmmTail, so that they meet their like again; a range up to an eighth larger than asked for is taken too).  When a creation
// fails for want of memory the idle ranges are unmapped and their chunks released, but their ADDRESSES are never given back:
// no later mapping can land where an old one was.
namespace {
constexpr size_t kVmmTail = size_t(64) << 20;
struct IdleRange {
    void *va;
    size_t total, chunk;
    std::vector<hipMemGenericAllocationHandle_t> chunks;
};
struct RangePool {
    std::mutex mu;
    std::vector<IdleRange> idle[kSmallDevices];
};
RangePool &range_pool() {
    static RangePool *p = new RangePool;
    return *p;
}
bool range_take(int dev, size_t total, size_t chunk, IdleRange *out) {
    RangePool &rp = range_pool();
    std::lock_guard<std::mutex> lock(rp.mu);
    auto &v = rp.idle[dev];
    size_t best = v.size();
    for (size_t i = 0; i < v.size(); i++)
        if (v[i].chunk == chunk && v[i].total >= total && v[i].total - total <= total / 8 && (best == v.size() || v[i].total < v[best].total)) best = i;
    if (best == v.size()) return false;
    *out = std::move(v[best]);
    v.erase(v.begin() + static_cast<std::ptrdiff_t>(best));
    return true;
}
void range_drop(IdleRange &r) {                            // memory back to the driver, the addresses stay reserved
    size_t k = 0;
    for (size_t off = 0; off < r.total; off += r.chunk, k++) {
        (void)hipMemUnmap(static_cast<char *>(r.va) + off, std::min(r.chunk, r.total - off));
        if (k < r.chunks.size()) (void)hipMemRelease(r.chunks[k]);
    }
}
// idle ranges kept per device at most (the oldest go first).  Not more: a process that holds -- or has just given back -- a hundred
// gigabytes makes the NEXT process's first allocations take 0.5 s longer (bench.py's iterator legs run in children)
constexpr size_t kVmmKeepBytes = size_t(16) << 30;
void range_give(int dev, IdleRange &&r) {
    RangePool &rp = range_pool();
    std::lock_guard<std::mutex> lock(rp.mu);
    auto &v = rp.idle[dev];
    v.push_back(std::move(r));
    size_t held = 0;
    for (const IdleRange &e : v) held += e.total;
    while (held > kVmmKeepBytes && v.size() > 1) {
        held -= v.front().total;
        range_drop(v.front());
        v.erase(v.begin());
    }
}
void range_trim(int dev) {                                 // the memory of everything idle goes back to the driver (not the addresses)
    RangePool &rp = range_pool();
    std::lock_guard<std::mutex> lock(rp.mu);
    for (IdleRange &r : rp.idle[dev]) range_drop(r);
    rp.idle[dev].clear();
}
}  // namespace

bool DevBuf::alloc_mapped(size_t bytes) {
    int dev = 0;
    size_t gran = 0;
    if (!hip_ok(hipGetDevice(&dev)) || dev < 0 || dev >= kSmallDevices || !vmm_usable(dev, &gran)) return false;
    size_t chunk = kVmmChunk;
    if (const char *ce = hook_env("NAFGPU_VMM_CHUNK_MIB")) {        // (experiments: tools/placement_probe.sh)
        const size_t want = static_cast<size_t>(std::strtoull(ce, nullptr, 10)) << 20;
        if (want >= gran && want % gran == 0) chunk = want;
    }
    const size_t tail_unit = kVmmTail % gran == 0 && chunk % kVmmTail == 0 ? kVmmTail : gran;
    const size_t total = (bytes + tail_unit - 1) / tail_unit * tail_unit;
    {
        IdleRange r;
        if (!hook_env("NAFGPU_VMM_NO_POOL") && range_take(dev, total, chunk, &r)) {
            ptr_ = r.va;
            size_ = bytes;
            reserved_ = r.total;
            chunk_bytes_ = r.chunk;
            chunks_ = std::move(r.chunks);
            return true;
        }
    }
    void *va = nullptr;
    if (!hip_ok(hipMemAddressReserve(&va, total, gran, nullptr, 0)) || !va) return false;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t mapped = 0;
    bool ok = true;
    for (size_t off = 0; off < total && ok; off += chunk) {
        const size_t n = total - off < chunk ? total - off : chunk;
        hipMemGenericAllocationHandle_t h;
        if (!hip_ok(hipMemCreate(&h, n, &prop, 0))) {
            (void)hipGetLastError();
            range_trim(dev);                               // (what waits for a buffer of another size gives its memory back first)
            if (!hip_ok(hipMemCreate(&h, n, &prop, 0))) { ok = false; break; }
        }
        if (!hip_ok(hipMemMap(static_cast<char *>(va) + off, n, 0, h, 0))) {
            (void)hipMemRelease(h);
            ok = false;
            break;
        }
        chunks_.push_back(h);
        mapped = off + n;
    }
    if (ok) {
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        ok = hip_ok(hipMemSetAccess(va, total, &acc, 1));
    }
    if (!ok) {                                   // e.g. out of device memory: undo, the caller reports the failure of hipMalloc
        for (size_t off = 0; off < mapped; off += chunk) (void)hipMemUnmap(static_cast<char *>(va) + off, std::min(chunk, mapped - off));
        for (auto h : chunks_) (void)hipMemRelease(h);
        chunks_.clear();
        (void)hipMemAddressFree(va, total);
        (void)hipGetLastError();
        return false;
    }
    ptr_ = va;
    size_ = bytes;
    reserved_ = total;
    chunk_bytes_ = chunk;
    return true;
}
#endif

#if !defined(NAFGPU_EMU) || defined(NAFGPU_EMU_CACHE)   // (NAFGPU_EMU_CACHE: a harness build WITH the cache, to chase what depends on it)
#define NAFGPU_SMALL_CACHE 1
#endif
#ifdef NAFGPU_SMALL_CACHE
// Small device buffers outlive their decoders.  Opening, decoding and closing one of the reference's fixtures makes a hundred
// hipMalloc calls and as many hipFree calls, each of which waits for the device (rocprofv3 --hip-trace, tools/small_api_trace.sh:
// 0.75 ms of a 3.6 ms cycle): buffers of up to 256 KiB come in power-of-two size classes, and one whose OWNER goes away (the
// destructor: by then the owner's streams are drained, ~ArchiveJob) is kept for the next decoder on the same device -- up to
// 64 MiB of them.  A buffer given up while its owner lives on (alloc() growing it) is freed as before: hipFree's wait is what
// makes that safe.  Never torn down (the runtime may be gone before a static destructor runs).  The CPU harness does without:
// rounded-up sizes would hide small overruns from the sanitizer.
namespace {
constexpr size_t kSmallMax = size_t(256) << 10, kSmallMin = 256, kSmallKeepBytes = size_t(64) << 20;
constexpr int kSmallClasses = 11;                          // 256 B .. 256 KiB
struct SmallCache {
    std::mutex mu;
    std::vector<void *> idle[kSmallDevices][kSmallClasses];
    size_t bytes = 0;
};
SmallCache &small_cache() {
    static SmallCache *c = new SmallCache;
    return *c;
}
int small_class(size_t bytes) {
    int c = 0;
    while ((kSmallMin << c) < bytes) c++;
    return c;
}
}  // namespace
#endif

void DevBuf::view(void *p, size_t bytes) {
    release();
    ptr_ = p;
    size_ = bytes;
    view_ = true;
}

bool DevBuf::alloc(size_t bytes) {
    if (ptr_ && !view_ && bytes <= size_) return true;
    release();
    // (nafgpu_test_hooks + NAFGPU_ALLOC_PLAIN=1: everything from hipMalloc, for A/B runs -- tools/placement_probe.sh; any value:
    //  no small-buffer cache)
    const char *plain = hook_env("NAFGPU_ALLOC_PLAIN");
#ifndef NAFGPU_EMU
    if (bytes >= kVmmMinBytes && !(plain && plain[0] == '1') && alloc_mapped(bytes)) return true;
#endif
#ifdef NAFGPU_SMALL_CACHE
    int dev = -1;
    if (bytes <= kSmallMax && !plain && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < kSmallDevices) {
        const int c = small_class(bytes);
        void *p = nullptr;
        {
            SmallCache &sc = small_cache();
            std::lock_guard<std::mutex> lock(sc.mu);
            std::vector<void *> &v = sc.idle[dev][c];
            if (!v.empty()) {
                p = v.back();
                v.pop_back();
                sc.bytes -= kSmallMin << c;
            }
        }
        if (!p && !hip_ok(hipMalloc(&p, kSmallMin << c))) {
            (void)hipGetLastError();
            range_trim(dev);                               // (idle ranges give their memory back before anything fails for want of it)
            if (!hip_ok(hipMalloc(&p, kSmallMin << c))) return false;
        }
        ptr_ = p;
        size_ = kSmallMin << c;
        cache_dev_ = dev;
        return true;
    }
#endif
    void *p = nullptr;
    if (!hip_ok(hipMalloc(&p, bytes ? bytes : 16))) {
#ifndef NAFGPU_EMU
        (void)hipGetLastError();
        int d = 0;
        if (!hip_ok(hipGetDevice(&d)) || d < 0 || d >= kSmallDevices) return false;
        range_trim(d);
        if (!hip_ok(hipMalloc(&p, bytes ? bytes : 16))) return false;
#else
        return false;
#endif
    }
    ptr_ = p;
    size_ = bytes ? bytes : 16;
    return true;
}

bool DevBuf::upload(const void *host, size_t bytes, hipStream_t stream) {
    if (!alloc(bytes)) return false;
    if (bytes == 0) return true;
    return hip_ok(hipMemcpyAsync(ptr_, host, bytes, hipMemcpyHostToDevice, stream));
}

void DevBuf::release(bool dying) {
    if (view_) {
        ptr_ = nullptr;
        size_ = 0;
        view_ = false;
        return;
    }
#ifdef NAFGPU_SMALL_CACHE
    if (ptr_ && cache_dev_ >= 0) {
        if (dying) {
            SmallCache &sc = small_cache();
            std::lock_guard<std::mutex> lock(sc.mu);
            if (sc.bytes + size_ <= kSmallKeepBytes) {
                sc.idle[cache_dev_][small_class(size_)].push_back(ptr_);
                sc.bytes += size_;
                ptr_ = nullptr;
            }
        }
        cache_dev_ = -1;
        if (!ptr_) {
            size_ = 0;
            return;
        }
    }
#endif
#ifndef NAFGPU_EMU
    if (ptr_ && reserved_) {
        // (range_pool above: why the range is kept as it is rather than unmapped)
        int dev = 0;
        if (hip_ok(hipGetDevice(&dev)) && dev >= 0 && dev < kSmallDevices && !hook_env("NAFGPU_VMM_NO_POOL")) {
            if (!dying) (void)hipDeviceSynchronize();      // (a living owner: whatever it still has in flight is done before another takes the range)
            IdleRange r{ptr_, reserved_, chunk_bytes_ ? chunk_bytes_ : reserved_, std::move(chunks_)};
            range_give(dev, std::move(r));
        } else {                                           // chunk by chunk, as they were mapped
            if (hook_env("NAFGPU_VMM_SYNC_UNMAP")) (void)hipDeviceSynchronize();   // (experiment: is anything still in flight on the range?)
            const size_t step = chunk_bytes_ ? chunk_bytes_ : reserved_;
            size_t k = 0;
            for (size_t off = 0; off < reserved_; off += step, k++) {
                (void)hipMemUnmap(static_cast<char *>(ptr_) + off, std::min(step, reserved_ - off));
                if (k < chunks_.size()) (void)hipMemRelease(chunks_[k]);
            }
            (void)hipMemAddressFree(ptr_, reserved_);
        }
        chunks_.clear();
        ptr_ = nullptr;
    }
#endif
    if (ptr_) (void)hipFree(ptr_);
    ptr_ = nullptr;
    size_ = reserved_ = 0;
}

void trim_device_memory(int device) {
#ifndef NAFGPU_EMU
    if (device < 0 || device >= kSmallDevices) return;
    (void)hipDeviceSynchronize();
    range_trim(device);
    std::vector<void *> gone;
    {
        SmallCache &sc = small_cache();
        std::lock_guard<std::mutex> lock(sc.mu);
        for (int c = 0; c < kSmallClasses; c++) {
            for (void *p : sc.idle[device][c]) {
                gone.push_back(p);
                sc.bytes -= kSmallMin << c;
            }
            sc.idle[device][c].clear();
        }
    }
    for (void *p : gone) (void)hipFree(p);
#else
    (void)device;
#endif
}

// ------------------------------------------------------------------ StageTimer
StageTimer::~StageTimer() {
    for (hipEvent_t e : pool_) (void)hipEventDestroy(e);
    if (t0_) (void)hipEventDestroy(t0_);
    if (t1_) (void)hipEventDestroy(t1_);
}

hipEvent_t StageTimer::get() {
    if (used_ == pool_.size()) {
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        pool_.push_back(e);
    }
    return pool_[used_++];
}

void StageTimer::reset() {
    spans_.clear();
    used_ = 0;
    open_ = false;
}

void StageTimer::begin(hipStream_t s, Cat c) {
    Span sp{get(), get(), c};
    (void)hipEventRecord(sp.a, s);
    spans_.push_back(sp);
    open_ = true;
}

void StageTimer::end(hipStream_t s) {
    if (!open_) return;
    (void)hipEventRecord(spans_.back().b, s);
    open_ = false;
}

void StageTimer::mark_total_begin(hipStream_t s) {
    if (!t0_) (void)hipEventCreate(&t0_);
    if (!t1_) (void)hipEventCreate(&t1_);
    (void)hipEventRecord(t0_, s);
}

void StageTimer::mark_total_end(hipStream_t s) { (void)hipEventRecord(t1_, s); }

StageTimes StageTimer::collect() {
    StageTimes t;
    for (const Span &sp : spans_) {
        float ms = 0;
        if (!hip_ok(hipEventElapsedTime(&ms, sp.a, sp.b))) continue;
        switch (sp.cat) {
        case kHuf: t.huf += ms; t.huf_launches++; break;
        case kSeqLz: t.seq_lz += ms; break;
        case kUnpack: t.unpack += ms; break;
        default: t.other += ms; break;
        }
    }
    if (t0_ && t1_) (void)hipEventElapsedTime(&t.total, t0_, t1_);
    return t;
}

// ------------------------------------------------------------------ SectionJob
// Host -> device for the compressed bytes of a section.  One hipMemcpyAsync out of ordinary or mapped memory moves at PCIe rate in
// one process and at a quarter of it in the next -- the runtime's copy engines again (see k_copy_out: 10 GB in 0.2 s or in 1.2 s,
// tools/iter_regime_probe.py) -- so large uploads take the same road as the read-back: kStageThreads host threads copy their
// chunks of the source into pinned buffers (two of 16 MiB each, so the memcpy of one overlaps the transfer of the other; the
// page faults of a file mapping spread over the threads as well), and the GPU fetches every chunk itself (k_copy_out with
// the pinned buffer as its source).  Returns when every byte is on the device.
#ifndef NAFGPU_EMU
namespace {
constexpr size_t kStageChunk = size_t(16) << 20;
constexpr unsigned kStageThreads = 8;
constexpr size_t kStageMin = size_t(256) << 20;            // smaller uploads: one plain copy
struct StageSlot {
    uint8_t *buf[2] = {nullptr, nullptr};
    hipStream_t stream = nullptr;
    hipEvent_t done[2] = {nullptr, nullptr};
};
std::mutex g_stage_mu;                                     // one staged upload at a time per process (the buffers are shared)
StageSlot g_stage[kStageThreads];
bool g_stage_ready = false, g_stage_failed = false;
bool stage_init() {
    if (g_stage_ready) return true;
    if (g_stage_failed) return false;
    for (unsigned t = 0; t < kStageThreads; t++) {
        StageSlot &sl = g_stage[t];
        bool ok = hipStreamCreate(&sl.stream) == hipSuccess;
        for (int k = 0; k < 2 && ok; k++)
            ok = hipHostMalloc(reinterpret_cast<void **>(&sl.buf[k]), kStageChunk) == hipSuccess &&
                 hipEventCreateWithFlags(&sl.done[k], hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            g_stage_failed = true;                         // (what was allocated stays: a plain copy serves from here on)
            return false;
        }
    }
    g_stage_ready = true;
    return true;
}
}  // namespace
#endif

bool upload_staged(uint8_t *d_dst, const uint8_t *src, size_t n, hipStream_t stream, size_t stage_min) {
#ifndef NAFGPU_EMU
    if (n >= (stage_min ? stage_min : kStageMin) && !hook_env("NAFGPU_NO_STAGING")) {
        std::lock_guard<std::mutex> guard(g_stage_mu);
        if (stage_init() && hipStreamSynchronize(stream) == hipSuccess) {   // (what was enqueued in front -- the pad memsets -- is done)
            int dev = 0;
            (void)hipGetDevice(&dev);
            const size_t n_chunks = (n + kStageChunk - 1) / kStageChunk;
            std::atomic<bool> failed{false};
            const bool sdma = hook_env("NAFGPU_STAGE_SDMA") != nullptr;   // (experiment: the copy engines fetch the chunks, not a kernel)
            auto worker = [&](unsigned t) {
                (void)hipSetDevice(dev);
                StageSlot &sl = g_stage[t];
                unsigned k = 0;
                for (size_t c = t; c < n_chunks && !failed.load(); c += kStageThreads, k ^= 1u) {
                    const size_t off = c * kStageChunk, len = std::min(kStageChunk, n - off);
                    if (hipEventSynchronize(sl.done[k]) != hipSuccess) failed = true;   // (the kernel that last read this buffer)
                    std::memcpy(sl.buf[k], src + off, len);
                    if (sdma) {
                        if (hipMemcpyAsync(d_dst + off, sl.buf[k], len, hipMemcpyHostToDevice, sl.stream) != hipSuccess) failed = true;
                    } else {
                        launch_copy_out(sl.stream, d_dst + off, sl.buf[k], len);
                    }
                    if (hipGetLastError() != hipSuccess || hipEventRecord(sl.done[k], sl.stream) != hipSuccess) failed = true;
                }
                if (hipStreamSynchronize(sl.stream) != hipSuccess) failed = true;
            };
            std::vector<std::thread> pool;
            unsigned started = 1;
            try {
                for (; started < kStageThreads; started++) pool.emplace_back(worker, started);
            } catch (...) {                                // no more threads to be had: their chunks are done here
            }
            worker(0);
            for (unsigned t = started; t < kStageThreads; t++) worker(t);
            for (std::thread &th : pool) th.join();
            return !failed.load();
        }
    }
#endif
    return hipMemcpyAsync(d_dst, src, n, hipMemcpyHostToDevice, stream) == hipSuccess;
}

// tiles whose compressed bytes are fewer travel with load_tile itself (NAFGPU_SRC_PREFETCH_MIN: tests lower it)
static size_t src_prefetch_min() {
    if (const char *e = hook_env("NAFGPU_SRC_PREFETCH_MIN")) return static_cast<size_t>(std::strtoull(e, nullptr, 10));
    return size_t(1) << 20;
}

void SectionJob::walk(const uint8_t *host_payload, size_t n) {
    drain();
    const double t0 = now_ms();
    master_ = ZPlan();
    walk_truncated_ = false;
    walk_err_ = walk_zstd(host_payload, n, &master_, &walk_truncated_);
    plan_ms_ = static_cast<float>(now_ms() - t0);
    walked_payload_ = host_payload;
    walked_n_ = n;
}

void SectionJob::drain() {
    for (SrcSlot &sl : src_slot_) {
        if (sl.pending.valid()) sl.pending.wait();
        sl.pending = std::future<bool>();
        sl.tile = 0xFFFFFFFFu;
    }
}

// The compressed bytes of tile t on their way into slot t & 1, on a thread of their own (the staged upload keeps its caller
// until the last byte is across).  Tile t - 2's bytes were there: its kernels are long done (decode_tile synchronises).
bool SectionJob::start_source_upload(uint32_t t) {
    SrcSlot &sl = src_slot_[t & 1];
    if (sl.pending.valid()) sl.pending.wait();             // (an upload nobody came for)
    sl.pending = std::future<bool>();
    sl.tile = 0xFFFFFFFFu;
    const Tile tile = tiles_[t];
    const uint64_t lo = master_.blk_off[tile.b0], n = master_.blk_off[tile.b1] - lo;
    if (!sl.buf.alloc(kSrcFrontPad + static_cast<size_t>(std::max(src_cap_, n)) + kSrcBackPad)) return false;
    if (!prefetch_stream_ && !hip_ok(hipStreamCreateWithFlags(&prefetch_stream_, hipStreamNonBlocking))) return false;
    int dev = 0;
    (void)hipGetDevice(&dev);
    uint8_t *dst = sl.buf.bytes();
    const uint8_t *src = host_payload_ + lo;
    hipStream_t st = prefetch_stream_;
#ifdef NAFGPU_EMU
    const std::launch how = std::launch::deferred;         // (the CPU harness is one thread: the copy runs when it is asked for)
#else
    const std::launch how = std::launch::async;
#endif
    sl.tile = t;
    if (hook_env("NAFGPU_DEBUG_TIMES")) std::fprintf(stderr, "[nafgpu] tile %u: %llu source bytes on their way\n", t, static_cast<unsigned long long>(n));
    auto send = [dst, src, n, dev, st] {
        (void)hipSetDevice(dev);
        bool ok = hip_ok(hipMemsetAsync(dst, 0, kSrcFrontPad, st)) && hip_ok(hipMemsetAsync(dst + kSrcFrontPad + n, 0, kSrcBackPad, st));
        ok = ok && (n == 0 || upload_staged(dst + kSrcFrontPad, src, static_cast<size_t>(n), st, size_t(32) << 20));
        return ok && hip_ok(hipStreamSynchronize(st));
    };
    try {
        sl.pending = std::async(how, send);
    } catch (...) {                                        // no thread to be had: the copy runs when the tile is loaded
        sl.pending = std::async(std::launch::deferred, send);
    }
    return true;
}

Failure SectionJob::prepare(const uint8_t *host_payload, size_t n, uint64_t expect_size, hipStream_t stream,
                            const SectionOptions &opt) {
    ready_ = false;
    drain();
    opt_ = opt;
    t_char_ = opt.t_char;
    host_payload_ = host_payload;
    plan_ = ZPlan();
    tiles_.clear();
    loaded_tile_ = 0xFFFFFFFFu;
    tiles_done_ = 0;
    {   // sub-streams (zplan.cpp: g_split_target): a section is cut when its streams are fewer than half the lanes named here --
        // a quarter of what the chip keeps resident (eight waves per CU), the two passes in front being work too
        static const uint32_t lanes = [] {
            hipDeviceProp_t p;
            int dev = 0;
            (void)hipGetDevice(&dev);
            const int cus = hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
            return static_cast<uint32_t>(cus) * 256u;
        }();
        const char *fs = hook_env("NAFGPU_HUF_SPLIT");     // tests: that many parts per stream whatever the size (0: never)
        const uint32_t force = fs ? static_cast<uint32_t>(std::atoi(fs)) : 0u;
        set_huf_split(fs && force == 0 ? 0u : lanes, force);
    }
    const double t0 = now_ms();
    if (walked_payload_ != host_payload || walked_n_ != n) walk(host_payload, n);
    walked_payload_ = nullptr;                             // (the result is used up: load_tile may move master_ away)
    const bool truncated = walk_truncated_;
    const std::string err = std::move(walk_err_);
    walk_err_.clear();
    const bool trace = hook_env("NAFGPU_DEBUG_TIMES") != nullptr;
    if (trace) std::fprintf(stderr, "[nafgpu] prepare: walk of %zu bytes %.1f ms\n", n, plan_ms_);
    if (!err.empty())
        return Failure::io(truncated ? NAFGPU_IO_UNEXPECTED_EOF : NAFGPU_IO_INVALID_DATA, "zstd: " + err);
    expect_ = expect_size;
    master_blocks_ = master_.blk_size.size();
    master_streams_ = master_.streams.size();
    // The size the archive announces is not something the reference checks (its zstd reader is never told it; a section that
    // decodes to less fails at the record that needs the missing bytes, mod.rs:373-385 over reader.rs:104-119): without LZ
    // sequences the walk knows what the section decodes to, and that is what is decoded; with sequences the announced size
    // bounds the output, less is found out on the device (check()), more is an error.
    if (master_.seq_blocks.empty()) expect_size = expect_ = master_.known_out;
    if (expect_size > static_cast<uint64_t>(master_.blk_size.size()) * kBlockMax)     // untrusted: no block decodes to more than 128 KiB
        return Failure::io(NAFGPU_IO_INVALID_DATA, "zstd: decoded size differs from the size recorded in the archive");
    // ---- this process's block range: a shard of a section without LZ sequences -- or, when the ranks run the shard
    // protocol, of any section -- else everything
    uint32_t b0 = 0, b1 = static_cast<uint32_t>(master_blocks_);
    sharded_ = shard_range(master_, opt.shard_rank, opt.shard_count, &b0, &b1, opt.shard_protocol);
    has_lz_ = !master_.seq_blocks.empty();
    proto_lz_ = sharded_ && opt.shard_protocol && has_lz_;
    out0_ = 0;
    out1_ = expect_size;
    if (sharded_ && !has_lz_) {
        for (uint32_t b = 0; b < b0; b++) out0_ += master_.blk_size[b];
        out1_ = out0_;
        for (uint32_t b = b0; b < b1; b++) out1_ += master_.blk_size[b];
    }
    ranges_.clear();
    frame_first_.clear();
    seq_blk_.clear();
    seq_frame_.clear();
    if (sharded_ && opt.shard_protocol) {                  // what the placement needs of the walk once the master plan is gone
        for (uint32_t r = 0; r < opt.shard_count; r++) {
            uint32_t a = 0, e = 0;
            shard_range(master_, r, opt.shard_count, &a, &e, true);
            ranges_.push_back({a, e});
        }
        for (const ZPlan::Frame &f : master_.frames) frame_first_.push_back(f.first_blk);
        for (const SeqBlock &sb : master_.seq_blocks) {
            seq_blk_.push_back(sb.blk);
            seq_frame_.push_back(sb.frame_first_blk);
        }
    }
    if (proto_lz_) {                                       // where the range begins in the decoded section is learnt from the other ranks
        out0_ = out1_ = 0;
        decoded_ = 0;
    }
    // ---- tiles
    // (a range that exchanges windows with its neighbours is decoded in one piece)
    const uint64_t tb = opt.tile_blocks && !proto_lz_ ? opt.tile_blocks : (static_cast<uint64_t>(b1 - b0) ? b1 - b0 : 1);
    for (uint64_t b = b0; b < b1 || tiles_.empty(); b += tb) {
        tiles_.push_back(Tile{static_cast<uint32_t>(b), static_cast<uint32_t>(std::min<uint64_t>(b + tb, b1))});
        if (b1 == b0) break;
    }
    sec_known_ = master_.known_out;
    sec_seqs_ = master_.n_sequences;
    const bool lz = has_lz_;
    // the window of a tile that follows another: its matches may reach that far back
    halo_cap_ = (lz && (tiles_.size() > 1 || (proto_lz_ && b0 > 0))) ? std::min<uint64_t>(std::max<uint64_t>(master_.window_max, 1), expect_size) : 0;
    if (halo_cap_ > 0xF0000000ull) return Failure::io(NAFGPU_IO_INVALID_DATA, "zstd: window too large to decode in tiles");
    tile_cap_ = 0;
    const uint64_t mult = t_char_ ? 2 : 1;
    out_shift_ = 0;
    halo_pending_ = false;
    lz_args_valid_ = false;
    send_elems_ = 0;
    bool ok = d_status_.alloc(64) && d_counters_.alloc(256);
    if (proto_lz_) {
        // [room for the window in front][the range: at most a full block per block]
        out_shift_ = halo_cap_ * mult;
        ok = ok && d_out_.alloc_items(halo_cap_ + static_cast<uint64_t>(b1 - b0) * kBlockMax, mult, 64);
    } else if (tiled_output()) {
        uint64_t most = 0;
        for (const Tile &t : tiles_) most = std::max<uint64_t>(most, t.b1 - t.b0);
        tile_cap_ = most * kBlockMax;
        ok = ok && d_out_.alloc_items(halo_cap_ + tile_cap_, mult, 64);
        src_cap_ = 0;
        for (const Tile &t : tiles_) src_cap_ = std::max<uint64_t>(src_cap_, master_.blk_off[t.b1] - master_.blk_off[t.b0]);
    } else {
        // Tiles of a section with LZ sequences into the WHOLE output: a tile's element count is known to the host as an upper
        // bound only (a full block per block), and the dense route's kernels walk D -- and look at the output behind it -- up
        // to that bound (k_pj_fill zeroes the words past the tile's real end, a sweep takes a zero for a literal in place and
        // reads it).  Inside the buffer that is the next tile's room; behind the last tile it must be room too.
        uint64_t slack = 0;
        if (lz && tiles_.size() > 1)
            for (const Tile &t : tiles_) slack = std::max<uint64_t>(slack, static_cast<uint64_t>(t.b1 - t.b0) * kBlockMax);
        ok = ok && d_out_.alloc_items(out1_ - out0_ + slack, mult, 64);
    }
    if (!ok) return Failure::make(NAFGPU_E_DEVICE, "out of device memory while preparing a section");
    tile_pos0_ = out0_;
    tile_len_ = 0;
    rep_carry_[0] = 1;
    rep_carry_[1] = 4;
    rep_carry_[2] = 8;
    carry_frame_ = 0xFFFFFFFFu;
    carry_same_frame_ = false;
    ready_ = true;
    if (trace) std::fprintf(stderr, "[nafgpu] prepare: output buffer at %.1f ms\n", now_ms() - t0);
    if (tiles_.size() == 1) {
        Failure f = load_tile(0, stream);
        if (trace) std::fprintf(stderr, "[nafgpu] prepare: tile loaded at %.1f ms\n", now_ms() - t0);
        if (!f.ok()) {
            ready_ = false;
            return f;
        }
    }
    return Failure();
}

// The launchable plan of tile t: selected from the walk, uploaded together with the compressed bytes it reads.
Failure SectionJob::load_tile(uint32_t t, hipStream_t stream) {
    if (t == loaded_tile_) return Failure();
    const Tile tile = tiles_[t];
    const bool lz = has_lz_;
    // LZ window in front of the tile: min(window, decoded so far in this process's range)
    halo_elems_ = (lz && t > 0) ? std::min<uint64_t>(halo_cap_, tile_pos0_ - out0_) : 0;
    if (proto_lz_) halo_elems_ = 0;                        // (learnt in shard_place; the pseudo block is there from the start)
    const double t0 = now_ms();
    if (tiles_.size() == 1 && !sharded_) {
        plan_ = std::move(master_);                        // the whole section: nothing to re-base, nothing selected again
        master_ = ZPlan();
        plan_.src_lo = 0;
        plan_.src_hi = plan_.blk_off.empty() ? 0 : plan_.blk_off.back();
        if (const char *tl = hook_env("NAFGPU_TASK_LANES")) set_task_lanes(static_cast<uint32_t>(std::atoi(tl)));
        if (const char *ds = hook_env("NAFGPU_DICT_SLOTS")) set_dict_slots(static_cast<uint32_t>(std::atoi(ds)));
        pack_tasks_public(&plan_);
    } else {
        select_zplan(master_, tile.b0, tile.b1, halo_elems_, &plan_, proto_lz_ && halo_cap_ != 0);
    }
    plan_ms_ += static_cast<float>(now_ms() - t0);
    n_blocks_ = plan_.blk_size.size();
    n_streams_ = plan_.streams.size();
    n_tasks_ = plan_.tasks.size();
    n_copies_ = plan_.copies.size();
    n_seq_blocks_ = plan_.seq_blocks.size();
    cells_cap_ = 0;
    for (const SeqBlock &sb : plan_.seq_blocks)
        cells_cap_ = std::max<uint32_t>(cells_cap_, (1u << sb.ll_al) + (1u << sb.of_al) + (1u << sb.ml_al));
    classes_ = plan_.classes;                              // launch classes of the Huffman tasks (plan.h: HufClass)
    // frames with a Content_Checksum: the pieces of them this tile holds (a frame begun in front of a shard's range
    // cannot be verified by this process)
    xxh_segs_.clear();
    if (t == 0) xxh_live_ = false;
    if (plan_.has_checksum) {
        const std::vector<ZPlan::Frame> &frames = plan_.frames;
        bool live = false;
        for (const ZPlan::Frame &f : frames) {
            if (f.end_blk <= tile.b0) continue;
            if (f.first_blk >= tile.b1) break;
            const bool begins = f.first_blk >= tile.b0, ends = f.end_blk <= tile.b1;
            if (!f.has_checksum || (!begins && !xxh_live_)) continue;
            xxh_segs_.push_back(XxhSeg{std::max(f.first_blk, tile.b0) - tile.b0 + plan_.halo, std::min(f.end_blk, tile.b1) - tile.b0 + plan_.halo,
                                       f.checksum, (begins ? 1u : 0u) | (ends ? 2u : 0u)});
            live = !ends;
        }
        xxh_live_ = live;
    }
    if (hook_env("NAFGPU_DEBUG_PLAN")) {
        std::fprintf(stderr, "[nafgpu] section plan: tile %u of %zu, %zu blocks, %zu streams, %zu seq blocks, %llu sequences, literal buffer %llu B, source %llu B; task classes:",
                     t, tiles_.size(), n_blocks_, n_streams_, n_seq_blocks_, static_cast<unsigned long long>(plan_.n_sequences),
                     static_cast<unsigned long long>(plan_.lit_bytes), static_cast<unsigned long long>(plan_.src_hi - plan_.src_lo));
        for (const HufClass &c : classes_)
            std::fprintf(stderr, " {%u tasks, tbl %u, %s%s, lds %u B, %u parts}", c.n_tasks, c.tbl, c.to_lit ? "lit" : "out", c.seg ? "+seg" : "", c.lds_bytes, c.split);
        uint64_t cells = 0;
        for (const SeqBlock &sb : plan_.seq_blocks) cells += (1u << sb.ll_al) + (1u << sb.of_al) + (1u << sb.ml_al);
        uint64_t fresh = 0;                                    // tables that are not the block in front's (Repeat_Mode)
        for (size_t k = 0; k < plan_.seq_blocks.size(); k++) {
            const SeqBlock &sb = plan_.seq_blocks[k];
            if (k == 0 || sb.ll_tbl != plan_.seq_blocks[k - 1].ll_tbl) fresh++;
            if (k == 0 || sb.of_tbl != plan_.seq_blocks[k - 1].of_tbl) fresh++;
            if (k == 0 || sb.ml_tbl != plan_.seq_blocks[k - 1].ml_tbl) fresh++;
        }
        std::fprintf(stderr, "; FSE cells per block: %llu on average, %u at most; %llu of %zu tables differ from the block in front's\n",
                     static_cast<unsigned long long>(n_seq_blocks_ ? cells / n_seq_blocks_ : 0), cells_cap_, static_cast<unsigned long long>(fresh), 3 * plan_.seq_blocks.size());
    }
    // ---- the compressed bytes the tile's tasks read, with the padding k_huf_decode's whole-line loads may touch
    const uint64_t src_n = plan_.src_hi - plan_.src_lo;
    src_resident_ = src_n;
    // Small tiles (the reference's own fixtures, the text sections of most archives): the source bytes and the ten task lists
    // travel as ONE buffer in ONE copy -- a dozen copies of a few hundred bytes each cost 24 us of host time apiece, a third of
    // what a small archive took.  The lists are pieces of d_pack_ then (DevBuf::view).
    struct Piece {
        DevBuf *buf;
        const void *src;
        size_t n, front, back, off;
    };
    Piece pieces[] = {
        {&d_src_buf_, host_payload_ + plan_.src_lo, static_cast<size_t>(src_n), kSrcFrontPad, kSrcBackPad, 0},
        {&d_blk_size_, plan_.blk_size.data(), n_blocks_ * sizeof(uint32_t), 0, 0, 0},
        {&d_streams_, plan_.streams.data(), n_streams_ * sizeof(HufStream), 0, 0, 0},
        {&d_tasks_, plan_.tasks.data(), n_tasks_ * sizeof(HufTask), 0, 0, 0},
        {&d_tbl_copies_, plan_.tbl_copies.data(), plan_.tbl_copies.size() * sizeof(HufTblCopy), 0, 0, 0},
        {&d_pool_, plan_.huf_pool.data(), plan_.huf_pool.size() * sizeof(uint16_t), 0, 0, 0},
        {&d_dicts_, plan_.dict_pool.data(), plan_.dict_pool.size(), 0, 0, 0},
        {&d_copies_, plan_.copies.data(), n_copies_ * sizeof(CopyTask), 0, 0, 0},
        {&d_seq_blocks_, plan_.seq_blocks.data(), n_seq_blocks_ * sizeof(SeqBlock), 0, 0, 0},
        {&d_cells_, plan_.fse_pool.data(), plan_.fse_pool.size() * sizeof(SeqCell), 0, 0, 0},
        {&d_xxh_segs_, xxh_segs_.data(), xxh_segs_.size() * sizeof(XxhSeg), 0, 0, 0},
    };
    size_t pack_total = 0;
    for (Piece &pc : pieces) {
        pc.off = pack_total;
        pack_total += (pc.front + pc.n + pc.back + 16 + 255) & ~size_t(255);
    }
    // (up to 256 KiB: a larger copy out of ordinary heap memory takes the runtime's pin-the-pages path -- 17 ms for 1.4 MB measured --
    // where the same bytes straight from the file mapping took 0.4)
    constexpr size_t kPackMax = size_t(256) << 10;
    bool ok = true;
    if (pack_total <= kPackMax) {
        pack_host_.assign(pack_total, 0);                      // (the padding around the source bytes is part of it)
        for (const Piece &pc : pieces)
            if (pc.n) std::memcpy(pack_host_.data() + pc.off + pc.front, pc.src, pc.n);
        ok = d_pack_.alloc(pack_total) &&
             hip_ok(hipMemcpyAsync(d_pack_.bytes(), pack_host_.data(), pack_total, hipMemcpyHostToDevice, stream));
        if (ok)
            for (const Piece &pc : pieces) pc.buf->view(d_pack_.bytes() + pc.off, pc.front + pc.n + pc.back);
        d_src_ = d_src_buf_.bytes() + kSrcFrontPad - plan_.src_lo;         // kernels address the payload by its offsets
        ok = ok && d_blk_base_.alloc((n_blocks_ + 1) * sizeof(uint64_t)) && d_scan_tmp_.alloc(scan_tmp_bytes(n_blocks_));
    } else {
        if (tiled_output() && src_n >= src_prefetch_min()) {      // (start_source_upload: it may have travelled ahead)
            SrcSlot &sl = src_slot_[t & 1];
            ok = sl.tile == t || start_source_upload(t);
            if (ok && sl.pending.valid()) ok = sl.pending.get();
            if (ok) {
                d_src_buf_.view(sl.buf.bytes(), sl.buf.size());
                d_src_ = d_src_buf_.bytes() + kSrcFrontPad - plan_.src_lo;
            } else {
                sl.tile = 0xFFFFFFFFu;
            }
        } else if ((ok = d_src_buf_.alloc(kSrcFrontPad + static_cast<size_t>(src_n) + kSrcBackPad))) {
            (void)hipMemsetAsync(d_src_buf_.bytes(), 0, kSrcFrontPad, stream);
            (void)hipMemsetAsync(d_src_buf_.bytes() + kSrcFrontPad + src_n, 0, kSrcBackPad, stream);
            if (src_n) ok = upload_staged(d_src_buf_.bytes() + kSrcFrontPad, host_payload_ + plan_.src_lo, static_cast<size_t>(src_n), stream);
            d_src_ = d_src_buf_.bytes() + kSrcFrontPad - plan_.src_lo;     // kernels address the payload by its offsets
        }
        ok = ok && d_blk_size_.upload(plan_.blk_size.data(), n_blocks_ * sizeof(uint32_t), stream) &&
             d_blk_base_.alloc((n_blocks_ + 1) * sizeof(uint64_t)) && d_scan_tmp_.alloc(scan_tmp_bytes(n_blocks_)) &&
             d_streams_.upload(plan_.streams.data(), n_streams_ * sizeof(HufStream), stream) &&
             d_tasks_.upload(plan_.tasks.data(), n_tasks_ * sizeof(HufTask), stream) &&
             d_tbl_copies_.upload(plan_.tbl_copies.data(), plan_.tbl_copies.size() * sizeof(HufTblCopy), stream) &&
             d_pool_.upload(plan_.huf_pool.data(), plan_.huf_pool.size() * sizeof(uint16_t), stream) &&
             d_dicts_.upload(plan_.dict_pool.data(), plan_.dict_pool.size(), stream) &&
             d_copies_.upload(plan_.copies.data(), n_copies_ * sizeof(CopyTask), stream) &&
             d_seq_blocks_.upload(plan_.seq_blocks.data(), n_seq_blocks_ * sizeof(SeqBlock), stream) &&
             d_cells_.upload(plan_.fse_pool.data(), plan_.fse_pool.size() * sizeof(SeqCell), stream) &&
             d_xxh_segs_.upload(xxh_segs_.data(), xxh_segs_.size() * sizeof(XxhSeg), stream);
    }
    ok = ok &&
         d_lit_.alloc(static_cast<size_t>(plan_.lit_bytes) + 64) &&
         d_seqs_.alloc_items(plan_.n_sequences, sizeof(Seq), 16) &&
         d_meta_.alloc_items(plan_.n_sequences, sizeof(SeqMeta), 16) &&
         d_rep_final_.alloc(n_seq_blocks_ * 12 + 16) && d_rep_init_.alloc(n_seq_blocks_ * 12 + 16) &&
         d_rep_scratch_.alloc((n_seq_blocks_ / 64 + 1) * 24 + 16) &&
         d_blk_pending_.alloc(n_seq_blocks_ * 4 + 16) &&
         (xxh_segs_.empty() || d_xxh_carry_.alloc(2 * sizeof(XxhCarry)));
    bool parts = false;                                        // streams in parts (plan.h: HufStream::sub): a HufSync record per part
    for (const HufClass &c : plan_.classes) parts = parts || c.split > 1;
    ok = ok && (!parts || d_huf_sync_.alloc(n_streams_ * sizeof(HufSync) + 16));
    if (!ok) return Failure::make(NAFGPU_E_DEVICE, "out of device memory while preparing a section");
    // the host vectors were consumed by asynchronous copies: keep them until the stream drains
    if (!hip_ok(hipStreamSynchronize(stream))) return Failure::make(NAFGPU_E_DEVICE, "upload of task lists failed");
    if (hook_env("NAFGPU_DEBUG_VERIFY_UPLOAD") && src_n) {     // (chasing a bad upload: the source bytes read back and compared)
        std::vector<uint8_t> back(static_cast<size_t>(src_n));
        (void)hipMemcpy(back.data(), d_src_ + plan_.src_lo, back.size(), hipMemcpyDeviceToHost);
        size_t bad = 0, first = 0, last = 0;
        for (size_t i = 0; i < back.size(); i++)
            if (back[i] != host_payload_[plan_.src_lo + i]) {
                if (!bad) first = i;
                last = i;
                bad++;
            }
        std::fprintf(stderr, "[nafgpu] tile %u: %zu source bytes on the device, %zu differ (first at %zu, last at %zu)\n", t, back.size(), bad, first, last);
        if (bad) {                                             // where in the payload do the wrong bytes come from?
            const size_t probe = ((first + 4096) & ~size_t(15));
            const void *hit = probe + 64 <= back.size() ? memmem(host_payload_, plan_.src_hi, back.data() + probe, 64) : nullptr;
            std::fprintf(stderr, "[nafgpu]   the 64 bytes at %zu are the payload's bytes at %lld; d_src_buf_ %p size %zu, slots %p %p\n", probe,
                         hit ? static_cast<long long>(static_cast<const uint8_t *>(hit) - host_payload_) : -1ll, static_cast<void *>(d_src_buf_.bytes()),
                         d_src_buf_.size(), static_cast<void *>(src_slot_[0].buf.bytes()), static_cast<void *>(src_slot_[1].buf.bytes()));
            {   // the first 64 KiB: runs of wrong bytes at 16-byte granularity, and where the first wrong 32 bytes come from
                size_t r0 = 0, shown = 0;
                bool w_in = false;
                for (size_t i = 0; i < std::min<size_t>(back.size(), 1 << 16); i += 16) {
                    const bool w = std::memcmp(back.data() + i, host_payload_ + plan_.src_lo + i, std::min<size_t>(16, back.size() - i)) != 0;
                    if (w && !w_in) r0 = i;
                    if (!w && w_in && shown++ < 10) std::fprintf(stderr, "[nafgpu]   wrong bytes %zu .. %zu\n", r0, i);
                    w_in = w;
                }
                const size_t m = first & ~size_t(15);
                const void *h2 = m + 32 <= back.size() ? memmem(host_payload_, plan_.src_hi, back.data() + m, 32) : nullptr;
                std::fprintf(stderr, "[nafgpu]   the 32 device bytes at %zu are the payload's at %lld; first bytes there %02x %02x %02x %02x, expected %02x %02x %02x %02x\n", m,
                             h2 ? static_cast<long long>(static_cast<const uint8_t *>(h2) - host_payload_) : -1ll, back[m], back[m + 1], back[m + 2], back[m + 3],
                             host_payload_[plan_.src_lo + m], host_payload_[plan_.src_lo + m + 1], host_payload_[plan_.src_lo + m + 2], host_payload_[plan_.src_lo + m + 3]);
            }
            size_t run0 = first, runs = 0;                     // the runs of wrong bytes, 1 MiB granularity
            bool in = false;
            for (size_t i = 0; i < back.size(); i += size_t(1) << 20) {
                const size_t e = std::min(back.size(), i + (size_t(1) << 20));
                const bool w = std::memcmp(back.data() + i, host_payload_ + plan_.src_lo + i, e - i) != 0;
                if (w && !in) run0 = i;
                if (!w && in && runs++ < 12) std::fprintf(stderr, "[nafgpu]   wrong from MiB %zu to %zu\n", run0 >> 20, i >> 20);
                in = w;
            }
            if (in) std::fprintf(stderr, "[nafgpu]   wrong from MiB %zu to the end\n", run0 >> 20);
        }
    }
    // only the counts are needed from here on
    std::vector<HufStream>().swap(plan_.streams);
    std::vector<HufTableRef>().swap(plan_.stream_ref);
    std::vector<uint16_t>().swap(plan_.huf_pool);
    std::vector<SeqCell>().swap(plan_.fse_pool);
    std::vector<CopyTask>().swap(plan_.copies);
    std::vector<uint64_t>().swap(plan_.blk_off);
    carry_same_frame_ = n_seq_blocks_ > 0 && carry_frame_ != 0xFFFFFFFFu && plan_.first_seq_frame == carry_frame_;
    loaded_tile_ = t;
    return Failure();
}

// Tiles are decoded in order: tile t needs the position, the repeat offsets and -- when the output is held one
// tile at a time -- the last window bytes the tiles before it left.
Failure SectionJob::decode_tile(uint32_t t, hipStream_t stream, StageTimer *timer, hipStream_t aux) {
    if (!ready_) return Failure();
    if (t == 0 && tiles_done_ != 0) {
        // a new pass over the section (nafgpu_decode_all_device re-runs the kernels on every call): everything the
        // tiles carry from one to the next goes back to its state in front of the first
        tiles_done_ = 0;
        tile_pos0_ = out0_;
        tile_len_ = 0;
        rep_carry_[0] = 1;
        rep_carry_[1] = 4;
        rep_carry_[2] = 8;
        carry_frame_ = 0xFFFFFFFFu;
        carry_same_frame_ = false;
        xxh_live_ = false;
    }
    if (t >= tiles_.size() || t != tiles_done_) return Failure::make(NAFGPU_E_INVALID_ARG, "tiles are decoded in order");
    const uint64_t mult = t_char_ ? 2 : 1;
    if (t > 0) {
        // ---- move on: the next tile starts where this one ended
        const uint64_t done = tile_pos0_ + tile_len_ - out0_;              // decoded so far in this process's range
        if (tiled_output() && halo_cap_) {                                 // keep the last window bytes in front of the tile area
            const uint64_t keep = std::min<uint64_t>(halo_cap_, done);
            uint8_t *base = d_out_.bytes();
            uint8_t *src = base + (halo_cap_ + tile_len_ - keep) * mult, *dst = base + (halo_cap_ - keep) * mult;
            if (tile_len_ >= keep) {
                (void)hipMemcpyAsync(dst, src, keep * mult, hipMemcpyDeviceToDevice, stream);
            } else if (tile_len_ > 0) {                                    // tiny tile: source and destination overlap
                if (!d_halo_tmp_.alloc_items(keep, mult)) return Failure::make(NAFGPU_E_DEVICE, "out of device memory");
                (void)hipMemcpyAsync(d_halo_tmp_.bytes(), src, keep * mult, hipMemcpyDeviceToDevice, stream);
                (void)hipMemcpyAsync(dst, d_halo_tmp_.bytes(), keep * mult, hipMemcpyDeviceToDevice, stream);
            }
        }
        tile_pos0_ += tile_len_;
        tile_len_ = 0;
    }
    Failure f = load_tile(t, stream);
    if (!f.ok()) return f;
    if (tiled_output() && halo_elems_ + static_cast<uint64_t>(n_blocks_ - plan_.halo) * kBlockMax > halo_cap_ + tile_cap_)
        return Failure::make(NAFGPU_E_DEVICE, "internal: tile larger than its buffer");
    run(stream, timer, aux);
    // what the tile decoded to, the repeat offsets it ends with, its status
    uint64_t total = 0;
    uint32_t rep[3] = {1, 4, 8};
    if (!hip_ok(hipMemcpyAsync(&total, d_blk_base_.as<uint64_t>() + n_blocks_, sizeof total, hipMemcpyDeviceToHost, stream)) ||
        !hip_ok(hipMemcpyAsync(rep, d_counters_.bytes() + 64, sizeof rep, hipMemcpyDeviceToHost, stream)))
        return Failure::make(NAFGPU_E_DEVICE, "tile read-back failed");
    f = check(stream);                                                     // synchronises
    if (!f.ok()) return f;
    tile_len_ = total - halo_elems_;
    if (n_seq_blocks_) {
        for (int k = 0; k < 3; k++) rep_carry_[k] = rep[k];
        carry_frame_ = plan_.last_seq_frame;
    }
    tiles_done_ = t + 1;
    if (tile_pos0_ + tile_len_ > out1_) return Failure::io(NAFGPU_IO_INVALID_DATA, status_text(kStSizeMismatch));
    if (tiled_output() && t + 1 < tiles_.size() &&
        master_.blk_off[tiles_[t + 1].b1] - master_.blk_off[tiles_[t + 1].b0] >= src_prefetch_min())
        (void)start_source_upload(t + 1);                                   // (a failure shows when the tile is loaded)
    if (tiles_done_ == tiles_.size()) out1_ = tile_pos0_ + tile_len_;                  // (less than announced: see prepare)
    return Failure();
}

// Device address of the loaded selection's local position 0 (its pseudo block when it has one): kernels
// address the output as base + blk_base[block] + offset inside the block.
uint8_t *SectionJob::tile_out_base() const {
    const uint64_t mult = t_char_ ? 2 : 1;
    if (tiled_output() || proto_lz_) return d_out_.bytes() + (halo_cap_ - halo_elems_) * mult;
    return d_out_.bytes() + (tile_pos0_ - halo_elems_ - out0_) * mult;
}

const uint8_t *SectionJob::tile_data() const {
    const uint64_t mult = t_char_ ? 2 : 1;
    return tile_out_base() + halo_elems_ * mult;
}

SectionJob::~SectionJob() {
    drain();
    if (prefetch_stream_) (void)hipStreamDestroy(prefetch_stream_);
    if (ev_fork_) (void)hipEventDestroy(ev_fork_);
    if (ev_join_) (void)hipEventDestroy(ev_join_);
    if (ev_early_fork_) (void)hipEventDestroy(ev_early_fork_);
    if (ev_early_join_) (void)hipEventDestroy(ev_early_join_);
    if (ev_k2_) (void)hipEventDestroy(ev_k2_);
}

void SectionJob::run_k2_ahead(hipStream_t st) {
    if (!ready_ || !n_seq_blocks_) return;
    if (!ev_k2_ && !hip_ok(hipEventCreateWithFlags(&ev_k2_, hipEventDisableTiming))) return;
    uint32_t *status = d_status_.as<uint32_t>();
    (void)hipMemsetAsync(status, 0, 64, st);
    launch_seq_decode(st, d_src_, d_seq_blocks_.as<SeqBlock>(), static_cast<uint32_t>(n_seq_blocks_), d_cells_.as<SeqCell>(),
                      d_seqs_.as<Seq>(), d_meta_.as<SeqMeta>(), d_blk_size_.as<uint32_t>(), d_rep_final_.as<uint32_t>(), status,
                      cells_cap_, static_cast<long long>(plan_.src_lo) - static_cast<long long>(kSrcFrontPad));
    k2_ahead_ = hip_ok(hipEventRecord(ev_k2_, st));
    if (!k2_ahead_) (void)hipStreamSynchronize(st);         // (cannot happen; run() then simply does it again)
}

void SectionJob::run(hipStream_t stream, StageTimer *timer, hipStream_t aux) {
    if (!ready_) return;
    bool early = false;
    run_front(stream, timer, aux, &early);
    run_back(stream, timer, aux, early, 0);
}

// What needs nothing but the compressed bytes: the status reset, K1's classes bound for the literal buffer (on `aux`), K2.
void SectionJob::run_front(hipStream_t stream, StageTimer *timer, hipStream_t aux, bool *early_out) {
    uint32_t *status = d_status_.as<uint32_t>();
    const bool k2_done = k2_ahead_ && hip_ok(hipStreamWaitEvent(stream, ev_k2_, 0));
    k2_ahead_ = false;
    if (!k2_done) (void)hipMemsetAsync(status, 0, 64, stream);
    {   // streams in parts (plan.h: HufStream::sub): where the parts begin -- once, in front of every class of k_huf_decode
        uint32_t sync_lds = 0;
        bool parts = false;
        for (const HufClass &c : classes_) {
            parts = parts || c.split > 1;
            sync_lds = std::max(sync_lds, c.sync_lds);
        }
        if (parts) {
            if (timer) timer->begin(stream, StageTimer::kHuf);
            launch_huf_parts(stream, d_src_, d_tasks_.as<HufTask>(), static_cast<uint32_t>(n_tasks_), d_tbl_copies_.as<HufTblCopy>(),
                             d_streams_.as<HufStream>(), d_pool_.as<uint16_t>(), d_huf_sync_.as<HufSync>(), sync_lds, status);
            if (timer) timer->end(stream);
        }
    }
    // Streams bound for the literal buffer need nothing from K2 (their destinations are the plan's): they start on `aux`
    // now, beside k_seq_states -- a chain per block that keeps one wave per CU busy and leaves the rest of the chip idle.
    bool early = false;
    if (n_seq_blocks_ && aux && !hook_env("NAFGPU_NO_EARLY_K1")) {   // (the hook: K2 with the chip to itself, for traces)
        bool any = false;
        for (const HufClass &c : classes_) any = any || c.to_lit;
        if (any) {
            if (!ev_early_fork_) (void)hipEventCreateWithFlags(&ev_early_fork_, hipEventDisableTiming);
            if (!ev_early_join_) (void)hipEventCreateWithFlags(&ev_early_join_, hipEventDisableTiming);
            early = ev_early_fork_ && ev_early_join_ && hip_ok(hipEventRecord(ev_early_fork_, stream)) &&
                    hip_ok(hipStreamWaitEvent(aux, ev_early_fork_, 0));
            if (early) {
                for (const HufClass &c : classes_)
                    if (c.to_lit)
                        launch_huf_decode(aux, d_src_, d_tasks_.as<HufTask>(), c, d_tbl_copies_.as<HufTblCopy>(), d_streams_.as<HufStream>(),
                                          d_pool_.as<uint16_t>(), d_blk_base_.as<uint64_t>(), nullptr, d_lit_.bytes(), d_seq_blocks_.as<SeqBlock>(),
                                          d_seqs_.as<Seq>(), d_dicts_.bytes(), t_char_ != 0, t_char_, status);
                (void)hipEventRecord(ev_early_join_, aux);
            }
        }
    }
    if (n_seq_blocks_ && !k2_done) {
        if (timer) timer->begin(stream, StageTimer::kSeqLz);
        // blk_size of blocks with sequences is rewritten in full by k_seq_values: re-runs are idempotent
        launch_seq_decode(stream, d_src_, d_seq_blocks_.as<SeqBlock>(), static_cast<uint32_t>(n_seq_blocks_),
                          d_cells_.as<SeqCell>(), d_seqs_.as<Seq>(), d_meta_.as<SeqMeta>(), d_blk_size_.as<uint32_t>(),
                          d_rep_final_.as<uint32_t>(),
                          status, cells_cap_, static_cast<long long>(plan_.src_lo) - static_cast<long long>(kSrcFrontPad));
        if (timer) timer->end(stream);
    }
    *early_out = early;
}

// The arguments of K4 for the loaded selection (phase: see LzArgs); allocates the scratch of the chosen route.
void SectionJob::fill_lz_args(LzArgs *out, uint32_t phase) {
    uint32_t *status = d_status_.as<uint32_t>();
    LzArgs la{};
    la.blocks = d_seq_blocks_.as<SeqBlock>();
    la.n_blocks = static_cast<uint32_t>(n_seq_blocks_);
    la.n_sequences = plan_.n_sequences;
    la.n_elems = proto_lz_ ? halo_elems_ + decoded_
                           : (tiles_.size() == 1 ? expect_ : halo_elems_ + static_cast<uint64_t>(n_blocks_ - plan_.halo) * kBlockMax);   // (tile: an upper bound)
    la.seqs = d_seqs_.as<Seq>();
    la.lit = d_lit_.bytes();
    la.blk_base = d_blk_base_.as<uint64_t>();
    la.rep_final = d_rep_final_.as<uint32_t>();
    la.rep_init = d_rep_init_.as<uint32_t>();
    la.rep_scratch = d_rep_scratch_.as<uint32_t>();
    // repeat offsets in front of the first block: the frame's initial ones, what the tile before left behind, or
    // (shard protocol) what the ranks in front leave behind
    const bool continues = plan_.first_frame_continues && loaded_tile_ > 0 && carry_same_frame_;
    la.rep_continues = plan_.first_frame_continues ? 1u : 0u;
    for (int k = 0; k < 3; k++) la.rep_carry[k] = proto_lz_ ? shard_carry_[k] : (continues ? rep_carry_[k] : (k == 0 ? 1u : (k == 1 ? 4u : 8u)));
    la.rep_out = reinterpret_cast<uint32_t *>(d_counters_.bytes() + 64);
    la.meta = d_meta_.as<SeqMeta>();
    la.blk_pending = d_blk_pending_.as<uint32_t>();
    la.out = tile_out_base();
    la.t_char = t_char_;
    la.status = status;
    la.counters = d_counters_.as<unsigned long long>();
    la.phase = phase;
    la.n_sel_blocks = static_cast<uint32_t>(n_blocks_);
    la.halo_wait = proto_lz_ ? halo_elems_ : 0;
    la.tail_elems = send_elems_;
    // Dense or sparse?  Where the matches are a good part of the output (level-3 DNA, quality strings) the frame is
    // swept element by element (one word of scratch per output element, allocated on the first run and kept);
    // a handful of matches in gigabytes of literals (real genomes at level 1) are visited one by one.
    const uint64_t sec_known = sec_known_, sec_seqs = sec_seqs_;                            // (the whole section's figures: every tile decides alike)
    const uint64_t match_elems = expect_ > sec_known ? expect_ - sec_known : 0;             // known_out = everything but the match bytes
    const char *force = hook_env("NAFGPU_LZ_MODE");                                         // tests: "dense" / "sparse"
    // Swept: matches are a fifth of the output or more (level-3 DNA, a quarter of it matches at random distances: 16.3 ms
    // swept against 17.5 match by match) and there is enough of them to pay for the sweeps -- many sequences, or a
    // megabyte of match bytes in a few long ones (the Length section of equal-length reads is one block-long run per
    // block, each copying from the block before: a chain no fixed number of passes gets through).
    bool dense = (sec_seqs >= 4096 || match_elems >= (1u << 20)) && match_elems * 5 >= expect_;
    if (force) dense = force[0] == 'd';
    lz_dense_ = false;
    if (dense && d_pj_dist_.alloc_items(la.n_elems, sizeof(uint32_t), 64) && d_pj_tiles_.alloc_items(lz_pj_tiles(la.n_elems), sizeof(uint32_t), 64)) {
        la.pj_dist = d_pj_dist_.as<uint32_t>();
        la.pj_tiles = d_pj_tiles_.as<uint32_t>();
        // the late sweeps work from lists of what is pending (a third of the elements at most); without memory for
        // them the sweeps simply go on over all of D
        la.pj_list_cap = la.n_elems / 3 + 4096;
        if (la.n_elems < (1ull << 32) && d_pj_list_[0].alloc_items(la.pj_list_cap, sizeof(uint32_t), 64) &&
            d_pj_list_[1].alloc_items(la.pj_list_cap, sizeof(uint32_t), 64)) {
            la.pj_list[0] = d_pj_list_[0].as<uint32_t>();
            la.pj_list[1] = d_pj_list_[1].as<uint32_t>();
        }
        // Strip-wise sweeps where chains are shallow: text / qualities (their matches come from near by) with a tenth of the
        // elements or more literals.  (Measured on the 10 M-read probe: level 1, 22 % literals, 44.6 -> 38.0 ms; level 3, 2 %
        // literals -- chains forty links deep -- 90.8 -> 100 ms.)
        // ... and the first sweep lists what it leaves pending where that will be little: half of the elements or more literals
        // (level-3 DNA, 75 %: 7.1 -> 6.6 ms; the probe's level-1 reads and qualities, 22 %: 33.5 -> 37.5 ms, so not those)
        la.shallow = sec_known * 2 >= expect_ ? 1u : 0u;
        la.strips = (!t_char_ && sec_known * 10 >= expect_) ? 1u : 0u;   // (nucleotides, strip-wise: 33.8 against 34.6 ms on the probe's reads, 8.10 against 7.94 on level-3 DNA)
        if (const char *e = hook_env("NAFGPU_PJ_STRIPS")) la.strips = la.shallow = e[0] == '1' ? 1u : 0u;   // (tests: both, on any data)
        lz_dense_ = true;
    } else {
        // the pending lists are an accelerator: without memory for them every pass walks the blocks
        const bool lists = plan_.n_sequences < (1ull << 40) && n_seq_blocks_ < (1ull << 24) &&
                           d_lz_list_[0].alloc_items(plan_.n_sequences, sizeof(uint64_t), 16) &&
                           d_lz_list_[1].alloc_items(plan_.n_sequences, sizeof(uint64_t), 16);
        la.plist[0] = lists ? d_lz_list_[0].as<uint64_t>() : nullptr;
        la.plist[1] = lists ? d_lz_list_[1].as<uint64_t>() : nullptr;
        la.roff = d_roff_.alloc_items(plan_.n_sequences, sizeof(uint32_t), 16) ? d_roff_.as<uint32_t>() : nullptr;
        // the index of "first sequence at or after every 128th element" pays when sequences are everywhere; a few
        // thousand sequences in gigabytes of output are found by binary search
        const bool index = plan_.n_sequences < 0xFFFFFFFFull && plan_.n_sequences * 1024 >= la.n_elems &&
                           d_lz_index_.alloc_items((la.n_elems >> 7) + 2, sizeof(uint32_t));
        la.cidx = index ? d_lz_index_.as<uint32_t>() : nullptr;
        la.n_idx_chunks = (la.n_elems >> 7) + 2;
    }
    *out = la;
}

// Everything that addresses the output: the block bases, raw / RLE copies, K1's remaining classes, K4, frame checksums.
// phase: 0 = the selection is complete in itself; 1 = (shard protocol) the window in front of it has not arrived yet.
void SectionJob::run_back(hipStream_t stream, StageTimer *timer, hipStream_t aux, bool early, uint32_t phase) {
    uint32_t *status = d_status_.as<uint32_t>();
    auto launch_class = [&](const HufClass &c, hipStream_t st, uint8_t *out_base) {
        launch_huf_decode(st, d_src_, d_tasks_.as<HufTask>(), c, d_tbl_copies_.as<HufTblCopy>(), d_streams_.as<HufStream>(),
                          d_pool_.as<uint16_t>(), d_blk_base_.as<uint64_t>(), out_base, d_lit_.bytes(), d_seq_blocks_.as<SeqBlock>(),
                          d_seqs_.as<Seq>(), d_dicts_.bytes(), t_char_ != 0, t_char_, status);
    };
    if (timer) timer->begin(stream, StageTimer::kOther);
    // total of the selection's block sizes: known for the whole section and for ranges without LZ sequences; a tile
    // of a section with sequences is added up by the host (decode_tile), a shard's by the ranks together (shard_place)
    const uint64_t expect_sel = tiles_.size() == 1 && !sharded_ ? expect_
                                : (!has_lz_ ? halo_elems_ + plan_.known_out : ~0ull);
    launch_scan_blocks(stream, d_blk_size_.as<uint32_t>(), n_blocks_, d_blk_base_.as<uint64_t>(), d_scan_tmp_.bytes(),
                       expect_sel, status);
    const bool ascii = t_char_ != 0;
    // kernels address the output as base + position inside the loaded selection
    uint8_t *const out_base = tile_out_base();
    launch_copy_fill(stream, d_src_, d_copies_.as<CopyTask>(), static_cast<uint32_t>(n_copies_),
                     d_blk_base_.as<uint64_t>(), out_base, d_lit_.bytes(), ascii, t_char_, status);
    if (timer) timer->end(stream);
    // K1, one launch per class.  Streams that write the section output (blocks without sequences, and -- segment by
    // segment -- blocks with a few) and streams that feed the literal buffer (blocks with many sequences).
    // The classes are independent of each other.  The one with the most tasks goes to `stream`, the others to `aux`
    // beside it: a class of a few tasks (the tail of a section; streams bound for the literal buffer) takes a whole
    // task's time -- 2.5 ms on one CU for 64 streams -- which back to back was a fifth of a real-genome decode, and
    // two large classes share the chip instead of each ending in a half-empty tail.  K4 waits for both.
    std::vector<const HufClass *> rest;                     // (what did not start early)
    for (const HufClass &c : classes_)
        if (!(early && c.to_lit)) rest.push_back(&c);
    if (!rest.empty()) {
        if (timer) timer->begin(stream, StageTimer::kHuf);
        size_t big = 0;
        for (size_t c = 1; c < rest.size(); c++)
            if (rest[c]->n_tasks > rest[big]->n_tasks) big = c;
        bool forked = false;
        if (rest.size() > 1 && aux) {
            if (!ev_fork_) (void)hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming);
            if (!ev_join_) (void)hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming);
            forked = ev_fork_ && ev_join_ && hip_ok(hipEventRecord(ev_fork_, stream)) && hip_ok(hipStreamWaitEvent(aux, ev_fork_, 0));
        }
        if (forked) {
            for (size_t c = 0; c < rest.size(); c++)
                if (c != big) launch_class(*rest[c], aux, out_base);
            (void)hipEventRecord(ev_join_, aux);
            launch_class(*rest[big], stream, out_base);
            (void)hipStreamWaitEvent(stream, ev_join_, 0);
        } else {
            for (const HufClass *c : rest) launch_class(*c, stream, out_base);
        }
        if (timer) timer->end(stream);
    }
    if (early) (void)hipStreamWaitEvent(stream, ev_early_join_, 0);   // the literal buffer is complete from here on
    lz_args_valid_ = false;
    if (n_seq_blocks_) {
        if (timer) timer->begin(stream, StageTimer::kSeqLz);
        fill_lz_args(&la_, phase);
        la_ascii_ = ascii;
        if (!la_.pj_dist && !la_.roff) {                       // (cannot happen short of a device out of memory: flag the section)
            (void)hipMemsetAsync(status, 0xFF, 4, stream);
        } else {
            launch_lz_execute(stream, la_, ascii);
            lz_args_valid_ = true;
        }
        if (timer) timer->end(stream);
    }
    if (!xxh_segs_.empty()) {                                  // Content_Checksum of the frames that carry one
        if (timer) timer->begin(stream, StageTimer::kOther);
        XxhCarry *carry = d_xxh_carry_.as<XxhCarry>();
        const uint32_t t = loaded_tile_ & 1u;
        launch_xxh64_frames(stream, d_xxh_segs_.as<XxhSeg>(), static_cast<uint32_t>(xxh_segs_.size()), d_blk_base_.as<uint64_t>(),
                            out_base, ascii, t_char_, carry + t, carry + (t ^ 1u), status);
        if (timer) timer->end(stream);
    }
}

// ------------------------------------------------------------------ the shard protocol (one section)
void SectionJob::range_of(uint32_t rank, uint32_t *b0, uint32_t *b1) const {
    *b0 = ranges_[rank].first;
    *b1 = ranges_[rank].second;
}

// The repeat offsets rank `rank`'s first block with sequences inherits: {1, 4, 8} at the start of a frame, else what
// the last block with sequences in front of it -- same frame, some rank before -- leaves behind: that rank's map
// applied to what IT inherits.
void SectionJob::carry_for(uint32_t rank, const ShardSummary *all, uint32_t out[3]) const {
    out[0] = 1;
    out[1] = 4;
    out[2] = 8;
    const uint32_t b0 = ranges_[rank].first, b1 = ranges_[rank].second;
    const size_t k = std::lower_bound(seq_blk_.begin(), seq_blk_.end(), b0) - seq_blk_.begin();   // its first block with sequences
    if (k == seq_blk_.size() || seq_blk_[k] >= b1 || k == 0) return;
    if (seq_frame_[k - 1] != seq_frame_[k]) return;          // that block begins its frame's sequences
    uint32_t q = rank;                                       // the rank that holds the block with sequences in front of it
    while (q > 0) {
        q--;
        if (seq_blk_[k - 1] >= ranges_[q].first && seq_blk_[k - 1] < ranges_[q].second) break;
    }
    uint32_t in[3];
    carry_for(q, all, in);
    bool bad = false;
    for (int j = 0; j < 3; j++) out[j] = rep_compose(all[q].rep_map[j], in, &bad);
    if (bad) out[0] = out[1] = out[2] = 0;                   // (corrupt: an offset of zero is flagged by the kernels)
}

void SectionJob::shard_begin(hipStream_t stream, StageTimer *timer, hipStream_t aux) {
    if (!ready_) return;
    if (!proto_lz_) return;                                  // everything about this range is known from the walk: shard_place runs it
    // the pseudo block in front has no size yet: positions are relative to the range's first element
    halo_elems_ = 0;
    halo_word_ = 0;
    if (plan_.halo) (void)hipMemcpyAsync(d_blk_size_.bytes(), &halo_word_, 4, hipMemcpyHostToDevice, stream);
    begin_early_ = false;
    run_front(stream, timer, aux, &begin_early_);            // (run_back waits for the classes that started early)
    uint32_t *status = d_status_.as<uint32_t>();
    launch_scan_blocks(stream, d_blk_size_.as<uint32_t>(), n_blocks_, d_blk_base_.as<uint64_t>(), d_scan_tmp_.bytes(), ~0ull, status);
    if (n_seq_blocks_)
        launch_rep_map(stream, d_seq_blocks_.as<SeqBlock>(), static_cast<uint32_t>(n_seq_blocks_), d_rep_final_.as<uint32_t>(),
                       d_rep_scratch_.as<uint32_t>(), plan_.first_frame_continues ? 1u : 0u,
                       reinterpret_cast<uint32_t *>(d_counters_.bytes() + 192), status);
}

Failure SectionJob::shard_summary(hipStream_t stream, ShardSummary *mine) {
    *mine = ShardSummary();
    if (!ready_) return Failure();
    const uint32_t b0 = ranges_[opt_.shard_rank].first, b1 = ranges_[opt_.shard_rank].second;
    // the last frame that begins inside the range
    uint32_t last_frame = 0xFFFFFFFFu;
    for (uint32_t f : frame_first_)
        if (f >= b0 && f < b1) last_frame = f;
    if (!proto_lz_) {
        mine->decoded = out1_ - out0_;
        mine->frame_tail = mine->decoded;                    // (no sequences anywhere: nobody looks at it)
        return Failure();
    }
    uint64_t total = 0, at_frame = 0;
    uint32_t map[3] = {mine->rep_map[0], mine->rep_map[1], mine->rep_map[2]};
    bool ok = hip_ok(hipMemcpyAsync(&total, d_blk_base_.as<uint64_t>() + n_blocks_, 8, hipMemcpyDeviceToHost, stream));
    if (last_frame != 0xFFFFFFFFu)
        ok = ok && hip_ok(hipMemcpyAsync(&at_frame, d_blk_base_.as<uint64_t>() + (last_frame - b0 + plan_.halo), 8, hipMemcpyDeviceToHost, stream));
    if (n_seq_blocks_) ok = ok && hip_ok(hipMemcpyAsync(map, d_counters_.bytes() + 192, sizeof map, hipMemcpyDeviceToHost, stream));
    if (!ok) {
        (void)hipStreamSynchronize(stream);                  // (copies already enqueued write into this frame's variables)
        return Failure::make(NAFGPU_E_DEVICE, "shard summary read-back failed");
    }
    Failure f = check(stream);                               // synchronises
    if (!f.ok()) {
        mine->failed = true;
        return f;
    }
    decoded_ = total;
    mine->decoded = total;
    mine->frame_tail = last_frame != 0xFFFFFFFFu ? total - at_frame : total;
    for (int k = 0; k < 3; k++) mine->rep_map[k] = map[k];
    return Failure();
}

Failure SectionJob::shard_place(const ShardSummary *all, uint32_t n_ranks, hipStream_t stream, StageTimer *timer, hipStream_t aux) {
    if (!ready_) return Failure();
    if (n_ranks != opt_.shard_count || ranges_.size() != n_ranks) return Failure::make(NAFGPU_E_INVALID_ARG, "shard summaries of another world size");
    const uint32_t me = opt_.shard_rank;
    for (uint32_t r = 0; r < n_ranks; r++)
        if (all[r].failed) return Failure::io(NAFGPU_IO_INVALID_DATA, "zstd: another rank could not decode its part of the section");
    if (!proto_lz_) {                                        // a section without sequences: ranges known from the walk, nothing to wait for
        if (tiles_.size() == 1) {
            bool early = false;
            run_front(stream, timer, aux, &early);
            run_back(stream, timer, aux, early, 0);
            return Failure();
        }
        for (uint32_t t = 0; t < tiles_.size(); t++) {       // (a range that does not fit the device beside its output: tile after tile)
            Failure f = decode_tile(t, stream, timer, aux);
            if (!f.ok()) return f;
        }
        return Failure();
    }
    uint64_t before = 0, sum = 0;
    for (uint32_t r = 0; r < n_ranks; r++) {
        if (r < me) before += all[r].decoded;
        sum += all[r].decoded;
    }
    if (sum > expect_) return Failure::io(NAFGPU_IO_INVALID_DATA, status_text(kStSizeMismatch));   // (less: see prepare)
    out0_ = before;
    out1_ = before + decoded_;
    tile_pos0_ = out0_;
    tile_len_ = decoded_;
    // how much of the window in front of rank r exists: the decoded elements between the start of the frame its first
    // block continues and that block -- nothing when the block begins a frame
    auto window_of = [&](uint32_t r) -> uint64_t {
        if (r == 0 || r >= n_ranks) return 0;
        const uint32_t rb0 = ranges_[r].first;
        if (rb0 >= master_blocks_) return 0;                 // (an empty range at the end)
        for (uint32_t f : frame_first_)
            if (f == rb0) return 0;
        uint64_t avail = 0;
        for (uint32_t q = r; q-- > 0;) {
            const uint32_t qb0 = ranges_[q].first, qb1 = ranges_[q].second;
            bool starts = false;
            for (uint32_t f : frame_first_) starts = starts || (f >= qb0 && f < qb1);
            avail += starts ? all[q].frame_tail : all[q].decoded;
            if (starts) break;
        }
        return std::min<uint64_t>(avail, halo_cap_ ? halo_cap_ : std::min<uint64_t>(std::max<uint64_t>(plan_.window_max, 1), expect_));
    };
    halo_elems_ = plan_.halo ? window_of(me) : 0;
    send_elems_ = window_of(me + 1);
    if (send_elems_ > halo_elems_ + decoded_) return Failure::make(NAFGPU_E_INVALID_ARG, "internal: the next rank's window exceeds what this rank holds");
    carry_for(me, all, shard_carry_);
    halo_word_ = static_cast<uint32_t>(halo_elems_);
    if (plan_.halo) (void)hipMemcpyAsync(d_blk_size_.bytes(), &halo_word_, 4, hipMemcpyHostToDevice, stream);
    halo_pending_ = halo_elems_ != 0;
    run_back(stream, timer, aux, begin_early_, 1);
    return Failure();
}

Failure SectionJob::tail_ready(hipStream_t stream, bool *ready) {
    *ready = true;
    if (!ready_ || !proto_lz_ || !send_elems_) return Failure();
    if (!halo_pending_) {                                    // nothing was left waiting (or it has been finished since)
        if (!hip_ok(hipStreamSynchronize(stream))) return Failure::make(NAFGPU_E_DEVICE, "device failure in the shard protocol");
        return Failure();
    }
    unsigned long long flag = 0;
    if (n_seq_blocks_ && lz_args_valid_) {
        if (!hip_ok(hipMemcpyAsync(&flag, d_counters_.as<unsigned long long>() + 19, 8, hipMemcpyDeviceToHost, stream)) ||
            !hip_ok(hipStreamSynchronize(stream)))
            return Failure::make(NAFGPU_E_DEVICE, "device failure in the shard protocol");
    }
    // (a range smaller than the next rank's window passes part of its own window on)
    *ready = flag == 0 && send_elems_ <= decoded_;
    return Failure();
}

Failure SectionJob::export_tail(void *dst, uint64_t n, hipStream_t stream) {
    if (!ready_ || !proto_lz_ || n != tail_send_bytes()) return Failure::make(NAFGPU_E_INVALID_ARG, "tail size differs from what the next rank waits for");
    if (!n) return Failure();
    const uint64_t mult = t_char_ ? 2 : 1;
    const uint8_t *end = d_out_.bytes() + out_shift_ + decoded_ * mult;
    if (!hip_ok(hipMemcpyAsync(dst, end - n, n, hipMemcpyDefault, stream)) || !hip_ok(hipStreamSynchronize(stream)))
        return Failure::make(NAFGPU_E_DEVICE, "copy of the tail failed");
    return Failure();
}

Failure SectionJob::import_halo(const void *src, uint64_t n, hipStream_t stream, StageTimer *timer) {
    if (!ready_ || !proto_lz_ || n != halo_recv_bytes()) return Failure::make(NAFGPU_E_INVALID_ARG, "window size differs from what this rank waits for");
    if (!n) return Failure();
    if (!hip_ok(hipMemcpyAsync(tile_out_base(), src, n, hipMemcpyDefault, stream))) return Failure::make(NAFGPU_E_DEVICE, "copy of the window failed");
    if (halo_pending_ && n_seq_blocks_ && lz_args_valid_) {
        if (timer) timer->begin(stream, StageTimer::kSeqLz);
        la_.phase = 2;
        launch_lz_execute(stream, la_, la_ascii_);
        if (timer) timer->end(stream);
    }
    halo_pending_ = false;
    // (the source buffer is the caller's: it may go as soon as this returns)
    if (!hip_ok(hipStreamSynchronize(stream))) return Failure::make(NAFGPU_E_DEVICE, "device failure while finishing a shard");
    return Failure();
}

// slot layout: [0, 8) status words, [8, 16) what the section decoded to, [16, 80) the LZ counters
bool SectionJob::check_begin(hipStream_t stream, uint8_t *slot) {
    if (!ready_) return true;
    std::memset(slot, 0, kCheckSlotBytes);
    bool ok = hip_ok(hipMemcpyAsync(slot, d_status_.bytes(), 8, hipMemcpyDeviceToHost, stream));
    if (has_lz_ && tiles_.size() == 1 && !sharded_)
        ok = ok && hip_ok(hipMemcpyAsync(slot + 8, d_blk_base_.as<uint64_t>() + n_blocks_, 8, hipMemcpyDeviceToHost, stream));
    if (n_seq_blocks_) ok = ok && hip_ok(hipMemcpyAsync(slot + 16, d_counters_.bytes(), 64, hipMemcpyDeviceToHost, stream));
    return ok;
}

Failure SectionJob::check_end(const uint8_t *slot) {
    if (!ready_) return Failure();
    uint32_t st[2];
    uint64_t total = 0;
    unsigned long long cnt[8];
    std::memcpy(st, slot, sizeof st);
    std::memcpy(&total, slot + 8, sizeof total);
    std::memcpy(cnt, slot + 16, sizeof cnt);
    // a section with LZ sequences may decode to less than the archive announces (prepare): what there is, is what counts
    if (has_lz_ && tiles_.size() == 1 && !sharded_ && st[0] == 0 && total <= expect_) out1_ = out0_ + total;
    lz_residue_ = 0;
    if (n_seq_blocks_) lz_residue_ = lz_dense_ ? plan_.n_sequences : cnt[1];   // statistics: matches the launched passes did not finish
    if (st[0] == kStInternal)
        return Failure::io(NAFGPU_IO_INVALID_DATA, std::string(status_text(st[0])) + " (detail " + std::to_string(st[1]) + ")");
    if (st[0] != 0) return Failure::io(NAFGPU_IO_INVALID_DATA, status_text(st[0]));
    return Failure();
}

Failure SectionJob::check(hipStream_t stream) {
    if (!ready_) return Failure();
    uint32_t st[2] = {0, 0};
    if (!hip_ok(hipMemcpyAsync(st, d_status_.bytes(), sizeof st, hipMemcpyDeviceToHost, stream)) ||
        !hip_ok(hipStreamSynchronize(stream)))
        return Failure::make(NAFGPU_E_DEVICE, "device status read-back failed");
    // a section with LZ sequences may decode to less than the archive announces (prepare): what there is, is what counts
    if (has_lz_ && tiles_.size() == 1 && !sharded_ && st[0] == 0) {
        uint64_t total = 0;
        if (hip_ok(hipMemcpyAsync(&total, d_blk_base_.as<uint64_t>() + n_blocks_, sizeof total, hipMemcpyDeviceToHost, stream)) &&
            hip_ok(hipStreamSynchronize(stream)) && total <= expect_)
            out1_ = out0_ + total;
    }
    lz_residue_ = 0;
    if (n_seq_blocks_) {                                       // statistics: matches the launched passes did not finish
        unsigned long long cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hip_ok(hipMemcpyAsync(cnt, d_counters_.bytes(), sizeof cnt, hipMemcpyDeviceToHost, stream)) && hip_ok(hipStreamSynchronize(stream)))
            lz_residue_ = lz_dense_ ? plan_.n_sequences : cnt[1];
    }
    if (st[0] == kStInternal)
        return Failure::io(NAFGPU_IO_INVALID_DATA, std::string(status_text(st[0])) + " (detail " + std::to_string(st[1]) + ")");
    if (st[0] != 0) return Failure::io(NAFGPU_IO_INVALID_DATA, status_text(st[0]));
    return Failure();
}

// ------------------------------------------------------------------ ArchiveJob
namespace {
// Streams outlive decoders.  hipStreamCreate takes 2.7 ms and hipStreamDestroy 2.4 ms on this stack (rocprofv3 --hip-trace of
// tools/small_probe.py: 80 % of the 14-18 ms that opening, decoding and closing ONE of the reference's fixtures took, whatever
// its size -- a decoder owns three streams), so a closed decoder hands its streams to the next one on the same device.  A
// stream is idle when it comes back (synchronised), a few per device are kept, and the pool is never torn down: at process
// exit the runtime may be gone before a static destructor runs.
class StreamPool {
public:
    static StreamPool &instance() {
        static StreamPool *pool = new StreamPool;
        return *pool;
    }
    hipStream_t get(int device) {                          // the current device is `device`
        {
            std::lock_guard<std::mutex> lock(mu_);
            std::vector<hipStream_t> &v = idle_[device];
            if (!v.empty()) {
                hipStream_t s = v.back();
                v.pop_back();
                return s;
            }
        }
        hipStream_t s = nullptr;
        if (!hip_ok(hipStreamCreate(&s))) return nullptr;
        return s;
    }
    void put(int device, hipStream_t s) {
        if (!s) return;
        (void)hipStreamSynchronize(s);
        {
            std::lock_guard<std::mutex> lock(mu_);
            std::vector<hipStream_t> &v = idle_[device];
            if (v.size() < kKeep) {
                v.push_back(s);
                return;
            }
        }
        (void)hipStreamDestroy(s);
    }

private:
    static constexpr size_t kKeep = 12;
    std::mutex mu_;
    std::map<int, std::vector<hipStream_t>> idle_;
};
}  // namespace

ArchiveJob::~ArchiveJob() {
    if (stream_ || aux_stream_ || k2_stream_) (void)hipSetDevice(device_);
    if (ev_fork_) (void)hipEventDestroy(ev_fork_);
    if (ev_join_) (void)hipEventDestroy(ev_join_);
    StreamPool::instance().put(device_, aux_stream_);
    StreamPool::instance().put(device_, k2_stream_);
    StreamPool::instance().put(device_, stream_);
}

Failure ArchiveJob::init(int device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (!hip_ok(e) || count <= 0)
        return Failure::make(NAFGPU_E_DEVICE, std::string("no HIP device available: libnafgpu decodes on the GPU only (hipGetDeviceCount: ") +
                                                  hipGetErrorString(e) + ", " + std::to_string(count) + " devices)");
    if (device < 0) {
        if (!hip_ok(hipGetDevice(&device))) device = 0;
    }
    if (device >= count) return Failure::make(NAFGPU_E_INVALID_ARG, "device ordinal out of range");
    e = hipSetDevice(device);
    if (!hip_ok(e)) return dev_fail("hipSetDevice", e);
    device_ = device;
    if (!stream_) {
        stream_ = StreamPool::instance().get(device);
        if (!stream_) return dev_fail("hipStreamCreate", hipGetLastError());
        aux_stream_ = StreamPool::instance().get(device);   // optional: K1 then stays on one stream
        k2_stream_ = StreamPool::instance().get(device);    // optional: K2 of a section beside the section before
    }
    return Failure();
}

void ArchiveJob::prewalk(const uint8_t *bytes, size_t n, const SectionInfo sec[kNumSections], const bool want[kNumSections]) {
    // (a section's walk is one thread following its block headers: the large sections side by side)
    std::vector<std::thread> side;
    for (int s = 0; s < kNumSections; s++) {
        if (!(sec[s].present && want[s] && sec[s].offset <= n && sec[s].compressed_size <= n - sec[s].offset)) continue;
        const uint8_t *payload = bytes + sec[s].offset;
        const size_t len = static_cast<size_t>(sec[s].compressed_size);
        SectionJob *j = &job_[s];
        bool aside = false;
        if (len >= (size_t(16) << 20) && s != kQuality) {
            try {
                side.emplace_back([j, payload, len] { j->walk(payload, len); });
                aside = true;
            } catch (...) {                                // no thread to be had
            }
        }
        if (!aside) j->walk(payload, len);
    }
    for (std::thread &t : side) t.join();
}

void ArchiveJob::drain() {
    for (SectionJob &j : job_) j.drain();
}

Failure ArchiveJob::upload(const uint8_t *bytes, size_t n, const nafgpu_header &h, const SectionInfo sec[kNumSections],
                           const ArchiveOptions &opt) {
    (void)hipSetDevice(device_);
    h_ = h;
    opt_ = opt;
    is_nuc_ = h.sequence_type <= 1;
    // ---- per section: host walk, then the compressed bytes of this process's block range (and only those: the
    // container header, sections that are switched off and the other ranks' shards never cross the bus) and the
    // task lists go to HBM
    const double t0 = now_ms();
    plan_ms_ = 0;
    compressed_ = 0;
    for (int s = 0; s < kNumSections; s++) {
        fail_[s] = Failure();
        if (!sec[s].present || !opt.want[s]) continue;
        if (sec[s].offset > n || sec[s].compressed_size > n - sec[s].offset) {
            fail_[s] = Failure::io(NAFGPU_IO_UNEXPECTED_EOF, "archive ends inside a section");
            continue;
        }
        uint64_t expect = sec[s].original_size;
        if (s == kSequence && is_nuc_) expect = (expect + 1) / 2;     // nucleotides -> packed bytes
        SectionOptions so;
        so.t_char = (s == kSequence && is_nuc_) ? (h.sequence_type == 1 ? 'U' : 'T') : 0;
        if ((s == kSequence || (s == kQuality && opt.shard_protocol)) && opt.shard_count > 1) {
            so.shard_rank = opt.shard_rank;
            so.shard_count = opt.shard_count;
            so.shard_protocol = opt.shard_protocol;
        }
        if (s == kSequence || s == kQuality) {                        // the two sections that can be large
            so.tile_blocks = opt.tile_blocks;
            so.tiled_output = opt.tiled_output;
        }
        fail_[s] = job_[s].prepare(bytes + sec[s].offset, static_cast<size_t>(sec[s].compressed_size), expect, stream_, so);
        plan_ms_ += job_[s].host_plan_ms();
        if (fail_[s].status == NAFGPU_E_DEVICE) return fail_[s];
        if (fail_[s].ok()) compressed_ += job_[s].n_tiles() == 1 ? job_[s].source_bytes() : sec[s].compressed_size;
    }
    h2d_ms_ = std::max(0.0f, static_cast<float>(now_ms() - t0) - plan_ms_);
    if (hook_env("NAFGPU_DEBUG_TIMES")) std::fprintf(stderr, "[nafgpu] upload: sections %.1f ms of which host walk %.1f\n", now_ms() - t0, plan_ms_);
    // ---- derived tables
    bool ok = d_totals_.alloc(8 * sizeof(ScanTotals)) && d_status_.alloc(64) && d_hash_.alloc(16);
    // The iterator never hands out more strings than records, and a section holds no more NULs than bytes:
    // number_of_sequences is an untrusted varint (up to 2^64 - 1) and must not size anything on its own.
    id_cap_ = job_[kIds].ready() ? std::min<uint64_t>(h.number_of_sequences, job_[kIds].size()) : 0;
    com_cap_ = job_[kComments].ready() ? std::min<uint64_t>(h.number_of_sequences, job_[kComments].size()) : 0;
    if (job_[kIds].ready()) ok = ok && d_id_ends_.alloc_items(id_cap_ + 1, sizeof(uint64_t));
    if (job_[kComments].ready()) ok = ok && d_com_ends_.alloc_items(com_cap_ + 1, sizeof(uint64_t));
    if (job_[kLengths].ready()) {
        rec_cap_ = job_[kLengths].size() / 4;
        ok = ok && d_rec_ends_.alloc_items(rec_cap_ + 1, sizeof(uint64_t));
    }
    mask_total_bases_ = sec[kSequence].present ? sec[kSequence].original_size : 0;   // mod.rs:236,241,250
    if (job_[kMask].ready()) {
        mask_cap_ = job_[kMask].size();
        ok = ok && d_mask_ends_.alloc_items(mask_cap_ + 1, sizeof(uint64_t));
    }
    const uint64_t scan_max = std::max({rec_cap_, mask_cap_, static_cast<uint64_t>(job_[kIds].ready() ? job_[kIds].size() : 0),
                                        static_cast<uint64_t>(job_[kComments].ready() ? job_[kComments].size() : 0)});
    ok = ok && d_scan_tmp_.alloc(scan_tmp_bytes(scan_max));
    if (!ok) return Failure::make(NAFGPU_E_DEVICE, "out of device memory");
    return Failure();
}

void ArchiveJob::decode_front() {
    (void)hipSetDevice(device_);
    timer_.reset();
    uint32_t *status = d_status_.as<uint32_t>();
    ScanTotals *totals = d_totals_.as<ScanTotals>();
    (void)hipMemsetAsync(status, 0, 64, stream_);
    (void)hipMemsetAsync(totals, 0, 8 * sizeof(ScanTotals), stream_);
    timer_.mark_total_begin(stream_);
    // Sections in file order.  The record table (LengthReader) and the mask run table (MaskReader) only need
    // the small Length / Mask sections, which come before the sequence: their scans run on the second stream
    // beside the sequence decode and are joined before the mask is applied.
    want_mask_ = job_[kMask].ready() && job_[kSequence].ready() && job_[kLengths].ready();
    for (int s = 0; s <= kMask; s++) job_[s].run(stream_, &timer_, aux_stream_);
    scans_forked_ = false;
    // (with a Mask section the run table is tens of megabytes: its scan beside the sequence decode slows K1 down by more
    //  than the scan takes alone -- 12.8 against 11.95 + 0.5 ms -- so then the scans wait until the sequence is done)
    if (aux_stream_ && job_[kSequence].ready() && !want_mask_) {
        if (!ev_fork_) (void)hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming);
        if (!ev_join_) (void)hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming);
        scans_forked_ = ev_fork_ && ev_join_ && hip_ok(hipEventRecord(ev_fork_, stream_)) &&
                        hip_ok(hipStreamWaitEvent(aux_stream_, ev_fork_, 0));
    }
    if (scans_forked_) {
        run_table_scans(aux_stream_);
        (void)hipEventRecord(ev_join_, aux_stream_);
    }
}

void ArchiveJob::run_table_scans(hipStream_t st) {
    uint32_t *status = d_status_.as<uint32_t>();
    ScanTotals *totals = d_totals_.as<ScanTotals>();
    if (job_[kLengths].ready())                                // LengthReader, reader.rs:48-67
        launch_scan_runs_u32(st, job_[kLengths].out(), rec_cap_, d_rec_ends_.as<uint64_t>(), rec_cap_, d_scan_tmp_.bytes(),
                             &totals[0], status);
    if (want_mask_)                                            // MaskReader, reader.rs:198-231
        launch_scan_runs_u8(st, job_[kMask].out(), mask_cap_, d_mask_ends_.as<uint64_t>(), mask_cap_, d_scan_tmp_.bytes(),
                            &totals[1], status);
}

Failure ArchiveJob::decode() {
    decode_front();
    // The sequence and quality sections: one resident tile each (kernels only), or tile after tile -- all of them
    // now when the whole output is resident, the first one when the output is held a tile at a time (the
    // iterator asks for the next ones, advance_tile).
    // K2 of every later single-tile section starts now, on its own stream: the sequence chains of a Quality section take
    // as long as the whole Sequence section in front of it, on one wave per CU.
    if (k2_stream_ && !hook_env("NAFGPU_NO_K2_AHEAD")) {   // (the hook: K2 of a section in its own place, for traces)
        bool first = true;
        for (int s = kMask + 1; s < kNumSections; s++) {
            if (!job_[s].ready() || job_[s].n_tiles() != 1) continue;
            if (!first) job_[s].run_k2_ahead(k2_stream_);
            first = false;
        }
    }
    for (int s = kMask + 1; s < kNumSections; s++) {
        if (!job_[s].ready()) continue;
        if (job_[s].n_tiles() == 1) {
            job_[s].run(stream_, &timer_, aux_stream_);
            continue;
        }
        const uint32_t upto = job_[s].tiled_output() ? 1u : job_[s].n_tiles();
        for (uint32_t t = 0; t < upto; t++) {
            Failure f = job_[s].decode_tile(t, stream_, &timer_, aux_stream_);
            if (f.status == NAFGPU_E_DEVICE) return f;
            if (!f.ok()) {
                fail_[s] = f;
                break;
            }
        }
    }
    return decode_back();
}

namespace {
// Pinned host blocks for small read-backs (a copy into ordinary memory keeps its caller until it is done, 37 us apiece; into
// pinned memory it is enqueued in 5): blocks of 16 KiB, handed back when done, never freed.
constexpr size_t kPinnedBlock = size_t(16) << 10;
class PinnedBlocks {
public:
    static PinnedBlocks &instance() {
        static PinnedBlocks *p = new PinnedBlocks;
        return *p;
    }
    uint8_t *take() {
        {
            std::lock_guard<std::mutex> lock(mu_);
            if (!idle_.empty()) {
                uint8_t *p = idle_.back();
                idle_.pop_back();
                return p;
            }
        }
        void *p = nullptr;
        return hipHostMalloc(&p, kPinnedBlock) == hipSuccess ? static_cast<uint8_t *>(p) : nullptr;
    }
    void give(uint8_t *p) {
        if (!p) return;
        std::lock_guard<std::mutex> lock(mu_);
        idle_.push_back(p);
    }

private:
    std::mutex mu_;
    std::vector<uint8_t *> idle_;
};
}  // namespace

Failure ArchiveJob::copy_small_to_host(const SmallCopy *copies, int count) {
    (void)hipSetDevice(device_);
    size_t total = 0;
    for (int k = 0; k < count; k++) total += (copies[k].n + 15) & ~size_t(15);
    uint8_t *blk = total && total <= kPinnedBlock ? PinnedBlocks::instance().take() : nullptr;
    bool ok = true;
    size_t off = 0;
    for (int k = 0; k < count && ok; k++) {
        if (!copies[k].n) continue;
        ok = hip_ok(hipMemcpyAsync(blk ? static_cast<void *>(blk + off) : copies[k].dst, copies[k].d_src, copies[k].n, hipMemcpyDeviceToHost, stream_));
        off += (copies[k].n + 15) & ~size_t(15);
    }
    ok = hip_ok(hipStreamSynchronize(stream_)) && ok;
    if (ok && blk) {
        off = 0;
        for (int k = 0; k < count; k++) {
            if (copies[k].n) std::memcpy(copies[k].dst, blk + off, copies[k].n);
            off += (copies[k].n + 15) & ~size_t(15);
        }
    }
    PinnedBlocks::instance().give(blk);
    return ok ? Failure() : Failure::make(NAFGPU_E_DEVICE, "device-to-host copy failed");
}

Failure ArchiveJob::decode_back() {
    uint32_t *status = d_status_.as<uint32_t>();
    ScanTotals *totals = d_totals_.as<ScanTotals>();
    timer_.begin(stream_, StageTimer::kOther);
    if (scans_forked_)
        (void)hipStreamWaitEvent(stream_, ev_join_, 0);
    else
        run_table_scans(stream_);
    // (nucleotide sequence sections come out of their SectionJob already expanded to ASCII:
    //  SequenceReader::read_nucleotide, reader.rs:121-172, is fused into the zstd kernels)
    if (want_mask_) apply_mask_to_held();                          // mod.rs:386-388, 402-441
    timer_.end(stream_);
    // ids / comments: CStringReader (reader.rs:22-30) as a scan; UTF-8 validity of every text section
    // (into_string().expect at mod.rs:362,368; from_utf8 at reader.rs:108-109) as one flag word
    uint32_t *utf8 = status + 8;
    timer_.begin(stream_, StageTimer::kOther);
    if (job_[kIds].ready()) {
        launch_scan_nul(stream_, job_[kIds].out(), job_[kIds].size(), d_id_ends_.as<uint64_t>(), id_cap_, d_scan_tmp_.bytes(),
                        &totals[2], status);
        launch_utf8_check(stream_, job_[kIds].out(), job_[kIds].size(), utf8, kIds);
    }
    if (job_[kComments].ready()) {
        launch_scan_nul(stream_, job_[kComments].out(), job_[kComments].size(), d_com_ends_.as<uint64_t>(), com_cap_,
                        d_scan_tmp_.bytes(), &totals[3], status);
        launch_utf8_check(stream_, job_[kComments].out(), job_[kComments].size(), utf8, kComments);
    }
    // (sections whose output is held a tile at a time are validated record by record by the iterator instead)
    if (job_[kSequence].ready() && !is_nuc_ && !job_[kSequence].tiled_output())
        launch_utf8_check(stream_, job_[kSequence].out(), job_[kSequence].size(), utf8, kSequence);
    if (job_[kQuality].ready() && !job_[kQuality].tiled_output())
        launch_utf8_check(stream_, job_[kQuality].out(), job_[kQuality].size(), utf8, kQuality);
    timer_.end(stream_);
    timer_.mark_total_end(stream_);
    // Everything the host wants to know -- the flag word, the scan totals, every section's status, size and counters -- comes back
    // through ONE pinned block with ONE wait for the stream (a read-back and a wait apiece were 0.55 ms of a small archive's 3.6).
    ScanTotals host_totals[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    utf8_invalid_ = 0;
    uint8_t *blk = PinnedBlocks::instance().take();
    constexpr size_t kTotalsAt = 16, kSlotsAt = kTotalsAt + sizeof host_totals;
    static_assert(kSlotsAt + kNumSections * SectionJob::kCheckSlotBytes <= kPinnedBlock, "pinned block too small");
    bool ok = blk != nullptr;
    if (ok) {
        ok = hip_ok(hipMemcpyAsync(blk, utf8, sizeof utf8_invalid_, hipMemcpyDeviceToHost, stream_)) &&
             hip_ok(hipMemcpyAsync(blk + kTotalsAt, totals, sizeof host_totals, hipMemcpyDeviceToHost, stream_));
        for (int s = 0; s < kNumSections && ok; s++)
            if (job_[s].ready() && job_[s].n_tiles() == 1) ok = job_[s].check_begin(stream_, blk + kSlotsAt + s * SectionJob::kCheckSlotBytes);
    }
    ok = hip_ok(hipStreamSynchronize(stream_)) && ok;
    if (!ok) {
        PinnedBlocks::instance().give(blk);
        return Failure::make(NAFGPU_E_DEVICE, std::string("decode failed: ") + hipGetErrorString(hipGetLastError()));
    }
    std::memcpy(&utf8_invalid_, blk, sizeof utf8_invalid_);
    std::memcpy(host_totals, blk + kTotalsAt, sizeof host_totals);
    rec_totals_ = host_totals[0];
    mask_totals_ = host_totals[1];
    id_totals_ = host_totals[2];
    com_totals_ = host_totals[3];
    times_ = timer_.collect();
    for (int s = 0; s < kNumSections; s++) {
        if (!job_[s].ready() || job_[s].n_tiles() > 1) continue;   // (tiles were checked one by one)
        Failure f = job_[s].check_end(blk + kSlotsAt + s * SectionJob::kCheckSlotBytes);
        if (!f.ok() && fail_[s].ok()) fail_[s] = f;
    }
    PinnedBlocks::instance().give(blk);
    return Failure();
}

// ------------------------------------------------------------------ the shard protocol (whole archive)
static int proto_section(int which) { return which == 0 ? kSequence : kQuality; }

Failure ArchiveJob::shard_begin(ShardSummary mine[2]) {
    decode_front();
    for (int w = 0; w < 2; w++) job_[proto_section(w)].shard_begin(stream_, &timer_, aux_stream_);
    for (int w = 0; w < 2; w++) {
        const int s = proto_section(w);
        mine[w] = ShardSummary();
        if (!job_[s].ready()) continue;
        if (!job_[s].in_protocol()) return Failure::make(NAFGPU_E_INVALID_ARG, "the decoder was not opened for the shard protocol (opts.shard_protocol, shard_count > 1)");
        Failure f = job_[s].shard_summary(stream_, &mine[w]);
        if (f.status == NAFGPU_E_DEVICE) return f;
        if (!f.ok()) {
            fail_[s] = f;
            mine[w].failed = true;
        }
    }
    return Failure();
}

Failure ArchiveJob::shard_place(const ShardSummary *seq_all, const ShardSummary *qual_all, uint32_t n_ranks) {
    (void)hipSetDevice(device_);
    for (int w = 0; w < 2; w++) {
        const int s = proto_section(w);
        if (!job_[s].ready()) continue;
        Failure f = job_[s].shard_place(w == 0 ? seq_all : qual_all, n_ranks, stream_, &timer_, aux_stream_);
        if (f.status == NAFGPU_E_DEVICE || f.status == NAFGPU_E_INVALID_ARG) return f;
        if (!f.ok() && fail_[s].ok()) fail_[s] = f;
    }
    return Failure();
}

Failure ArchiveJob::shard_halo(int which, uint64_t *recv_bytes, uint64_t *send_bytes, bool *tail_ready) {
    (void)hipSetDevice(device_);
    SectionJob &j = job_[proto_section(which)];
    *recv_bytes = *send_bytes = 0;
    *tail_ready = true;
    if (!j.ready() || !j.in_protocol()) return Failure();
    *recv_bytes = j.halo_recv_bytes();
    *send_bytes = j.tail_send_bytes();
    return j.tail_ready(stream_, tail_ready);
}

Failure ArchiveJob::shard_export(int which, void *dst, uint64_t n) {
    (void)hipSetDevice(device_);
    return job_[proto_section(which)].export_tail(dst, n, stream_);
}

Failure ArchiveJob::shard_import(int which, const void *src, uint64_t n) {
    (void)hipSetDevice(device_);
    return job_[proto_section(which)].import_halo(src, n, stream_, &timer_);
}

Failure ArchiveJob::shard_finish() {
    (void)hipSetDevice(device_);
    return decode_back();
}

// Lower-cases the masked runs inside the part of the sequence that is in HBM right now: this process's whole
// range, or the tile just decoded.
void ArchiveJob::apply_mask_to_held() {
    const SectionJob &j = job_[kSequence];
    const uint64_t mult = is_nuc_ ? 2 : 1;
    uint64_t lo, hi;
    uint8_t *seq;
    if (j.tiled_output()) {
        lo = j.tile_pos0() * mult;
        hi = (j.tile_pos0() + j.tile_len()) * mult;
        seq = const_cast<uint8_t *>(j.tile_data()) - lo;           // addressed by global base index
    } else {
        lo = j.shard_out0() * mult;
        hi = j.shard_out1() * mult;
        seq = j.out_mut() - lo;
    }
    const uint64_t n_all = is_nuc_ ? mask_total_bases_ : std::min<uint64_t>(mask_total_bases_, j.total_size());
    launch_mask_apply(stream_, seq, n_all, lo, hi, d_mask_ends_.as<uint64_t>(), &d_totals_.as<ScanTotals>()[1], d_rec_ends_.as<uint64_t>(),
                      &d_totals_.as<ScanTotals>()[0], mask_cap_, opt_.spec_mask ? 1 : 0, d_status_.as<uint32_t>());
}

// The iterator has consumed the tile section s holds: decode the next one into its place.
Failure ArchiveJob::advance_tile(int s) {
    (void)hipSetDevice(device_);
    SectionJob &j = job_[s];
    if (!j.ready() || !j.tiled_output() || j.tiles_done() >= j.n_tiles()) return Failure::make(NAFGPU_E_INVALID_ARG, "no tile left");
    const double t0 = now_ms();
    const uint32_t t = j.tiles_done();
    Failure f = j.decode_tile(t, stream_, nullptr, aux_stream_);
    if (hook_env("NAFGPU_DEBUG_TIMES")) std::fprintf(stderr, "[nafgpu] tile %u of section %d: %.1f ms\n", t, s, now_ms() - t0);
    if (!f.ok()) {
        if (f.status != NAFGPU_E_DEVICE) fail_[s] = f;
        return f;
    }
    if (s == kSequence && want_mask_) {
        apply_mask_to_held();
        if (!hip_ok(hipStreamSynchronize(stream_))) return Failure::make(NAFGPU_E_DEVICE, "mask application failed");
    }
    return Failure();
}

Failure ArchiveJob::format_text(bool with_ids, bool with_comments, bool with_quality, uint64_t n_rec, const uint8_t **d_text,
                                uint64_t *n_text, float *ms) {
    (void)hipSetDevice(device_);
    *d_text = nullptr;
    *n_text = 0;
    *ms = 0;
    if (!job_[kSequence].ready() || !job_[kLengths].ready())
        return Failure::make(NAFGPU_E_INVALID_ARG, "text output needs the Sequence and Length sections");
    if (job_[kSequence].sharded()) return Failure::make(NAFGPU_E_INVALID_ARG, "text output is not available on a shard");
    if (job_[kSequence].tiled_output() || job_[kQuality].tiled_output())
        return Failure::make(NAFGPU_E_INVALID_ARG, "text output needs the whole sequence in HBM (tiled output is for the record iterator)");
    n_rec = std::min<uint64_t>(n_rec, rec_totals_.count);
    if (n_rec == 0) return Failure();
    FmtText t{};
    t.seq = job_[kSequence].out();
    t.qual = with_quality && job_[kQuality].ready() ? job_[kQuality].out() : nullptr;
    t.rec_end = d_rec_ends_.as<uint64_t>();
    if (with_ids && job_[kIds].ready()) {
        t.ids = job_[kIds].out();
        t.id_end = d_id_ends_.as<uint64_t>();
        t.n_ids = n_ids();
    }
    if (with_comments && job_[kComments].ready()) {
        t.com = job_[kComments].out();
        t.com_end = d_com_ends_.as<uint64_t>();
        t.n_com = n_comments();
    }
    t.n_rec = n_rec;
    t.line_length = h_.line_length;
    t.sep = static_cast<uint8_t>(h_.name_separator);
    // the records must lie inside what was decoded (lengths may promise more than the sequence holds)
    uint64_t last_end = 0;
    if (!hip_ok(hipMemcpyAsync(&last_end, t.rec_end + (n_rec - 1), sizeof last_end, hipMemcpyDeviceToHost, stream_)) ||
        !hip_ok(hipStreamSynchronize(stream_)))
        return Failure::make(NAFGPU_E_DEVICE, "record table read-back failed");
    const uint64_t have = is_nuc_ ? std::min<uint64_t>(n_sequence_bytes(), mask_total_bases_) : job_[kSequence].total_size();
    if (last_end > have || (t.qual && last_end > job_[kQuality].size()))
        return Failure::io(NAFGPU_IO_UNEXPECTED_EOF, "record lengths exceed the decoded sequence");
    if (!d_fmt_sizes_.alloc(n_rec * sizeof(uint64_t)) || !d_fmt_off_.alloc((n_rec + 1) * sizeof(uint64_t)) ||
        !d_scan_tmp_.alloc(scan_tmp_bytes(std::max<uint64_t>(n_rec, 1))))
        return Failure::make(NAFGPU_E_DEVICE, "out of device memory");
    ScanTotals *totals = d_totals_.as<ScanTotals>();
    uint32_t *status = d_status_.as<uint32_t>();
    hipEvent_t e0 = nullptr, e1 = nullptr;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, stream_);
    (void)hipMemsetAsync(&totals[4], 0, sizeof(ScanTotals), stream_);
    launch_fmt_sizes(stream_, t, d_fmt_sizes_.as<uint64_t>());
    launch_scan_excl_u64(stream_, d_fmt_sizes_.as<uint64_t>(), n_rec, d_fmt_off_.as<uint64_t>(), d_scan_tmp_.bytes(), &totals[4], status);
    ScanTotals tot{0, 0};
    if (!hip_ok(hipMemcpyAsync(&tot, &totals[4], sizeof tot, hipMemcpyDeviceToHost, stream_)) || !hip_ok(hipStreamSynchronize(stream_)))
        return Failure::make(NAFGPU_E_DEVICE, "text size read-back failed");
    if (!d_text_.alloc(static_cast<size_t>(tot.sum) + 64)) return Failure::make(NAFGPU_E_DEVICE, "out of device memory for the text");
    launch_fmt_write(stream_, t, d_fmt_off_.as<uint64_t>(), tot.sum, d_text_.bytes());
    (void)hipEventRecord(e1, stream_);
    if (!hip_ok(hipStreamSynchronize(stream_)))
        return Failure::make(NAFGPU_E_DEVICE, std::string("text formatting failed: ") + hipGetErrorString(hipGetLastError()));
    (void)hipEventElapsedTime(ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *d_text = d_text_.bytes();
    *n_text = tot.sum;
    return Failure();
}

const uint8_t *ArchiveJob::d_sequence() const {
    if (!job_[kSequence].ready()) return nullptr;
    return job_[kSequence].out();
}

uint64_t ArchiveJob::sequence_offset() const {
    if (!job_[kSequence].ready()) return 0;
    return job_[kSequence].shard_out0() * (is_nuc_ ? 2 : 1);
}

uint64_t ArchiveJob::n_sequence_bytes() const {
    if (!job_[kSequence].ready()) return 0;
    return job_[kSequence].out_bytes();
}

Failure ArchiveJob::copy_to_host(void *dst, const void *d_src, size_t n) {
    if (!n) return Failure();
    (void)hipSetDevice(device_);
    if (!hip_ok(hipMemcpyAsync(dst, d_src, n, hipMemcpyDeviceToHost, stream_)) || !hip_ok(hipStreamSynchronize(stream_)))
        return Failure::make(NAFGPU_E_DEVICE, "device-to-host copy failed");
    return Failure();
}

// ... into memory the caller got from hipHostMalloc (the record iterator's window): the GPU stores it there itself (k_copy_out)
Failure ArchiveJob::copy_to_pinned(void *dst_pinned, const void *d_src, size_t n) {
    if (!n) return Failure();
    (void)hipSetDevice(device_);
    if (n < (size_t(1) << 20) || hook_env("NAFGPU_D2H_MEMCPY")) return copy_to_host(dst_pinned, d_src, n);   // small: one call is one call
    launch_copy_out(stream_, static_cast<uint8_t *>(dst_pinned), static_cast<const uint8_t *>(d_src), n);
    if (!hip_ok(hipGetLastError()) || !hip_ok(hipStreamSynchronize(stream_)))
        return Failure::make(NAFGPU_E_DEVICE, "device-to-host copy failed");
    return Failure();
}

bool ArchiveJob::copy_to_pinned_begin(void *dst_pinned, const void *d_src, size_t n) {
    if (!aux_stream_ || !n || hook_env("NAFGPU_D2H_MEMCPY")) return false;
    (void)hipSetDevice(device_);
    launch_copy_out(aux_stream_, static_cast<uint8_t *>(dst_pinned), static_cast<const uint8_t *>(d_src), n);
    return hip_ok(hipGetLastError());
}

Failure ArchiveJob::copy_to_pinned_end() {
    (void)hipSetDevice(device_);
    if (aux_stream_ && !hip_ok(hipStreamSynchronize(aux_stream_))) return Failure::make(NAFGPU_E_DEVICE, "device-to-host copy failed");
    return Failure();
}

Failure ArchiveJob::hash_device(const void *d_ptr, uint64_t n, uint64_t first_chunk, uint64_t *out) {
    (void)hipSetDevice(device_);
    unsigned long long *acc = d_hash_.as<unsigned long long>();
    (void)hipMemsetAsync(acc, 0, 8, stream_);
    launch_hash64(stream_, static_cast<const uint8_t *>(d_ptr), n, first_chunk, acc);
    unsigned long long v = 0;
    Failure f = copy_to_host(&v, acc, 8);
    *out = v;
    return f;
}

}  // namespace nafgpu
