// api.cpp -- the C-ABI of include/nafgpu.h on top of ArchiveJob.
//
// Host-side mirror of nafcodec's Decoder iterator (nafcodec/src/decoder/mod.rs:285-461):
// open = header + section table only (cheap, errors as the reference raises them at open);
// the first next() / decode_all_device() runs the whole GPU decode; next() then slices
// records out of the decoded sections exactly as next_record (mod.rs:356-399) zips its readers.
#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <future>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/nafgpu.h"
#include "container.h"
#include "engine.h"
#include "hash64.h"

using namespace nafgpu;

namespace {

// Pinned read-back window over one decoded section in HBM (sequence / quality): records are
// consumed front to back, so one sliding window replaces the reference's per-record Strings.
// The section may be resident whole, or a tile at a time (SectionJob::tiled_output): the window then
// asks for the next tile whenever it runs off the end of the one in HBM -- positions only move forward.
// Pinned memory outlives decoders: hipHostMalloc + hipHostFree of the window were 1.5 of the 5 ms a 5-Mbase archive takes open to
// close (a caller that walks a directory of genomes pays them per file).  A closed decoder's windows go to a small per-process
// pool -- at most kPinnedPoolKeep buffers -- and the next decoder takes the smallest one that is large enough.
class PinnedPool {
public:
    static PinnedPool &get() {
        static PinnedPool p;
        return p;
    }
    uint8_t *take(uint64_t want, uint64_t *cap) {
        std::lock_guard<std::mutex> g(mu_);
        int best = -1;
        for (int i = 0; i < n_; i++)
            if (cap_[i] >= want && (best < 0 || cap_[i] < cap_[best])) best = i;
        if (best < 0) return nullptr;
        uint8_t *p = buf_[best];
        *cap = cap_[best];
        buf_[best] = buf_[n_ - 1];
        cap_[best] = cap_[n_ - 1];
        n_--;
        return p;
    }
    void give(uint8_t *p, uint64_t cap) {
        {
            std::lock_guard<std::mutex> g(mu_);
            if (n_ < kPinnedPoolKeep) {
                buf_[n_] = p;
                cap_[n_] = cap;
                n_++;
                return;
            }
            int small = 0;                                 // full: the smallest buffer makes room for a larger one
            for (int i = 1; i < n_; i++)
                if (cap_[i] < cap_[small]) small = i;
            if (cap_[small] < cap) {
                std::swap(buf_[small], p);
                std::swap(cap_[small], cap);
            }
        }
        (void)hipHostFree(p);
    }

private:
    static constexpr int kPinnedPoolKeep = 4;
    std::mutex mu_;
    uint8_t *buf_[kPinnedPoolKeep] = {nullptr, nullptr, nullptr, nullptr};
    uint64_t cap_[kPinnedPoolKeep] = {0, 0, 0, 0};
    int n_ = 0;
};

// The bytes of one section the iterator hands out, on the host: a window of pinned memory that moves forward over the
// section (the reference's BufReader over its zstd stream, mod.rs:223).  While the caller works through one window the next
// one is already on its way into a second buffer (`ahead`): the link is then busy all the time, where waiting for each window
// in turn cost 0.14 ms of launch and wake-up per window on top of its 1.3 ms.
class HostWindow {
public:
    ~HostWindow() {
        settle();
        if (buf_) PinnedPool::get().give(buf_, cap_);
        if (nbuf_) PinnedPool::get().give(nbuf_, ncap_);
    }
    void bind(ArchiveJob *job, int section, uint64_t mult, uint64_t total, uint64_t window) {
        settle();
        job_ = job;
        section_ = section;
        mult_ = mult;
        total_ = total;
        window_ = window;
        lo_ = hi_ = 0;
        base_ = buf_;
    }
    // bytes [start, start+len) are on the host already (a record in front of a tile that failed is still handed out)
    bool holds(uint64_t start, uint64_t len) const { return start >= lo_ && start + len <= hi_ && hi_ > lo_; }
    // pointer to bytes [start, start+len) on the host; nullptr on failure
    const uint8_t *get(uint64_t start, uint64_t len, Failure *f) {
        if (len == 0) return reinterpret_cast<const uint8_t *>("");
        if (start >= lo_ && start + len <= hi_) return base_ + (start - lo_);
        if (ahead_n_) {
            // the window in flight begins where this one ends: a record that straddles the two has its first bytes copied into
            // the room in front of it
            const uint64_t a_lo = ahead_lo_, a_n = ahead_n_;
            *f = job_->copy_to_pinned_end();
            ahead_n_ = 0;
            if (!f->ok()) return nullptr;
            if (start >= lo_ && start <= hi_ && hi_ == a_lo && hi_ - start <= kHead && start + len <= a_lo + a_n) {
                const uint64_t keep = hi_ - start;
                uint8_t *nb = nbuf_ + kHead - keep;
                if (keep) std::memcpy(nb, base_ + (start - lo_), keep);
                std::swap(buf_, nbuf_);
                std::swap(cap_, ncap_);
                base_ = nb;
                lo_ = start;
                hi_ = a_lo + a_n;
                send_ahead();
                return base_;
            }
        }
        const uint64_t want = std::max(len, std::min(window_, total_ - start));
        const uint64_t keep = (start >= lo_ && start < hi_) ? hi_ - start : 0;   // already on the host (a record straddling the old window)
        // (a section of several windows: room for the head of a straddling record in front, see above)
        const uint64_t room = total_ > window_ ? std::max(want, window_ + kHead) : want;
        if (room > cap_) {
            uint64_t cap = room;
            void *p = PinnedPool::get().take(room, &cap);
            if (!p && hipHostMalloc(&p, room) != hipSuccess) {
                *f = Failure::make(NAFGPU_E_DEVICE, "cannot allocate the pinned read-back window");
                return nullptr;
            }
            if (keep) std::memcpy(p, base_ + (start - lo_), keep);
            if (buf_) PinnedPool::get().give(buf_, cap_);
            buf_ = static_cast<uint8_t *>(p);
            cap_ = cap;
        } else if (keep) {
            std::memmove(buf_, base_ + (start - lo_), keep);
        }
        base_ = buf_;
        const SectionJob &sj = job_->job(section_);
        uint64_t pos = start + keep;                       // next byte to fetch
        const uint64_t end = start + want;
        while (pos < end) {
            uint64_t held0 = 0, held1 = total_;            // bytes of the section in HBM right now
            const uint8_t *d_held = sj.out();
            if (sj.tiled_output()) {
                held0 = sj.tile_pos0() * mult_;
                held1 = (sj.tile_pos0() + sj.tile_len()) * mult_;
                d_held = sj.tile_data();
                if (pos >= held1) {                        // off the end of the tile: the next one takes its place
                    if (sj.tiles_done() >= sj.n_tiles()) break;
                    *f = job_->advance_tile(section_);
                    if (!f->ok()) {
                        // a tile BEHIND the record asked for failed while the window was being filled ahead: the record itself is
                        // whole -- it is handed out, and the error comes with the first record that needs the bad tile (the
                        // reference meets a corrupt block when its stream gets there, mod.rs:356-399)
                        if (pos >= start + len && f->status != NAFGPU_E_DEVICE) {
                            *f = Failure();
                            break;
                        }
                        return nullptr;
                    }
                    continue;
                }
                if (pos < held0) {
                    *f = Failure::make(NAFGPU_E_INVALID_ARG, "internal: the read-back window moved backwards over a tile");
                    return nullptr;
                }
            }
            const uint64_t n = std::min(end, held1) - pos;
            if (n == 0) break;
            *f = job_->copy_to_pinned(buf_ + (pos - start), d_held + (pos - held0), n);
            if (!f->ok()) return nullptr;
            pos += n;
        }
        lo_ = start;
        hi_ = pos;
        if (start + len > hi_) {
            *f = Failure::io(NAFGPU_IO_UNEXPECTED_EOF, "section ends before the record does");
            return nullptr;
        }
        send_ahead();
        return base_;
    }

    // waits for the window in flight, if any (before the buffers, the job or the device bytes it reads go away)
    void settle() {
        if (ahead_n_ && job_) (void)job_->copy_to_pinned_end();
        ahead_n_ = 0;
    }

private:
    static constexpr uint64_t kHead = uint64_t(8) << 20;   // room in front of a window in flight (records that straddle by more wait for their window)
    // the window after this one, as far as it lies in the bytes held in HBM right now (a tile's end is crossed by get())
    void send_ahead() {
        if (hi_ >= total_ || hook_env("NAFGPU_NO_AHEAD")) return;      // (the hook: A/B runs, tools/iter_ahead_probe.py)
        const SectionJob &sj = job_->job(section_);
        uint64_t held0 = 0, held1 = total_;
        const uint8_t *d_held = sj.out();
        if (sj.tiled_output()) {
            held0 = sj.tile_pos0() * mult_;
            held1 = (sj.tile_pos0() + sj.tile_len()) * mult_;
            d_held = sj.tile_data();
        }
        if (hi_ < held0 || hi_ >= held1) return;
        const uint64_t n = std::min(window_, held1 - hi_);
        if (n < std::min<uint64_t>(uint64_t(1) << 20, window_)) return;   // (the last bytes of a tile: the plain way)
        if (kHead + n > ncap_) {
            uint64_t cap = kHead + window_;
            void *p = PinnedPool::get().take(cap, &cap);
            if (!p && hipHostMalloc(&p, kHead + window_) != hipSuccess) return;
            if (nbuf_) PinnedPool::get().give(nbuf_, ncap_);
            nbuf_ = static_cast<uint8_t *>(p);
            ncap_ = cap;
        }
        if (!job_->copy_to_pinned_begin(nbuf_ + kHead, d_held + (hi_ - held0), n)) return;
        ahead_lo_ = hi_;
        ahead_n_ = n;
    }
    ArchiveJob *job_ = nullptr;
    int section_ = 0;
    uint64_t mult_ = 1, total_ = 0, window_ = 0, lo_ = 0, hi_ = 0, cap_ = 0, ncap_ = 0;
    uint64_t ahead_lo_ = 0, ahead_n_ = 0;                  // bytes [ahead_lo_, + ahead_n_) of the section are on their way to nbuf_ + kHead
    uint8_t *buf_ = nullptr, *nbuf_ = nullptr, *base_ = nullptr;   // base_: where byte lo_ is (inside buf_)
};

}  // namespace

struct nafgpu_decoder {
    nafgpu_opts opts{};
    nafgpu_header header{};
    SectionInfo sec[kNumSections];
    // archive bytes: borrowed (open_bytes), a read-only file mapping (open_path: no copy, pages come in as the
    // host walk and the uploads touch them) or an anonymous mapping filled from a reader (open_io)
    void *map = nullptr;
    size_t map_len = 0;
    const uint8_t *bytes = nullptr;
    size_t n_bytes = 0;
    ArchiveJob job;
    ~nafgpu_decoder() {
        job.drain();                     // (a tile's source bytes may still be travelling out of the mapping)
        if (map) (void)munmap(map, map_len);
    }
    bool device_ready = false, decoded = false;
    uint64_t tile_blocks = 0;            // > 0: the sequence / quality sections are decoded in tiles of this many zstd blocks
    bool tiled_output = false;           // ... and their output is held a tile at a time (record iterator)
    Failure fatal;                       // device failure: every later call reports it
    // the shard protocol is a sequence: begin -> place -> halo / export_tail / import_halo (any order the exchange needs) -> finish
    enum ShardPhase { kShardIdle, kShardBegun, kShardPlaced } shard_phase = kShardIdle;
    Failure last;
    // iterator state (mod.rs:285-296)
    uint64_t n = 0;                      // records yielded
    uint64_t ids_pos = 0, com_pos = 0;   // CStringReader cursors
    uint64_t rec_idx = 0;                // LengthReader cursor
    std::vector<uint8_t> ids, comments;
    std::vector<uint64_t> rec_ends;
    HostWindow seq_win, qual_win;
    uint64_t mask_covered = 0;           // bases the mask section covers (see next())
    bool use[kNumSections] = {false, false, false, false, false, false};
};

namespace {

constexpr uint64_t kIterTileFrom = uint64_t(4) << 30;    // the iterator holds a section's output a tile at a time from this size on ...
constexpr uint64_t kIterTileBytes = uint64_t(2) << 30;   // ... in tiles of this many decoded bytes
constexpr size_t kWalkBesideInit = size_t(64) << 20;     // archives from this size on are walked beside the start of the HIP runtime

Failure ensure_uploaded(nafgpu_decoder *d, bool for_iterator);
Failure after_decode(nafgpu_decoder *d);

// Decoded output of this many bytes per tile (0: no tiling): nafgpu_opts.tile_mib (tests: NAFGPU_TILE_KIB), or -- when the
// selected sections would not fit beside each other in the device's free memory -- a sixteenth of that memory.
uint64_t tile_blocks_for(const nafgpu_decoder *d, bool for_iterator) {
    uint64_t tile_bytes = static_cast<uint64_t>(d->opts.tile_mib > 0 ? d->opts.tile_mib : 0) << 20;
    if (const char *e = hook_env("NAFGPU_TILE_KIB")) tile_bytes = std::strtoull(e, nullptr, 10) << 10;   // tests: tiles far below a MiB
    if (!tile_bytes && for_iterator) {
        // The record iterator reads every decoded byte back over the link, 20 ms a GiB, the decode of that GiB taking a third of a
        // millisecond: a large section goes tile by tile -- the first records leave after one tile's upload and decode instead of
        // the whole archive's, the compressed bytes of the next tile travel while this one is read back (engine.cpp:
        // start_source_upload), and the device holds one tile of output, not tens of gigabytes (whose address ranges and chunks
        // took 0.01 s in one process and 0.8 s in the next).  (NAFGPU_ITER_TILE_MIB: experiments; 0 = whole output)
        uint64_t big = 0, tile = kIterTileBytes, from = kIterTileFrom;
        for (int s : {kSequence, kQuality})
            if (d->use[s]) big = std::max<uint64_t>(big, d->sec[s].original_size);
        if (const char *e = hook_env("NAFGPU_ITER_TILE_MIB")) from = tile = std::strtoull(e, nullptr, 10) << 20;
        if (tile && big >= from) tile_bytes = tile;
    }
    if (!tile_bytes) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 0;
        uint64_t need = 0;
        for (int s = 0; s < kNumSections; s++)
            if (d->use[s]) {
                const uint64_t out = s == kSequence && d->header.sequence_type <= 1 ? d->sec[s].original_size : d->sec[s].original_size;
                need += d->sec[s].compressed_size + out + out / 8;      // compressed + decoded + scratch
            }
        if (need < static_cast<uint64_t>(free_b) / 10 * 8) return 0;
        tile_bytes = std::max<uint64_t>(static_cast<uint64_t>(free_b) / 16, uint64_t(64) << 20);
    }
    const uint64_t per_block = kBlockMax * (d->header.sequence_type <= 1 ? 2 : 1);
    return std::max<uint64_t>(1, tile_bytes / per_block);
}

Failure ensure_decoded(nafgpu_decoder *d, bool for_iterator) {
    if (!d->fatal.ok()) return d->fatal;
    if (d->decoded) return Failure();
    d->seq_win.settle();                                   // (a window on its way reads what is about to be decoded again)
    d->qual_win.settle();
    const bool trace = hook_env("NAFGPU_DEBUG_TIMES") != nullptr;   // (experiments: what the first call is made of)
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    Failure f = ensure_uploaded(d, for_iterator);
    if (!f.ok()) return f;
    const double t1 = now();
    f = d->job.decode();
    if (!f.ok()) return d->fatal = f;
    const double t2 = now();
    f = after_decode(d);
    if (!f.ok()) return f;
    if (trace) std::fprintf(stderr, "[nafgpu] first decode: init + walk + upload %.1f ms, decode %.1f, tables back %.1f\n", t1 - t0, t2 - t1, now() - t2);
    d->decoded = true;
    return Failure();
}

Failure ensure_uploaded(nafgpu_decoder *d, bool for_iterator) {
    if (!d->fatal.ok()) return d->fatal;
    const uint64_t tile_blocks = d->device_ready ? d->tile_blocks : 0;
    if (d->device_ready && tile_blocks && d->tiled_output != for_iterator) d->device_ready = false;   // the other way of holding the output: prepare again
    if (!d->device_ready) {
        d->seq_win.settle();
        d->qual_win.settle();
        const bool want[kNumSections] = {d->opts.id != 0, d->opts.comment != 0, true, d->opts.mask != 0,
                                         d->opts.sequence != 0, d->opts.quality != 0};
        Failure f;
#ifndef NAFGPU_EMU
        if (d->n_bytes >= kWalkBesideInit) {
            // a process's first HIP call takes 0.15-0.2 s, the walk of a 10 GB archive out of a fresh file mapping as long (a page
            // fault per block): side by side
            std::future<Failure> started;
            try {
                started = std::async(std::launch::async, [d] { return d->job.init(d->opts.device); });
            } catch (...) {                                // no thread to be had: one after the other
            }
            d->job.prewalk(d->bytes, d->n_bytes, d->sec, want);
            f = started.valid() ? started.get() : d->job.init(d->opts.device);
        } else
#endif
            f = d->job.init(d->opts.device);
        if (!f.ok()) return d->fatal = f;
        ArchiveOptions ao;
        for (int s = 0; s < kNumSections; s++) ao.want[s] = want[s];
        ao.spec_mask = d->opts.spec_mask != 0;
        ao.shard_count = d->opts.shard_count > 1 ? static_cast<uint32_t>(d->opts.shard_count) : 1u;
        ao.shard_rank = d->opts.shard_rank > 0 ? static_cast<uint32_t>(d->opts.shard_rank) : 0u;
        ao.shard_protocol = d->opts.shard_protocol != 0 && ao.shard_count > 1;
        d->tile_blocks = tile_blocks_for(d, for_iterator);
        d->tiled_output = d->tile_blocks != 0 && for_iterator;
        ao.tile_blocks = d->tile_blocks;
        ao.tiled_output = d->tiled_output;
        f = d->job.upload(d->bytes, d->n_bytes, d->header, d->sec, ao);
        if (!f.ok()) return d->fatal = f;
        d->device_ready = true;
    }
    return Failure();
}

int fail(nafgpu_decoder *d, const Failure &f) {
    d->last = f;
    return f.status;
}

// CStringReader::next (reader.rs:22-30) over a fully decoded section
int cstring_next(const std::vector<uint8_t> &buf, uint64_t *pos, nafgpu_field *out, Failure *f) {
    if (*pos >= buf.size()) return 0;   // read_until -> Ok(0) -> None
    const uint8_t *start = buf.data() + *pos;
    const void *nul = std::memchr(start, 0, buf.size() - static_cast<size_t>(*pos));
    if (!nul) {
        *f = Failure::make(NAFGPU_E_PANIC, "string section does not end with NUL (reference: expect() panic)");
        return -1;
    }
    out->ptr = start;
    out->len = static_cast<uint64_t>(static_cast<const uint8_t *>(nul) - start);
    out->present = 1;
    *pos += out->len + 1;
    if (!utf8_valid(out->ptr, out->len)) {
        *f = Failure::make(NAFGPU_E_PANIC, "invalid UTF-8 in id/comment (reference: expect(\"TODO\") panic, mod.rs:362)");
        return -1;
    }
    return 1;
}

void set_opts(nafgpu_decoder *d, const nafgpu_opts *opts) {
    if (opts)
        d->opts = *opts;
    else
        nafgpu_opts_default(&d->opts);
    if (d->opts.shard_count <= 0) d->opts.shard_count = 1;
}

int open_common(std::unique_ptr<nafgpu_decoder> d, nafgpu_decoder **out, nafgpu_error *err, const NeedFn *need = nullptr) {
    Failure f = parse_archive(d->bytes, d->n_bytes, &d->header, d->sec, need);
    if (!f.ok()) {
        f.to_c(err);
        return f.status;
    }
    const bool want[kNumSections] = {d->opts.id != 0, d->opts.comment != 0, true, d->opts.mask != 0,
                                     d->opts.sequence != 0, d->opts.quality != 0};
    for (int s = 0; s < kNumSections; s++) {
        d->use[s] = d->sec[s].present && want[s];
        // the payloads of the selected sections (the reference seeks over the others, mod.rs:228)
        if (need && d->use[s] && d->sec[s].offset < d->n_bytes)
            (*need)(static_cast<size_t>(d->sec[s].offset),
                    static_cast<size_t>(std::min<uint64_t>(d->sec[s].compressed_size, d->n_bytes - d->sec[s].offset)));
    }
    *out = d.release();
    if (err) Failure().to_c(err);
    return NAFGPU_OK;
}

Failure errno_failure(const char *what, int e) {
    int kind = NAFGPU_IO_OTHER;
    if (e == ENOENT) kind = NAFGPU_IO_NOT_FOUND;
    else if (e == EISDIR) kind = NAFGPU_IO_IS_A_DIRECTORY;
    else if (e == EACCES || e == EPERM) kind = NAFGPU_IO_PERMISSION_DENIED;
    return Failure::io(kind, std::string(what) + ": " + std::strerror(e), e);
}

// the small sections and the record table come back to the host; windows over sequence / quality are bound
Failure after_decode(nafgpu_decoder *d) {
    // ids, comments and the record table: one wait for the three of them (engine.cpp: copy_small_to_host)
    ArchiveJob::SmallCopy copies[3];
    int n_copies = 0;
    auto fetch = [&](int s, std::vector<uint8_t> *dst) {
        if (!d->job.job(s).ready() || !d->job.section_failure(s).ok()) return;
        dst->resize(static_cast<size_t>(d->job.section_size(s)));
        copies[n_copies++] = {dst->data(), d->job.d_section(s), dst->size()};
    };
    fetch(kIds, &d->ids);
    fetch(kComments, &d->comments);
    if (d->job.job(kLengths).ready() && d->job.section_failure(kLengths).ok()) {
        d->rec_ends.resize(static_cast<size_t>(d->job.n_records()));
        copies[n_copies++] = {d->rec_ends.data(), d->job.d_rec_ends(), d->rec_ends.size() * sizeof(uint64_t)};
    }
    Failure f = d->job.copy_small_to_host(copies, n_copies);
    if (!f.ok()) return d->fatal = f;
    uint64_t window = std::max<uint64_t>(d->opts.buffer_size, uint64_t(64) << 20);
    if (const char *e = hook_env("NAFGPU_WINDOW_KIB")) window = std::max<uint64_t>(1, std::strtoull(e, nullptr, 10)) << 10;   // (tests: many windows over a small section)
    d->seq_win.bind(&d->job, kSequence, d->header.sequence_type <= 1 ? 2 : 1, d->job.n_sequence_bytes(), window);
    d->qual_win.bind(&d->job, kQuality, 1, d->job.section_size(kQuality), window);
    // MaskReader yields units until their sum reaches the nucleotide count (reader.rs:200-202);
    // a record ending beyond what the units cover raises "failed to get mask unit" (mod.rs:430-435)
    const uint64_t total = d->sec[kSequence].present ? d->sec[kSequence].original_size : 0;
    d->mask_covered = d->job.mask_sum() >= total ? UINT64_MAX : d->job.mask_sum();
    return Failure();
}

void fill_result(nafgpu_decoder *d, nafgpu_device_result *out) {
    std::memset(out, 0, sizeof *out);
    const ArchiveJob &j = d->job;
    out->d_sequence = d->use[kSequence] ? j.d_sequence() : nullptr;
    const bool nuc = d->header.sequence_type <= 1;
    // bases held by this decoder: the whole section, or (sharded) global bases [base0, base1)
    const uint64_t total_bases = d->use[kSequence] ? (nuc ? d->sec[kSequence].original_size : j.job(kSequence).total_size()) : 0;
    const uint64_t base0 = std::min(j.sequence_offset(), total_bases);
    const uint64_t base1 = std::min(j.sequence_offset() + j.n_sequence_bytes(), total_bases);
    out->n_bases = base1 - base0;
    out->base_offset = base0;
    out->sharded = j.job(kSequence).ready() && j.job(kSequence).sharded() ? 1 : 0;
    if (d->use[kLengths] && !d->rec_ends.empty()) {
        // first record that STARTS in this shard (start_k = end_{k-1}); carry: the shard begins inside a record
        const auto it = std::lower_bound(d->rec_ends.begin(), d->rec_ends.end(), base0);   // first end >= base0
        uint64_t k = static_cast<uint64_t>(it - d->rec_ends.begin());                     // records ending before base0 ... k-1
        const bool at_start = base0 == 0 || (k > 0 && d->rec_ends[k - 1] == base0) || (it != d->rec_ends.end() && *it == base0);
        if (it != d->rec_ends.end() && *it == base0) k += 1;      // record k ends exactly here: the next one starts here
        out->carry = at_start ? 0 : 1;
        out->first_record = at_start ? k : k + 1;
    }
    out->d_quality = j.d_section(kQuality);
    out->n_quality = j.section_size(kQuality);
    out->d_record_end = d->use[kLengths] ? j.d_rec_ends() : nullptr;
    out->n_records = j.n_records();
    out->d_ids = j.d_section(kIds);
    out->n_ids_bytes = j.section_size(kIds);
    out->d_comments = j.d_section(kComments);
    out->n_comments_bytes = j.section_size(kComments);
    out->packed_bytes = j.packed_bytes();
    out->compressed_bytes = j.compressed_bytes();
    // compressed bytes of the sequence section this process reads: its block range's when the section is sharded
    out->seq_compressed_bytes = !d->use[kSequence] ? 0
                                : (j.job(kSequence).ready() && j.job(kSequence).n_tiles() == 1 ? j.job(kSequence).source_bytes()
                                                                                              : d->sec[kSequence].compressed_size);
    out->n_zstd_blocks = j.job(kSequence).n_blocks();
    out->n_huf_streams = j.job(kSequence).n_streams();
    const StageTimes &t = j.times();
    out->ms_total = t.total;
    out->ms_huf = t.huf;
    out->ms_unpack = t.unpack;
    out->ms_seq_lz = t.seq_lz;
    out->ms_other = t.other;
    out->n_huf_launches = t.huf_launches;
    for (int s = 0; s < kNumSections; s++) out->lz_residue_matches += j.job(s).lz_residue();
    out->ms_host_plan = j.host_plan_ms();
    out->ms_h2d = j.h2d_ms();
    out->d_id_end = d->use[kIds] ? j.d_id_ends() : nullptr;
    out->d_comment_end = d->use[kComments] ? j.d_com_ends() : nullptr;
    out->n_ids = d->use[kIds] ? j.n_ids() : 0;
    out->n_comments = d->use[kComments] ? j.n_comments() : 0;
    out->utf8_invalid = j.utf8_invalid();
    out->quality_offset = j.quality_offset();
}

}  // namespace

extern "C" {

void nafgpu_opts_default(nafgpu_opts *o) {
    std::memset(o, 0, sizeof *o);
    o->id = o->comment = o->sequence = o->quality = o->mask = 1;   // mod.rs:67-76
    o->buffer_size = 4096;
    o->device = -1;
    o->shard_rank = 0;
    o->shard_count = 1;
}

void nafgpu_opts_from_flags(nafgpu_opts *o, uint8_t flags) {
    nafgpu_opts_default(o);                                        // mod.rs:93-101: `id` is left on
    o->quality = (flags & 0x01) != 0;
    o->sequence = (flags & 0x02) != 0;
    o->mask = (flags & 0x04) != 0;
    o->comment = (flags & 0x10) != 0;
}

int nafgpu_open_bytes(const uint8_t *bytes, size_t n, const nafgpu_opts *opts, nafgpu_decoder **out, nafgpu_error *err) {
    if (!out || (!bytes && n)) return NAFGPU_E_INVALID_ARG;
    std::unique_ptr<nafgpu_decoder> d(new (std::nothrow) nafgpu_decoder);
    if (!d) return NAFGPU_E_DEVICE;
    d->bytes = bytes;
    d->n_bytes = n;
    set_opts(d.get(), opts);
    return open_common(std::move(d), out, err);
}

int nafgpu_open_path(const char *path, const nafgpu_opts *opts, nafgpu_decoder **out, nafgpu_error *err) {
    if (!out || !path) return NAFGPU_E_INVALID_ARG;
    std::unique_ptr<nafgpu_decoder> d(new (std::nothrow) nafgpu_decoder);
    if (!d) return NAFGPU_E_DEVICE;
    const int fd = ::open(path, O_RDONLY | O_CLOEXEC);              // File::open, mod.rs:163
    if (fd < 0) {
        Failure f = errno_failure(path, errno);
        f.to_c(err);
        return f.status;
    }
    struct stat stt;
    std::memset(&stt, 0, sizeof stt);
    if (fstat(fd, &stt) != 0) {
        Failure f = errno_failure(path, errno);
        ::close(fd);
        f.to_c(err);
        return f.status;
    }
    if (S_ISDIR(stt.st_mode)) {
        ::close(fd);
        Failure f = errno_failure(path, EISDIR);
        f.to_c(err);
        return f.status;
    }
    // One read-only mapping replaces the IoSlice lock+seek+read refills (ioslice.rs:28-41): nothing is copied on
    // the host; the frame walk touches a page per zstd block and the uploads stream the payloads to HBM.
    // (Not a regular file -- a pipe, a character device: drain it through read().)
    if (S_ISREG(stt.st_mode) && stt.st_size > 0) {
        void *m = mmap(nullptr, static_cast<size_t>(stt.st_size), PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) {
            Failure f = errno_failure(path, errno);
            ::close(fd);
            f.to_c(err);
            return f.status;
        }
        (void)madvise(m, static_cast<size_t>(stt.st_size), MADV_SEQUENTIAL);
        d->map = m;
        d->map_len = static_cast<size_t>(stt.st_size);
        d->bytes = static_cast<const uint8_t *>(m);
        d->n_bytes = d->map_len;
        ::close(fd);
    } else if (!S_ISREG(stt.st_mode)) {
        std::vector<uint8_t> buf;
        uint8_t chunk[1 << 16];
        for (;;) {
            const ssize_t got = ::read(fd, chunk, sizeof chunk);
            if (got < 0) {
                if (errno == EINTR) continue;
                Failure f = errno_failure(path, errno);
                ::close(fd);
                f.to_c(err);
                return f.status;
            }
            if (got == 0) break;
            buf.insert(buf.end(), chunk, chunk + got);
        }
        ::close(fd);
        if (!buf.empty()) {
            void *m = mmap(nullptr, buf.size(), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (m == MAP_FAILED) {
                Failure f = errno_failure("mmap", errno);
                f.to_c(err);
                return f.status;
            }
            std::memcpy(m, buf.data(), buf.size());
            d->map = m;
            d->map_len = buf.size();
            d->bytes = static_cast<const uint8_t *>(m);
            d->n_bytes = buf.size();
        }
    } else {
        ::close(fd);                                               // empty file: Io(UnexpectedEof) from the header parse
    }
    set_opts(d.get(), opts);
    return open_common(std::move(d), out, err);
}

int nafgpu_open_io(nafgpu_read_fn read, nafgpu_seek_fn seek, void *ctx, const nafgpu_opts *opts, nafgpu_decoder **out,
                   nafgpu_error *err) {
    if (!out || !read) return NAFGPU_E_INVALID_ARG;
    std::unique_ptr<nafgpu_decoder> d(new (std::nothrow) nafgpu_decoder);
    if (!d) return NAFGPU_E_DEVICE;
    set_opts(d.get(), opts);
    auto io_fail = [&](const char *what, int64_t rc) {
        Failure f = errno_failure(what, static_cast<int>(-rc));
        f.to_c(err);
        return f.status;
    };
    // R: Read + Seek (mod.rs:169-172).  The archive starts at the reader's current position, as in the reference
    // (fill_buf from wherever the reader stands).  With a working seek only the header, the section table and the
    // payloads of the SELECTED sections are read -- with_reader seeks over the others (mod.rs:228) -- into an
    // anonymous mapping as long as the archive (untouched pages cost nothing); without one the reader is drained.
    int64_t pos0 = -1, end = -1;
    if (seek) {
        pos0 = seek(ctx, 0, SEEK_CUR);
        if (pos0 >= 0) end = seek(ctx, 0, SEEK_END);
        if (pos0 >= 0 && end >= pos0) {
            const int64_t back = seek(ctx, pos0, SEEK_SET);
            if (back < 0) return io_fail("seek", back);
        } else {
            pos0 = end = -1;                                       // not seekable after all: drain
        }
    }
    if (pos0 < 0) {
        std::vector<uint8_t> buf, chunk(1 << 20);
        for (;;) {
            const int64_t got = read(ctx, chunk.data(), chunk.size());
            if (got < 0) return io_fail("read", got);
            if (got == 0) break;
            buf.insert(buf.end(), chunk.begin(), chunk.begin() + std::min<int64_t>(got, static_cast<int64_t>(chunk.size())));
        }
        if (!buf.empty()) {
            void *m = mmap(nullptr, buf.size(), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (m == MAP_FAILED) return io_fail("mmap", -static_cast<int64_t>(errno));
            std::memcpy(m, buf.data(), buf.size());
            d->map = m;
            d->map_len = buf.size();
            d->bytes = static_cast<const uint8_t *>(m);
            d->n_bytes = buf.size();
        }
        return open_common(std::move(d), out, err);
    }
    const size_t total = static_cast<size_t>(end - pos0);
    if (total == 0) return open_common(std::move(d), out, err);
    void *m = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (m == MAP_FAILED) return io_fail("mmap", -static_cast<int64_t>(errno));
    d->map = m;
    d->map_len = total;
    d->bytes = static_cast<const uint8_t *>(m);
    d->n_bytes = total;
    // loaded state per 64 KiB chunk; a request loads every missing chunk it overlaps, one seek + reads per gap
    constexpr size_t kChunk = size_t(64) << 10;
    std::vector<uint8_t> have((total + kChunk - 1) / kChunk, 0);
    int64_t io_rc = 0;                                             // first reader error (reported after the parse)
    uint8_t *base = static_cast<uint8_t *>(m);
    NeedFn need = [&](size_t off, size_t len) {
        if (io_rc < 0 || len == 0 || off >= total) return;
        const size_t c0 = off / kChunk, c1 = (std::min(off + len, total) - 1) / kChunk;
        for (size_t c = c0; c <= c1; c++) {
            if (have[c]) continue;
            size_t ce = c;
            while (ce + 1 <= c1 && !have[ce + 1]) ce++;
            const size_t lo = c * kChunk, hi = std::min((ce + 1) * kChunk, total);
            int64_t rc = seek(ctx, pos0 + static_cast<int64_t>(lo), SEEK_SET);
            size_t at = lo;
            while (rc >= 0 && at < hi) {
                rc = read(ctx, base + at, hi - at);
                if (rc == 0) break;                                // shorter than seek(End) promised: the rest reads as zeros
                if (rc > 0) at += static_cast<size_t>(std::min<int64_t>(rc, static_cast<int64_t>(hi - at)));
            }
            if (rc < 0) {
                io_rc = rc;
                return;
            }
            for (size_t k = c; k <= ce; k++) have[k] = 1;
            c = ce;
        }
    };
    const int rc = open_common(std::move(d), out, err, &need);
    if (io_rc < 0) {
        if (rc == NAFGPU_OK) {
            nafgpu_close(*out);
            *out = nullptr;
        }
        return io_fail("read", io_rc);
    }
    return rc;
}

void nafgpu_get_header(const nafgpu_decoder *d, nafgpu_header *out) { *out = d->header; }

uint64_t nafgpu_remaining(const nafgpu_decoder *d) { return d->header.number_of_sequences - d->n; }

void nafgpu_close(nafgpu_decoder *d) { delete d; }

void nafgpu_last_error(const nafgpu_decoder *d, nafgpu_error *err) {
    if (d) d->last.to_c(err);
}

// One record (next_record, mod.rs:356-399).  `in_batch`: the record is not the first of a nafgpu_next_batch call -- the views of
// the records in front of it must stay valid, so a record whose bytes are not in the host windows yet is left for the next
// call (NAFGPU_END + 1 = "not now": nothing consumed).
constexpr int kNotNow = NAFGPU_END + 1;
static int next_one(nafgpu_decoder *d, nafgpu_record *rec, bool in_batch) {
    if (d->n >= d->header.number_of_sequences) return NAFGPU_END;          // mod.rs:447-449
    if (d->opts.shard_count > 1)
        return fail(d, Failure::make(NAFGPU_E_INVALID_ARG, "record iteration needs the whole archive (shard_count == 1)"));
    Failure f = ensure_decoded(d, true);
    if (!f.ok()) return fail(d, f);
    if (in_batch && d->use[kLengths] && d->rec_idx < d->rec_ends.size()) {
        const uint64_t s0 = d->rec_idx ? d->rec_ends[d->rec_idx - 1] : 0, l0 = d->rec_ends[d->rec_idx] - s0;
        if (l0 && ((d->use[kSequence] && !d->seq_win.holds(s0, l0)) || (d->use[kQuality] && !d->qual_win.holds(s0, l0)))) return kNotNow;
    }
    std::memset(rec, 0, sizeof *rec);
    // state consumed before an error is not rolled back, as in the reference (mod.rs:391)
    if (d->use[kIds]) {
        if (!d->job.section_failure(kIds).ok()) return fail(d, d->job.section_failure(kIds));
        if (cstring_next(d->ids, &d->ids_pos, &rec->id, &f) < 0) return fail(d, f);
    }
    if (d->use[kComments]) {
        if (!d->job.section_failure(kComments).ok()) return fail(d, d->job.section_failure(kComments));
        if (cstring_next(d->comments, &d->com_pos, &rec->comment, &f) < 0) return fail(d, f);
    }
    uint64_t start = 0, end = 0;
    if (d->use[kLengths]) {
        if (!d->job.section_failure(kLengths).ok()) return fail(d, d->job.section_failure(kLengths));
        if (d->rec_idx < d->rec_ends.size()) {                             // else LengthReader -> None
            start = d->rec_idx ? d->rec_ends[d->rec_idx - 1] : 0;
            end = d->rec_ends[d->rec_idx];
            d->rec_idx++;
            rec->has_length = 1;
            rec->length = end - start;
        }
    }
    if (rec->has_length) {                                                 // mod.rs:373
        const uint64_t l = rec->length;
        if (d->use[kSequence]) {
            if (!d->job.section_failure(kSequence).ok() && !(d->tiled_output && d->seq_win.holds(start, l)))
                return fail(d, d->job.section_failure(kSequence));
            if (end > d->job.n_sequence_bytes())
                return fail(d, Failure::io(NAFGPU_IO_UNEXPECTED_EOF, "sequence section ends before the record does"));
            const uint8_t *p = d->seq_win.get(start, l, &f);
            if (!p) return fail(d, f);
            if (d->header.sequence_type > 1 && !utf8_valid(p, l))          // reader.rs:108-109
                return fail(d, Failure::io(NAFGPU_IO_INVALID_DATA, "invalid UTF-8 in sequence"));
            rec->sequence.ptr = p;
            rec->sequence.len = l;
            rec->sequence.present = 1;
        }
        if (d->use[kQuality]) {
            if (!d->job.section_failure(kQuality).ok() && !(d->tiled_output && d->qual_win.holds(start, l)))
                return fail(d, d->job.section_failure(kQuality));
            if (end > d->job.section_size(kQuality))
                return fail(d, Failure::io(NAFGPU_IO_UNEXPECTED_EOF, "quality section ends before the record does"));
            const uint8_t *p = d->qual_win.get(start, l, &f);
            if (!p) return fail(d, f);
            if (!utf8_valid(p, l)) return fail(d, Failure::io(NAFGPU_IO_INVALID_DATA, "invalid UTF-8 in quality"));
            rec->quality.ptr = p;
            rec->quality.len = l;
            rec->quality.present = 1;
        }
        if (rec->sequence.present && d->use[kMask]) {                      // mod.rs:386-388
            if (!d->job.section_failure(kMask).ok()) return fail(d, d->job.section_failure(kMask));
            if (end > d->mask_covered)
                return fail(d, Failure::io(NAFGPU_IO_UNEXPECTED_EOF, "failed to get mask unit"));
        }
    }
    d->n += 1;
    return NAFGPU_OK;
}

int nafgpu_next(nafgpu_decoder *d, nafgpu_record *rec) {
    if (!d || !rec) return NAFGPU_E_INVALID_ARG;
    return next_one(d, rec, false);
}

int nafgpu_next_batch(nafgpu_decoder *d, nafgpu_record *recs, uint64_t cap, uint64_t *n_out) {
    if (!d || !n_out || (cap && !recs)) return NAFGPU_E_INVALID_ARG;
    *n_out = 0;
    while (*n_out < cap) {
        const int rc = next_one(d, recs + *n_out, *n_out != 0);
        if (rc == kNotNow) break;                                          // the next record needs the host window moved: next call
        if (rc == NAFGPU_END) return *n_out ? NAFGPU_OK : NAFGPU_END;
        if (rc != NAFGPU_OK) return rc;                                    // recs[0 .. *n_out) are good; the error is this call's result
        *n_out += 1;
    }
    return NAFGPU_OK;
}

int nafgpu_decode_all_device(nafgpu_decoder *d, nafgpu_device_result *out) {
    if (!d || !out) return NAFGPU_E_INVALID_ARG;
    if (d->opts.shard_protocol && d->opts.shard_count > 1)
        return fail(d, Failure::make(NAFGPU_E_INVALID_ARG, "this decoder was opened for the shard protocol: use nafgpu_shard_begin / _place / _finish"));
    d->decoded = false;                                                    // every call re-runs the kernels
    Failure f = ensure_decoded(d, false);
    if (!f.ok()) return fail(d, f);
    for (int s = 0; s < kNumSections; s++)
        if (d->use[s] && !d->job.section_failure(s).ok()) return fail(d, d->job.section_failure(s));
    fill_result(d, out);
    return NAFGPU_OK;
}

// ---- the shard protocol (include/nafgpu.h) -------------------------------------------------------------------------
int nafgpu_shard_begin(nafgpu_decoder *d, nafgpu_shard_summary *mine) {
    if (!d || !mine) return NAFGPU_E_INVALID_ARG;
    if (!d->opts.shard_protocol || d->opts.shard_count <= 1)
        return fail(d, Failure::make(NAFGPU_E_INVALID_ARG, "the shard protocol needs opts.shard_protocol = 1 and shard_count > 1"));
    if (!d->fatal.ok()) return fail(d, d->fatal);
    d->decoded = false;
    d->shard_phase = nafgpu_decoder::kShardIdle;
    Failure f = ensure_uploaded(d, false);
    if (!f.ok()) return fail(d, f);
    ShardSummary m[2];
    f = d->job.shard_begin(m);
    if (!f.ok()) return fail(d, f.status == NAFGPU_E_DEVICE ? (d->fatal = f) : f);
    d->shard_phase = nafgpu_decoder::kShardBegun;
    std::memset(mine, 0, sizeof *mine);
    for (int w = 0; w < 2; w++) {
        mine->decoded[w] = m[w].decoded;
        mine->frame_tail[w] = m[w].frame_tail;
        for (int k = 0; k < 3; k++) mine->rep_map[w][k] = m[w].rep_map[k];
        mine->failed[w] = m[w].failed ? 1 : 0;
    }
    return NAFGPU_OK;
}

int nafgpu_shard_place(nafgpu_decoder *d, const nafgpu_shard_summary *all, int n_ranks) {
    if (!d || !all || n_ranks != d->opts.shard_count) return NAFGPU_E_INVALID_ARG;
    if (!d->fatal.ok()) return fail(d, d->fatal);
    if (d->shard_phase != nafgpu_decoder::kShardBegun)
        return fail(d, Failure::make(NAFGPU_E_INVALID_ARG, "nafgpu_shard_place needs a successful nafgpu_shard_begin in front of it"));
    std::vector<ShardSummary> sq(static_cast<size_t>(n_ranks)), ql(static_cast<size_t>(n_ranks));
    for (int r = 0; r < n_ranks; r++)
        for (int w = 0; w < 2; w++) {
            ShardSummary &t = w == 0 ? sq[static_cast<size_t>(r)] : ql[static_cast<size_t>(r)];
            t.decoded = all[r].decoded[w];
            t.frame_tail = all[r].frame_tail[w];
            for (int k = 0; k < 3; k++) t.rep_map[k] = all[r].rep_map[w][k];
            t.failed = all[r].failed[w] != 0;
        }
    Failure f = d->job.shard_place(sq.data(), ql.data(), static_cast<uint32_t>(n_ranks));
    if (!f.ok()) {
        d->shard_phase = nafgpu_decoder::kShardIdle;
        return fail(d, f.status == NAFGPU_E_DEVICE ? (d->fatal = f) : f);
    }
    d->shard_phase = nafgpu_decoder::kShardPlaced;
    return NAFGPU_OK;
}

int nafgpu_shard_halo(nafgpu_decoder *d, int section, uint64_t *recv_bytes, uint64_t *send_bytes, int *tail_ready) {
    if (!d || !recv_bytes || !send_bytes || !tail_ready || section < 0 || section > 1) return NAFGPU_E_INVALID_ARG;
    if (!d->fatal.ok()) return fail(d, d->fatal);
    if (d->shard_phase != nafgpu_decoder::kShardPlaced)
        return fail(d, Failure::make(NAFGPU_E_INVALID_ARG, "the shard protocol is begin, place, then halo / export_tail / import_halo, then finish"));
    bool ready = true;
    Failure f = d->job.shard_halo(section, recv_bytes, send_bytes, &ready);
    *tail_ready = ready ? 1 : 0;
    if (!f.ok()) return fail(d, f.status == NAFGPU_E_DEVICE ? (d->fatal = f) : f);
    return NAFGPU_OK;
}

int nafgpu_shard_export_tail(nafgpu_decoder *d, int section, void *dst, uint64_t n) {
    if (!d || section < 0 || section > 1 || (n && !dst)) return NAFGPU_E_INVALID_ARG;
    if (!d->fatal.ok()) return fail(d, d->fatal);
    if (d->shard_phase != nafgpu_decoder::kShardPlaced)
        return fail(d, Failure::make(NAFGPU_E_INVALID_ARG, "the shard protocol is begin, place, then halo / export_tail / import_halo, then finish"));
    Failure f = d->job.shard_export(section, dst, n);
    if (!f.ok()) return fail(d, f.status == NAFGPU_E_DEVICE ? (d->fatal = f) : f);
    return NAFGPU_OK;
}

int nafgpu_shard_import_halo(nafgpu_decoder *d, int section, const void *src, uint64_t n) {
    if (!d || section < 0 || section > 1 || (n && !src)) return NAFGPU_E_INVALID_ARG;
    if (!d->fatal.ok()) return fail(d, d->fatal);
    if (d->shard_phase != nafgpu_decoder::kShardPlaced)
        return fail(d, Failure::make(NAFGPU_E_INVALID_ARG, "the shard protocol is begin, place, then halo / export_tail / import_halo, then finish"));
    Failure f = d->job.shard_import(section, src, n);
    if (!f.ok()) return fail(d, f.status == NAFGPU_E_DEVICE ? (d->fatal = f) : f);
    return NAFGPU_OK;
}

int nafgpu_shard_finish(nafgpu_decoder *d, nafgpu_device_result *out) {
    if (!d || !out) return NAFGPU_E_INVALID_ARG;
    if (!d->fatal.ok()) return fail(d, d->fatal);
    if (d->shard_phase != nafgpu_decoder::kShardPlaced)
        return fail(d, Failure::make(NAFGPU_E_INVALID_ARG, "the shard protocol is begin, place, then halo / export_tail / import_halo, then finish"));
    d->shard_phase = nafgpu_decoder::kShardIdle;
    Failure f = d->job.shard_finish();
    if (!f.ok()) return fail(d, f.status == NAFGPU_E_DEVICE ? (d->fatal = f) : f);   // (a corrupt range is this decode's error, not the decoder's end)
    f = after_decode(d);
    if (!f.ok()) return fail(d, f);
    d->decoded = true;
    for (int s = 0; s < kNumSections; s++)
        if (d->use[s] && !d->job.section_failure(s).ok()) return fail(d, d->job.section_failure(s));
    fill_result(d, out);
    return NAFGPU_OK;
}

int nafgpu_format_device(nafgpu_decoder *d, nafgpu_text_result *out) {
    if (!d || !out) return NAFGPU_E_INVALID_ARG;
    std::memset(out, 0, sizeof *out);
    if (d->decoded && d->tiled_output) d->decoded = false;                 // text needs the whole output in HBM
    Failure f = ensure_decoded(d, false);
    if (!f.ok()) return fail(d, f);
    for (int s = 0; s < kNumSections; s++)
        if (d->use[s] && !d->job.section_failure(s).ok()) return fail(d, d->job.section_failure(s));
    if (!d->use[kSequence] || !d->use[kLengths])
        return fail(d, Failure::make(NAFGPU_E_INVALID_ARG, "text output needs the sequence and length fields"));
    const uint64_t n_rec = std::min<uint64_t>(d->header.number_of_sequences, d->job.n_records());
    const bool fastq = d->use[kQuality] && d->job.job(kQuality).ready();
    f = d->job.format_text(d->use[kIds], d->use[kComments], fastq, n_rec, &out->d_text, &out->n_text, &out->ms);
    if (!f.ok()) return fail(d, f);
    out->n_records = n_rec;
    out->fastq = fastq ? 1 : 0;
    return NAFGPU_OK;
}

int nafgpu_copy_to_host(nafgpu_decoder *d, const void *d_ptr, uint64_t n, void *dst) {
    if (!d || (n && (!d_ptr || !dst))) return NAFGPU_E_INVALID_ARG;
    Failure f = d->job.copy_to_host(dst, d_ptr, static_cast<size_t>(n));
    return f.ok() ? NAFGPU_OK : fail(d, f);
}

int nafgpu_upload(nafgpu_decoder *d) {
    if (!d) return NAFGPU_E_INVALID_ARG;
    Failure f = ensure_uploaded(d, false);
    return f.ok() ? NAFGPU_OK : fail(d, f);
}

int nafgpu_device_synchronize(int device) {
    if (device >= 0 && hipSetDevice(device) != hipSuccess) return NAFGPU_E_DEVICE;
    return hipDeviceSynchronize() == hipSuccess ? NAFGPU_OK : NAFGPU_E_DEVICE;
}

int nafgpu_trim_device_memory(int device) {
    if (device >= 0 && hipSetDevice(device) != hipSuccess) return NAFGPU_E_DEVICE;
    if (device < 0 && hipGetDevice(&device) != hipSuccess) return NAFGPU_E_DEVICE;
    trim_device_memory(device);
    return NAFGPU_OK;
}

int nafgpu_hash64_device(const nafgpu_decoder *d, const void *d_ptr, uint64_t n, uint64_t *out) {
    return nafgpu_hash64_device_at(d, d_ptr, n, 0, out);
}

int nafgpu_hash64_device_at(const nafgpu_decoder *d, const void *d_ptr, uint64_t n, uint64_t first_chunk, uint64_t *out) {
    if (!d || !out) return NAFGPU_E_INVALID_ARG;
    Failure f = const_cast<nafgpu_decoder *>(d)->job.hash_device(d_ptr, n, first_chunk, out);
    return f.status;
}

uint64_t nafgpu_hash64_host(const uint8_t *p, uint64_t n) { return hash64_host(p, n); }

uint64_t nafgpu_hash64_host_at(const uint8_t *p, uint64_t n, uint64_t first_chunk) { return hash64_host(p, n, first_chunk); }

int nafgpu_zstd_decompress(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *produced, int device,
                           nafgpu_error *err) {
    // one section = one job on a private stream (zstd::stream::read::Decoder, mod.rs:221-223)
    Failure f;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        f = Failure::make(NAFGPU_E_DEVICE, "no HIP device available: libnafgpu decodes on the GPU only");
        f.to_c(err);
        return f.status;
    }
    if (device >= 0) (void)hipSetDevice(device);
    hipStream_t stream = nullptr;
    if (hipStreamCreate(&stream) != hipSuccess) {
        f = Failure::make(NAFGPU_E_DEVICE, "hipStreamCreate failed");
        f.to_c(err);
        return f.status;
    }
    {
        SectionJob job;
        // the decoded size is not part of a NAF-less call: take it from a host-side walk when the
        // frame is literal-only, else trust `cap` as the exact size
        ZPlan probe;
        bool trunc = false;
        std::string perr = build_zplan(src, n, &probe, &trunc);
        if (!perr.empty()) {
            f = Failure::io(trunc ? NAFGPU_IO_UNEXPECTED_EOF : NAFGPU_IO_INVALID_DATA, "zstd: " + perr);
        } else {
            const uint64_t expect = probe.seq_blocks.empty() ? probe.known_out : cap;
            if (expect > cap) {
                f = Failure::io(NAFGPU_IO_INVALID_DATA, "zstd: destination buffer too small");
            } else {
                f = job.prepare(src, n, expect, stream, SectionOptions());
                if (f.ok()) {
                    job.run(stream, nullptr);
                    f = job.check(stream);
                }
                if (f.ok() && expect) {
                    if (hipMemcpyAsync(dst, job.out(), expect, hipMemcpyDeviceToHost, stream) != hipSuccess ||
                        hipStreamSynchronize(stream) != hipSuccess)
                        f = Failure::make(NAFGPU_E_DEVICE, "device-to-host copy failed");
                }
                if (f.ok() && produced) *produced = static_cast<size_t>(expect);
            }
        }
    }
    (void)hipStreamDestroy(stream);
    f.to_c(err);
    return f.status;
}

int nafgpu_abi_version(void) { return NAFGPU_ABI_VERSION; }

void nafgpu_test_hooks(int enable) { set_test_hooks(enable != 0); }

int nafgpu_device_info(int device, char *name, size_t cap, uint64_t *hbm_bytes, int *compute_units) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return NAFGPU_E_DEVICE;
    if (device < 0) device = 0;
    if (device >= count) return NAFGPU_E_INVALID_ARG;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) return NAFGPU_E_DEVICE;
    if (name && cap) {
        std::snprintf(name, cap, "%s (%s)", p.name, p.gcnArchName);
    }
    if (hbm_bytes) *hbm_bytes = p.totalGlobalMem;
    if (compute_units) *compute_units = p.multiProcessorCount;
    return NAFGPU_OK;
}

}  // extern "C"
