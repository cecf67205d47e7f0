// engine.cpp -- see engine.h
#include "engine.h"

#include <cstdio>
#include <cstdlib>

#include <chrono>
#include <cstring>

namespace nafgpu {

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

bool DevBuf::alloc(size_t bytes) {
    if (ptr_ && bytes <= size_) return true;
    release();
    void *p = nullptr;
    if (!hip_ok(hipMalloc(&p, bytes ? bytes : 16))) return false;
    ptr_ = p;
    size_ = bytes ? bytes : 16;
    return true;
}

bool DevBuf::upload(const void *host, size_t bytes, hipStream_t stream) {
    if (!alloc(bytes)) return false;
    if (bytes == 0) return true;
    return hip_ok(hipMemcpyAsync(ptr_, host, bytes, hipMemcpyHostToDevice, stream));
}

void DevBuf::release() {
    if (ptr_) (void)hipFree(ptr_);
    ptr_ = nullptr;
    size_ = 0;
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
Failure SectionJob::prepare(const uint8_t *host_payload, size_t n, uint64_t expect_size, const uint8_t *d_payload,
                            hipStream_t stream, uint32_t ascii_t_char, uint32_t shard_rank, uint32_t shard_count) {
    ready_ = false;
    t_char_ = ascii_t_char;
    plan_ = ZPlan();
    const double t0 = now_ms();
    bool truncated = false;
    std::string err = build_zplan(host_payload, n, &plan_, &truncated, shard_rank, shard_count);
    plan_ms_ = static_cast<float>(now_ms() - t0);
    if (!err.empty())
        return Failure::io(truncated ? NAFGPU_IO_UNEXPECTED_EOF : NAFGPU_IO_INVALID_DATA, "zstd: " + err);
    expect_ = expect_size;
    d_src_ = d_payload;
    n_blocks_ = plan_.blk_size.size();
    n_streams_ = plan_.streams.size();
    n_tasks_ = plan_.tasks.size();
    n_copies_ = plan_.copies.size();
    n_seq_blocks_ = plan_.seq_blocks.size();
    if ((plan_.seq_blocks.empty() && plan_.known_out != expect_size) ||
        expect_size > static_cast<uint64_t>(plan_.blk_size.size()) * kBlockMax)     // untrusted: no block decodes to more than 128 KiB
        return Failure::io(NAFGPU_IO_INVALID_DATA, "zstd: decoded size differs from the size recorded in the archive");
    out0_ = plan_.sharded ? plan_.shard_out0 : 0;
    out1_ = plan_.sharded ? plan_.shard_out1 : expect_size;
    classes_ = plan_.classes;                              // launch classes of the Huffman tasks (plan.h: HufClass)
    if (std::getenv("NAFGPU_DEBUG_PLAN")) {
        std::fprintf(stderr, "[nafgpu] section plan: %zu blocks, %zu streams, %zu seq blocks, %llu sequences, literal buffer %llu B; task classes:",
                     n_blocks_, n_streams_, n_seq_blocks_, static_cast<unsigned long long>(plan_.n_sequences),
                     static_cast<unsigned long long>(plan_.lit_bytes));
        for (const HufClass &c : classes_)
            std::fprintf(stderr, " {%u tasks, tbl %u, %s%s, lds %u B}", c.n_tasks, c.tbl, c.to_lit ? "lit" : "out", c.seg ? "+seg" : "", c.lds_bytes);
        std::fprintf(stderr, "\n");
    }
    bool ok = d_out_.alloc(static_cast<size_t>(out_bytes()) + 64) && d_status_.alloc(64) &&
              d_blk_size_.upload(plan_.blk_size.data(), n_blocks_ * sizeof(uint32_t), stream) &&
              d_blk_base_.alloc((n_blocks_ + 1) * sizeof(uint64_t)) && d_scan_tmp_.alloc(scan_tmp_bytes(n_blocks_)) &&
              d_streams_.upload(plan_.streams.data(), n_streams_ * sizeof(HufStream), stream) &&
              d_tasks_.upload(plan_.tasks.data(), n_tasks_ * sizeof(HufTask), stream) &&
              d_tbl_copies_.upload(plan_.tbl_copies.data(), plan_.tbl_copies.size() * sizeof(HufTblCopy), stream) &&
              d_pool_.upload(plan_.huf_pool.data(), plan_.huf_pool.size() * sizeof(uint16_t), stream) &&
              d_dicts_.upload(plan_.dict_pool.data(), plan_.dict_pool.size(), stream) &&
              d_copies_.upload(plan_.copies.data(), n_copies_ * sizeof(CopyTask), stream) &&
              d_seq_blocks_.upload(plan_.seq_blocks.data(), n_seq_blocks_ * sizeof(SeqBlock), stream) &&
              d_cells_.upload(plan_.fse_pool.data(), plan_.fse_pool.size() * sizeof(SeqCell), stream) &&
              d_lit_.alloc(static_cast<size_t>(plan_.lit_bytes) + 64) &&
              d_seqs_.alloc(static_cast<size_t>(plan_.n_sequences) * sizeof(Seq) + 16) &&
              d_meta_.alloc(static_cast<size_t>(plan_.n_sequences) * sizeof(SeqMeta) + 16) &&
              d_rep_final_.alloc(n_seq_blocks_ * 12 + 16) && d_rep_init_.alloc(n_seq_blocks_ * 12 + 16) &&
              d_rep_scratch_.alloc((n_seq_blocks_ / 64 + 1) * 24 + 16) &&
              d_blk_pending_.alloc(n_seq_blocks_ * 4 + 16) && d_counters_.alloc(64);
    if (!ok) return Failure::make(NAFGPU_E_DEVICE, "out of device memory while preparing a section");
    // the host vectors were consumed by asynchronous copies: keep them until the stream drains
    if (!hip_ok(hipStreamSynchronize(stream))) return Failure::make(NAFGPU_E_DEVICE, "upload of task lists failed");
    // only the counts are needed from here on
    std::vector<HufStream>().swap(plan_.streams);
    std::vector<uint16_t>().swap(plan_.huf_pool);
    std::vector<SeqCell>().swap(plan_.fse_pool);
    std::vector<CopyTask>().swap(plan_.copies);
    ready_ = true;
    return Failure();
}

SectionJob::~SectionJob() {
    if (ev_fork_) (void)hipEventDestroy(ev_fork_);
    if (ev_join_) (void)hipEventDestroy(ev_join_);
}

void SectionJob::run(hipStream_t stream, StageTimer *timer, hipStream_t aux) {
    if (!ready_) return;
    uint32_t *status = d_status_.as<uint32_t>();
    (void)hipMemsetAsync(status, 0, 64, stream);
    if (n_seq_blocks_) {
        if (timer) timer->begin(stream, StageTimer::kSeqLz);
        // blk_size of blocks with sequences is rewritten in full by k_seq_decode: re-runs are idempotent
        launch_seq_decode(stream, d_src_, d_seq_blocks_.as<SeqBlock>(), static_cast<uint32_t>(n_seq_blocks_),
                          d_cells_.as<SeqCell>(), d_seqs_.as<Seq>(), d_blk_size_.as<uint32_t>(), d_rep_final_.as<uint32_t>(),
                          status);
        if (timer) timer->end(stream);
    }
    if (timer) timer->begin(stream, StageTimer::kOther);
    launch_scan_blocks(stream, d_blk_size_.as<uint32_t>(), n_blocks_, d_blk_base_.as<uint64_t>(), d_scan_tmp_.bytes(),
                       expect_, status);
    const bool ascii = t_char_ != 0;
    // kernels address the section output as out + position; a shard holds positions [out0_, out1_)
    uint8_t *const out_base = d_out_.bytes() - out0_ * (ascii ? 2 : 1);
    launch_copy_fill(stream, d_src_, d_copies_.as<CopyTask>(), static_cast<uint32_t>(n_copies_),
                     d_blk_base_.as<uint64_t>(), out_base, d_lit_.bytes(), ascii, t_char_, status);
    if (timer) timer->end(stream);
    // K1, one launch per class.  Streams that write the section output (blocks without sequences, and -- segment by
    // segment -- blocks with a few) and streams that feed the literal buffer (blocks with many sequences).  When
    // both kinds exist the literal-buffer classes run on `aux` beside the others, so the launches share the chip
    // instead of each ending in a half-empty tail; K4 then waits for both.  One timed span covers the phase.
    auto launch_class = [&](const HufClass &c, hipStream_t st) {
        launch_huf_decode(st, d_src_, d_tasks_.as<HufTask>(), c, d_tbl_copies_.as<HufTblCopy>(), d_streams_.as<HufStream>(),
                          d_pool_.as<uint16_t>(), d_blk_base_.as<uint64_t>(), out_base, d_lit_.bytes(), d_seq_blocks_.as<SeqBlock>(),
                          d_seqs_.as<Seq>(), d_dicts_.bytes(), ascii, t_char_, status);
    };
    bool have_direct = false, have_lit = false;
    for (const HufClass &c : classes_) (c.to_lit ? have_lit : have_direct) = true;
    if (have_direct || have_lit) {
        if (timer) timer->begin(stream, StageTimer::kHuf);
        bool forked = false;
        if (have_direct && have_lit && aux) {
            if (!ev_fork_) (void)hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming);
            if (!ev_join_) (void)hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming);
            forked = ev_fork_ && ev_join_ && hip_ok(hipEventRecord(ev_fork_, stream)) && hip_ok(hipStreamWaitEvent(aux, ev_fork_, 0));
        }
        if (forked) {
            for (const HufClass &c : classes_)
                if (c.to_lit) launch_class(c, aux);
            (void)hipEventRecord(ev_join_, aux);
            for (const HufClass &c : classes_)
                if (!c.to_lit) launch_class(c, stream);
            (void)hipStreamWaitEvent(stream, ev_join_, 0);
        } else {
            for (const HufClass &c : classes_) launch_class(c, stream);
        }
        if (timer) timer->end(stream);
    }
    if (n_seq_blocks_) {
        if (timer) timer->begin(stream, StageTimer::kSeqLz);
        LzArgs la{};
        la.blocks = d_seq_blocks_.as<SeqBlock>();
        la.n_blocks = static_cast<uint32_t>(n_seq_blocks_);
        la.n_sequences = plan_.n_sequences;
        la.n_elems = expect_;
        la.seqs = d_seqs_.as<Seq>();
        la.lit = d_lit_.bytes();
        la.blk_base = d_blk_base_.as<uint64_t>();
        la.rep_final = d_rep_final_.as<uint32_t>();
        la.rep_init = d_rep_init_.as<uint32_t>();
        la.rep_scratch = d_rep_scratch_.as<uint32_t>();
        la.meta = d_meta_.as<SeqMeta>();
        la.blk_pending = d_blk_pending_.as<uint32_t>();
        la.out = out_base;
        la.t_char = t_char_;
        la.status = status;
        la.counters = d_counters_.as<unsigned long long>();
        // Dense or sparse?  Where the matches are a good part of the output (level-3 DNA, quality strings) the frame is
        // swept element by element (one word of scratch per output element, allocated on the first run and kept);
        // a handful of matches in gigabytes of literals (real genomes at level 1) are visited one by one.
        const uint64_t match_elems = expect_ > plan_.known_out ? expect_ - plan_.known_out : 0;   // known_out = everything but the match bytes
        const char *force = std::getenv("NAFGPU_LZ_MODE");                                      // tests: "dense" / "sparse"
        bool dense = plan_.n_sequences >= 4096 && match_elems * 2 >= expect_;                      // (level-3 DNA, a quarter of it matches at random
                                                                                                   //  distances, is faster match by match: 38 against 46 ms)
        if (force) dense = force[0] == 'd';
        lz_dense_ = false;
        if (dense && d_pj_dist_.alloc_items(expect_, sizeof(uint32_t), 64) && d_pj_tiles_.alloc_items(lz_pj_tiles(expect_), sizeof(uint32_t), 64)) {
            la.pj_dist = d_pj_dist_.as<uint32_t>();
            la.pj_tiles = d_pj_tiles_.as<uint32_t>();
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
            const bool index = plan_.n_sequences < 0xFFFFFFFFull && plan_.n_sequences * 1024 >= expect_ &&
                               d_lz_index_.alloc_items((expect_ >> 7) + 2, sizeof(uint32_t));
            la.cidx = index ? d_lz_index_.as<uint32_t>() : nullptr;
            la.n_idx_chunks = (static_cast<uint64_t>(expect_) >> 7) + 2;
            if (!la.roff) {                                    // (cannot happen short of a device out of memory: flag the section)
                (void)hipMemsetAsync(status, 0xFF, 4, stream);
            }
        }
        if (la.pj_dist || la.roff) launch_lz_execute(stream, la, ascii);
        if (timer) timer->end(stream);
    }
}

Failure SectionJob::check(hipStream_t stream) {
    if (!ready_) return Failure();
    uint32_t st[2] = {0, 0};
    if (!hip_ok(hipMemcpyAsync(st, d_status_.bytes(), sizeof st, hipMemcpyDeviceToHost, stream)) ||
        !hip_ok(hipStreamSynchronize(stream)))
        return Failure::make(NAFGPU_E_DEVICE, "device status read-back failed");
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
ArchiveJob::~ArchiveJob() {
    if (ev_fork_) (void)hipEventDestroy(ev_fork_);
    if (ev_join_) (void)hipEventDestroy(ev_join_);
    if (aux_stream_) (void)hipStreamDestroy(aux_stream_);
    if (stream_) (void)hipStreamDestroy(stream_);
}

Failure ArchiveJob::init(int device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (!hip_ok(e) || count <= 0)
        return Failure::make(NAFGPU_E_DEVICE, "no HIP device available: libnafgpu decodes on the GPU only");
    if (device < 0) {
        if (!hip_ok(hipGetDevice(&device))) device = 0;
    }
    if (device >= count) return Failure::make(NAFGPU_E_INVALID_ARG, "device ordinal out of range");
    e = hipSetDevice(device);
    if (!hip_ok(e)) return dev_fail("hipSetDevice", e);
    device_ = device;
    if (!stream_) {
        e = hipStreamCreate(&stream_);
        if (!hip_ok(e)) return dev_fail("hipStreamCreate", e);
        if (!hip_ok(hipStreamCreate(&aux_stream_))) aux_stream_ = nullptr;   // optional: K1 then stays on one stream
    }
    return Failure();
}

Failure ArchiveJob::upload(const uint8_t *bytes, size_t n, const nafgpu_header &h, const SectionInfo sec[kNumSections],
                           const ArchiveOptions &opt) {
    (void)hipSetDevice(device_);
    h_ = h;
    opt_ = opt;
    is_nuc_ = h.sequence_type <= 1;
    // ---- archive bytes -> HBM, once
    const double t0 = now_ms();
    if (!d_archive_.alloc(kSrcFrontPad + n + kSrcBackPad)) return Failure::make(NAFGPU_E_DEVICE, "out of device memory");
    (void)hipMemsetAsync(d_archive_.bytes(), 0, kSrcFrontPad, stream_);
    (void)hipMemsetAsync(d_archive_.bytes() + kSrcFrontPad + n, 0, kSrcBackPad, stream_);
    if (n && !hip_ok(hipMemcpyAsync(d_archive_.bytes() + kSrcFrontPad, bytes, n, hipMemcpyHostToDevice, stream_)))
        return Failure::make(NAFGPU_E_DEVICE, "archive upload failed");
    if (!hip_ok(hipStreamSynchronize(stream_))) return Failure::make(NAFGPU_E_DEVICE, "archive upload failed");
    h2d_ms_ = static_cast<float>(now_ms() - t0);
    // ---- per-section host plan + task upload
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
        const uint32_t t_char = (s == kSequence && is_nuc_) ? (h.sequence_type == 1 ? 'U' : 'T') : 0;
        const bool shard_this = s == kSequence && opt.shard_count > 1;
        fail_[s] = job_[s].prepare(bytes + sec[s].offset, static_cast<size_t>(sec[s].compressed_size), expect,
                                   d_archive_.bytes() + kSrcFrontPad + sec[s].offset, stream_, t_char,
                                   shard_this ? opt.shard_rank : 0, shard_this ? opt.shard_count : 1);
        plan_ms_ += job_[s].host_plan_ms();
        if (fail_[s].status == NAFGPU_E_DEVICE) return fail_[s];
        if (fail_[s].ok()) compressed_ += sec[s].compressed_size;
    }
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

Failure ArchiveJob::decode() {
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
    const bool want_mask = job_[kMask].ready() && job_[kSequence].ready() && job_[kLengths].ready();
    for (int s = 0; s <= kMask; s++) job_[s].run(stream_, &timer_, aux_stream_);
    bool scans_forked = false;
    auto scans = [&](hipStream_t st) {
        if (job_[kLengths].ready())                                // LengthReader, reader.rs:48-67
            launch_scan_runs_u32(st, job_[kLengths].out(), rec_cap_, d_rec_ends_.as<uint64_t>(), rec_cap_, d_scan_tmp_.bytes(),
                                 &totals[0], status);
        if (want_mask)                                             // MaskReader, reader.rs:198-231
            launch_scan_runs_u8(st, job_[kMask].out(), mask_cap_, d_mask_ends_.as<uint64_t>(), mask_cap_, d_scan_tmp_.bytes(),
                                &totals[1], status);
    };
    if (aux_stream_ && job_[kSequence].ready()) {
        if (!ev_fork_) (void)hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming);
        if (!ev_join_) (void)hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming);
        scans_forked = ev_fork_ && ev_join_ && hip_ok(hipEventRecord(ev_fork_, stream_)) &&
                       hip_ok(hipStreamWaitEvent(aux_stream_, ev_fork_, 0));
    }
    if (scans_forked) {
        scans(aux_stream_);
        (void)hipEventRecord(ev_join_, aux_stream_);
    }
    for (int s = kMask + 1; s < kNumSections; s++) job_[s].run(stream_, &timer_, aux_stream_);
    timer_.begin(stream_, StageTimer::kOther);
    if (scans_forked)
        (void)hipStreamWaitEvent(stream_, ev_join_, 0);
    else
        scans(stream_);
    // (nucleotide sequence sections come out of their SectionJob already expanded to ASCII:
    //  SequenceReader::read_nucleotide, reader.rs:121-172, is fused into the zstd kernels)
    if (want_mask) {                                               // mod.rs:386-388, 402-441
        const uint64_t mult = is_nuc_ ? 2 : 1;
        const uint64_t lo = job_[kSequence].shard_out0() * mult, hi = job_[kSequence].shard_out1() * mult;   // bases held here
        uint8_t *seq = job_[kSequence].out_mut() - lo;         // addressed by global base index
        const uint64_t n_all = is_nuc_ ? mask_total_bases_ : std::min<uint64_t>(mask_total_bases_, job_[kSequence].total_size());
        launch_mask_apply(stream_, seq, n_all, lo, hi, d_mask_ends_.as<uint64_t>(), &totals[1], d_rec_ends_.as<uint64_t>(),
                          &totals[0], mask_cap_, opt_.spec_mask ? 1 : 0, status);
    }
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
    if (job_[kSequence].ready() && !is_nuc_) launch_utf8_check(stream_, job_[kSequence].out(), job_[kSequence].size(), utf8, kSequence);
    if (job_[kQuality].ready()) launch_utf8_check(stream_, job_[kQuality].out(), job_[kQuality].size(), utf8, kQuality);
    timer_.end(stream_);
    timer_.mark_total_end(stream_);
    ScanTotals host_totals[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    utf8_invalid_ = 0;
    (void)hipMemcpyAsync(&utf8_invalid_, utf8, sizeof utf8_invalid_, hipMemcpyDeviceToHost, stream_);
    if (!hip_ok(hipMemcpyAsync(host_totals, totals, sizeof host_totals, hipMemcpyDeviceToHost, stream_)) ||
        !hip_ok(hipStreamSynchronize(stream_)))
        return Failure::make(NAFGPU_E_DEVICE, std::string("decode failed: ") + hipGetErrorString(hipGetLastError()));
    rec_totals_ = host_totals[0];
    mask_totals_ = host_totals[1];
    id_totals_ = host_totals[2];
    com_totals_ = host_totals[3];
    times_ = timer_.collect();
    for (int s = 0; s < kNumSections; s++) {
        if (!job_[s].ready()) continue;
        Failure f = job_[s].check(stream_);
        if (f.status == NAFGPU_E_DEVICE) return f;
        if (!f.ok()) fail_[s] = f;
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
