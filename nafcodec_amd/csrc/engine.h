// engine.h -- device-side orchestration: buffers in HBM, kernel sequence per section, timing.
//
// Data layout in HBM (hipMalloc'ed, from 32 MiB on address ranges backed by hipMemCreate chunks; 256-byte aligned, sized for a 288 GB device):
//   archive   [256 B pad][archive bytes][64 B pad]      compressed input, uploaded once
//   per zstd section:
//     out       decoded bytes (packed 4-bit for DNA/RNA sequence, text otherwise)
//     lit       literal buffer, only for blocks that have LZ sequences (16-B aligned per block)
//     seqs      {ll, ml, offset_value} triples of those blocks
//     blk_size  u32 per zstd block / blk_base u64 exclusive scan (+ total)
//     task lists (plan.h) and the Huffman / FSE table pools
//   ascii     one byte per base, masked, contiguous over all records   (DNA/RNA only)
//   rec_ends  u64 inclusive prefix sum of record lengths
//   mask_ends u64 inclusive prefix sum of mask runs
#pragma once
#include <algorithm>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <future>
#include <vector>

#include "container.h"
#include "kernels.h"
#include "zplan.h"

namespace nafgpu {

class DevBuf {
public:
    DevBuf() = default;
    ~DevBuf() { release(true); }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    bool alloc(size_t bytes);                 // contents undefined
    bool alloc_items(uint64_t count, uint64_t item_bytes, uint64_t extra_bytes = 0);   // count * item_bytes + extra_bytes, overflow-checked
    bool upload(const void *host, size_t bytes, hipStream_t stream);   // alloc + async H2D
    void release(bool dying = false);       // dying: the owner goes away (its streams are drained): a small buffer goes to the cache below
    void view(void *p, size_t bytes);         // a piece of another buffer: not owned, release() only forgets it
    template <class T>
    T *as() const { return static_cast<T *>(ptr_); }
    uint8_t *bytes() const { return static_cast<uint8_t *>(ptr_); }
    size_t size() const { return size_; }

private:
    bool alloc_mapped(size_t bytes);          // an address range backed by hipMemCreate chunks (engine.cpp: why)
    void *ptr_ = nullptr;
    size_t size_ = 0, reserved_ = 0;          // reserved_ != 0: ptr_ is such a range
    bool view_ = false;
    int cache_dev_ = -1;                      // >= 0: ptr_ is a size-class buffer of that device's cache (engine.cpp: SmallCache)
#ifndef NAFGPU_EMU
    std::vector<hipMemGenericAllocationHandle_t> chunks_;
    size_t chunk_bytes_ = 0;                  // every chunk but the last maps this many bytes
#endif
};

struct StageTimes {          // milliseconds, summed over launches of the last run
    float huf = 0, seq_lz = 0, unpack = 0, other = 0, total = 0;
    uint32_t huf_launches = 0;
};

// Records HIP events around kernel groups on one stream and sums them per category afterwards.
class StageTimer {
public:
    enum Cat { kHuf = 0, kSeqLz, kUnpack, kOther, kNumCats };
    ~StageTimer();
    void begin(hipStream_t s, Cat c);
    void end(hipStream_t s);
    void mark_total_begin(hipStream_t s);
    void mark_total_end(hipStream_t s);
    StageTimes collect();                     // call after the stream is synchronised
    void reset();

private:
    struct Span {
        hipEvent_t a, b;
        Cat cat;
    };
    hipEvent_t get();
    std::vector<Span> spans_;
    std::vector<hipEvent_t> pool_;
    size_t used_ = 0;
    hipEvent_t t0_ = nullptr, t1_ = nullptr;
    bool open_ = false;
};

// How one section is cut up: which block range this process decodes (multi-GPU shards) and in how many
// pieces (tiles), so that neither the compressed bytes nor the scratch memory -- nor, on the iterator
// path, the output -- have to be resident whole.
struct SectionOptions {
    uint32_t t_char = 0;             // != 0: nucleotide sequence section -- every byte bound for the section output is expanded to its
                                     // two IUPAC characters on the way ('T' / 'U' for code 1); the packed form never exists in HBM
    uint32_t shard_rank = 0, shard_count = 1;   // decode only this rank's contiguous block range (sections without LZ sequences)
    uint64_t tile_blocks = 0;        // > 0: decode in tiles of at most this many zstd blocks
    bool tiled_output = false;       // the output buffer holds ONE tile (and the LZ window in front of it) at a time
    bool shard_protocol = false;     // the ranks exchange what their ranges need from each other (shard_begin / shard_place / ...):
                                     // a section with LZ sequences is sharded too
};

// What one rank tells the others about its block range of one section (nafgpu_shard_summary holds two of these, spread
// over its arrays).
struct ShardSummary {
    uint64_t decoded = 0;            // elements the range decodes to
    uint64_t frame_tail = 0;         // of which: behind the start of the last frame that begins inside the range
    uint32_t rep_map[3] = {kRepToken | (0u << 24), kRepToken | (1u << 24), kRepToken | (2u << 24)};   // identity
    bool failed = false;
};

// One Zstandard-compressed NAF section on the device.
class SectionJob {
public:
    SectionJob() = default;
    SectionJob(const SectionJob &) = delete;
    SectionJob &operator=(const SectionJob &) = delete;
    ~SectionJob();
    // Walks the payload on the host (zplan), chooses the block range and the tiles, allocates the output.
    // One tile (the usual case): its compressed bytes and task lists are uploaded here and stay resident --
    // run() then only enqueues kernels.  Several tiles: each is uploaded when it is decoded (decode_tile).
    Failure prepare(const uint8_t *host_payload, size_t n, uint64_t expect_size, hipStream_t stream, const SectionOptions &opt);
    // The host walk alone, ahead of prepare() (which then finds it done): no HIP call, so it can run beside the start of the runtime.
    void walk(const uint8_t *host_payload, size_t n);
    // Waits for what a background thread may still be reading or writing (the source bytes of the next tile).
    void drain();
    // Enqueues the decode kernels of the resident tile.  Results: out() holds size() bytes once the stream is done.
    // aux: a second stream (or null) on which the literal-buffer Huffman tasks run beside the direct ones
    void run(hipStream_t stream, StageTimer *timer, hipStream_t aux = nullptr);
    // K2 of this section enqueued on `st` ahead of run(): a chain per block that leaves most of the chip idle, so it
    // goes beside the section decoded before this one; run() then waits for it instead of launching it.
    void run_k2_ahead(hipStream_t st);
    // After synchronisation: device status -> Failure
    Failure check(hipStream_t stream);
    // the same in two steps, so that the read-backs of several sections wait for the stream ONCE: check_begin enqueues them into
    // `slot` (pinned host memory, kCheckSlotBytes of it), check_end reads the slot after the caller has synchronised the stream
    static constexpr size_t kCheckSlotBytes = 96;
    bool check_begin(hipStream_t stream, uint8_t *slot);
    Failure check_end(const uint8_t *slot);
    // Tiles: uploads tile t (they are decoded in order), decodes it, synchronises, keeps what the next one needs
    // (repeat offsets, position, the LZ window).  Tile t's output then is tile_data() .. + tile_len() elements.
    uint32_t n_tiles() const { return static_cast<uint32_t>(tiles_.size()); }
    Failure decode_tile(uint32_t t, hipStream_t stream, StageTimer *timer, hipStream_t aux = nullptr);
    // ---- the shard protocol (SectionOptions.shard_protocol; include/nafgpu.h has the sequence of calls)
    // enqueues what needs nothing from the other ranks: entropy decode, block sizes, the range's repeat-offset map
    void shard_begin(hipStream_t stream, StageTimer *timer, hipStream_t aux);
    // synchronises; what the other ranks need to know
    Failure shard_summary(hipStream_t stream, ShardSummary *mine);
    // all[r] = rank r's summary of this section: places the range, enqueues literals and every match that does not
    // depend on the window in front of the range
    Failure shard_place(const ShardSummary *all, uint32_t n_ranks, hipStream_t stream, StageTimer *timer, hipStream_t aux);
    uint64_t halo_recv_bytes() const { return halo_elems_ * (t_char_ ? 2 : 1); }   // (valid after shard_place)
    uint64_t tail_send_bytes() const { return send_elems_ * (t_char_ ? 2 : 1); }
    Failure tail_ready(hipStream_t stream, bool *ready);           // synchronises
    Failure export_tail(void *dst, uint64_t n, hipStream_t stream);   // the last tail_send_bytes() of [window in front | this range's output]
    Failure import_halo(const void *src, uint64_t n, hipStream_t stream, StageTimer *timer);   // ... and finishes what waited for it
    bool in_protocol() const { return ready_ && opt_.shard_protocol && opt_.shard_count > 1; }
    uint32_t tiles_done() const { return tiles_done_; }
    uint64_t tile_pos0() const { return tile_pos0_; }             // decoded-byte position (in the section) of the loaded tile's first byte
    uint64_t tile_len() const { return tile_len_; }               // decoded bytes of the loaded tile (known once it is decoded)
    const uint8_t *tile_data() const;                             // device address of the loaded tile's first output byte

    const uint8_t *out() const { return d_out_.bytes() + out_shift_; }
    uint8_t *out_mut() const { return d_out_.bytes() + out_shift_; }
    uint64_t size() const { return out1_ - out0_; }                 // decoded zstd bytes of this process's range (the whole section unless sharded)
    uint64_t total_size() const { return expect_; }                 // decoded zstd bytes of the whole section
    uint64_t shard_out0() const { return out0_; }
    uint64_t shard_out1() const { return out1_; }
    bool sharded() const { return sharded_; }
    bool tiled_output() const { return opt_.tiled_output && tiles_.size() > 1; }
    uint64_t out_bytes() const { return size() * (t_char_ ? 2 : 1); }   // bytes behind out() in whole-output mode
    bool ascii() const { return t_char_ != 0; }
    bool ready() const { return ready_; }
    const ZPlan &plan() const { return plan_; }
    uint64_t n_blocks() const { return master_blocks_; }
    uint64_t n_streams() const { return master_streams_; }
    float host_plan_ms() const { return plan_ms_; }
    uint64_t lz_residue() const { return lz_residue_; }
    uint64_t source_bytes() const { return src_resident_; }       // compressed bytes uploaded for the resident tile

private:
    struct Tile {
        uint32_t b0, b1;
    };
    Failure load_tile(uint32_t t, hipStream_t stream);
    // Output held a tile at a time: the compressed bytes of tile t travel on a background thread into one of two buffers while
    // the tile in front is being read back (the two directions of the link are independent).
    struct SrcSlot {
        DevBuf buf;
        uint32_t tile = 0xFFFFFFFFu;     // whose bytes it holds (or will, once `pending` is done)
        std::future<bool> pending;
    };
    bool start_source_upload(uint32_t t);
    SrcSlot src_slot_[2];
    uint64_t src_cap_ = 0;               // the most compressed bytes any tile reads
    hipStream_t prefetch_stream_ = nullptr;
    const uint8_t *walked_payload_ = nullptr;   // walk() ran for this payload: master_, walk_err_, walk_truncated_ are its result
    size_t walked_n_ = 0;
    std::string walk_err_;
    bool walk_truncated_ = false;
    uint8_t *tile_out_base() const;                               // address of the loaded selection's local position 0
    void run_front(hipStream_t stream, StageTimer *timer, hipStream_t aux, bool *early);   // status reset, K1's literal-buffer classes beside K2
    void run_back(hipStream_t stream, StageTimer *timer, hipStream_t aux, bool early, uint32_t phase);   // scan, copies, K1, K4, checksums
    void fill_lz_args(LzArgs *la, uint32_t phase);
    void range_of(uint32_t rank, uint32_t *b0, uint32_t *b1) const;
    void carry_for(uint32_t rank, const ShardSummary *all, uint32_t out[3]) const;
    ZPlan master_, plan_;                                         // the walk; the loaded tile's launchable plan
    std::vector<Tile> tiles_;
    SectionOptions opt_;
    const uint8_t *host_payload_ = nullptr;
    uint32_t loaded_tile_ = 0xFFFFFFFFu, tiles_done_ = 0;
    uint64_t tile_pos0_ = 0, tile_len_ = 0, halo_elems_ = 0, halo_cap_ = 0, tile_cap_ = 0;
    uint32_t rep_carry_[3] = {1, 4, 8};
    uint32_t carry_frame_ = 0xFFFFFFFFu;                          // master frame (its first block) of the last block with sequences decoded so far
    bool carry_same_frame_ = false;                               // ... and the loaded tile's first block with sequences belongs to it
    bool has_lz_ = false;                                         // the section has blocks with LZ sequences
    uint64_t sec_known_ = 0, sec_seqs_ = 0;                       // the whole section: decoded bytes that are not match bytes; sequences
    uint64_t lz_residue_ = 0;
    uint64_t out0_ = 0, out1_ = 0;
    // shard protocol
    bool proto_lz_ = false;                                       // this range waits for (or feeds) a window: section with LZ sequences, several ranks
    uint64_t out_shift_ = 0;                                      // bytes in front of the range's output in d_out_ (room for the window)
    uint64_t send_elems_ = 0, decoded_ = 0;                       // elements the next rank's window takes from here; elements of this range
    uint32_t shard_carry_[3] = {1, 4, 8};                         // the repeat offsets the range inherits
    uint32_t halo_word_ = 0;                                      // (staging of the pseudo block's size)
    bool halo_pending_ = false, lz_args_valid_ = false, begin_early_ = false;
    std::vector<std::pair<uint32_t, uint32_t>> ranges_;           // every rank's block range
    std::vector<uint32_t> frame_first_;                           // first block of every frame of the section
    std::vector<uint32_t> seq_blk_, seq_frame_;                   // block / frame (its first block) of every block with sequences
    LzArgs la_{};
    bool la_ascii_ = false;
    uint64_t expect_ = 0, n_blocks_ = 0, n_streams_ = 0, n_tasks_ = 0, n_copies_ = 0, n_seq_blocks_ = 0;
    uint64_t master_blocks_ = 0, master_streams_ = 0, src_resident_ = 0;
    std::vector<HufClass> classes_;
    uint32_t t_char_ = 0, cells_cap_ = 0;
    bool sharded_ = false;
    hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr;   // K1 on two streams (created on first use, destroyed with the job)
    hipEvent_t ev_early_fork_ = nullptr, ev_early_join_ = nullptr;   // literal-buffer classes of K1 beside K2
    hipEvent_t ev_k2_ = nullptr;                                      // run_k2_ahead -> run
    bool k2_ahead_ = false;
    const uint8_t *d_src_ = nullptr;                     // device address of payload offset 0 (only [src_lo, src_hi) of the tile is behind it)
    float plan_ms_ = 0;
    bool ready_ = false;
    DevBuf d_src_buf_, d_out_, d_lit_, d_seqs_, d_blk_size_, d_blk_base_, d_scan_tmp_, d_status_;
    DevBuf d_pack_;                                // small tiles: the source bytes and every task list in ONE buffer, one copy (load_tile)
    std::vector<uint8_t> pack_host_;
    DevBuf d_meta_, d_rep_final_, d_rep_init_, d_rep_scratch_, d_lz_index_, d_blk_pending_, d_roff_, d_counters_;
    DevBuf d_lz_list_[2];
    DevBuf d_halo_tmp_;
    std::vector<XxhSeg> xxh_segs_;                // frame checksums: the (pieces of) checksummed frames in the loaded tile
    bool xxh_live_ = false;                       // ... and whether the frame the next tile continues was begun in this process's range
    DevBuf d_xxh_segs_, d_xxh_carry_;
    DevBuf d_pj_list_[2];                          // dense LZ sections: lists of pending elements for the late sweeps (optional)
    DevBuf d_pj_dist_, d_pj_tiles_;               // dense LZ sections: one word per output element + one per tile (allocated on first use, kept)
    bool lz_dense_ = false;
    DevBuf d_streams_, d_tasks_, d_tbl_copies_, d_pool_, d_dicts_, d_copies_, d_seq_blocks_, d_cells_;
    DevBuf d_huf_sync_;                  // one HufSync per part, when the plan's streams come in parts (plan.h)
};

struct ArchiveOptions {
    bool want[kNumSections] = {true, true, true, true, true, true};
    bool spec_mask = false;
    uint32_t shard_rank = 0, shard_count = 1;   // block-range sharding of the sequence section
    bool shard_protocol = false;                // ... of the quality section too, and of sections with LZ sequences (shard_begin / ...)
    uint64_t tile_blocks = 0;                   // > 0: sequence / quality sections are decoded in tiles of at most this many zstd blocks
    bool tiled_output = false;                  // ... whose output is held one tile at a time (iterator path; not for decode_all_device)
};

bool upload_staged(uint8_t *d_dst, const uint8_t *src, size_t n, hipStream_t stream, size_t stage_min = 0);
void trim_device_memory(int device);             // engine.cpp: the idle mapped ranges and small buffers of `device` go back to the driver   // engine.cpp: large host -> device copies

// A whole archive on one GPU: sections -> record table -> ASCII bases.
class ArchiveJob {
public:
    ~ArchiveJob();
    Failure init(int device);
    // the host walks of the wanted sections, ahead of upload() and without a HIP call (beside init() on another thread)
    void prewalk(const uint8_t *bytes, size_t n, const SectionInfo sec[kNumSections], const bool want[kNumSections]);
    // waits for background uploads (a tile's source bytes travelling ahead): before `bytes` goes away
    void drain();
    // several small device arrays to the host with ONE wait for the stream (through a pinned block when they fit one)
    struct SmallCopy {
        void *dst;
        const void *d_src;
        size_t n;
    };
    Failure copy_small_to_host(const SmallCopy *copies, int count);
    // bytes must stay valid until upload() returns -- with tiles (ArchiveOptions.tile_blocks) until the last tile is decoded
    Failure upload(const uint8_t *bytes, size_t n, const nafgpu_header &h, const SectionInfo sec[kNumSections],
                   const ArchiveOptions &opt);
    // (re)runs every kernel; synchronises; fills times
    Failure decode();
    // The shard protocol (include/nafgpu.h: nafgpu_shard_*): the same decode in steps, between which the ranks exchange
    // what their block ranges need from each other.  Sections: 0 = Sequence, 1 = Quality.
    Failure shard_begin(ShardSummary mine[2]);
    Failure shard_place(const ShardSummary *seq_all, const ShardSummary *qual_all, uint32_t n_ranks);
    Failure shard_halo(int which, uint64_t *recv_bytes, uint64_t *send_bytes, bool *tail_ready);
    Failure shard_export(int which, void *dst, uint64_t n);
    Failure shard_import(int which, const void *src, uint64_t n);
    Failure shard_finish();
    uint64_t quality_offset() const { return job_[kQuality].ready() ? job_[kQuality].shard_out0() : 0; }
    // output held a tile at a time (iterator path): decodes the next tile of section s into the tile buffer
    Failure advance_tile(int s);
    SectionJob &job_mut(int s) { return job_[s]; }

    // device results (valid after decode())
    const uint8_t *d_sequence() const;       // ASCII (nucleotides) or text
    uint64_t n_sequence_bytes() const;       // nucleotides: 2 * packed bytes (incl. a possible pad nibble)
    uint64_t packed_bytes() const { return is_nuc_ && job_[kSequence].ready() ? job_[kSequence].size() : 0; }
    // first base / byte of the sequence section held by this shard (0 unless sharded)
    uint64_t sequence_offset() const;
    const uint8_t *d_section(int s) const { return job_[s].ready() ? job_[s].out() : nullptr; }
    uint64_t section_size(int s) const { return job_[s].ready() ? job_[s].size() : 0; }
    const uint64_t *d_rec_ends() const { return d_rec_ends_.as<uint64_t>(); }
    uint64_t n_records() const { return rec_totals_.count; }
    uint64_t sum_lengths() const { return rec_totals_.sum; }
    uint64_t mask_sum() const { return mask_totals_.sum; }
    const SectionJob &job(int s) const { return job_[s]; }
    Failure section_failure(int s) const { return fail_[s]; }
    const StageTimes &times() const { return times_; }
    float host_plan_ms() const { return plan_ms_; }
    float h2d_ms() const { return h2d_ms_; }
    uint64_t compressed_bytes() const { return compressed_; }
    hipStream_t stream() const { return stream_; }
    int device() const { return device_; }
    // ids / comments split on NUL on the device (CStringReader): offsets just past each NUL; at most number_of_sequences strings
    const uint64_t *d_id_ends() const { return job_[kIds].ready() ? d_id_ends_.as<uint64_t>() : nullptr; }
    const uint64_t *d_com_ends() const { return job_[kComments].ready() ? d_com_ends_.as<uint64_t>() : nullptr; }
    uint64_t n_ids() const { return std::min<uint64_t>(id_totals_.count, id_cap_); }
    uint64_t n_comments() const { return std::min<uint64_t>(com_totals_.count, com_cap_); }
    // bit s set: section s (kIds, kComments, kSequence as text, kQuality) is not valid UTF-8 (Error::Utf8 in the reference)
    uint32_t utf8_invalid() const { return utf8_invalid_; }
    // FASTA (or FASTQ when `with_quality`) text of the first n_rec records, built on the device after decode()
    Failure format_text(bool with_ids, bool with_comments, bool with_quality, uint64_t n_rec, const uint8_t **d_text, uint64_t *n_text,
                        float *ms);
    Failure copy_to_host(void *dst, const void *d_src, size_t n);
    Failure copy_to_pinned(void *dst_pinned, const void *d_src, size_t n);   // dst from hipHostMalloc: the GPU writes it (k_copy_out)
    // the same copy enqueued on the second stream, not waited for (false: there is no such stream); ..._end() waits for all of them
    bool copy_to_pinned_begin(void *dst_pinned, const void *d_src, size_t n);
    Failure copy_to_pinned_end();
    Failure hash_device(const void *d_ptr, uint64_t n, uint64_t first_chunk, uint64_t *out);

private:
    int device_ = -1;
    hipStream_t stream_ = nullptr, aux_stream_ = nullptr, k2_stream_ = nullptr;
    hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr;   // record / mask table scans beside the sequence decode
    StageTimer timer_;
    StageTimes times_;
    nafgpu_header h_{};
    ArchiveOptions opt_;
    bool is_nuc_ = false;
    void apply_mask_to_held();
    bool want_mask_ = false;
    void decode_front();                        // status reset, the small sections, the record / mask table scans beside what follows
    Failure decode_back();                      // scans joined, mask, names, UTF-8; synchronises; totals and per-section checks
    bool scans_forked_ = false;
    void run_table_scans(hipStream_t st);
    SectionJob job_[kNumSections];
    Failure fail_[kNumSections];
    DevBuf d_rec_ends_, d_mask_ends_, d_scan_tmp_, d_totals_, d_status_, d_hash_;
    DevBuf d_id_ends_, d_com_ends_, d_fmt_sizes_, d_fmt_off_, d_text_;
    uint64_t id_cap_ = 0, com_cap_ = 0;
    ScanTotals id_totals_{0, 0}, com_totals_{0, 0};
    uint32_t utf8_invalid_ = 0;
    uint64_t rec_cap_ = 0, mask_cap_ = 0, mask_total_bases_ = 0;
    ScanTotals rec_totals_{0, 0}, mask_totals_{0, 0};
    float plan_ms_ = 0, h2d_ms_ = 0;
    uint64_t compressed_ = 0;
};

}  // namespace nafgpu
