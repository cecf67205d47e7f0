// zplan.h -- host front end of the GPU Zstandard decoder.
//
// Replaces the control side of zstd::stream::read::Decoder (constructed at
// nafcodec/src/decoder/mod.rs:221-223) for one NAF section: walks the magicless frame(s),
// parses every block / literals / sequences header, resolves `treeless` and `repeat` modes
// to concrete tables, builds the Huffman and FSE decode tables ONCE per block, and emits the
// flat task lists (plan.h) that the HIP kernels consume.  No payload byte is decoded here.
// Format reference: RFC 8878 as digested in SURVEY.md Appendix B.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "plan.h"

namespace nafgpu {

struct HufTableRef {                     // one Huffman tree in the table pool
    uint32_t pool_off = 0;
    uint16_t n_syms = 0;                 // symbols with a code
    uint8_t max_bits = 0;
    bool valid = false;
};

struct ZPlan {
    // per zstd block
    std::vector<uint32_t> blk_size;      // decoded size; blocks with sequences: literal bytes only
                                         // (k_seq_values adds the match bytes on the device)
    // Huffman literals
    std::vector<HufStream> streams;
    std::vector<HufTask> tasks;          // grouped by launch class
    std::vector<HufClass> classes;       // tasks that are launched together (table format x destination x segment-aware)
    std::vector<HufTblCopy> tbl_copies;
    std::vector<uint8_t> dict_pool;      // kTblDict tasks: kHufDictSyms symbol values per task
    std::vector<uint16_t> huf_pool;      // decode tables, (len << 8 | symbol)
    // raw / RLE
    std::vector<CopyTask> copies;
    // sequences
    std::vector<SeqBlock> seq_blocks;
    std::vector<SeqCell> fse_pool;
    uint64_t n_sequences = 0;
    uint64_t lit_bytes = 0;              // literal buffer size (blocks with many sequences only)
    uint64_t known_out = 0;              // sum of blk_size
    uint32_t n_frames = 0;
    uint32_t n_huf_tables = 0;
    uint64_t window_max = 0;
    bool has_checksum = false;
    // ---- what the walk keeps besides (master plan only)
    std::vector<HufTableRef> stream_ref; // tree of each stream (parallel to `streams` until the tasks are packed)
    std::vector<uint64_t> blk_off;       // payload offset of every block header; one more entry: the end of the last block
    struct Frame {                       // one Zstandard frame of the section
        uint32_t first_blk, end_blk;     // its blocks
        uint32_t checksum;               // low 32 bits of XXH64 of the decoded frame, when has_checksum
        bool has_checksum;
    };
    std::vector<Frame> frames;
    // ---- a selection (select_zplan): blocks [sel_blk0, sel_blk1) of the master, re-based to start at block `halo`
    bool sharded = false;                // the selection is one of several shards (multi-GPU)
    uint32_t sel_blk0 = 0, sel_blk1 = 0;
    uint32_t halo = 0;                   // 1: block 0 is a pseudo block standing for the decoded bytes in front of the selection
    uint64_t src_lo = 0, src_hi = 0;     // payload bytes the selection's tasks read
    uint64_t shard_out0 = 0, shard_out1 = 0;   // decoded-byte range of the selection when no block has sequences (else unknown: 0)
    bool first_frame_continues = false;  // the selection's first block with sequences continues a frame begun in front of it
    uint32_t first_seq_frame = 0xFFFFFFFFu, last_seq_frame = 0xFFFFFFFFu;   // master frame (its first block) of the first / last block with sequences
};

// The walk: every block / literals / sequences header of the section's frame(s), tables built, nothing packed.
// Returns "" on success, else a description of the first malformed field.
// `truncated` is set when the payload ends early (maps to Io(UnexpectedEof)).
std::string walk_zstd(const uint8_t *payload, size_t n, ZPlan *master, bool *truncated);
// The shard_rank-th of shard_count contiguous block ranges balanced by decoded bytes (block boundaries only).
// A master without LZ sequences: every block is independent once treeless chains are resolved, which the walk has
// done.  With sequences (with_lz; else false is returned for such a master) the decoded size of a block is not
// known on the host -- a block with sequences counts as a full one -- and the shards then depend on each other
// through the LZ window and the repeat offsets (engine.cpp: the shard protocol).
bool shard_range(const ZPlan &master, uint32_t shard_rank, uint32_t shard_count, uint32_t *b0, uint32_t *b1, bool with_lz = false);
// The launchable plan of blocks [b0, b1): task lists re-based to the range (block indices, literal-buffer and
// sequence offsets, table pools holding only what the range uses), tasks packed into launch classes.
// halo_elems > 0: block 0 of the selection is a pseudo block of that size standing for the decoded bytes in
// front of the range (the LZ window of a tile); frames begun in front of the range then start at it.
// force_halo: the pseudo block exists whatever halo_elems says (a shard learns its size later).
void select_zplan(const ZPlan &master, uint32_t b0, uint32_t b1, uint64_t halo_elems, ZPlan *out, bool force_halo = false);
// packs the tasks of a plan that holds a whole walk (no selection needed)
void pack_tasks_public(ZPlan *plan);
// sub-streams (zplan.cpp: g_split_target): sections below `target_lanes` / 2 streams are cut; force: that many parts, always (tests)
void set_huf_split(uint32_t target_lanes, uint32_t force);
void set_dict_slots(uint32_t slots);   // LDS budget of a dictionary-format task in 2-byte slots (experiments)
void set_task_lanes(uint32_t lanes);   // streams per K1 task (experiments; 64 otherwise)
// walk + (shard) + select: the whole section, or one shard of it, in one call
std::string build_zplan(const uint8_t *payload, size_t n, ZPlan *plan, bool *truncated, uint32_t shard_rank = 0,
                        uint32_t shard_count = 1);

}  // namespace nafgpu
