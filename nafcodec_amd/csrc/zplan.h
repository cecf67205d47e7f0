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

struct ZPlan {
    // per zstd block
    std::vector<uint32_t> blk_size;      // decoded size; blocks with sequences: literal bytes only
                                         // (k_seq_decode adds the match bytes on the device)
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
    uint64_t lit_bytes = 0;              // literal buffer size (blocks with sequences only)
    uint64_t known_out = 0;              // sum of blk_size
    uint32_t n_frames = 0;
    uint32_t n_huf_tables = 0;
    uint64_t window_max = 0;
    bool has_checksum = false;
    // multi-GPU sharding: this plan covers zstd blocks [shard_blk0, shard_blk1) only
    bool sharded = false;
    uint32_t shard_blk0 = 0, shard_blk1 = 0;
    uint64_t shard_out0 = 0, shard_out1 = 0;   // decoded-byte range of those blocks
};

// Returns "" on success, else a description of the first malformed field.
// `truncated` is set when the payload ends early (maps to Io(UnexpectedEof)).
// shard_count > 1 restricts the task lists to the shard_rank-th of shard_count contiguous block
// ranges balanced by decoded bytes.  That is only possible when no block has LZ sequences (every
// block is then independent once treeless chains are resolved, which the walk has done); otherwise
// the plan stays complete and `sharded` stays false (every rank decodes the whole section).
std::string build_zplan(const uint8_t *payload, size_t n, ZPlan *plan, bool *truncated, uint32_t shard_rank = 0,
                        uint32_t shard_count = 1);

}  // namespace nafgpu
