// kernels.h -- host-callable launchers of the gfx950 kernels in kernels.hip.
// All pointers are device pointers; every launch is asynchronous on `stream`.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "plan.h"

namespace nafgpu {

// Test / measurement switches (NAFGPU_LZ_MODE, NAFGPU_PJ_MAX_DIST, NAFGPU_TILE_KIB, NAFGPU_K2_LANES, NAFGPU_DEBUG_PLAN) are
// read from the environment only after nafgpu_test_hooks(1): a variable that leaks into the environment of another
// caller of the library changes nothing.  Returns the variable's value, or null.
const char *hook_env(const char *name);
void set_test_hooks(bool on);

struct ScanTotals {          // written by the scan kernels
    uint64_t sum;            // sum of all values
    uint64_t count;          // number of run terminators (runs modes) or elements (exclusive mode)
};

// K2: FSE sequence decode in two kernels -- k_seq_states (one lane per block with sequences: the FSE states and the bit
// cursor of every sequence) and k_seq_values (one wave per block: values, positions, repeat offsets).  Adds match bytes to
// blk_size[].  `meta` is scratch here (the 8-byte state records): nothing else uses it before k_lz_literals.
// rep_final: 3 u32 per block -- the repeat offsets the block ends with (concrete or kRepToken tokens)
// cells_cap: most cells (LL + OF + ML tables together) any of the blocks has; src_min: lowest payload offset that may be read
// (both for the variant of k_seq_states that keeps tables and bitstream in LDS; cells_cap = 0: never that one)
void launch_seq_decode(hipStream_t stream, const uint8_t *src, const SeqBlock *blocks, uint32_t n_blocks,
                       const SeqCell *cells, Seq *seqs, SeqMeta *meta, uint32_t *blk_size, uint32_t *rep_final, uint32_t *status,
                       uint32_t cells_cap = 0, long long src_min = 0);

// K3: exclusive scan of u32 block sizes -> u64 bases (n+1 entries; the last is the total).
// `tile_tmp` needs scan_tmp_bytes(n) bytes.  Flags status if the total differs from `expect_total`.
size_t scan_tmp_bytes(uint64_t n);
void launch_scan_blocks(hipStream_t stream, const uint32_t *blk_size, uint64_t n, uint64_t *blk_base, void *tile_tmp,
                        uint64_t expect_total, uint32_t *status);

// K6 / mask runs: values with a continuation sentinel (0xFFFFFFFF words / 0xFF bytes).
// ends[k] = position (inclusive prefix sum) where run k ends; totals->count = number of runs.
void launch_scan_runs_u32(hipStream_t stream, const uint8_t *words, uint64_t n_words, uint64_t *ends, uint64_t cap,
                          void *tile_tmp, ScanTotals *totals, uint32_t *status);
void launch_scan_runs_u8(hipStream_t stream, const uint8_t *bytes, uint64_t n_bytes, uint64_t *ends, uint64_t cap,
                         void *tile_tmp, ScanTotals *totals, uint32_t *status);

// ids / comments: ends[k] = offset just past the k-th NUL (CStringReader, reader.rs:22-30); strings beyond `cap` are dropped
void launch_scan_nul(hipStream_t stream, const uint8_t *bytes, uint64_t n_bytes, uint64_t *ends, uint64_t cap, void *tile_tmp,
                     ScanTotals *totals, uint32_t *status);
// exclusive prefix sums of n u64 items (out: n entries; totals->sum = grand total)
void launch_scan_excl_u64(hipStream_t stream, const uint64_t *items, uint64_t n, uint64_t *out, void *tile_tmp, ScanTotals *totals,
                          uint32_t *status);
// sets bit `bit` of *flags when p[0..n) is not valid UTF-8 (mod.rs:362,368; reader.rs:108-109)
void launch_utf8_check(hipStream_t stream, const uint8_t *p, uint64_t n, uint32_t *flags, uint32_t bit);

// FASTA / FASTQ text from the decoded buffers
struct FmtText {
    const uint8_t *seq;          // ASCII bases / text, mask applied
    const uint8_t *qual;         // non-null: FASTQ
    const uint64_t *rec_end;
    const uint8_t *ids;
    const uint64_t *id_end;
    uint64_t n_ids;
    const uint8_t *com;
    const uint64_t *com_end;
    uint64_t n_com;
    uint64_t n_rec;
    uint64_t line_length;
    uint32_t sep;
};
void launch_fmt_sizes(hipStream_t stream, const FmtText &t, uint64_t *sizes);
void launch_fmt_write(hipStream_t stream, const FmtText &t, const uint64_t *off, uint64_t n_text, uint8_t *text);

// raw / RLE blocks and literal sections
// ascii = true (nucleotide sequence sections): bytes bound for `out` are expanded to two IUPAC
// characters each on the way (t_char = 'T' DNA / 'U' RNA); bytes bound for `lit` stay packed.
void launch_copy_fill(hipStream_t stream, const uint8_t *src, const CopyTask *tasks, uint32_t n_tasks,
                      const uint64_t *blk_base, uint8_t *out, uint8_t *lit, bool ascii, uint32_t t_char,
                      uint32_t *status);

// K1: Huffman literal streams, one lane per stream, one wave per task.  One launch per class (plan.h:
// HufClass): `tasks` is the section's whole task list, cls names the run to launch, its table format,
// destination and whether the segment-aware variant is needed (streams of blocks with a few sequences:
// seq_blocks / seqs are then read, so k_seq_values must have run).
void launch_huf_parts(hipStream_t stream, const uint8_t *src, const HufTask *tasks, uint32_t n_tasks, const HufTblCopy *copies,
                      HufStream *streams, const uint16_t *pool, HufSync *sync, uint32_t sync_lds, uint32_t *status);
void launch_huf_decode(hipStream_t stream, const uint8_t *src, const HufTask *tasks, const HufClass &cls,
                       const HufTblCopy *copies, const HufStream *streams, const uint16_t *pool,
                       const uint64_t *blk_base, uint8_t *out, uint8_t *lit, const SeqBlock *seq_blocks, const Seq *seqs,
                       const uint8_t *dicts, bool ascii, uint32_t t_char, uint32_t *status);

// K4: LZ77 sequence execution: repeat-offset chain, parallel literal scatter, then -- all enqueued at once, no host
// round trip: every stage returns at once when the one before it left nothing --
//   dense sections (pj_dist != null: matches are a good part of the output): every element learns the distance to the
//     element it copies, then the frame is swept until every element is final (pointer jumping, k_pj_sweep);
//   sparse sections: match passes over the list of what is still pending, one workgroup for the last few;
//   a frame-order walk for whatever neither resolves (chains the distances cannot express / thousands of links deep).
struct LzArgs {
    const SeqBlock *blocks;
    uint32_t n_blocks;
    uint64_t n_sequences;
    uint64_t n_elems;            // output elements of the section (bytes; packed bytes when the output is expanded to ASCII)
    const Seq *seqs;
    const uint8_t *lit;
    const uint64_t *blk_base;
    const uint32_t *rep_final;   // 3 per block, from k_seq_values
    uint32_t *rep_init;          // 3 per block
    uint32_t rep_carry[3];       // repeat offsets in front of the first block ({1, 4, 8} unless a tile in front left others)
    uint32_t rep_continues;      // 1: the first block continues a frame begun in front of this tile (no reset to {1, 4, 8})
    uint32_t *rep_out;           // 3 words: the repeat offsets behind the last block
    uint32_t *rep_scratch;       // 6 words per chunk of 64 blocks (k_rep_partial / k_rep_scan)
    uint32_t *cidx;              // sparse: sequence index per 128 output elements (null: plain binary search)
    uint64_t n_idx_chunks;       // entries of cidx
    SeqMeta *meta;               // per sequence: output position and length of its match, pass that completed it (0 = pending)
    uint32_t *blk_pending;       // per block: matches still pending
    uint32_t *roff;              // sparse: per sequence, resolved offset of a match that is still pending
    uint64_t *plist[2];          // sparse: two lists of pending matches (n_sequences entries each; null: block-wise passes only)
    uint32_t *pj_dist;           // dense: D, one word per output element (n_elems, 16-byte aligned)
    uint32_t *pj_tiles;          // dense: pending elements per tile of 2048 (lz_pj_tiles(n_elems) words)
    uint32_t *pj_list[2];        // dense, optional: two lists of pending element indices, pj_list_cap entries each (k_pj_list)
    uint64_t pj_list_cap;
    unsigned long long *counters;// 32 words: [0] matches still pending (sparse), [1] matches left to the one-workgroup stage, [4..6] stage counters, ... (kernels.hip: kCtr*)
    uint8_t *out;
    uint32_t t_char;
    uint32_t *status;
    // the shard protocol (engine.cpp): see lz_execute
    uint32_t phase;              // 0: everything; 1: the window in front has not arrived yet; 2: it has -- finish
    uint32_t n_sel_blocks;       // blocks of the loaded selection (blk_base[n_sel_blocks] = its elements)
    uint64_t halo_wait;          // phase 1 / 2: elements of the pseudo block in front
    uint64_t tail_elems;         // phase 1: the next shard waits for the last this-many elements
    uint32_t strips;             // dense: the sweeps go strip-wise (k_pj_sweep<., kPjWin>): shallow chains, matches from near by
    uint32_t shallow;            // dense: chains are shallow (a tenth of the elements or more are literals): the FIRST sweep already lists what it leaves pending
};
uint64_t lz_pj_tiles(uint64_t n_elems);
// The repeat-offset map of a whole run of blocks (shard protocol): what the three offsets behind the last block are in
// terms of the three in front of the first -- values, or kRepToken tokens -- from k_seq_values' per-block maps.
// rep_scratch as in LzArgs; map_out: 3 words on the device.
void launch_rep_map(hipStream_t stream, const SeqBlock *blocks, uint32_t n_blocks, const uint32_t *rep_final, uint32_t *rep_scratch,
                    uint32_t continues, uint32_t *map_out, uint32_t *status);
void launch_lz_execute(hipStream_t stream, const LzArgs &args, bool ascii);

// soft-mask: lower-case the masked runs (odd-numbered runs of mask_ends) honouring record ends.
// `ascii` is addressed by global base index; only bases [lo_clamp, hi_clamp) are touched (a shard).
void launch_mask_apply(hipStream_t stream, uint8_t *ascii, uint64_t n_bases, uint64_t lo_clamp, uint64_t hi_clamp,
                       const uint64_t *mask_ends, const ScanTotals *mask_totals, const uint64_t *rec_ends,
                       const ScanTotals *rec_totals,
                       uint64_t max_runs, int spec_mask, uint32_t *status);

// Content_Checksum of the frames (pieces of frames) in `segs`: XXH64 over the decoded bytes at out + blk_base[..]
// (ascii: the section output holds two characters per decoded byte, packed again on the fly); a mismatch flags
// kStChecksum.  carry_in / carry_out: running state of a frame that spans tiles.
void launch_xxh64_frames(hipStream_t stream, const XxhSeg *segs, uint32_t n_segs, const uint64_t *blk_base, const uint8_t *out,
                         bool ascii, uint32_t t_char, const XxhCarry *carry_in, XxhCarry *carry_out, uint32_t *status);

// order-sensitive checksum of a device buffer (see hash64.h); *result must be zeroed first.
// first_chunk: index of the buffer's first 4 KiB chunk in the whole object (shards add up)
void launch_hash64(hipStream_t stream, const uint8_t *p, uint64_t n, uint64_t first_chunk, unsigned long long *result);
void launch_copy_out(hipStream_t stream, uint8_t *dst_pinned, const uint8_t *d_src, uint64_t n);

}  // namespace nafgpu
