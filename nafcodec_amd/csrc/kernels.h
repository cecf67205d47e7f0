// kernels.h -- host-callable launchers of the gfx950 kernels in kernels.hip.
// All pointers are device pointers; every launch is asynchronous on `stream`.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "plan.h"

namespace nafgpu {

struct ScanTotals {          // written by the scan kernels
    uint64_t sum;            // sum of all values
    uint64_t count;          // number of run terminators (runs modes) or elements (exclusive mode)
};

// K2: FSE sequence decode, one lane per block with sequences.  Adds match bytes to blk_size[].
// rep_final: 3 u32 per block -- the repeat offsets the block ends with (concrete or kRepToken tokens)
void launch_seq_decode(hipStream_t stream, const uint8_t *src, const SeqBlock *blocks, uint32_t n_blocks,
                       const SeqCell *cells, Seq *seqs, uint32_t *blk_size, uint32_t *rep_final, uint32_t *status);

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
// seq_blocks / seqs are then read, so k_seq_decode must have run).
void launch_huf_decode(hipStream_t stream, const uint8_t *src, const HufTask *tasks, const HufClass &cls,
                       const HufTblCopy *copies, const HufStream *streams, const uint16_t *pool,
                       const uint64_t *blk_base, uint8_t *out, uint8_t *lit, const SeqBlock *seq_blocks, const Seq *seqs,
                       const uint8_t *dicts, bool ascii, uint32_t t_char, uint32_t *status);

// K4: LZ77 sequence execution: repeat-offset chain, parallel literal scatter, multi-pass match
// resolution over all blocks, ordered fallback
struct LzArgs {
    const SeqBlock *blocks;
    uint32_t n_blocks;
    uint64_t n_sequences;
    uint32_t mean_ml;            // mean match length of the section (lanes per match in the pointer-jumping steps)
    const Seq *seqs;
    const uint8_t *lit;
    const uint64_t *blk_base;
    const uint32_t *rep_final;   // 3 per block, from k_seq_decode
    uint32_t *rep_init;          // 3 per block
    uint32_t *rep_scratch;       // 6 words per chunk of 64 blocks (k_rep_partial / k_rep_scan)
    uint32_t *cidx;              // sequence index per 128 output elements (null: plain binary search)
    uint64_t n_idx_chunks;       // entries of cidx
    SeqMeta *meta;               // per sequence: output position and length of its match, pass that completed it (0 = pending)
    uint32_t *blk_pending;       // per block: matches still pending
    uint32_t *roff;              // per sequence: resolved offset of a match that is still pending
    uint64_t *plist[2];          // two lists of pending matches (n_sequences entries each; null: block-wise passes only)
    unsigned long long *counters;// [0] matches still pending after the passes, [1] pointer-jumping list length, [2] its flag, [4] [5] pending-list lengths
    uint8_t *out;
    uint32_t t_char;
    uint32_t *status;
};
// rep chain + literal scatter + the fixed number of match passes (asynchronous)
void launch_lz_execute(hipStream_t stream, const LzArgs &args, bool ascii);
// a short pending list (<= lz_few_pending() matches: a few long chains of long matches) is walked on, `n` more passes
// at a time: cur = counters[3] (list that holds what is pending), first_pass = number of the next pass
uint32_t lz_passes_done();
uint32_t lz_few_pending();
void launch_lz_more_passes(hipStream_t stream, const LzArgs &args, bool ascii, uint32_t cur, uint32_t first_pass, uint32_t n);
// what the passes left pending (args.counters[0] != 0):
//   pointer jumping: collect the pending list, then op 0 (init D), op 1 (jump, repeat while *changed), op 2 (copy)
void launch_pj_collect(hipStream_t stream, const LzArgs &args, uint64_t *list);
// stamp: one byte per list entry (zeroed before the first jump step); step = 1, 2, ... for the jump steps
void launch_pj_step(hipStream_t stream, const LzArgs &args, bool ascii, const uint64_t *list, uint64_t n_list, uint32_t *D,
                    int op, uint32_t *changed, uint8_t *stamp, uint32_t step);
//   or, without scratch memory for D: one workgroup in frame order
void launch_lz_ordered(hipStream_t stream, const LzArgs &args, bool ascii);

// K5: 4-bit -> IUPAC ASCII; t_char = 'T' (DNA) or 'U' (RNA)
void launch_unpack4(hipStream_t stream, const uint8_t *packed, uint64_t n_packed, uint8_t *ascii, uint64_t n_bases,
                    uint32_t t_char, uint32_t *status);

// soft-mask: lower-case the masked runs (odd-numbered runs of mask_ends) honouring record ends.
// `ascii` is addressed by global base index; only bases [lo_clamp, hi_clamp) are touched (a shard).
void launch_mask_apply(hipStream_t stream, uint8_t *ascii, uint64_t n_bases, uint64_t lo_clamp, uint64_t hi_clamp,
                       const uint64_t *mask_ends, const ScanTotals *mask_totals, const uint64_t *rec_ends,
                       const ScanTotals *rec_totals,
                       uint64_t max_runs, int spec_mask, uint32_t *status);

// order-sensitive checksum of a device buffer (see hash64.h); *result must be zeroed first.
// first_chunk: index of the buffer's first 4 KiB chunk in the whole object (shards add up)
void launch_hash64(hipStream_t stream, const uint8_t *p, uint64_t n, uint64_t first_chunk, unsigned long long *result);

}  // namespace nafgpu
