// plan.h -- plain-data descriptors shared by the host front end (zplan.cpp) and the HIP
// kernels (kernels.hip).  The host walks a section's Zstandard frame(s) ONCE, builds the
// entropy tables per block, and hands the GPU flat task lists; nothing here owns memory.
#pragma once
#include <cstdint>

namespace nafgpu {

constexpr uint32_t kBlockMax = 128u << 10;   // zstd Block_Maximum_Size
constexpr int kHufWave = 64;                 // one Huffman stream per lane, one wave per workgroup
constexpr uint32_t kHufLdsEntries = 2048;    // 8-byte decode-table entries a single-tree wave task may stage in LDS
constexpr uint32_t kHufLdsEntries4 = 6144;   // 4-byte entries a task with several trees may stage (compact tables, see HufTask)
constexpr uint32_t kHufLdsSlots2 = 5632;     // 2-byte slots a task of small-alphabet trees may stage (dictionary tables, see HufTask)
constexpr uint32_t kHufDictSyms = 64;        // most symbols the trees of a task may use between them to take the dictionary format
constexpr uint32_t kHufDictSlots = 32;       // 2-byte slots in front of a dictionary table: the tree's code length of each of the 64 dictionary symbols
constexpr uint32_t kDirectSeqMax = 64;       // blocks with at most this many LZ sequences have their Huffman literals written straight
                                             // to their final positions by k_huf_decode (segment by segment); more: literal buffer + K4
constexpr uint32_t kHufTaskSpan = 1u << 30;   // a wave task's streams lie within this many bytes of input and of output
constexpr uint32_t kSrcFrontPad = 512;       // bytes readable in front of any device source buffer (k_huf_decode: ring look-ahead + whole lines)
constexpr uint32_t kSrcBackPad = 256;        // ... and behind it (k_huf_decode reads whole 128-byte lines)

// One Huffman-coded literal stream = one lane of k_huf_decode.
struct alignas(16) HufStream {
    uint64_t src_end;    // offset one past the stream's last byte, from the section payload base
    uint64_t dst;        // flags&1: absolute offset in the literal buffer; else bits 0-31: index of the stream's first literal in the
                         // block's literal section (= offset inside block `blk` when the block has no sequences), bits 32-63:
                         // 1 + index of the block's SeqBlock when its literals are interleaved with matches (flags&2), else 0
    uint32_t src_len;    // bytes in the stream (>= 1)
    uint32_t n_syms;     // symbols to regenerate
    uint32_t blk;        // zstd block index (for blk_base[])
    uint16_t tbl_lds;    // first entry of this stream's table inside the task's LDS table area
    uint8_t max_bits;    // index width W of the staged table (6..8, chosen per task by the host)
    uint8_t flags;       // bit0: write to the literal buffer (block has many sequences); bit1: block has a few sequences, the
                         // literals go to their final positions segment by segment; bits 4-7: tree max_bits - W (0 = no escapes)
    // ---- sub-streams (sections with fewer streams than the chip has lanes: pack_tasks).  A stream is then `S` records
    // in a row -- the fields above the same in all of them -- and lane k decodes the k-th part of it: the device finds
    // where the parts begin (k_huf_sync, k_huf_bounds) and writes the three words below; the host only sets `sub`.
    uint32_t sub;        // 0: a whole stream; else k | S << 8: part k of S
    uint32_t sub_start;  // bits of the stream below this part's first symbol (part 0: all of them; the next part's: where this one ends)
    uint32_t sub_syms;   // symbols in this part
    uint32_t sub_first;  // symbols of the stream in front of it
};
static_assert(sizeof(HufStream) == 48, "HufStream layout");

// What k_huf_sync learns about one part decoded from its GUESSED first bit (an even split of the stream's bits), which
// is a symbol boundary only by luck: prefix codes synchronise -- two decodes of the same bits from different starts
// soon stand on the same boundaries -- so the guess is wrong for the part's first few symbols only.  At `kHufSyncMarks`
// marks along the part it notes the first boundary at or below the mark and the symbols before it; k_huf_bounds
// decodes from the TRUE first bit until it finds itself on a noted boundary, and takes the rest from here.
constexpr uint32_t kHufSyncMarks = 32;
constexpr uint32_t kHufSyncBad = 0xFFFFFFFFu;
struct HufSync {
    uint32_t end_pos;    // the first boundary at or below the next part's guessed first bit (kHufSyncBad: the decode ran into an invalid code)
    uint32_t total;      // symbols in front of it
    uint16_t cnt[kHufSyncMarks];   // symbols in front of the boundary noted at mark j
    uint8_t off[kHufSyncMarks];    // mark j minus that boundary (< 12)
};
static_assert(sizeof(HufSync) == 104, "HufSync layout");
constexpr uint32_t kHufSplitMax = 16;        // most parts per stream (a power of two: 64 lanes hold whole streams)

struct HufTblCopy {      // build the two-symbol table of pool[pool_off ..) at LDS entry lds_off
    uint32_t pool_off;   // 2^max_bits single-symbol entries (len << 8 | sym) in the pool
    uint32_t lds_off;    // first staged entry (8-byte entries)
    uint32_t n_entries;  // 2^W main entries + one 2^(max_bits - W) sub-table per escaping prefix
                         // (kTblDict: lds_off and n_entries count 2-byte slots and include the kHufDictSlots in front)
    uint32_t bits;       // max_bits | W << 8
};

// One workgroup of k_huf_decode: <= 64 streams and the decode tables of their trees.  Tasks come in three
// table formats, launched separately:
//   kTblBaked   8-byte entries with the output characters baked in (one tree per task: the fast path)
//   kTblCompact 4-byte entries {sym1, sym2, bits, bits of sym1, two} whose characters come from a 512-byte
//               look-up table shared by the wave (several trees, some with more than kHufDictSyms symbols)
//   kTblDict    2-byte entries {symbols, bits, index of sym1, index of sym2} into ONE dictionary of
//               <= kHufDictSyms symbols shared by the trees of the task (several small-alphabet trees -- what
//               real genomes give: what counts there is LDS per lane, because the lanes resident per CU set
//               the throughput; and a dictionary shared by the wave is read conflict-free -- the few hot
//               symbols are broadcast)
enum HufTblKind : uint32_t { kTblBaked = 0, kTblCompact = 1, kTblDict = 2 };
struct HufTask {
    uint32_t first_stream, n_streams, first_copy, n_copies;
    uint32_t dict_off;       // kTblDict: first of the task's kHufDictSyms dictionary symbols in the dictionary pool (short codes first)
    uint32_t n_dict;         // symbols in it
    uint32_t pad[2];
};
// A run of tasks launched together: same table format, same destination, same kernel variant.
struct HufClass {
    uint32_t first_task, n_tasks;
    uint32_t tbl;            // HufTblKind
    uint32_t to_lit;         // streams feed the literal buffer (never expanded to ASCII)
    uint32_t seg;            // some stream is interleaved with matches (flags&2): the segment-aware kernel variant
    uint32_t lds_bytes;      // dynamic LDS: the largest staged-table footprint over the tasks
    uint32_t split;          // > 1: the class's streams come in that many parts each (HufStream::sub): k_huf_sync and k_huf_bounds run first
    uint32_t sync_lds;       // ... and their dynamic LDS: one length byte per entry of the task's trees
};

struct CopyTask {        // Raw/RLE blocks and Raw/RLE literal sections
    uint64_t src_off;    // payload offset of the first source byte (fill: the byte itself)
    uint64_t dst;        // as HufStream::dst
    uint32_t len;
    uint32_t blk;
    uint32_t flags;      // bit0: to literal buffer, bit1: fill with one byte (RLE)
    uint32_t pad;
};

struct SeqCell {         // one FSE state of an LL / OF / ML table with its code->value baked in:
    uint16_t next_base;  //   next state = next_base + read(nb)           (App. B "FSE table build")
    uint8_t nb;
    uint8_t extra_bits;  //   value = base_value + read(extra_bits)       (App. B "Code->value")
    uint32_t base_value;
};
static_assert(sizeof(SeqCell) == 8, "SeqCell layout");

struct SeqBlock {        // one compressed block with nbSeq > 0
    uint64_t bits_off;   // payload offset of the sequence bitstream
    uint64_t lit_off;    // absolute offset of this block's literals in the literal buffer
    uint64_t seq_first;  // index of its first sequence in the sequence buffer
    uint32_t bits_len;
    uint32_t n_seq;
    uint32_t ll_tbl, of_tbl, ml_tbl;   // SeqCell pool offsets
    uint32_t blk;
    uint32_t lit_size;
    uint32_t frame_first_blk;          // first block of the frame this block belongs to
    uint8_t ll_al, of_al, ml_al;
    uint8_t direct;                    // 1: k_huf_decode already put the literals at their final positions (<= kDirectSeqMax sequences)
    uint32_t pad2;
};
static_assert(sizeof(SeqBlock) == 64, "SeqBlock layout");

struct Seq {             // one decoded sequence
    uint32_t ll, ml;
    uint32_t off;        // match offset; bit 31 set: symbolic (see kRepToken) -- depends on the
                         // repeat-offset history the block inherits from its predecessor
    uint32_t opos;       // output position of its literals, relative to the block's first byte
    uint32_t lpos;       // position of its literals in the block's literal section
};

struct alignas(16) SeqMeta {   // what the match passes need to know about a sequence as a PRODUCER of bytes, in one 16-byte load
    uint64_t pos;        // output position of its match (element index in the section)
    uint32_t ml;
    uint32_t flag;       // 0 = pending, else the pass that completed the match
};
static_assert(sizeof(SeqMeta) == 16, "SeqMeta layout");

// Repeat offsets (App. B): a block starts from the three offsets its predecessor ends with, which
// k_seq_values does not know (blocks decode in parallel).  It tracks them symbolically instead:
//   token = kRepToken | slot << 24 | d   means   (initial rep[slot]) - d
// and k_rep_chain later walks the blocks in frame order to turn every block's final triple into
// the next block's initial one.
constexpr uint32_t kRepToken = 0x80000000u;
// `tok` (an offset, or "inherited rep[slot] - d") applied after the map / triple `f`: maps compose.  The host side of
// the kernels' rep_apply_entry (shard protocol: the ranks' maps are put together on the host).
inline uint32_t rep_compose(uint32_t tok, const uint32_t *f, bool *bad) {
    if (!(tok & kRepToken)) return tok;
    const uint32_t slot = (tok >> 24) & 3u, d = tok & 0xFFFFFFu;
    if (slot > 2u) {                                        // (three repeat offsets: a token never names a fourth)
        *bad = true;
        return 1;
    }
    const uint32_t fv = f[slot];
    if (!(fv & kRepToken)) {
        if (fv <= d) {
            *bad = true;
            return 1;
        }
        return fv - d;
    }
    const uint32_t d2 = (fv & 0xFFFFFFu) + d;
    if (d2 > 0xFFFFFFu) *bad = true;
    return (fv & 0xFF000000u) | (d2 & 0xFFFFFFu);
}

// device status words
enum : uint32_t {
    kStOk = 0,
    kStHufBadEnd = 1,      // Huffman stream did not end exactly at its start / no sentinel
    kStSeqBadEnd = 2,      // sequence bitstream over/under-run, bad code
    kStSeqLiterals = 3,    // sum of literal lengths exceeds the block's literals
    kStBadOffset = 4,      // match offset reaches before the frame start
    kStSizeMismatch = 5,   // decoded size != section original size / block too large
    kStRunsOverflow = 6,
    kStInternal = 7,       // a host-side guarantee did not hold (k_huf_decode: task spans > kHufTaskSpan)
    kStChecksum = 8,       // a frame's Content_Checksum differs from XXH64 of what it decoded to
};

// Frame checksums (RFC 8878 3.1.1: low 32 bits of XXH64, seed 0, of the frame's decoded bytes).  One piece of a
// frame per k_xxh64_frames workgroup: the blocks of the frame that the loaded selection holds.  A frame that
// continues into the next tile leaves its running state (XxhCarry) behind.
struct XxhSeg {
    uint32_t blk0, blk1;   // blocks [blk0, blk1) of the loaded selection
    uint32_t expected;     // the stored checksum
    uint32_t flags;        // 1: the frame begins with blk0; 2: it ends with blk1 - 1
};
struct XxhCarry {
    uint64_t v[4];
    uint64_t total;        // bytes hashed so far
    uint32_t n_mem, pad;
    uint8_t mem[32];       // bytes of the stripe not yet complete
};

}  // namespace nafgpu
