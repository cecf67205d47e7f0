// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the NAF decode path.
//
// Everything here is integer / bit manipulation bound by HBM and LDS, not by MFMA:
//   k_huf_decode   K1  Huffman literal streams (zstd literals section)        lane = stream
//   k_seq_states / k_seq_values   K2  FSE sequence decode (LL / OF / ML)   lane = block, then wave = block
//   k_scan_*       K3/K6  tile scans: block bases, record ends, mask run ends
//   k_copy_fill        Raw / RLE blocks and literal sections
//   k_rep_partial/scan/apply, k_lz_literals, k_lz_index, k_lz_match_pass, k_lz_matches_ordered   K4
//   k_mask_apply       soft-mask lower-casing incl. the record-end rule (mod.rs:402-441)
//   k_hash64           checksum used by full-size parity tests
// Reference counterparts are cited per kernel.  Format: RFC 8878 / SURVEY.md Appendix B.
#include <hip/hip_runtime.h>
#ifdef NAFGPU_EMU
#include <cstdio>
#include <vector>
#endif

#include <cstdlib>
#include <type_traits>

#include "hash64.h"
#include "kernels.h"
#include "plan.h"

namespace nafgpu {

namespace {

// Timing ablations (tools/ablate.sh builds experiment libraries with -DNAFGPU_ABLATE=<mask>; results are then
// wrong by construction and no error is flagged).  The product is built without it: every `kAblate` test folds to nothing.
#ifndef NAFGPU_ABLATE
#define NAFGPU_ABLATE 0
#endif
constexpr uint32_t kAblate = NAFGPU_ABLATE;   // k_huf_decode: 1 no output stores, 2 no look-ups (rows fill instantly), 4 no input loads,
                                              // 8 no table staging, 16 stores only to a small window, 32 no dictionary reads;
                                              // k_mask_apply: 512 no atomics on shared chunks, 1024 no record search

__device__ inline void flag_error(uint32_t *status, uint32_t code, uint32_t detail) {
    if (kAblate) return;
    if (atomicCAS(&status[0], 0u, code) == 0u) status[1] = detail;
}

// ======================================================================================
// K1  Huffman literal streams
// ======================================================================================
// One wave per task, one lane per stream.  What bounds it on MI355X, in this order:
// the HBM request pattern (610 k scattered input and output fronts), LDS capacity (which caps the
// waves per CU) and the dependent chain of one look-up.  Hence:
//   * two-symbol table: while a task's Huffman tables are staged into LDS they are widened to
//     W = max(max_bits, 8) index bits and every entry describes the next one or two symbols.  Three
//     entry formats (plan.h: HufTblKind), one kernel instantiation each:
//       baked    8 bytes, the bytes to emit inside the entry (for DNA/RNA: the 2 x 2 IUPAC characters of
//                the two packed bytes -- SequenceReader::read_nucleotide/decode, reader.rs:121-172, costs
//                nothing per symbol and the 4-bit intermediate never reaches HBM).  Tasks with ONE tree.
//       compact  4 bytes {sym1, sym2, bits, bits of sym1, two}; characters from a 512-byte table shared by the
//                wave.  Tasks with several trees, one of them with more than 64 symbols.
//       dict     2 bytes {index of sym1, index of sym2, bits - 1, two} into a per-tree dictionary of <= 64
//                symbols holding their output bytes.  Tasks with several small-alphabet trees: what real
//                genomes give (one tree per 128 KiB block, A C G T plus a few IUPAC codes).  Half the LDS
//                of the compact format per tree: the lanes resident per CU set the throughput there.
//   * bit window: {hi, lo, nw} are three consecutive 32-bit words of the backward stream,
//     s = 32 - (bits of hi consumed) in [0, 31]; peek = v_alignbit_b32(hi, lo, s); consuming len
//     bits is s -= len, a borrow meaning "advance one word", s &= 31.
//   * input: every lane keeps the current 128-byte line of its stream in 32 VGPRs and feeds a
//     16-word ring in LDS from it (stored transposed, word x of lane l at x*64 + l: any mix of
//     positions is bank-conflict free), one 32-byte piece per round at most.  Invariant (a look-up
//     consumes <= 11 bits, so a round <= 6 words, + 2 words of look-ahead): >= 9 staged words past
//     the cursor after each service.  Each line of compressed input is requested exactly once.
//   * output: each lane appends to its own row in LDS whose byte 0 is unit-aligned in the destination
//     (one unaligned ds_write_b32 per look-up, no wrap arithmetic).  Once per round the rows holding a
//     complete unit are listed (ballot + rank) and written out 16 bytes per lane, a unit per lane
//     group: every store instruction writes whole 128-byte lines (baked tables: 128-byte units, eight
//     rows per instruction) or whole 64-byte half lines (compact / dict tables, sixteen rows per
//     instruction: 64 bytes less LDS per lane, where lanes per CU matter more than the store pattern).
//     The row's owner moves its leftover (< 64 B) down.
//   * a round is [flush what earlier rounds completed] -> [16 look-ups] -> [land the next piece].
//   * SEG: streams of a block that has a few LZ sequences (plan.h: kDirectSeqMax) write their literals
//     straight to their final positions: the stream's symbols are cut into segments by the block's
//     decoded sequences (k_seq_values ran before); at the end of a segment the lane drains its row
//     byte-wise, like at the end of a stream, and re-bases it at the next literal run.  The literal
//     buffer round trip (K1 -> lit -> k_lz_literals -> out) disappears for those blocks.
// DESIGN.md section 4 has the measurements behind each of these choices.
constexpr uint32_t kRingWords = 16;
template <int TBL>
struct HufGeom {
#ifndef NAFGPU_K1_UNIT64
#define NAFGPU_K1_UNIT64 0
#endif
    static constexpr bool kBig = TBL == kTblBaked && !NAFGPU_K1_UNIT64;       // (128-byte units for the other formats: measured, no faster; 4 KB more LDS per wave)
    static constexpr uint32_t kUnit = kBig ? 128u : 64u;               // output bytes flushed per row at a time
    static constexpr uint32_t kUnitShift = kBig ? 7u : 6u;
    static constexpr uint32_t kPitch = kUnit + 64u + 8u;               // row pitch: unit + one round's worth + slack, 8-byte aligned, 2 (mod 32)-dword stride
    static constexpr uint32_t kLanesPerRow = kUnit / 16u;              // lanes that store one row's unit
    static constexpr uint32_t kRowsPerStore = 64u / kLanesPerRow;      // rows served by one store instruction
    static constexpr uint32_t kStoreIters = 64u / kRowsPerStore;       // store instructions per flush at most
};
#ifndef NAFGPU_EMU
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// a volatile access that stays an LDS instruction (a volatile access through a generic pointer becomes flat_store)
#define NAFGPU_LDS_VOLATILE(T, p) ((__attribute__((address_space(3))) volatile T *)(p))
#else
#define NAFGPU_LDS_VOLATILE(T, p) (reinterpret_cast<volatile T *>(p))
#endif
constexpr uint32_t kBufRange = 0x80000000u;   // bytes a wave's buffer descriptors cover
constexpr uint32_t kBufOff = 0xFFFFFF00u;     // an offset outside that range: switches the lane off
constexpr int kBufWord3 = 0x00020000;         // raw buffer descriptor word 3 (gfx9 family: DATA_FORMAT = 32)

// Ordering point for LDS traffic inside a ONE-WAVE workgroup.  LDS instructions of a wave execute in
// issue order, so no s_waitcnt / s_barrier is needed -- only the compiler must not move LDS accesses
// across this point.  Unlike __syncthreads() this does not wait for outstanding global stores
// (vmcnt), which is what lets the flush stores stay in flight behind the next look-ups.
__device__ inline void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// baked table entry (uint2):
//   .x  bytes to emit: first symbol in the low half, second symbol in the high half
//       (ASCII: two characters per symbol; plain: one byte per symbol, in bits 0-7 and 16-23)
//   .y  bits  0-7  bits consumed when everything the entry offers is taken
//       bits  8-15 output bytes produced then
//       bits 16-23 bits consumed by the first symbol alone (stream tail)
//       bit  24    entry holds two symbols
//       bit  25    escape: .x = first entry of the sub-table
// compact entry (uint32): sym1 | sym2 << 8 | bits << 16 | bits of sym1 << 20 | two << 24 | escape << 25 (then bits 0-15: sub-table)
// dict entry (uint16):    symbols (0, 1, 2) | index of sym1 << 2 | index of sym2 << 8 | bits << 12 (indices into the task's dictionary;
//                         a second symbol is only taken when its index is < 16 -- the dictionary lists short codes first; the
//                         fields sit where one AND / shift turns them into LDS byte offsets);
//                         0 symbols / 0 bits marks an escape (sub-table number in bits 2-11): the cursor does not move on it, the
//                         lane idles for the rest of its group of four look-ups and resolves it at the start of the next;
//                         sub-table entries hold bits - W

// "-TGKCYSBAWRDMHVN"[nib], with index 1 = t_char ('T' for DNA, 'U' for RNA)  (reader.rs:152-172)
__device__ inline uint32_t nib_char(uint32_t nib, uint32_t t_char) {
    const uint64_t lo = 0x425359434B47002Dull | (static_cast<uint64_t>(t_char) << 8);   // - T G K C Y S B
    const uint64_t hi = 0x4E56484D44525741ull;                                          // A W R D M H V N
    return static_cast<uint32_t>(((nib & 8u) ? hi : lo) >> (8u * (nib & 7u))) & 0xFFu;
}
// both characters of one packed byte: low nibble first (reader.rs:131-136)
__device__ inline uint32_t byte_chars(uint32_t b, uint32_t t_char) {
    return nib_char(b & 15u, t_char) | (nib_char(b >> 4, t_char) << 8);
}

struct HufLane {
    uint32_t hi, lo, nw;   // bit window: hi = word under the cursor, lo = next, nw = the one after
    uint32_t s;            // 32 - (bits of hi consumed), in [0, 31]
    uint32_t ra;           // LDS byte offset (inside s_ring) of the look-ahead word: ((rp + 2) & 15) << 8 | lane << 2
};

// consume `len` bits: on a borrow shift the window by one word and fetch the next look-ahead
__device__ inline void huf_advance(HufLane &L, const uint8_t *ring_bytes, uint32_t len) {
    const uint32_t s2 = L.s - len;
    const bool adv = static_cast<int32_t>(s2) < 0;
    L.s = s2 & 31u;
    L.hi = adv ? L.lo : L.hi;
    L.lo = adv ? L.nw : L.lo;
    L.ra = (L.ra + (adv ? 256u : 0u)) & 0xFFFu;
    L.nw = *reinterpret_cast<const uint32_t *>(ring_bytes + L.ra);
}

struct HufLook {           // what one table look-up says about the next one or two symbols
    uint32_t x;            // bytes to emit: first symbol in the low half, second in the high half
    uint32_t bits;         // bits consumed when both are taken
    uint32_t two;          // 1: the entry holds two symbols
    uint32_t first;        // format-specific handle for "bits of the first symbol alone" (stream tail only)
};

// ---- sub-streams (plan.h: HufStream::sub, HufSync): where the parts of a stream begin ----------------------------------
// Both kernels take the tasks of k_huf_decode as they are -- a wave per task, the task's trees staged once -- but stage ONE
// BYTE per table entry, the code length: all that is asked here is where symbols begin.  Bits come straight from memory
// (an unaligned 8-byte load per symbol: the sections this is for are small, the loads hit L2).
struct HufLens {
    const uint8_t *lut;       // 2^mb code lengths of the lane's tree (LDS)
    uint32_t mb;
    const uint8_t *bits;      // first byte of the stream
    // ---- the bits, a WINDOW at a time: 32 bytes in registers serve kHufLensBatch symbols (of <= 11 bits each), so that a
    // memory latency is paid once per batch and not once per symbol -- every lane of the wave loads its window at the same
    // place in the code (uniform control flow: one wait for all), then the symbols come out of a 64-bit buffer that is
    // topped up a dword at a time from the window.
    uint32_t w0, w1, w2, w3, w4, w5;   // the window's lower 24 bytes (the upper 8 go into buf at once); scalars, not an array: registers
    uint64_t buf;             // the next bits, first bit on top
    int32_t nbits;            // valid bits in buf
    int32_t j;                // next window dword to append (7 .. 0, then none)
    __device__ void load(uint32_t pos) {                                   // the window that ends with the byte holding bit pos - 1
        const int64_t e = (static_cast<int64_t>(pos) + 7) >> 3;
        uint4 qa, qb;
        __builtin_memcpy(&qa, bits + (e - 32), 16);                          // (in front of the stream: readable, kSrcFrontPad)
        __builtin_memcpy(&qb, bits + (e - 16), 16);
        w0 = qa.x; w1 = qa.y; w2 = qa.z; w3 = qa.w;
        w4 = qb.x; w5 = qb.y;
        const uint32_t drop = static_cast<uint32_t>(8 * e - static_cast<int64_t>(pos));   // bits of the top byte above the cursor (0 .. 7)
        buf = ((static_cast<uint64_t>(qb.w) << 32) | qb.z) << drop;
        nbits = 64 - static_cast<int32_t>(drop);
        j = 5;
    }
    // length of the code under the cursor; 0: invalid, or longer than the `left` bits the stream still has.
    // (Measured and dropped: a second table with as many symbols as the index holds whole -- two per look-up on DNA -- made
    //  the walk no faster, 2.4 against 2.0 ms for 8 K symbols: at one wave per SIMD the walk runs at the latency of its
    //  forty-odd dependent instructions and five branches per step, not at the rate of its look-ups.)
    __device__ uint32_t len_next(uint32_t left) const {
        const uint32_t len = lut[static_cast<uint32_t>(buf >> (64u - mb))];
        return len <= left ? len : 0u;
    }
    __device__ void consume(uint32_t len) {
        buf <<= len;
        nbits -= static_cast<int32_t>(len);
        if (nbits <= 32 && j >= 0) {
            // the next dword is always w5: the window moves down a register (indexing w by j -- even as a chain of selects --
            // is turned into an indexed load from a stack array by hipcc: a trip to scratch memory per top-up)
            const uint32_t x = w5;
            w5 = w4;
            w4 = w3;
            w3 = w2;
            w2 = w1;
            w1 = w0;
            buf |= static_cast<uint64_t>(x) << (32 - nbits);
            nbits += 32;
            j--;
        }
    }
};
constexpr uint32_t kHufLensBatch = 20;       // steps per window: 20 x 11 bits + a look-up's 11 < the 249 bits a window holds at least
// the task's trees as length bytes, tree c at s_len + lens_off(c); returns the lane's table through its copy's lds_off
__device__ inline void huf_stage_lens(const HufTask &task, const HufTblCopy *copies, const uint16_t *pool, uint8_t *s_len) {
    uint32_t at = 0;
    for (uint32_t k = 0; k < task.n_copies; k++) {
        const HufTblCopy cp = copies[task.first_copy + k];
        const uint32_t mb = cp.bits & 0xFFu;
        const uint16_t *x1 = pool + cp.pool_off;
        for (uint32_t i = threadIdx.x; i < (1u << mb); i += blockDim.x) s_len[at + i] = static_cast<uint8_t>(x1[i] >> 8);
        at += (1u << mb) < 16u ? 16u : (1u << mb);
    }
}
__device__ inline HufLens huf_lens_of(const HufTask &task, const HufTblCopy *copies, const uint8_t *s_len, const HufStream &st, const uint8_t *src) {
    HufLens L{};
    L.lut = s_len;
    L.mb = 1;
    L.bits = src + st.src_end - st.src_len;                 // (a lane without a stream: never dereferenced)
    uint32_t at = 0;
    for (uint32_t k = 0; k < task.n_copies; k++) {
        const HufTblCopy cp = copies[task.first_copy + k];
        const uint32_t mb = cp.bits & 0xFFu;
        if (cp.lds_off == st.tbl_lds) {
            L.lut = s_len + at;
            L.mb = mb;
        }
        at += (1u << mb) < 16u ? 16u : (1u << mb);
    }
    return L;
}
__device__ inline uint32_t huf_stream_bits(const HufStream &st, const uint8_t *src) {   // data bits of the stream (0: no end mark)
    const uint32_t lastb = src[st.src_end - 1];
    if (lastb == 0) return 0;
    return (st.src_len - 1u) * 8u + (31u - static_cast<uint32_t>(__clz(static_cast<int>(lastb))));
}
// part k of S begins (as a guess) at bit total * (S - k) / S; the marks of a part are `step` bits apart
__device__ inline uint32_t huf_part_guess(uint32_t total, uint32_t k, uint32_t S) {
    return static_cast<uint32_t>(static_cast<uint64_t>(total) * (S - k) / S);
}
__device__ inline uint32_t huf_mark_step(uint32_t total, uint32_t S) {
    const uint32_t part = total / S + 1u;
    const uint32_t step = (part + kHufSyncMarks - 1u) / kHufSyncMarks;
    return step < 64u ? 64u : step;
}

// lane = part: decode from the guessed first bit, note the boundaries at the marks and where the next part's guess is crossed.
// The walk is a chain of dependent instructions -- table look-up, shifts, comparisons, a handful of branches: some 600 cycles
// a symbol for a wave alone on its SIMD -- and a section that is cut has few tasks: the task's 64 parts are therefore spread
// over kHufSyncSpread waves (every kHufSyncSpread-th thread of the workgroup has a part), so that a SIMD has several walks to
// switch between.
constexpr uint32_t kHufSyncSpread = 4;
__global__ __launch_bounds__(64 * kHufSyncSpread) void k_huf_sync(const uint8_t *__restrict__ src, const HufTask *__restrict__ tasks,
                                                 const HufTblCopy *__restrict__ copies, const HufStream *__restrict__ streams,
                                                 const uint16_t *__restrict__ pool, HufSync *sync, const uint32_t *status) {
    HIP_DYNAMIC_SHARED(uint8_t, s_len)
    if (status[0] != 0) return;
    const uint32_t lane = threadIdx.x / kHufSyncSpread;
    const HufTask task = tasks[blockIdx.x];
    huf_stage_lens(task, copies, pool, s_len);
    __syncthreads();
    const bool have = threadIdx.x % kHufSyncSpread == 0 && lane < task.n_streams;
    HufStream st{};
    if (have) st = streams[task.first_stream + lane];
    const uint32_t k = st.sub & 0xFFu, S = st.sub >> 8;
    // (the record is written in place, mark by mark: a private copy indexed by `j` would live in scratch memory)
    HufSync *out = sync + task.first_stream + (have ? lane : 0u);
    uint32_t end_pos = kHufSyncBad, n_total = 0;
    const uint32_t total = have ? huf_stream_bits(st, src) : 0u;
    HufLens L = huf_lens_of(task, copies, s_len, st, src);
    const uint32_t g = huf_part_guess(total, k, S ? S : 1u), r = k + 1 < S ? huf_part_guess(total, k + 1, S) : 0u;
    const uint32_t step = huf_mark_step(total, S ? S : 1u);
    uint32_t pos = g, cnt = 0, j = 0;
    int64_t mark = static_cast<int64_t>(g) - step;             // mark j = g - (j + 1) step, as long as it lies above the part's end
    bool done = !(have && S > 1 && total != 0);
    auto at_boundary = [&]() {                                // pos is a symbol boundary: marks crossed, the end reached?
        while (j < kHufSyncMarks && mark > static_cast<int64_t>(r) && static_cast<int64_t>(pos) <= mark) {
            out->off[j] = static_cast<uint8_t>(mark - pos);
            out->cnt[j] = static_cast<uint16_t>(cnt);
            j++;
            mark -= step;
        }
        if (pos <= r) {
            end_pos = pos;
            n_total = cnt;
            done = true;
        }
    };
    // (the per-symbol path is a table look-up, two shifts and ONE comparison: `thr` is the next place where something is to
    //  be noted -- the next mark, or the part's end)
    auto next_thr = [&]() -> uint32_t { return j < kHufSyncMarks && mark > static_cast<int64_t>(r) ? static_cast<uint32_t>(mark) : r; };
    if (!done) at_boundary();
    uint32_t thr = next_thr();
    while (__any(done ? 0 : 1)) {                              // (every lane of the wave takes part in every window load)
        if (!done) L.load(pos);
#pragma unroll 1
        for (uint32_t i = 0; i < kHufLensBatch; i++) {
            const uint32_t len = done ? 0u : L.len_next(pos);
            if (!done && (len == 0 || cnt >= 0xFFFFu)) done = true;   // (a wrong start may run into anything: the part is then decoded from its true one)
            if (done) continue;
            L.consume(len);
            pos -= len;
            cnt++;
            if (pos <= thr) {
                at_boundary();
                thr = next_thr();
            }
        }
    }
    if (have) {
        out->end_pos = end_pos;
        out->total = n_total;
        for (; j < kHufSyncMarks; j++) out->off[j] = 0xFF;    // marks not reached: no boundary noted
    }
}

// lane = stream (the lane of its part 0): the parts one after the other from their TRUE first bits -- each a short decode,
// until it stands on a boundary the guessed decode noted -- and the three words k_huf_decode wants written into every part
__global__ __launch_bounds__(64) void k_huf_bounds(const uint8_t *__restrict__ src, const HufTask *__restrict__ tasks,
                                                   const HufTblCopy *__restrict__ copies, HufStream *streams,
                                                   const uint16_t *__restrict__ pool, const HufSync *__restrict__ sync, uint32_t *status) {
    HIP_DYNAMIC_SHARED(uint8_t, s_len)
    if (status[0] != 0) return;
    const uint32_t lane = threadIdx.x;
    const HufTask task = tasks[blockIdx.x];
    huf_stage_lens(task, copies, pool, s_len);
    __syncthreads();
    if (lane >= task.n_streams) return;
    const uint32_t me = task.first_stream + lane;
    const HufStream st = streams[me];
    const uint32_t S = st.sub >> 8;
    if (S <= 1 || (st.sub & 0xFFu) != 0) return;
    const uint32_t total = huf_stream_bits(st, src);
    HufLens L = huf_lens_of(task, copies, s_len, st, src);
    const uint32_t step = huf_mark_step(total, S);
    uint32_t t = total, before = 0;
    bool bad = total == 0;
    for (uint32_t k = 0; k < S && !bad; k++) {
        const HufSync *rec = sync + me + k;
        const uint32_t g = huf_part_guess(total, k, S), r = k + 1 < S ? huf_part_guess(total, k + 1, S) : 0u;
        const bool guess_ok = rec->end_pos != kHufSyncBad;
        uint32_t n_k = 0, e = 0;
        if (t == g && guess_ok) {                         // the guess was a boundary (part 0 always): the noted decode is the true one
            n_k = rec->total;
            e = rec->end_pos;
        } else {
            uint32_t pos = t, cnt = 0, j = 0;
            int64_t mark = static_cast<int64_t>(g) - step;
            bool done = false;
            auto at_boundary = [&]() {                        // pos is a symbol boundary: on one the guessed decode noted? at the end?
                while (j < kHufSyncMarks && mark > static_cast<int64_t>(r) && static_cast<int64_t>(pos) <= mark) {
                    if (guess_ok && rec->off[j] == static_cast<uint32_t>(mark - pos)) {   // the same boundary: the same decode from here on
                        n_k = cnt + (rec->total - rec->cnt[j]);
                        e = rec->end_pos;
                        done = true;
                        return;
                    }
                    j++;
                    mark -= step;
                }
                if (pos <= r) {
                    n_k = cnt;
                    e = pos;
                    done = true;
                }
            };
            auto next_thr = [&]() -> uint32_t { return j < kHufSyncMarks && mark > static_cast<int64_t>(r) ? static_cast<uint32_t>(mark) : r; };
            at_boundary();
            uint32_t thr = next_thr();
            while (!done && !bad) {
                L.load(pos);
#pragma unroll 1
                for (uint32_t i = 0; i < kHufLensBatch && !done; i++) {
                    const uint32_t len = L.len_next(pos);
                    if (len == 0) {
                        bad = true;
                        break;
                    }
                    L.consume(len);
                    pos -= len;
                    cnt++;
                    if (pos <= thr) {
                        at_boundary();
                        thr = next_thr();
                    }
                }
            }
        }
        if (bad) break;
        streams[me + k].sub_start = t;
        streams[me + k].sub_syms = n_k;
        streams[me + k].sub_first = before;
        before += n_k;
        t = e;
    }
    if (bad || t != 0 || before != st.n_syms) {           // (what k_huf_decode flags for a whole stream that does not end at its first bit)
        flag_error(status, kStHufBadEnd, me);
        for (uint32_t k = 0; k < S; k++) streams[me + k].sub_syms = 0;
    }
}

template <bool ASCII, int TBL, bool SEG>
__global__ __launch_bounds__(64) void k_huf_decode(const uint8_t *__restrict__ src, const HufTask *__restrict__ tasks,
                                                   const HufTblCopy *__restrict__ copies,
                                                   const HufStream *__restrict__ streams,
                                                   const uint16_t *__restrict__ pool,
                                                   const uint64_t *__restrict__ blk_base, uint8_t *out, uint8_t *lit,
                                                   const SeqBlock *__restrict__ sblocks, const Seq *__restrict__ seqs,
                                                   const uint8_t *__restrict__ dicts, uint32_t t_char, uint32_t *status) {
    using G = HufGeom<TBL>;
    constexpr uint32_t kUnit = G::kUnit, kPitch = G::kPitch, kLanesPerRow = G::kLanesPerRow;
    constexpr uint32_t kRowsPerStore = G::kRowsPerStore, kStoreIters = G::kStoreIters, kUnitShift = G::kUnitShift;
    constexpr uint32_t kOutB = ASCII ? 2 : 1;              // output bytes per symbol
    HIP_DYNAMIC_SHARED(uint2, s_tbl)
    __shared__ __attribute__((aligned(16))) uint8_t s_ring[kRingWords * 64 * 4];
    __shared__ __attribute__((aligned(16))) uint8_t s_out[64 * kPitch];
    __shared__ __attribute__((aligned(16))) uint32_t s_cunit[64];   // per round: descriptors of the ready rows, compacted (see publish)
    __shared__ uint32_t s_pv[64];      // incomplete units (publish): first valid byte of the row | bytes in the row << 8

    // compact tables: characters of a packed byte, shared by all the trees of the task
    __shared__ uint16_t s_lut[(TBL == kTblCompact && ASCII) ? 256 : 2];
    // dict tables: output bytes of the task's dictionary symbols (ASCII: two characters each), shared by all its trees
    __shared__ uint32_t s_dict[TBL == kTblDict ? kHufDictSyms : 2];

    if (status[0] != 0) return;
    const uint32_t lane = threadIdx.x;
    const HufTask task = tasks[blockIdx.x];
    if (TBL == kTblCompact && ASCII)
        for (uint32_t i = lane; i < 256; i += 64) s_lut[i] = static_cast<uint16_t>(byte_chars(i, t_char));

    // ---- stage the task's tables: 2^W two-symbol entries; W-bit prefixes whose first code is longer than
    // W bits point to a 2^(max_bits - W) entry sub-table
    if (TBL == kTblDict) {                                 // the task's dictionary and its inverse (in the still unused rows)
        uint8_t *sc = s_out;
        for (uint32_t i = lane; i < 256; i += 64) sc[i] = 0xFF;
        wave_sync();
        const uint32_t sym = lane < task.n_dict ? dicts[task.dict_off + lane] : 0u;
        if (lane < task.n_dict) sc[sym] = static_cast<uint8_t>(lane);
        s_dict[lane] = ASCII ? byte_chars(sym, t_char) : sym;
        wave_sync();
    }
    bool all_pairs = true;                                 // every main entry of every table of the task holds two symbols (uniform)
    for (uint32_t k = 0; k < ((kAblate & 8u) ? 0u : task.n_copies); k++) {
        const HufTblCopy cp = copies[task.first_copy + k];
        const uint32_t mb = cp.bits & 0xFFu, W = cp.bits >> 8;
        const uint16_t *x1 = pool + cp.pool_off;           // 2^mb entries of len << 8 | sym
        uint2 *t = s_tbl + cp.lds_off;
        uint32_t *t4 = reinterpret_cast<uint32_t *>(s_tbl) + cp.lds_off;
        uint8_t *lenlut = reinterpret_cast<uint8_t *>(reinterpret_cast<uint16_t *>(s_tbl) + cp.lds_off);   // dict format: this tree's code length of
        uint16_t *t2 = reinterpret_cast<uint16_t *>(s_tbl) + cp.lds_off + kHufDictSlots;                    //   each dictionary symbol, then the table
        const uint8_t *sc_idx = s_out;                     // symbol -> dictionary index (staged once per task, below)
        const uint32_t mbx = mb > W ? mb : W;              // bits that index x1 (zero-extended if mb < W)
        if (TBL == kTblDict) {
            for (uint32_t i = lane; i < (1u << mb); i += 64) {
                const uint32_t e = x1[i], len = e >> 8;
                if ((i & ((1u << (mb - len)) - 1u)) == 0) lenlut[sc_idx[e & 0xFFu] & 63u] = static_cast<uint8_t>(len);   // first entry of each code
            }
        }
        uint32_t n_esc = 0;
        for (uint32_t i = lane; i < (1u << W); i += 64) {
            const uint32_t v = i << (32u - W);             // the W index bits, left-aligned
            const uint32_t e1 = x1[v >> (32u - mb)];
            const uint32_t l1 = e1 >> 8;
            const bool esc = l1 > W;
            const unsigned long long m = __ballot(esc ? 1 : 0);
            const uint32_t rank = n_esc + static_cast<uint32_t>(__popcll(m & ((1ull << lane) - 1ull)));
            n_esc += static_cast<uint32_t>(__popcll(m));
            const uint32_t e2_ = esc ? 0u : x1[(v << l1) >> (32u - mb)];
            all_pairs = all_pairs && !__any((!esc && l1 + (e2_ >> 8) <= W) ? 0 : 1);
            if (!esc) {
                const uint32_t e2 = e2_;
                const uint32_t l2 = e2 >> 8;
                const uint32_t two = l1 + l2 <= W ? 1u : 0u;
                if (TBL == kTblDict) {
                    const uint32_t i1 = sc_idx[e1 & 0xFFu] & 63u, i2 = sc_idx[e2 & 0xFFu];
                    const bool pair = two && i2 < 16u;     // (the second index has four bits: short codes come first in the dictionary)
                    t2[i] = static_cast<uint16_t>((pair ? 2u : 1u) | (i1 << 2) | ((pair ? i2 : 0u) << 8) | ((pair ? l1 + l2 : l1) << 12));
                } else if (TBL == kTblCompact) {
                    t4[i] = (e1 & 0xFFu) | ((e2 & 0xFFu) << 8) | ((two ? l1 + l2 : l1) << 16) | (l1 << 20) | (two << 24);
                } else {
                    const uint32_t o1 = ASCII ? byte_chars(e1 & 0xFFu, t_char) : (e1 & 0xFFu);
                    const uint32_t o2 = ASCII ? byte_chars(e2 & 0xFFu, t_char) : (e2 & 0xFFu);
                    t[i] = make_uint2(o1 | (o2 << 16), (two ? l1 + l2 : l1) | (((1u + two) * kOutB) << 8) | (l1 << 16) | (two << 24));
                }
            } else {
                const uint32_t sub = (1u << W) + (rank << (mbx - W));
                if (TBL == kTblDict)
                    t2[i] = static_cast<uint16_t>(rank << 2);
                else if (TBL == kTblCompact)
                    t4[i] = sub | (1u << 25);
                else
                    t[i] = make_uint2(sub, 1u << 25);
                for (uint32_t j = 0; j < (1u << (mbx - W)); j++) {
                    const uint32_t es = x1[(i << (mbx - W)) | j];
                    if (TBL == kTblDict) {
                        t2[sub + j] = static_cast<uint16_t>(1u | ((sc_idx[es & 0xFFu] & 63u) << 2) | (((es >> 8) - W) << 12));
                    } else if (TBL == kTblCompact) {
                        t4[sub + j] = (es & 0xFFu) | ((es >> 8) << 16) | ((es >> 8) << 20);
                    } else {
                        const uint32_t o = ASCII ? byte_chars(es & 0xFFu, t_char) : (es & 0xFFu);
                        t[sub + j] = make_uint2(o, (es >> 8) | (kOutB << 8) | ((es >> 8) << 16));
                    }
                }
            }
        }
    }

    bool have = lane < task.n_streams;
    HufStream st{};
    if (have) st = streams[task.first_stream + lane];
    // a part of a stream (HufStream::sub): its symbols, and where they go; an empty part's lane sits the task out
    const bool is_part = (st.sub >> 8) > 1u;
    uint32_t part_end = 0;               // bits of the stream below the part's last symbol
    if (is_part) {
        if ((st.sub & 0xFFu) + 1u < (st.sub >> 8)) part_end = streams[task.first_stream + lane + 1].sub_start;
        st.n_syms = st.sub_syms;
        st.dst += st.sub_first;          // (the literal buffer holds a byte per symbol; else: the index of the stream's first literal)
        have = st.sub_syms != 0;
    }
    const uint32_t W = st.max_bits;                        // max_bits holds W here
    const uint32_t sh = 32u - W;
    const uint32_t esc_bits = st.flags >> 4;               // tree max_bits - W
    const uint32_t sh2 = sh - esc_bits, esc_mask = (1u << esc_bits) - 1u;
    const uint2 *tbl = s_tbl + st.tbl_lds;
    const uint32_t *tbl4 = reinterpret_cast<const uint32_t *>(s_tbl) + st.tbl_lds;
    const uint8_t *lenlut = reinterpret_cast<const uint8_t *>(reinterpret_cast<const uint16_t *>(s_tbl) + st.tbl_lds);
    const uint16_t *tbl2 = reinterpret_cast<const uint16_t *>(s_tbl) + st.tbl_lds + kHufDictSlots;
    uint8_t *const orow = s_out + lane * kPitch;

    HufLane L{0, 0, 0, 31, 0};
    // one table look-up on the next bits of the stream; rare long codes take a second, sub-table look-up
    // (tasks none of whose trees is deeper than W bits -- the common case -- run a copy of the loop without the test)
    const bool task_esc = __any(esc_bits != 0 ? 1 : 0) != 0;
    const bool narrow_task = !__any((have && W > 8u) ? 1 : 0);     // no look-up of the task takes more than 8 bits (pack_tasks: W is 6 .. 8)
    auto lookup = [&](auto esc) -> HufLook {
        const uint32_t peek = __builtin_amdgcn_alignbit(L.hi, L.lo, L.s);
        if (TBL == kTblBaked) {
            uint2 e = tbl[peek >> sh];
            if (decltype(esc)::value) {
                if (e.y & (1u << 25)) e = tbl[e.x + ((peek >> sh2) & esc_mask)];
            }
            return HufLook{e.x, e.y & 0xFFu, (e.y >> 24) & 1u, (e.y >> 16) & 0xFFu};
        }
        if (TBL == kTblCompact) {
            uint32_t c = tbl4[peek >> sh];
            if (decltype(esc)::value) {
                if (c & (1u << 25)) c = tbl4[(c & 0xFFFFu) + ((peek >> sh2) & esc_mask)];
            }
            const uint32_t s1 = c & 0xFFu, s2 = (c >> 8) & 0xFFu;
            const uint32_t x = ASCII ? (static_cast<uint32_t>(s_lut[s1]) | (static_cast<uint32_t>(s_lut[s2]) << 16)) : (s1 | (s2 << 16));
            return HufLook{x, (c >> 16) & 15u, (c >> 24) & 1u, (c >> 20) & 15u};
        }
        return HufLook{0, 0, 0, 0};                        // (dict tables have their own loop: decode_round_dict)
    };
    // dict tables: the raw entry under the cursor is lane state (fetched one look-up ahead, see decode_round_dict)
    uint32_t r = 0;
    auto dict_fetch = [&]() -> uint32_t { return tbl2[__builtin_amdgcn_alignbit(L.hi, L.lo, L.s) >> sh]; };
    uintptr_t ptop = 0;                  // address of the 32-byte piece holding the stream's last byte
    uint32_t wp = 0;                     // 32-byte pieces landed in the ring
    uint32_t c0 = 1, rp0 = 0, bits_total = 0;
    bool bad = false;
    uint8_t *wa = orow;                  // next write address inside the row
    uint32_t rbase = 0, end_abs = 0, h = 0;   // row coordinate of orow[0]; end and start of the stream (segment) in row coordinates
    // A wave addresses memory through two buffer descriptors (input, output) whose bases are the
    // lowest address any of its lanes touches; lanes use 32-bit offsets, and an offset outside the
    // descriptor's range switches a lane off (store is dropped, no memory traffic).
    uint64_t my_dst = ~0ull;             // destination offset of row coordinate 0 (kUnit aligned)
    uint64_t my_low = ~0ull;             // lowest input address this lane may load (128-byte aligned)
    // SEG: the stream's symbols are literals [lit0, lit0 + n_syms) of its block; sequence i of the block takes
    // literals [lpos_i, lpos_i + ll_i) to output elements opos_i .. and is followed by ml_i match elements
    const bool to_lit_lane = (st.flags & 1u) != 0;
    const uint32_t sb_idx = SEG ? static_cast<uint32_t>(st.dst >> 32) : 0u;       // 1 + SeqBlock index, 0: block without sequences
    const Seq *sq = nullptr;
    uint32_t sq_n = 0, sq_i = 0;         // the block's sequences; the one whose literal run holds the cursor (sq_n: the run after the last)
    uint32_t syms_after = 0;             // symbols of the stream that come after the current segment
    uint64_t blk_out = 0;                // first output element of the block
    if (SEG && have && sb_idx) {
        const SeqBlock sb = sblocks[sb_idx - 1];
        sq = seqs + sb.seq_first;
        sq_n = sb.n_seq;
    }
    // where literal number `lc` of the block goes: sets the segment [lc, seg_end) and its first output element
    auto segment_at = [&](uint32_t lc, uint32_t lit_end, uint32_t *seg_len) -> uint64_t {
        uint32_t run_end = 0xFFFFFFFFu, out_el = lc;
        if (sq_n) {
            Seq q{};
            bool in_run = false;
            while (sq_i < sq_n) {                          // (runs of length 0 are stepped over)
                q = sq[sq_i];
                if (q.lpos + q.ll > lc) {
                    in_run = true;
                    break;
                }
                sq_i++;
            }
            if (in_run) {
                run_end = q.lpos + q.ll;
                out_el = q.opos + (lc - q.lpos);
            } else {                                       // literals after the last sequence
                const Seq last = sq[sq_n - 1];
                out_el = last.opos + last.ll + last.ml + (lc - (last.lpos + last.ll));
            }
        }
        const uint32_t seg_end = run_end < lit_end ? run_end : lit_end;
        *seg_len = seg_end - lc;
        return blk_out + out_el;
    };
    if (have) {
        const uint8_t *lastp = src + st.src_end - 1;
        uint32_t lastb = *lastp;
        bad = lastb == 0;                                        // no end mark
        uint32_t hb = 31u - static_cast<uint32_t>(__clz(static_cast<int>(lastb | 1u)));
        if (is_part) {
            // the part begins sub_start bits above the stream's first bit: as if the stream ended there, with its end mark
            // at that bit -- and it is done part_end bits above the stream's first bit, not at it (bits_total: what it consumes)
            lastp = src + st.src_end - st.src_len + (st.sub_start >> 3);
            hb = st.sub_start & 7u;
            bad = false;
        }
        bits_total = is_part ? st.sub_start - part_end : (st.src_len - 1u) * 8u + hb;
        const uintptr_t a = reinterpret_cast<uintptr_t>(lastp);
        ptop = a - (a & 31);
        rp0 = 7u - static_cast<uint32_t>((a >> 2) & 7u);
        c0 = (3u - static_cast<uint32_t>(a & 3u)) * 8u + (8u - hb);   // bits of the top word already "consumed"
        L.s = 32u - c0;
        uint64_t dstart;
        uint32_t seg_len = st.n_syms;
        if (to_lit_lane) {
            dstart = st.dst;
        } else {
            blk_out = blk_base[st.blk];
            const uint32_t lit0 = static_cast<uint32_t>(st.dst);
            if (SEG && sq_n) {
                // first sequence whose literal run ends after lit0 (the runs are in literal order)
                uint32_t lo = 0, hi = sq_n;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    const Seq q = sq[mid];
                    if (q.lpos + q.ll > lit0)
                        hi = mid;
                    else
                        lo = mid + 1;
                }
                sq_i = lo;
                dstart = segment_at(lit0, lit0 + st.n_syms, &seg_len) * kOutB;
            } else {
                dstart = (blk_out + lit0) * kOutB;
            }
        }
        syms_after = st.n_syms - seg_len;
        h = static_cast<uint32_t>(dstart & (kUnit - 1));
        end_abs = h + seg_len * (to_lit_lane ? 1 : kOutB);
        wa = orow + h;
        my_dst = dstart - h;
        // look-ahead of the ring (<= 96 bytes past the stream start), the rest of that line and the line requested after it: inside kSrcFrontPad
        my_low = (reinterpret_cast<uintptr_t>(src) + st.src_end - st.src_len - 384u) & ~static_cast<uint64_t>(127);
    }
    wave_sync();                                           // (dict staging scratch is dead)
    {   // wave minima of my_dst / my_low through the (still unused) output rows
        uint64_t *scratch = reinterpret_cast<uint64_t *>(s_out);
        scratch[lane] = my_dst;
        scratch[64 + lane] = my_low;
        if (lane == 0) scratch[128] = st.flags & 1u;       // destination kind: uniform per launch (pack_tasks)
    }
    __syncthreads();                                       // tables, row metadata and the scratch are staged
    uint64_t dbase = ~0ull, sbase = ~0ull;
    for (uint32_t i = 0; i < 64; i++) {
        const uint64_t a = reinterpret_cast<const uint64_t *>(s_out)[i], b = reinterpret_cast<const uint64_t *>(s_out)[64 + i];
        dbase = a < dbase ? a : dbase;
        sbase = b < sbase ? b : sbase;
    }
    const bool to_lit = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<const uint64_t *>(s_out)[128])) != 0;
    __syncthreads();                                       // scratch is dead, rows may be written
    auto uniform64 = [](uint64_t v) -> uint64_t {          // the builtin returns int: cast before widening
        const uint32_t lo = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v)));
        const uint32_t hi = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32)));
        return (static_cast<uint64_t>(hi) << 32) | lo;
    };
    dbase = uniform64(dbase);
    sbase = uniform64(sbase);
    // pack_tasks keeps a task's streams within 1 GiB of input and of output; a lane that is not is a host bug
    // (SEG: later segments of a stream lie inside the same 128 KiB block as its first one)
    const bool in_range = !have || (my_dst - dbase < kBufRange - (1u << 20) &&
                                    ptop - sbase < kBufRange - (1u << 20));
    if (__any(in_range ? 0 : 1)) {
        if (!in_range) flag_error(status, kStInternal, (my_dst - dbase < kBufRange - (1u << 20) ? 0u : 1u << 31) | (lane << 24) | (blockIdx.x & 0xFFFFFFu));
        return;
    }
    uint32_t dst_rel = have ? static_cast<uint32_t>(my_dst - dbase) : 0u;
    const uint32_t src_rel = have ? static_cast<uint32_t>(ptop - sbase) : 0u;   // sbase is 128-byte aligned: offsets and addresses share their low 7 bits
    uint8_t *const obase = (to_lit ? lit : out) + dbase;
    const __amdgpu_buffer_rsrc_t rs_dst = __builtin_amdgcn_make_buffer_rsrc(obase, 0, static_cast<int>(kBufRange), kBufWord3);
#ifdef NAFGPU_EMU
    const __amdgpu_buffer_rsrc_t rs_src =
        __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(static_cast<uintptr_t>(sbase)), 0, static_cast<int>(kBufRange), kBufWord3);
#endif

    // ---- input.  The stream is consumed in 32-byte aligned PIECES, numbered backwards from the one
    // that holds its last byte (piece j = bytes [ptop - 32 j, ptop - 32 j + 32); ring word 8 j + i is the
    // dword at offset 28 - 4 i of piece j).  Pieces come out of a whole 128-byte LINE that the lane keeps
    // in 32 VGPRs: every line of compressed input is requested from L2 exactly once (with 16- or
    // 32-byte refills the line had been evicted before its next piece was needed: twice the fetch
    // traffic, four times the requests -- the refill loads then cost as much as the decode itself).
    //
    // The wait for a line is in the author's hands.  Vector memory operations retire in order, so hipcc's own wait in front of
    // the first use of a line -- requested a round earlier -- is vmcnt(0) whenever a data-dependent number of stores was issued
    // in between (the flush), and that also waits for those stores: written a moment ago, to 64 scattered fronts.  Here the
    // eight loads of a line are ONE asm statement whose outputs are tied to the registers that hold the line ("+v": the
    // load overwrites the old line in place, no copy can come between), the wait is `s_waitcnt vmcnt(k)` with k = the store
    // instructions issued since (a scalar branch picks the immediate; any other vector memory operation in between -- the
    // byte-wise units, SEG's sequence reads -- only makes the true count larger than k, and a wait for more is a safe wait),
    // and the registers pass through the wait statement, so no use of theirs can move in front of it.  hipcc does not know the
    // line registers are in flight between the two statements: nothing but land_piece reads them, and a last wait behind the
    // main loop keeps them out of the allocator's hands until the last load has landed.  (Round 1 found asm loads unsafe with
    // plain "=v" outputs -- the allocator reused them; tied operands and the wait that carries them are what changed.
    // Round 3 had measured the relaxed wait at 0.8 ms of the 10 GB archive's 10.8 with filler stores that cost 1.5.)
    u32x4 line[8];                       // line[2 q], line[2 q + 1] = piece at offset 32 q of the line
    uint32_t line_rel = 0;               // offset (from sbase) of the line held in `line`
#pragma unroll
    for (int i = 0; i < 8; i++) line[i] = u32x4{0, 0, 0, 0};
#ifndef NAFGPU_EMU
    u32x4 srd_src;                       // the input descriptor as four scalars (uniform: sbase is)
    srd_src[0] = static_cast<uint32_t>(sbase);
    srd_src[1] = static_cast<uint32_t>(sbase >> 32) & 0xFFFFu;
    srd_src[2] = kBufRange;
    srd_src[3] = static_cast<uint32_t>(kBufWord3);
    auto load_line = [&](uint32_t rel) {
        line_rel = rel;
        asm volatile("buffer_load_dwordx4 %0, %8, %9, 0 offen\n\t"
                     "buffer_load_dwordx4 %1, %8, %9, 0 offen offset:16\n\t"
                     "buffer_load_dwordx4 %2, %8, %9, 0 offen offset:32\n\t"
                     "buffer_load_dwordx4 %3, %8, %9, 0 offen offset:48\n\t"
                     "buffer_load_dwordx4 %4, %8, %9, 0 offen offset:64\n\t"
                     "buffer_load_dwordx4 %5, %8, %9, 0 offen offset:80\n\t"
                     "buffer_load_dwordx4 %6, %8, %9, 0 offen offset:96\n\t"
                     "buffer_load_dwordx4 %7, %8, %9, 0 offen offset:112"
                     : "+v"(line[0]), "+v"(line[1]), "+v"(line[2]), "+v"(line[3]), "+v"(line[4]), "+v"(line[5]), "+v"(line[6]), "+v"(line[7])
                     : "v"(rel), "s"(srd_src)
                     : "memory");
    };
    // `younger` (uniform): vector memory instructions KNOWN to have been issued since the loads that must have landed.
    // s_waitcnt takes an immediate only: a computed jump into a table of {s_waitcnt vmcnt(k); s_branch end} pairs, eight
    // bytes each (seven scalar instructions in all; hipcc's own lowering of a switch in this loop is a chain of twenty).
    // Behind it ONE empty statement carries the line registers, so that every use of theirs comes after it -- and, the
    // statements being volatile, after the wait.
    auto wait_line = [&](uint32_t younger) {
        asm volatile("s_min_u32 s92, %0, 16\n\t"
                     "s_getpc_b64 s[90:91]\n\t"            // = the address of the next instruction
                     "s_lshl3_add_u32 s92, s92, 16\n\t"     //   + 0
                     "s_add_u32 s90, s90, s92\n\t"          //   + 4
                     "s_addc_u32 s91, s91, 0\n\t"           //   + 8
                     "s_setpc_b64 s[90:91]\n\t"             //   + 12; the table begins at + 16
                     "s_waitcnt vmcnt(0)\n\ts_branch .Lnafgpu_wl_%=\n\t"
                     "s_waitcnt vmcnt(1)\n\ts_branch .Lnafgpu_wl_%=\n\t"
                     "s_waitcnt vmcnt(2)\n\ts_branch .Lnafgpu_wl_%=\n\t"
                     "s_waitcnt vmcnt(3)\n\ts_branch .Lnafgpu_wl_%=\n\t"
                     "s_waitcnt vmcnt(4)\n\ts_branch .Lnafgpu_wl_%=\n\t"
                     "s_waitcnt vmcnt(5)\n\ts_branch .Lnafgpu_wl_%=\n\t"
                     "s_waitcnt vmcnt(6)\n\ts_branch .Lnafgpu_wl_%=\n\t"
                     "s_waitcnt vmcnt(7)\n\ts_branch .Lnafgpu_wl_%=\n\t"
                     "s_waitcnt vmcnt(8)\n\ts_branch .Lnafgpu_wl_%=\n\t"
                     "s_waitcnt vmcnt(9)\n\ts_branch .Lnafgpu_wl_%=\n\t"
                     "s_waitcnt vmcnt(10)\n\ts_branch .Lnafgpu_wl_%=\n\t"
                     "s_waitcnt vmcnt(11)\n\ts_branch .Lnafgpu_wl_%=\n\t"
                     "s_waitcnt vmcnt(12)\n\ts_branch .Lnafgpu_wl_%=\n\t"
                     "s_waitcnt vmcnt(13)\n\ts_branch .Lnafgpu_wl_%=\n\t"
                     "s_waitcnt vmcnt(14)\n\ts_branch .Lnafgpu_wl_%=\n\t"
                     "s_waitcnt vmcnt(15)\n\ts_branch .Lnafgpu_wl_%=\n\t"
                     "s_waitcnt vmcnt(16)\n"
                     ".Lnafgpu_wl_%=:"
                     :
                     : "s"(__builtin_amdgcn_readfirstlane(younger))
                     : "s90", "s91", "s92", "scc", "memory");
        asm volatile(""
                     : "+v"(line[0]), "+v"(line[1]), "+v"(line[2]), "+v"(line[3]), "+v"(line[4]), "+v"(line[5]), "+v"(line[6]), "+v"(line[7])
                     :
                     : "memory");
    };
#else
    auto load_line = [&](uint32_t rel) {
        line_rel = rel;
#pragma unroll
        for (int i = 0; i < 8; i++) line[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_src, rel + 16u * i, 0, 0);
    };
    auto wait_line = [&](uint32_t) {};
#endif
    // piece `wp` -> ring words 8 wp .. 8 wp + 7 (mod 16); requests the next lower line once this one is used up
    auto land_piece = [&]() -> bool {                      // (the caller has waited for the line: wait_line)
        const uint32_t pa = src_rel - 32u * wp;
        const uint32_t q = (pa >> 5) & 3u;
        const bool b0 = q & 1u, b1 = q & 2u;
        auto sel = [&](uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3) { return b1 ? (b0 ? x3 : x2) : (b0 ? x1 : x0); };
        uint32_t *r = reinterpret_cast<uint32_t *>(s_ring) + lane + ((8 * wp) & (kRingWords - 1)) * 64;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            r[i * 64] = sel(line[1][3 - i], line[3][3 - i], line[5][3 - i], line[7][3 - i]);
            r[(4 + i) * 64] = sel(line[0][3 - i], line[2][3 - i], line[4][3 - i], line[6][3 - i]);
        }
        wp++;
        return q == 0;                                     // the line is used up: the caller requests the next lower one (load_line)
    };
    if (have) load_line(src_rel & ~127u);
    wait_line(0);
    if (have && land_piece()) load_line(line_rel - 128u);  // the ring holds two pieces
    wait_line(0);
    if (have) {
        if (land_piece()) load_line(line_rel - 128u);
        const uint32_t *rw = reinterpret_cast<const uint32_t *>(s_ring) + lane;
        L.hi = rw[rp0 * 64];
        L.lo = rw[(rp0 + 1) * 64];
        L.nw = rw[(rp0 + 2) * 64];
        L.ra = ((rp0 + 2) << 8) | (lane << 2);
        if (TBL == kTblDict) r = dict_fetch();
        if (TBL == kTblCompact) r = tbl4[__builtin_amdgcn_alignbit(L.hi, L.lo, L.s) >> sh];
    }
    wave_sync();

    // ---- flush: one unit per ready row, 16 bytes per lane.  A round produces <= 64 bytes per row, so
    // about half the rows (128-byte units) are ready in any round: the owners compact their
    // descriptors (position >> kUnitShift << 7 | row << 1 | full) into a list, and store instruction k
    // serves list entries kRowsPerStore k .. kRowsPerStore k + kRowsPerStore - 1 -- every instruction
    // writes whole units.  The list is stored transposed (entry i at dword (i % kRowsPerStore) * kStoreIters
    // + i / kRowsPerStore) so that the lanes of a group fetch their entries with one or two ds_read_b128.
    // The first and last unit of a stream or segment (bytes of a neighbour / not produced yet) are
    // written byte-wise.
    const uint32_t oct = lane % kLanesPerRow, grp = lane / kLanesPerRow;
    uint32_t n_part = 0;                                   // set by publish: rows whose unit is incomplete (uniform)
    bool took_partial = false;                             // set by publish: this lane's row goes out as an incomplete unit
    auto publish = [&](bool lane_final) -> uint32_t {      // returns the number of ready rows
        const uint32_t avail = static_cast<uint32_t>(wa - orow);
        const bool ready = have && (avail >= kUnit || (lane_final && avail > 0));
        const bool full = rbase >= h && avail >= kUnit;
        const unsigned long long m = __ballot(ready ? 1 : 0);
        const uint32_t rank = static_cast<uint32_t>(__popcll(m & ((1ull << lane) - 1ull)));
        if (ready)
            s_cunit[(rank % kRowsPerStore) * kStoreIters + (rank / kRowsPerStore)] =
                (((dst_rel + rbase) >> kUnitShift) << 7) | (lane << 1) | (full ? 1u : 0u);
        took_partial = ready && avail < kUnit;
        n_part = static_cast<uint32_t>(__popcll(__ballot(ready && !full ? 1 : 0)));
        if (n_part) s_pv[lane] = (h > rbase ? h - rbase : 0u) | (avail << 8);   // only the byte-wise path needs these
        return static_cast<uint32_t>(__popcll(m));
    };
    auto flush = [&](uint32_t n_ready) {                   // (issues exactly ceil(n_ready / kRowsPerStore) store instructions, and the byte-wise ones)
        uint32_t c[kStoreIters];
        {
            const uint4 c0v = *reinterpret_cast<const uint4 *>(&s_cunit[grp * kStoreIters]);
            c[0] = c0v.x;
            c[1] = c0v.y;
            c[2] = c0v.z;
            c[3] = c0v.w;
            if constexpr (kStoreIters == 8) {
                const uint4 c1v = *reinterpret_cast<const uint4 *>(&s_cunit[grp * kStoreIters + 4]);
                c[kStoreIters - 4] = c1v.x;
                c[kStoreIters - 3] = c1v.y;
                c[kStoreIters - 2] = c1v.z;
                c[kStoreIters - 1] = c1v.w;
            }
        }
        if (n_part) {                                      // first / last unit of a stream or segment: byte-wise, rare
#pragma unroll 1
            for (uint32_t k = 0; kRowsPerStore * k < n_ready; k++) {
                const uint32_t ck = s_cunit[grp * kStoreIters + k];
                if (kRowsPerStore * k + grp >= n_ready || (ck & 1u)) continue;
                const uint32_t row = (ck >> 1) & 63u, pos = (ck >> 7) << kUnitShift;
                const uint32_t pv = s_pv[row], rh = pv & 0xFFu, rq = pv >> 8;      // (unit-relative: the row's byte 0 is 0)
                const uint32_t lo_x = 16 * oct, hi_x = lo_x + 16;
                const uint32_t v_lo = lo_x > rh ? lo_x : rh, v_hi = hi_x < rq ? hi_x : rq;
                uint8_t *d = obase + pos + 16 * oct;
                const uint8_t *rowp = s_out + row * kPitch + 16 * oct;
#pragma clang loop vectorize(disable) unroll(disable)
                for (uint32_t x = v_lo; x < v_hi; x++) d[x - lo_x] = rowp[x - lo_x];
            }
        }
        uint2 w[kStoreIters][2];                           // all LDS reads first: the stores then go out back to back
#pragma unroll
        for (uint32_t k = 0; k < kStoreIters; k++) {
            if (kRowsPerStore * k >= n_ready) break;       // uniform
            const uint8_t *rowp = s_out + ((c[k] >> 1) & 63u) * kPitch + 16 * oct;
            w[k][0] = *reinterpret_cast<const uint2 *>(rowp);
            w[k][1] = *reinterpret_cast<const uint2 *>(rowp + 8);
        }
#pragma unroll
        for (uint32_t k = 0; k < kStoreIters; k++) {
            if (kRowsPerStore * k >= n_ready) break;       // uniform
            const uint32_t pos = (c[k] >> 7) << kUnitShift;
            const bool on = kRowsPerStore * k + grp < n_ready && (c[k] & 1u) && !(kAblate & 1u);
            const uint32_t voff = on ? ((kAblate & 16u) ? (pos & 0x1FFFC0u) : pos) + 16 * oct : kBufOff;
            u32x4 v;
            v[0] = w[k][0].x;
            v[1] = w[k][0].y;
            v[2] = w[k][1].x;
            v[3] = w[k][1].y;
            __builtin_amdgcn_raw_buffer_store_b128(v, rs_dst, voff, 0, 0);
        }
    };
    // ---- a look-up's bytes into the row.  A row position is only 2-byte (ASCII) / 1-byte aligned, and an LDS store must
    // not straddle a dword: ONE 4-byte store at a 2-byte aligned address costs 56 cycles per wave-instruction on
    // gfx950 against 6.5 for an aligned one and 11.5 for a 2-byte store (tools/ldsbench.hip) -- and hipcc fuses
    // adjacent narrow stores into exactly the straddling kind.  So: TWO narrow stores, kept apart on purpose
    // (volatile, LDS address space).  (Measured and dropped: two aligned dwords per look-up through ds_write2_b32
    // with the two characters below the position carried in a register -- 10 LDS cycles instead of 20, but the
    // 13 extra VALU instructions per look-up cost more than they saved.)
    // nsym = symbols the caller takes (0: none -- the bytes land beyond the row's content and are overwritten).
    auto put_adv = [&](uint32_t x, uint32_t nsym) {
        if (ASCII) {
            NAFGPU_LDS_VOLATILE(uint16_t, wa)[0] = static_cast<uint16_t>(x);
            NAFGPU_LDS_VOLATILE(uint16_t, wa)[1] = static_cast<uint16_t>(x >> 16);
        } else {
            NAFGPU_LDS_VOLATILE(uint8_t, wa)[0] = static_cast<uint8_t>(x);
            NAFGPU_LDS_VOLATILE(uint8_t, wa)[1] = static_cast<uint8_t>(x >> 16);
        }
        wa += nsym * kOutB;
    };

    // One round = [flush the units the previous rounds completed] -> [16 look-ups] -> [land the next
    // 32-byte piece if the ring has room].  The line loads are requested a round or two before their
    // first piece is needed, so they are older than the round's stores in the in-order VM counter.
    auto decode_round = [&](auto esc) {
        const uint32_t pos = rbase + static_cast<uint32_t>(wa - orow);
        if (kAblate & 2u) {
            if (pos < end_abs) wa += (end_abs - pos < 64u ? end_abs - pos : 64u);
            return;
        }
        // (wave-wide votes stay outside divergent code: every lane of the wave takes part)
        const bool dword_round = ASCII && TBL == kTblBaked && all_pairs &&
                                 !__any((pos < end_abs && (static_cast<uint32_t>(wa - orow) & 2u)) ? 1 : 0);
        if (pos + 32 * kOutB <= end_abs && !decltype(esc)::value && narrow_task) {
            // Four look-ups off ONE peek.  The 32 bits under the cursor serve four look-ups of <= 8 bits each: the next
            // index is the peek shifted left by what the entry consumed, and the bit window moves once per group, by the
            // sum -- a borrow still means "one word on" (the sum is <= 32).  Per look-up: a shift, an address, the table
            // read, the store and a quarter of an advance -- 6 vector instructions where a look-up with its own advance
            // takes 13, and the chain from one table read to the next is two instructions long instead of ten.
            // (Round 4, measured and dropped: the peek kept ROTATED so that the index is one AND-OR on a size-aligned table base and
            //  taking an entry one v_alignbit_b32 with the entry as shift operand -- two vector instructions per look-up instead of
            //  three, 16 of the round's ~207 gone: 10.40 against 10.47 ms.  The chain of LDS round trips is what a lane runs at.)
            auto group4 = [&](auto dword) {
                uint32_t p = __builtin_amdgcn_alignbit(L.hi, L.lo, L.s), tot = 0;
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    const uint2 e = tbl[p >> sh];
                    if (decltype(dword)::value) {
                        *reinterpret_cast<uint32_t *>(wa) = e.x;
                        wa += 4;
                    } else {
                        put_adv(e.x, 1u + ((e.y >> 24) & 1u));
                    }
                    p <<= (e.y & 31u);
                    tot += e.y;                            // (the fields above the low byte do not reach into it)
                }
                huf_advance(L, s_ring, tot & 0xFFu);
            };
            if (dword_round) {
#pragma unroll
                for (uint32_t g = 0; g < 4; g++) group4(std::true_type{});
            } else {
#pragma unroll
                for (uint32_t g = 0; g < 4; g++) group4(std::false_type{});
            }
        } else if (pos + 32 * kOutB <= end_abs) {          // 16 look-ups cannot overrun the stream (segment)
            if (dword_round) {
                // every entry of the task's table holds two symbols (codes of 4 bits or less: uniform ACGT) and every
                // lane stands on a dword boundary: each look-up is one aligned 4-byte store, nothing to carry
#pragma unroll
                for (uint32_t k = 0; k < 16; k++) {
                    const HufLook e = lookup(esc);
                    *reinterpret_cast<uint32_t *>(wa) = e.x;
                    wa += 4;
                    huf_advance(L, s_ring, e.bits);
                }
            } else {
#pragma unroll
                for (uint32_t k = 0; k < 16; k++) {
                    const HufLook e = lookup(esc);
                    put_adv(e.x, 1u + e.two);
                    huf_advance(L, s_ring, e.bits);
                }
            }
        } else if (pos < end_abs) {                        // tail of the stream (segment): never take more than is left
#pragma unroll 4
            for (uint32_t k = 0; k < 16; k++) {
                const HufLook e = lookup(esc);
                const uint32_t left = end_abs - (rbase + static_cast<uint32_t>(wa - orow));   // bytes
                const bool two = e.two && left >= 2 * kOutB;
                put_adv(e.x, left == 0 ? 0u : (two ? 2u : 1u));
                huf_advance(L, s_ring, left == 0 ? 0u : (two ? e.bits : e.first));
            }
        }
    };
    // compact tables, the same pipelining as the dict tables below: `r` is the entry under the cursor, fetched by the
    // look-up before; the next table read is issued BEFORE the two character reads of this one.
    auto compact_fetch = [&]() -> uint32_t { return tbl4[__builtin_amdgcn_alignbit(L.hi, L.lo, L.s) >> sh]; };
    auto compact_chars = [&](uint32_t c) -> uint32_t {
        const uint32_t s1 = c & 0xFFu, s2 = (c >> 8) & 0xFFu;
        return ASCII ? (static_cast<uint32_t>(s_lut[s1]) | (static_cast<uint32_t>(s_lut[s2]) << 16)) : (s1 | (s2 << 16));
    };
    auto decode_round_compact = [&](auto esc) {
        const uint32_t pos = rbase + static_cast<uint32_t>(wa - orow);
        if (kAblate & 2u) {
            if (pos < end_abs) wa += (end_abs - pos < 64u ? end_abs - pos : 64u);
            return;
        }
        auto entry = [&]() -> uint32_t {                   // the entry under the cursor, long codes resolved through their sub-table
            uint32_t c = r;
            if (decltype(esc)::value) {
                if (c & (1u << 25)) c = tbl4[(c & 0xFFFFu) + ((__builtin_amdgcn_alignbit(L.hi, L.lo, L.s) >> sh2) & esc_mask)];
            }
            return c;
        };
        if (pos + 32 * kOutB <= end_abs) {                 // 16 look-ups cannot overrun the stream (segment)
#pragma unroll
            for (uint32_t k = 0; k < 16; k++) {
                const uint32_t c = entry();
                huf_advance(L, s_ring, (c >> 16) & 15u);
                r = compact_fetch();
                put_adv(compact_chars(c), 1u + ((c >> 24) & 1u));
            }
        } else if (pos < end_abs) {                        // tail of the stream (segment): never take more than is left
#pragma unroll 4
            for (uint32_t k = 0; k < 16; k++) {
                const uint32_t c = entry();
                const uint32_t left = end_abs - (rbase + static_cast<uint32_t>(wa - orow));   // bytes
                const bool two = ((c >> 24) & 1u) && left >= 2 * kOutB;
                huf_advance(L, s_ring, left == 0 ? 0u : (two ? (c >> 16) & 15u : (c >> 20) & 15u));
                r = compact_fetch();
                put_adv(compact_chars(c), left == 0 ? 0u : (two ? 2u : 1u));
            }
        }
    };
    // dict tables.  The entry under the cursor (`r`) was fetched by the look-up before: a look-up extracts its
    // fields, moves the bit window, ISSUES the next table read and the two dictionary reads together, and only
    // then writes its characters -- one LDS latency per look-up on the dependent chain instead of two.  An escape
    // entry has 0 symbols / 0 bits: the lane stays on it (no branch in the loop) and resolves it before the
    // next round's look-ups with one sub-table read.
    const uint8_t *const dict_bytes = reinterpret_cast<const uint8_t *>(s_dict);
    auto resolve_escape = [&]() {                          // r is an escape entry: one symbol through its sub-table
        const uint32_t peek = __builtin_amdgcn_alignbit(L.hi, L.lo, L.s);
        const uint32_t c = tbl2[(1u << W) + (((r >> 2) & 0x3FFu) << esc_bits) + ((peek >> sh2) & esc_mask)];
        put_adv(*reinterpret_cast<const uint32_t *>(dict_bytes + (c & 0xFCu)), 1u);
        huf_advance(L, s_ring, (c >> 12) + W);
        r = dict_fetch();
    };
    // A round is four groups of four look-ups.  A group whose four look-ups cannot overrun the stream (segment)
    // runs them without any test; an escape met in a group is resolved IN PLACE OF the next group's first look-up (so a
    // round never produces more than 16 x 2 symbols, which is what the row -- unit + 64 bytes -- is sized for); the last
    // few symbols of a stream (segment) go one careful step at a time.
    auto decode_round_dict = [&]() {
        if (kAblate & 2u) {
            const uint32_t pos = rbase + static_cast<uint32_t>(wa - orow);
            if (pos < end_abs) wa += (end_abs - pos < 64u ? end_abs - pos : 64u);
            return;
        }
        auto step = [&]() {                                // one look-up: see above
            const uint32_t c = r;
            const uint32_t *a1 = reinterpret_cast<const uint32_t *>(dict_bytes + (c & 0xFCu));
            const uint32_t *a2 = reinterpret_cast<const uint32_t *>(dict_bytes + ((c >> 6) & 0x3Cu));
            huf_advance(L, s_ring, c >> 12);
            r = dict_fetch();
            const uint32_t x = (kAblate & 32u) ? c : (*a1 | (*a2 << 16));
            put_adv(x, c & 3u);
        };
        // ... off ONE peek where no look-up takes more than 8 bits (see group4 above): the index of the next entry is the peek
        // shifted left by what this one consumed, the bit window moves once per group
        auto gstep = [&](uint32_t &p, uint32_t &tot, bool fetch) {
            const uint32_t c = r;
            const uint32_t *a1 = reinterpret_cast<const uint32_t *>(dict_bytes + (c & 0xFCu));
            const uint32_t *a2 = reinterpret_cast<const uint32_t *>(dict_bytes + ((c >> 6) & 0x3Cu));
            p <<= ((c >> 12) & 31u);
            tot += c >> 12;
            if (fetch) r = tbl2[p >> sh];
            put_adv(*a1 | (*a2 << 16), c & 3u);
        };
#pragma unroll
        for (uint32_t g = 0; g < 4; g++) {
            const uint32_t pos = rbase + static_cast<uint32_t>(wa - orow);
            if (pos + 8 * kOutB <= end_abs && narrow_task) {
                uint32_t p, tot = 0;
                if (task_esc && (r & 3u) == 0) {           // (an escape entry: 0 bits -- the lane stood still on it since it met it)
                    resolve_escape();
                    p = __builtin_amdgcn_alignbit(L.hi, L.lo, L.s);
                } else {
                    p = __builtin_amdgcn_alignbit(L.hi, L.lo, L.s);
                    gstep(p, tot, true);
                }
                gstep(p, tot, true);
                gstep(p, tot, true);
                gstep(p, tot, false);
                huf_advance(L, s_ring, tot);
                r = dict_fetch();
            } else if (pos + 8 * kOutB <= end_abs) {       // four look-ups cannot overrun the stream (segment)
                if (task_esc && (r & 3u) == 0)
                    resolve_escape();
                else
                    step();
                step();
                step();
                step();
            } else if (pos < end_abs) {                    // never take more than is left
#pragma unroll 1
                for (uint32_t k = 0; k < 4; k++) {
                    const uint32_t left = end_abs - (rbase + static_cast<uint32_t>(wa - orow));   // bytes
                    if (left != 0 && (r & 3u) == 0) {
                        resolve_escape();
                    } else {
                        const uint32_t c = r;
                        const bool two = (c & 3u) == 2u && left >= 2 * kOutB;
                        const uint32_t bits = (c & 3u) == 2u && !two ? static_cast<uint32_t>(lenlut[(c >> 2) & 63u]) : c >> 12;
                        put_adv(*reinterpret_cast<const uint32_t *>(dict_bytes + (c & 0xFCu)) |
                                    (*reinterpret_cast<const uint32_t *>(dict_bytes + ((c >> 6) & 0x3Cu)) << 16),
                                left == 0 ? 0u : (two ? 2u : 1u));
                        huf_advance(L, s_ring, left == 0 ? 0u : bits);
                        r = dict_fetch();
                    }
                }
            }
        }
    };
    int any = 1;
    while (any) {
        // SEG: a lane whose segment is complete drains its row (whole units first, then the incomplete one) ...
        const bool seg_done = SEG && rbase + static_cast<uint32_t>(wa - orow) == end_abs && syms_after != 0;
        const uint32_t n_ready = publish(seg_done);
        wave_sync();
        flush(n_ready);
        const uint32_t n_stores = (n_ready + kRowsPerStore - 1u) / kRowsPerStore;   // store instructions the flush issued at least
        wave_sync();
        if (static_cast<uint32_t>(wa - orow) >= kUnit) {   // the flush above took this row's first unit: < 64 bytes stay
            const uint8_t *mv = orow + kUnit;
            const uint2 m0 = *reinterpret_cast<const uint2 *>(mv), m1 = *reinterpret_cast<const uint2 *>(mv + 8);
            const uint2 m2 = *reinterpret_cast<const uint2 *>(mv + 16), m3 = *reinterpret_cast<const uint2 *>(mv + 24);
            const uint2 m4 = *reinterpret_cast<const uint2 *>(mv + 32), m5 = *reinterpret_cast<const uint2 *>(mv + 40);
            const uint2 m6 = *reinterpret_cast<const uint2 *>(mv + 48), m7 = *reinterpret_cast<const uint2 *>(mv + 56);
            *reinterpret_cast<uint2 *>(orow + 0) = m0;
            *reinterpret_cast<uint2 *>(orow + 8) = m1;
            *reinterpret_cast<uint2 *>(orow + 16) = m2;
            *reinterpret_cast<uint2 *>(orow + 24) = m3;
            *reinterpret_cast<uint2 *>(orow + 32) = m4;
            *reinterpret_cast<uint2 *>(orow + 40) = m5;
            *reinterpret_cast<uint2 *>(orow + 48) = m6;
            *reinterpret_cast<uint2 *>(orow + 56) = m7;
            wa -= kUnit;
            rbase += kUnit;
        } else if (SEG && took_partial) {                  // the incomplete last unit of the segment went out: the row is empty
            rbase += static_cast<uint32_t>(wa - orow);
            wa = orow;
        }
        if (SEG && seg_done && wa == orow) {               // ... and re-bases it where the next literal run goes
            uint32_t seg_len = 0;
            const uint32_t lit_end = static_cast<uint32_t>(st.dst) + st.n_syms;
            const uint64_t dstart = segment_at(lit_end - syms_after, lit_end, &seg_len) * kOutB;
            syms_after -= seg_len;
            h = static_cast<uint32_t>(dstart & (kUnit - 1));
            dst_rel = static_cast<uint32_t>(dstart - h - dbase);
            rbase = 0;
            wa = orow + h;
            end_abs = h + seg_len * kOutB;
        }
        if (TBL == kTblDict) decode_round_dict();
        else if (TBL == kTblCompact && task_esc) decode_round_compact(std::true_type{});
        else if (TBL == kTblCompact) decode_round_compact(std::false_type{});
        else if (task_esc) decode_round(std::true_type{});
        else decode_round(std::false_type{});
        // ---- land the next piece, if the ring has room for it.
        // d = words between the oldest ring slot and the cursor; staged past the cursor = 16 - d;
        // 8 words fit once d >= 8.  A round uses <= 6 words + 2 of look-ahead, and d >= 8 whenever
        // fewer than 9 are staged, so the cursor never outruns the ring.
        {
            const uint32_t rp_mod = ((L.ra >> 8) - 2u) & 15u;
            const uint32_t d = (rp_mod - 8u * wp) & 15u;
            const bool lands = have && d >= 8u && !(kAblate & 4u);
            // The wave has ONE counter: the wait serves all the lanes that land now.  (Measured and dropped: leaving the loads
            // of the round before in flight as well -- vmcnt(stores + 8) unless a landing lane made its request only then --
            // so that a line has two rounds to arrive: 10.79 ms against 10.63 for the round-3 build in the same processes,
            // profiles/r04_k1_wait_experiments.log.  The landing does not wait for memory to any extent that matters.)
            if (__any(lands ? 1 : 0)) wait_line(n_stores);
            if (lands && land_piece()) load_line(line_rel - 128u);     // wanted one or two rounds from now
        }
        any = __any((rbase + static_cast<uint32_t>(wa - orow) < end_abs || (SEG && syms_after != 0)) ? 1 : 0);
    }
    wait_line(0);                                          // (a line requested by the last landing may still be on its way: see load_line)
    for (uint32_t t = 0; t < 2; t++) {                     // at most kUnit - 1 + 64 bytes are left in a row
        const uint32_t n_ready = publish(true);
        wave_sync();
        flush(n_ready);
        wave_sync();
        const uint32_t avail = static_cast<uint32_t>(wa - orow);
        if (avail > kUnit) {
            for (uint32_t j = 0; j < 64; j += 8) *reinterpret_cast<uint2 *>(orow + j) = *reinterpret_cast<const uint2 *>(orow + kUnit + j);
            wa -= kUnit;
            rbase += kUnit;
        } else {
            rbase += avail;
            wa = orow;
        }
    }
    if (have && !(kAblate & 6u)) {
        // words advanced: the ring holds absolute words [8 wp - 16, 8 wp)
        const uint32_t rp_mod = ((L.ra >> 8) - 2u) & 15u;
        const uint32_t rp_abs = 8u * wp - 16u + ((rp_mod - 8u * wp) & 15u);
        const uint32_t consumed = 32u * (rp_abs - rp0) + (32u - L.s) - c0;
        if (bad || consumed != bits_total) flag_error(status, kStHufBadEnd, task.first_stream + lane);
    }
}

// ======================================================================================
// K2  FSE sequence decode (App. B "Sequence decode loop")
// ======================================================================================

__device__ inline uint32_t rep_minus_one(uint32_t r) {      // rep - 1 for a concrete offset or a token
    return r + ((r & kRepToken) ? 1u : 0xFFFFFFFFu);
}

// `tok` (an offset, or "incoming rep[slot] - d") applied after the map `f` (three entries of the same kind): maps compose
__device__ inline uint32_t rep_apply_entry(uint32_t tok, const uint32_t *f, bool *bad) {
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

// K2 is two kernels.  The sequences of a block are a chain -- where the bits of sequence i+1 lie depends on the three FSE
// states after sequence i -- but only the STATES and the bit cursor are: with those known for every sequence, the values
// (literal length, match length, offset), the output positions and the repeat offsets are work for as many threads as
// there are sequences.  So
//   k_seq_states   lane = block walks the chain and does nothing else: three 4-byte cell reads, the 16-byte window, the
//                  state bits in ONE field (LL, ML, OF state updates are adjacent in the stream: <= 26 bits), one 8-byte
//                  record {cursor, states} per sequence.  (The first version did all of K2 in this loop: ~250 instructions
//                  per sequence, and with one wave per SIMD every one of them is exposed -- 7.5 ms for level-3 DNA,
//                  18-21 ms for the qualities of 10 M reads, the largest item of every LZ-heavy archive.)
//   k_seq_values   wave = block, four consecutive sequences per lane: values from the records, running sums for the
//                  positions, and the repeat-offset history as a scan of composable maps (rep_apply_entry) over the lanes.
// (Measured and dropped in the one-kernel version, twice: the three FSE tables of a block staged in LDS, 15 blocks per
//  workgroup -- level-3 DNA 16.2 -> 18.2 ms; then the bitstream too, through a 256-byte ring per block: 16.4 -> 19.0 ms.)
// `lanes` = blocks per wave.  Every sequence ends in four loads per lane and the wave goes on when the slowest of them is
// back: with fewer lanes in a wave each block runs closer to the mean latency than to the maximum.  The launcher aims at
// about one wave per CU and never goes below 16 lanes.
struct SeqRec {          // what k_seq_states leaves for k_seq_values
    int32_t pos;         // unread bits below the cursor before this sequence's extra bits
    uint32_t states;     // LL state | ML state << 9 | OF state << 18
};
static_assert(sizeof(SeqRec) == 8 && sizeof(SeqRec) <= sizeof(SeqMeta), "SeqRec layout (the records live in the SeqMeta buffer until k_lz_literals fills that)");

// One 16-byte window of a backward bitstream: a sequence reads at most 31 + 16 + 16 extra bits and 9 + 9 + 8 state
// bits = 89, the window ending at the byte that holds the cursor has > 120 unread bits, and its address depends on the
// cursor alone -- so the window load and the three table look-ups of a sequence are issued together and the dependent
// chain per sequence is ONE memory latency.  Bits below the stream's first bit are whatever precedes it in the archive
// (readable: kSrcFrontPad); a stream that reaches them has pos < 0 and is flagged.
struct SeqWindow {
    const uint8_t *p;     // first byte of the bitstream
    int32_t pos;          // unread bits below the cursor (negative: overrun)
    uint64_t lo, hi;      // bytes [off, off + 16) of the stream, off = ceil(pos / 8) - 16
    int32_t base;         // 8 * off
    __device__ void load() {
        const int32_t off = ((pos + 7) >> 3) - 16;
        base = off * 8;
        uint64_t w[2];
        __builtin_memcpy(w, p + off, 16);
        lo = w[0];
        hi = w[1];
    }
    __device__ uint32_t read(uint32_t nb) {          // nb <= 31; the window holds the bits [pos - nb, pos)
        pos -= nb;
        const uint32_t a = static_cast<uint32_t>(pos - base) & 127u;   // bit index of the field's lowest bit inside the window
        // branch-free (the lanes of a wave stand at 64 different bit positions: every branch is taken both ways)
        const bool up = a >= 64;
        const uint64_t x = up ? hi : lo, y = up ? 0ull : hi;
        const uint32_t t = a & 63u;
        const uint64_t v = (x >> t) | ((y << 1) << (63u - t));
        return static_cast<uint32_t>(v) & ((1u << nb) - 1u);
    }
};

__global__ __launch_bounds__(64) void k_seq_states(const uint8_t *__restrict__ src, const SeqBlock *__restrict__ blocks,
                                                   uint32_t n_blocks, const SeqCell *__restrict__ cells, SeqRec *recs,
                                                   uint32_t lanes, uint32_t *status) {
    if (status[0] != 0) return;
    if (threadIdx.x >= lanes) return;
    const uint32_t b = blockIdx.x * lanes + threadIdx.x;
    if (b >= n_blocks) return;
    const SeqBlock sb = blocks[b];
    const uint8_t *bits = src + sb.bits_off;
    const uint32_t lastb = bits[sb.bits_len - 1];   // host checked: non-zero
    SeqWindow r{bits, static_cast<int32_t>(sb.bits_len - 1) * 8 + (31 - __clz(static_cast<int>(lastb | 1u))), 0, 0, 0};
    r.load();
    const uint32_t s0 = r.read(sb.ll_al), s1 = r.read(sb.of_al), s2 = r.read(sb.ml_al);
    // the first dword of a cell: next_base | nb << 16 | extra_bits << 24 (read at 32-bit byte offsets from the pool's base)
    const char *cb = reinterpret_cast<const char *>(cells);
    const uint32_t bl = 8u * sb.ll_tbl, bo = 8u * sb.of_tbl, bm = 8u * sb.ml_tbl;
    uint32_t sl = s0, so = s1, sm = s2;
    int32_t pos = r.pos;
    SeqRec *dst = recs + sb.seq_first;
    const uint32_t n = sb.n_seq;
    // (everything loaded so far is used here, so that no load is pending when the loop is entered: otherwise the loop's first
    //  wait has to be vmcnt(0) on every trip, which also waits for the record store of the trip before)
#ifndef NAFGPU_EMU
    asm volatile("" ::"v"(bl), "v"(bo), "v"(bm), "v"(sl), "v"(so), "v"(sm), "v"(pos), "v"(n));
    // (the literal-buffer classes of K1 run beside this kernel: the chain goes first whenever it can issue)
    __builtin_amdgcn_s_setprio(3);
#endif
    // every sequence but the last: its record, its extra bits skipped, the three state updates (LL, ML, OF: one field)
    for (uint32_t i = 0; i + 1 < n; i++) {
        if (pos < 0) break;
        // the 12 bytes that end with the cursor's byte: at most 63 extra bits and 26 state bits lie between the
        // field's lowest bit and the cursor, so that bit is 0..96 bits above the window's first
        const int32_t off = ((pos + 7) >> 3) - 12;
        uint32_t w[3];
        __builtin_memcpy(w, bits + off, 12);
        const uint32_t cl = *reinterpret_cast<const uint32_t *>(cb + (bl + 8u * sl)), co = *reinterpret_cast<const uint32_t *>(cb + (bo + 8u * so)),
                       cm = *reinterpret_cast<const uint32_t *>(cb + (bm + 8u * sm));
        dst[i] = SeqRec{pos, sl | sm << 9 | so << 18};
        const uint32_t nbl = (cl >> 16) & 0xFFu, nbm = (cm >> 16) & 0xFFu, nbo = (co >> 16) & 0xFFu;
        pos -= static_cast<int32_t>((cl >> 24) + (co >> 24) + (cm >> 24) + nbl + nbm + nbo);
        const uint32_t a = static_cast<uint32_t>(pos - off * 8), k = a >> 5;
        const uint32_t x = k == 0 ? w[0] : (k == 1 ? w[1] : (k == 2 ? w[2] : 0u));
        const uint32_t y = k == 0 ? w[1] : (k == 1 ? w[2] : 0u);
        const uint32_t f = __builtin_amdgcn_alignbit(y, x, a & 31u);
        sl = (cl & 0xFFFFu) + __builtin_amdgcn_ubfe(f, nbm + nbo, nbl);
        sm = (cm & 0xFFFFu) + __builtin_amdgcn_ubfe(f, nbo, nbm);
        so = (co & 0xFFFFu) + __builtin_amdgcn_ubfe(f, 0, nbo);
    }
    if (pos >= 0) {                                      // the last: no state update after it
        const uint32_t cl = *reinterpret_cast<const uint32_t *>(cb + (bl + 8u * sl)), co = *reinterpret_cast<const uint32_t *>(cb + (bo + 8u * so)),
                       cm = *reinterpret_cast<const uint32_t *>(cb + (bm + 8u * sm));
        dst[n - 1] = SeqRec{pos, sl | sm << 9 | so << 18};
        pos -= static_cast<int32_t>((cl >> 24) + (co >> 24) + (cm >> 24));
    }
    if (pos != 0) flag_error(status, kStSeqBadEnd, sb.blk);
}

// k_seq_states with the chain's memory in LDS: the first dword of every cell of the block's three tables, and the
// bitstream through a 2 KiB ring per block -- no global load sits between two sequences any more: the dependent chain is
// two LDS latencies (~64 cycles each: the three cells, then the two ring dwords they point at) plus ~35 instructions, instead of an L2
// latency (~225) plus 55.  Every kSeqBatch sequences the wave tops the rings up together: a block whose cursor has left
// the upper half of its ring gets the KiB below the ring in its place, from registers that were loaded a batch earlier
// (one chunk per block is always on its way), so no wait sits in front of the top-up either.
// Only for sections whose blocks are all resident at once (launch_seq_decode: LDS per block = cells + ring, 160 KiB per
// CU): beyond that the blocks would run in batches, and two batches at half the time per sequence are no faster than
// one batch out of L2.
constexpr uint32_t kSeqRing = 2048;          // bytes of bitstream per block in LDS (+ a 16-byte guard: a dword pair never wraps)
constexpr uint32_t kSeqBatch = 88;           // sequences between top-ups: 88 x 89 bits + a dword pair < 1008 bytes, what a ring always holds below its cursor
constexpr uint32_t kSeqLdsLanes = 8;         // most blocks per wave

__global__ __launch_bounds__(64) void k_seq_states_lds(const uint8_t *__restrict__ src, const SeqBlock *__restrict__ blocks,
                                                       uint32_t n_blocks, const SeqCell *__restrict__ cells, SeqRec *recs,
                                                       uint32_t lanes, uint32_t cells_cap, long long src_min, uint32_t *status) {
    HIP_DYNAMIC_SHARED(uint32_t, s_seq)                   // per block: cells_cap cell dwords, then (kSeqRing + 16) / 4 ring dwords
    __shared__ uint64_t s_at[kSeqLdsLanes];               // payload offset of the chunk a block wants loaded
    __shared__ uint32_t s_cmd[kSeqLdsLanes], s_ro[kSeqLdsLanes];   // bit 0: write the chunk in flight at ring offset s_ro; bit 1: load the chunk at s_at
    if (status[0] != 0) return;
    const uint32_t tid = threadIdx.x;
    const uint32_t b0 = blockIdx.x * lanes;
    const uint32_t per = cells_cap + (kSeqRing + 16) / 4;
    // ---- the tables of the wave's blocks: first dword of every cell (next_base | nb << 16 | extra_bits << 24)
    for (uint32_t j = 0; j < lanes && b0 + j < n_blocks; j++) {
        const SeqBlock bj = blocks[b0 + j];
        uint32_t *t = s_seq + j * per;
        const uint32_t nl = 1u << bj.ll_al, no = 1u << bj.of_al, nm = 1u << bj.ml_al;
        for (uint32_t i = tid; i < nl; i += 64) t[i] = *reinterpret_cast<const uint32_t *>(cells + bj.ll_tbl + i);
        for (uint32_t i = tid; i < no; i += 64) t[nl + i] = *reinterpret_cast<const uint32_t *>(cells + bj.of_tbl + i);
        for (uint32_t i = tid; i < nm; i += 64) t[nl + no + i] = *reinterpret_cast<const uint32_t *>(cells + bj.ml_tbl + i);
    }
    const bool live = tid < lanes && b0 + tid < n_blocks;
    SeqBlock sb{};
    if (live) sb = blocks[b0 + tid];
    // Ring geometry.  Payload byte x lives at ring offset (x - lo0) mod kSeqRing, lo0 = the lowest byte of the first
    // fill (16-byte aligned; the stream's last byte in the ring's top 16 bytes).  The ring holds payload bytes
    // [lo, lo + kSeqRing); lo moves down a KiB per top-up, so chunks never wrap and the one at ring offset 0 refreshes the guard.
    const uint64_t end_off = sb.bits_off + sb.bits_len;                  // one past the stream's last byte
    const uint64_t lo0 = ((end_off + 15) & ~uint64_t(15)) - kSeqRing;    // (wraps below zero for a stream near the payload's start: never dereferenced there)
    uint64_t lo = lo0;
    // ---- first fill: both halves of every block's ring; and the first chunk below goes on its way
    if (live) s_at[tid] = lo0;
    wave_sync();
    uint4 fly[kSeqLdsLanes];                                             // the chunk in flight of every block (this thread's 16 bytes of it)
    auto load16 = [&](uint64_t at) -> uint4 {
        uint4 v = make_uint4(0, 0, 0, 0);
#ifdef NAFGPU_EMU
        if (static_cast<long long>(at) >= src_min) __builtin_memcpy(&v, src + at, 16);   // (the host compiler may take uint4 for aligned)
#else
        if (static_cast<long long>(at) >= src_min) v = *reinterpret_cast<const uint4 *>(src + at);
#endif
        return v;
    };
#pragma unroll
    for (uint32_t j = 0; j < kSeqLdsLanes; j++) {
        fly[j] = make_uint4(0, 0, 0, 0);
        if (j < lanes && b0 + j < n_blocks) {
            const uint64_t base = s_at[j];
            uint32_t *rj = s_seq + j * per + cells_cap;
            const uint4 h0 = load16(base + 16 * tid), h1 = load16(base + 1024 + 16 * tid);
            *reinterpret_cast<uint4 *>(rj + 4 * tid) = h0;
            *reinterpret_cast<uint4 *>(rj + 256 + 4 * tid) = h1;
            if (tid == 0) *reinterpret_cast<uint4 *>(rj + kSeqRing / 4) = h0;     // the guard mirrors ring bytes 0..15
            fly[j] = load16(base - 1024 + 16 * tid);
        }
    }
    wave_sync();
    const uint32_t *ring = s_seq + (tid < lanes ? tid : 0) * per + cells_cap;
    // stream bit p (0 = lowest bit of the stream's first byte) is ring bit (bit0 + p) mod (8 kSeqRing)
    const uint32_t bit0 = static_cast<uint32_t>((sb.bits_off - lo0) & (kSeqRing - 1)) * 8u;
    // the <= 26-bit field whose lowest bit is stream bit p_low: two ring dwords (the guard takes the one past the ring's end),
    // addressed once the cells have said where the field lies -- a second LDS latency on the chain, but no selects
    auto field = [&](uint32_t p_low) -> uint32_t {
        const uint32_t rb = (bit0 + p_low) & (kSeqRing * 8 - 1);
        const uint32_t *w = ring + (rb >> 5);
        return __builtin_amdgcn_alignbit(w[1], w[0], rb & 31u);
    };
    int32_t pos = -1;                                                    // unread bits below the cursor (as in k_seq_states)
    uint32_t sl = 0, so = 0, sm = 0, n = 0;
    SeqRec *dst = recs + sb.seq_first;
    if (live) {
        const uint32_t lastb = src[end_off - 1];                         // host checked: non-zero
        pos = static_cast<int32_t>(sb.bits_len - 1) * 8 + (31 - __clz(static_cast<int>(lastb | 1u)));
        n = sb.n_seq;
        const int32_t p1 = pos - static_cast<int32_t>(sb.ll_al + sb.of_al + sb.ml_al);   // the three initial states: <= 26 bits below the end mark
        if (p1 >= 0) {
            const uint32_t f = field(static_cast<uint32_t>(p1));
            sl = __builtin_amdgcn_ubfe(f, sb.of_al + sb.ml_al, sb.ll_al);
            so = __builtin_amdgcn_ubfe(f, sb.ml_al, sb.of_al);
            sm = __builtin_amdgcn_ubfe(f, 0, sb.ml_al);
        }
        pos = p1;
    }
#ifndef NAFGPU_EMU
    __builtin_amdgcn_s_setprio(3);
#endif
    const uint32_t *tl = s_seq + (tid < lanes ? tid : 0) * per, *tof = tl + (1u << sb.ll_al), *tm = tof + (1u << sb.of_al);
    uint32_t i = 0;
    bool pending = live;                                                 // a chunk is in flight for this block
    for (;;) {
        // ---- up to kSeqBatch sequences without a global load
        const uint32_t stop = i + kSeqBatch;
        for (; i < stop; i++) {
            const bool on = live && i + 1 < n && pos >= 0;
            if (!__any(on ? 1 : 0)) break;
            if (!on) continue;
            const uint32_t cl = tl[sl], co = tof[so], cm = tm[sm];
            dst[i] = SeqRec{pos, sl | sm << 9 | so << 18};
            const uint32_t nbl = (cl >> 16) & 0xFFu, nbm = (cm >> 16) & 0xFFu, nbo = (co >> 16) & 0xFFu;
            const int32_t p1 = pos - static_cast<int32_t>((cl >> 24) + (co >> 24) + (cm >> 24) + nbl + nbm + nbo);
            // (an overrun, p1 < 0, reads some dword of the ring: the states no longer matter, the block stops here and is flagged)
            const uint32_t f = field(static_cast<uint32_t>(p1 < 0 ? 0 : p1));
            sl = (cl & 0xFFFFu) + __builtin_amdgcn_ubfe(f, nbm + nbo, nbl);
            sm = (cm & 0xFFFFu) + __builtin_amdgcn_ubfe(f, nbo, nbm);
            so = (co & 0xFFFFu) + __builtin_amdgcn_ubfe(f, 0, nbo);
            pos = p1;
        }
        const bool more = live && i + 1 < n && pos >= 0;
        if (!__any(more ? 1 : 0)) break;
        // ---- top up.  A block whose cursor has left the upper half of its ring (the five dwords that end at the cursor too)
        // takes the chunk in flight in that half's place and sends for the next one.
        if (tid < kSeqLdsLanes) s_cmd[tid] = 0;
        wave_sync();
        if (more) {
            const uint64_t cur = sb.bits_off + (static_cast<uint32_t>(pos > 0 ? pos - 1 : 0) >> 3);   // payload offset of the byte the next bit read lies in
            uint32_t cmd = 0;
            // (a field's dword pair reaches up to eight bytes above the cursor's byte)
            if (pending && static_cast<long long>(cur) + 16 <= static_cast<long long>(lo) + kSeqRing / 2) {   // (signed: lo goes below the payload's start)
                lo -= kSeqRing / 2;
                s_ro[tid] = static_cast<uint32_t>((lo - lo0) & (kSeqRing - 1));
                cmd = 1;
                pending = false;
            }
            if (!pending) {
                s_at[tid] = lo - kSeqRing / 2;
                cmd |= 2;
                pending = true;
            }
            s_cmd[tid] = cmd;
        }
        wave_sync();
#pragma unroll
        for (uint32_t j = 0; j < kSeqLdsLanes; j++) {
            const uint32_t cmd = s_cmd[j];                               // (uniform)
            if (cmd & 1u) {
                uint32_t *rj = s_seq + j * per + cells_cap;
                const uint32_t ro = s_ro[j];
                *reinterpret_cast<uint4 *>(rj + ro / 4 + 4 * tid) = fly[j];
                if (ro == 0 && tid == 0) *reinterpret_cast<uint4 *>(rj + kSeqRing / 4) = fly[j];
            }
            if (cmd & 2u) fly[j] = load16(s_at[j] + 16 * tid);
        }
        wave_sync();
    }
    if (live) {
        if (pos >= 0 && n > 0) {                                         // the last: no state update after it
            const uint32_t cl = tl[sl], co = tof[so], cm = tm[sm];
            dst[n - 1] = SeqRec{pos, sl | sm << 9 | so << 18};
            pos -= static_cast<int32_t>((cl >> 24) + (co >> 24) + (cm >> 24));
        }
        if (pos != 0) flag_error(status, kStSeqBadEnd, sb.blk);
    }
}

// The same chain for sections of up to about twice as many blocks: 2-byte cells and a 512-byte ring (3 KB of LDS per block
// instead of 7).  A cell's next_base is a multiple of 2^nb (it is (next << nb) - table size), so next_base >> nb and nb fit one
// ten-bit number with a leading one -- (1 << (9 - nb)) | (next_base >> nb): the position of the top bit says nb -- and the
// extra-bit count takes five more.  Decoding that costs five instructions per table and sequence (0.15 us per sequence
// against 0.12, and 0.23 out of L2); the ring's half is 256 bytes, a batch twenty sequences, the chunk in flight one dword per
// thread, so a wave takes up to sixteen blocks.
constexpr uint32_t kSeqRingS = 512;          // bytes of bitstream per block in LDS (+ a 16-byte guard)
constexpr uint32_t kSeqBatchS = 20;          // 20 x 89 bits + a dword pair < 240 bytes
constexpr uint32_t kSeqLdsLanesS = 16;       // most blocks per wave

__device__ inline uint32_t seq_cell_pack(uint32_t c) {     // c = next_base | nb << 16 | extra_bits << 24
    const uint32_t nb = (c >> 16) & 0xFFu, base = c & 0xFFFFu, ex = c >> 24;
    return ((1u << (9u - nb)) | (base >> nb)) | (ex << 10);
}

__global__ __launch_bounds__(64) void k_seq_states_lds16(const uint8_t *__restrict__ src, const SeqBlock *__restrict__ blocks,
                                                         uint32_t n_blocks, const SeqCell *__restrict__ cells, SeqRec *recs,
                                                         uint32_t lanes, uint32_t cells_cap, long long src_min, uint32_t *status) {
    HIP_DYNAMIC_SHARED(uint32_t, s_seq)                   // per block: cells_cap 2-byte cells, then (kSeqRingS + 16) / 4 ring dwords
    __shared__ uint64_t s_at[kSeqLdsLanesS];
    __shared__ uint32_t s_cmd[kSeqLdsLanesS], s_ro[kSeqLdsLanesS];
    if (status[0] != 0) return;
    const uint32_t tid = threadIdx.x;
    const uint32_t b0 = blockIdx.x * lanes;
    const uint32_t cell_dw = cells_cap / 2;               // (cells_cap is even)
    const uint32_t per = cell_dw + (kSeqRingS + 16) / 4;
    for (uint32_t j = 0; j < lanes && b0 + j < n_blocks; j++) {
        const SeqBlock bj = blocks[b0 + j];
        uint16_t *t = reinterpret_cast<uint16_t *>(s_seq + j * per);
        const uint32_t nl = 1u << bj.ll_al, no = 1u << bj.of_al, nm = 1u << bj.ml_al;
        for (uint32_t i = tid; i < nl; i += 64) t[i] = static_cast<uint16_t>(seq_cell_pack(*reinterpret_cast<const uint32_t *>(cells + bj.ll_tbl + i)));
        for (uint32_t i = tid; i < no; i += 64) t[nl + i] = static_cast<uint16_t>(seq_cell_pack(*reinterpret_cast<const uint32_t *>(cells + bj.of_tbl + i)));
        for (uint32_t i = tid; i < nm; i += 64) t[nl + no + i] = static_cast<uint16_t>(seq_cell_pack(*reinterpret_cast<const uint32_t *>(cells + bj.ml_tbl + i)));
    }
    const bool live = tid < lanes && b0 + tid < n_blocks;
    SeqBlock sb{};
    if (live) sb = blocks[b0 + tid];
    // ring geometry as in k_seq_states_lds, with halves of 256 bytes: a chunk is one dword per thread
    const uint64_t end_off = sb.bits_off + sb.bits_len;
    const uint64_t lo0 = ((end_off + 15) & ~uint64_t(15)) - kSeqRingS;
    uint64_t lo = lo0;
    if (live) s_at[tid] = lo0;
    wave_sync();
    uint32_t fly[kSeqLdsLanesS];
    auto load4 = [&](uint64_t at) -> uint32_t {
        uint32_t v = 0;
        if (static_cast<long long>(at) >= src_min) __builtin_memcpy(&v, src + at, 4);
        return v;
    };
#pragma unroll
    for (uint32_t j = 0; j < kSeqLdsLanesS; j++) {
        fly[j] = 0;
        if (j < lanes && b0 + j < n_blocks) {
            const uint64_t base = s_at[j];
            uint32_t *rj = s_seq + j * per + cell_dw;
            const uint32_t h0 = load4(base + 4 * tid), h1 = load4(base + kSeqRingS / 2 + 4 * tid);
            rj[tid] = h0;
            rj[kSeqRingS / 8 + tid] = h1;
            if (tid < 4) rj[kSeqRingS / 4 + tid] = h0;                       // the guard mirrors ring bytes 0..15
            fly[j] = load4(base - kSeqRingS / 2 + 4 * tid);
        }
    }
    wave_sync();
    const uint32_t *ring = s_seq + (tid < lanes ? tid : 0) * per + cell_dw;
    const uint32_t bit0 = static_cast<uint32_t>((sb.bits_off - lo0) & (kSeqRingS - 1)) * 8u;
    auto field = [&](uint32_t p_low) -> uint32_t {
        const uint32_t rb = (bit0 + p_low) & (kSeqRingS * 8 - 1);
        const uint32_t *w = ring + (rb >> 5);
        return __builtin_amdgcn_alignbit(w[1], w[0], rb & 31u);
    };
    int32_t pos = -1;
    uint32_t sl = 0, so = 0, sm = 0, n = 0;
    SeqRec *dst = recs + sb.seq_first;
    if (live) {
        const uint32_t lastb = src[end_off - 1];                         // host checked: non-zero
        pos = static_cast<int32_t>(sb.bits_len - 1) * 8 + (31 - __clz(static_cast<int>(lastb | 1u)));
        n = sb.n_seq;
        const int32_t p1 = pos - static_cast<int32_t>(sb.ll_al + sb.of_al + sb.ml_al);
        if (p1 >= 0) {
            const uint32_t f = field(static_cast<uint32_t>(p1));
            sl = __builtin_amdgcn_ubfe(f, sb.of_al + sb.ml_al, sb.ll_al);
            so = __builtin_amdgcn_ubfe(f, sb.ml_al, sb.of_al);
            sm = __builtin_amdgcn_ubfe(f, 0, sb.ml_al);
        }
        pos = p1;
    }
#ifndef NAFGPU_EMU
    __builtin_amdgcn_s_setprio(3);
#endif
    const uint16_t *tl = reinterpret_cast<const uint16_t *>(s_seq + (tid < lanes ? tid : 0) * per), *tof = tl + (1u << sb.ll_al),
                   *tm = tof + (1u << sb.of_al);
    // a packed cell -> {next_base, nb}; the extra bits are c >> 10
    auto unpack = [](uint32_t c, uint32_t *nb) -> uint32_t {
        const uint32_t code = c & 1023u;
        const uint32_t msb = 31u - static_cast<uint32_t>(__clz(static_cast<int>(code | 1u)));
        *nb = 9u - msb;
        return (code ^ (1u << msb)) << (9u - msb);
    };
    uint32_t i = 0;
    bool pending = live;
    for (;;) {
        const uint32_t stop = i + kSeqBatchS;
        for (; i < stop; i++) {
            const bool on = live && i + 1 < n && pos >= 0;
            if (!__any(on ? 1 : 0)) break;
            if (!on) continue;
            const uint32_t cl = tl[sl], co = tof[so], cm = tm[sm];
            dst[i] = SeqRec{pos, sl | sm << 9 | so << 18};
            uint32_t nbl, nbm, nbo;
            const uint32_t bl = unpack(cl, &nbl), bm = unpack(cm, &nbm), bo = unpack(co, &nbo);
            const int32_t p1 = pos - static_cast<int32_t>((cl >> 10) + (co >> 10) + (cm >> 10) + nbl + nbm + nbo);
            const uint32_t f = field(static_cast<uint32_t>(p1 < 0 ? 0 : p1));
            sl = bl + __builtin_amdgcn_ubfe(f, nbm + nbo, nbl);
            sm = bm + __builtin_amdgcn_ubfe(f, nbo, nbm);
            so = bo + __builtin_amdgcn_ubfe(f, 0, nbo);
            pos = p1;
        }
        const bool more = live && i + 1 < n && pos >= 0;
        if (!__any(more ? 1 : 0)) break;
        if (tid < kSeqLdsLanesS) s_cmd[tid] = 0;
        wave_sync();
        if (more) {
            const uint64_t cur = sb.bits_off + (static_cast<uint32_t>(pos > 0 ? pos - 1 : 0) >> 3);
            uint32_t cmd = 0;
            if (pending && static_cast<long long>(cur) + 16 <= static_cast<long long>(lo) + kSeqRingS / 2) {
                lo -= kSeqRingS / 2;
                s_ro[tid] = static_cast<uint32_t>((lo - lo0) & (kSeqRingS - 1));
                cmd = 1;
                pending = false;
            }
            if (!pending) {
                s_at[tid] = lo - kSeqRingS / 2;
                cmd |= 2;
                pending = true;
            }
            s_cmd[tid] = cmd;
        }
        wave_sync();
#pragma unroll
        for (uint32_t j = 0; j < kSeqLdsLanesS; j++) {
            const uint32_t cmd = s_cmd[j];                               // (uniform)
            if (cmd & 1u) {
                uint32_t *rj = s_seq + j * per + cell_dw;
                const uint32_t ro = s_ro[j];
                rj[ro / 4 + tid] = fly[j];
                if (ro == 0 && tid < 4) rj[kSeqRingS / 4 + tid] = fly[j];
            }
            if (cmd & 2u) fly[j] = load4(s_at[j] + 4 * tid);
        }
        wave_sync();
    }
    if (live) {
        if (pos >= 0 && n > 0) {                                         // the last: no state update after it
            const uint32_t cl = tl[sl], co = tof[so], cm = tm[sm];
            dst[n - 1] = SeqRec{pos, sl | sm << 9 | so << 18};
            pos -= static_cast<int32_t>((cl >> 10) + (co >> 10) + (cm >> 10));
        }
        if (pos != 0) flag_error(status, kStSeqBadEnd, sb.blk);
    }
}

constexpr uint32_t kSeqPerLane = 4;

__global__ __launch_bounds__(64) void k_seq_values(const uint8_t *__restrict__ src, const SeqBlock *__restrict__ blocks,
                                                   uint32_t n_blocks, const SeqCell *__restrict__ cells,
                                                   const SeqRec *__restrict__ recs, Seq *seqs, uint32_t *blk_size,
                                                   uint32_t *rep_final, uint32_t cells_cap, uint32_t *status) {
    HIP_DYNAMIC_SHARED(uint2, s_cells)          // the block's three tables (cells_cap cells), when it has sequences enough to pay for them
    __shared__ uint32_t s_sum[2][2][64];        // [ping-pong][ll, ml][lane]
    __shared__ uint32_t s_map[2][3][64];        // [ping-pong][slot][lane]
    // a tile's 256 records on their way out: a lane's four are 80 contiguous bytes, and stored from registers every store
    // instruction touches 64 different lines; through LDS the wave writes its 5 KB as twenty runs of 256 contiguous bytes
    __shared__ uint32_t s_stage[64 * kSeqPerLane * 5];
    static_assert(sizeof(Seq) == 20, "Seq records are staged as five dwords");
    if (status[0] != 0) return;                 // (k_seq_states flagged a stream: its records are not all there)
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    if (b >= n_blocks) return;
    const SeqBlock sb = blocks[b];
    const uint8_t *bits = src + sb.bits_off;
    const SeqCell *tll = cells + sb.ll_tbl, *tof = cells + sb.of_tbl, *tml = cells + sb.ml_tbl;
    // Three cells per sequence at random places of the tables: out of L2 that is 64 different sectors per load instruction
    // (a block of 30 k sequences asks for its 10 KB of tables 90 k times).  A block with at least as many sequences as its
    // tables have cells stages them in LDS first.
    const uint32_t nl = 1u << sb.ll_al, nof = 1u << sb.of_al, nml = 1u << sb.ml_al;
    const bool staged = cells_cap != 0 && nl + nof + nml <= cells_cap && sb.n_seq >= nl + nof + nml;
    if (staged) {
        for (uint32_t i = lane; i < nl; i += 64) s_cells[i] = *reinterpret_cast<const uint2 *>(tll + i);
        for (uint32_t i = lane; i < nof; i += 64) s_cells[nl + i] = *reinterpret_cast<const uint2 *>(tof + i);
        for (uint32_t i = lane; i < nml; i += 64) s_cells[nl + nof + i] = *reinterpret_cast<const uint2 *>(tml + i);
        wave_sync();
    }
    auto cell_at = [&](const SeqCell *tbl, uint32_t lds_base, uint32_t idx) -> SeqCell {
        SeqCell c;
        if (staged) {
            const uint2 v = s_cells[lds_base + idx];
            __builtin_memcpy(&c, &v, 8);
        } else {
            c = tbl[idx];
        }
        return c;
    };
    const SeqRec *rec = recs + sb.seq_first;
    Seq *dst = seqs + sb.seq_first;
    const uint32_t id0 = kRepToken | (0u << 24), id1 = kRepToken | (1u << 24), id2 = kRepToken | (2u << 24);
    uint32_t carry[3] = {id0, id1, id2};        // the history at the start of the tile, in terms of the block's
    uint32_t run_ll = 0, run_ml = 0;            // sums over the tiles before (<= kBlockMax + one tile: checked per tile)
    bool bad = false;
    for (uint32_t base = 0; base < sb.n_seq; base += 64 * kSeqPerLane) {
        const uint32_t i0 = base + kSeqPerLane * lane;
        uint32_t ll[kSeqPerLane], ml[kSeqPerLane], offs[kSeqPerLane];
        uint32_t r0 = id0, r1 = id1, r2 = id2, sum_ll = 0, sum_ml = 0;
        for (uint32_t j = 0; j < kSeqPerLane; j++) {
            ll[j] = ml[j] = 0;
            offs[j] = 0;
            if (i0 + j >= sb.n_seq) continue;
            const SeqRec q = rec[i0 + j];
            const SeqCell cl = cell_at(tll, 0, q.states & 511u), cm = cell_at(tml, nl + nof, (q.states >> 9) & 511u),
                          co = cell_at(tof, nl, q.states >> 18);
            SeqWindow r{bits, q.pos, 0, 0, 0};
            r.load();
            const uint32_t ofv = co.base_value + r.read(co.extra_bits);   // extra bits in the order OF, ML, LL
            ml[j] = cm.base_value + r.read(cm.extra_bits);
            ll[j] = cl.base_value + r.read(cl.extra_bits);
            sum_ll += ll[j];
            sum_ml += ml[j];
            // repeat-offset history (App. B "Repeat offsets"), on tokens: "what this lane's first sequence inherits"
            uint32_t o;
            if (ofv > 3) {
                o = ofv - 3;
                bad = bad || (o & kRepToken);                // >= 2^31: beyond any legal window
                r2 = r1;
                r1 = r0;
                r0 = o;
            } else {
                const uint32_t idx = ofv - 1 + (ll[j] == 0 ? 1u : 0u);
                if (idx == 0) {
                    o = r0;
                } else {
                    o = idx == 1 ? r1 : (idx == 2 ? r2 : rep_minus_one(r0));
                    bad = bad || o == 0;
                    if (idx > 1) r2 = r1;
                    r1 = r0;
                    r0 = o;
                }
            }
            offs[j] = o;
        }
        // inclusive scans over the lanes: the sums, and the maps "history before my first sequence -> history after my last".
        // A map without tokens no longer depends on what came before, so the map rounds stop as soon as every lane's is
        // either closed or complete (with mostly fresh offsets: after the first round or two).
        uint32_t m[3] = {r0, r1, r2}, in_ll = sum_ll, in_ml = sum_ml;
        uint32_t pp = 0;
        for (uint32_t d = 1; d < 64; d <<= 1, pp ^= 1u) {
            s_sum[pp][0][lane] = in_ll;
            s_sum[pp][1][lane] = in_ml;
            wave_sync();
            if (lane >= d) {
                in_ll += s_sum[pp][0][lane - d];
                in_ml += s_sum[pp][1][lane - d];
            }
        }
        s_sum[pp][0][lane] = in_ll;
        s_sum[pp][1][lane] = in_ml;
        uint32_t pm = 0;
        for (uint32_t d = 1; d < 64; d <<= 1) {
            const bool need = lane >= d && ((m[0] | m[1] | m[2]) & kRepToken) != 0;
            if (!__any(need ? 1 : 0)) break;
            s_map[pm][0][lane] = m[0];
            s_map[pm][1][lane] = m[1];
            s_map[pm][2][lane] = m[2];
            wave_sync();
            if (need) {
                const uint32_t f[3] = {s_map[pm][0][lane - d], s_map[pm][1][lane - d], s_map[pm][2][lane - d]};
                const uint32_t n0 = rep_apply_entry(m[0], f, &bad), n1 = rep_apply_entry(m[1], f, &bad),
                               n2 = rep_apply_entry(m[2], f, &bad);
                m[0] = n0;
                m[1] = n1;
                m[2] = n2;
            }
            pm ^= 1u;
        }
        s_map[pm][0][lane] = m[0];
        s_map[pm][1][lane] = m[1];
        s_map[pm][2][lane] = m[2];
        wave_sync();
        // what my first sequence inherits: the lanes before me applied to the tile's start
        uint32_t init[3] = {carry[0], carry[1], carry[2]};
        uint32_t pre_ll = run_ll, pre_ml = run_ml;
        if (lane) {
            const uint32_t e[3] = {s_map[pm][0][lane - 1], s_map[pm][1][lane - 1], s_map[pm][2][lane - 1]};
            init[0] = rep_apply_entry(e[0], carry, &bad);
            init[1] = rep_apply_entry(e[1], carry, &bad);
            init[2] = rep_apply_entry(e[2], carry, &bad);
            pre_ll += s_sum[pp][0][lane - 1];
            pre_ml += s_sum[pp][1][lane - 1];
        }
        for (uint32_t j = 0; j < kSeqPerLane; j++) {
            if (i0 + j >= sb.n_seq) break;
            uint32_t *st = s_stage + (kSeqPerLane * lane + j) * 5u;   // {ll, ml, off, opos, lpos}: struct Seq
            st[0] = ll[j];
            st[1] = ml[j];
            st[2] = rep_apply_entry(offs[j], init, &bad);
            st[3] = pre_ll + pre_ml;
            st[4] = pre_ll;
            pre_ll += ll[j];
            pre_ml += ml[j];
        }
        wave_sync();
        {
            const uint32_t n_here = sb.n_seq - base < 64u * kSeqPerLane ? sb.n_seq - base : 64u * kSeqPerLane;
            uint32_t *d32 = reinterpret_cast<uint32_t *>(dst + base);
#pragma unroll
            for (uint32_t k = 0; k < kSeqPerLane * 5u; k++) {
                const uint32_t idx = lane + 64u * k;
                if (idx < n_here * 5u) d32[idx] = s_stage[idx];
            }
        }
        // the tile's totals and its whole map, for the next tile
        {
            const uint32_t e[3] = {s_map[pm][0][63], s_map[pm][1][63], s_map[pm][2][63]};
            const uint32_t n0 = rep_apply_entry(e[0], carry, &bad), n1 = rep_apply_entry(e[1], carry, &bad),
                           n2 = rep_apply_entry(e[2], carry, &bad);
            carry[0] = n0;
            carry[1] = n1;
            carry[2] = n2;
            run_ll += s_sum[pp][0][63];
            run_ml += s_sum[pp][1][63];
        }
        wave_sync();                                      // (the next tile writes the other half first, then this one)
        if (run_ll > kBlockMax || run_ml > kBlockMax) break;   // invalid: flagged below; keeps the 32-bit sums exact
    }
    const bool any_bad = __any(bad ? 1 : 0) != 0;
    if (lane != 0) return;
    if (run_ll > sb.lit_size) {
        flag_error(status, kStSeqLiterals, sb.blk);
        return;
    }
    if (any_bad) {
        flag_error(status, kStBadOffset, sb.blk);
        return;
    }
    rep_final[3 * b + 0] = carry[0];
    rep_final[3 * b + 1] = carry[1];
    rep_final[3 * b + 2] = carry[2];
    if (sb.lit_size + run_ml > kBlockMax) {
        flag_error(status, kStSizeMismatch, sb.blk);
        return;
    }
    blk_size[sb.blk] = sb.lit_size + run_ml;
}

// ======================================================================================
// K3 / K6  tile scans
// ======================================================================================
// One generic three-pass scan over "items" that carry a value and a terminator bit:
//   exclusive mode : item = u32 block size          -> out[i] = sum of items before i
//   runs modes     : item = length word / mask byte -> ends[k] = inclusive sum at the k-th
//                    terminator (value != sentinel).  Record k therefore spans
//                    [ends[k-1], ends[k]) -- LengthReader (reader.rs:48-67) and MaskReader
//                    (reader.rs:198-231) folded into one prefix sum.
constexpr uint32_t kScanThreads = 256;
constexpr uint32_t kScanItems = 8;
constexpr uint32_t kScanTile = kScanThreads * kScanItems;

struct TileAgg {
    uint64_t sum;
    uint64_t cnt;
};

enum ScanMode {
    kModeExcl = 0,      // u32 items -> exclusive prefix sums
    kModeRunsU32 = 1,   // length words, 0xFFFFFFFF continues a run
    kModeRunsU8 = 2,    // mask bytes, 0xFF continues a run
    kModeNul = 3,       // text bytes: ends[k] = offset just past the k-th NUL (CStringReader, reader.rs:22-30); extra strings are dropped
    kModeExclU64 = 4,   // u64 items -> exclusive prefix sums
};

template <int MODE>
__device__ inline void scan_item(const uint8_t *in, uint64_t i, uint64_t n, uint64_t *v, uint32_t *t) {
    if (i >= n) {
        *v = 0;
        *t = 0;
        return;
    }
    if (MODE == kModeRunsU8) {
        const uint32_t b = in[i];
        *v = b;
        *t = b != 0xFFu;
    } else if (MODE == kModeNul) {
        *v = 1;
        *t = in[i] == 0;
    } else if (MODE == kModeExclU64) {
        *v = reinterpret_cast<const uint64_t *>(in)[i];
        *t = 1;
    } else {
        const uint8_t *q = in + 4 * i;      // length words are not guaranteed 4-byte aligned in memory
        const uint32_t w = static_cast<uint32_t>(q[0]) | (static_cast<uint32_t>(q[1]) << 8) |
                           (static_cast<uint32_t>(q[2]) << 16) | (static_cast<uint32_t>(q[3]) << 24);
        *v = w;
        *t = MODE == kModeExcl ? 1u : (w != 0xFFFFFFFFu);
    }
}

// block-wide exclusive scan of one (sum, cnt) pair per thread; returns the block total in *tot
__device__ inline void block_scan_pair(uint64_t &sum, uint32_t &cnt, TileAgg *tot, uint64_t *s_sum, uint32_t *s_cnt) {
    const uint32_t t = threadIdx.x;
    s_sum[t] = sum;
    s_cnt[t] = cnt;
    __syncthreads();
    for (uint32_t d = 1; d < kScanThreads; d <<= 1) {
        uint64_t a = 0;
        uint32_t c = 0;
        if (t >= d) {
            a = s_sum[t - d];
            c = s_cnt[t - d];
        }
        __syncthreads();
        s_sum[t] += a;
        s_cnt[t] += c;
        __syncthreads();
    }
    tot->sum = s_sum[kScanThreads - 1];
    tot->cnt = s_cnt[kScanThreads - 1];
    const uint64_t incl_s = s_sum[t];
    const uint32_t incl_c = s_cnt[t];
    __syncthreads();
    sum = incl_s - sum;
    cnt = incl_c - cnt;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_scan_reduce(const uint8_t *in, uint64_t n, TileAgg *tiles) {
    // (no early-out on the status word: these kernels are memory-safe on any input, and a
    //  multi-wave workgroup must reach its barriers with uniform control flow)
    __shared__ uint64_t s_sum[kScanThreads];
    __shared__ uint32_t s_cnt[kScanThreads];
    const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kScanTile + threadIdx.x * kScanItems;
    uint64_t sum = 0;
    uint32_t cnt = 0;
    for (uint32_t k = 0; k < kScanItems; k++) {
        uint64_t v;
        uint32_t t;
        scan_item<MODE>(in, base + k, n, &v, &t);
        sum += v;
        cnt += t;
    }
    TileAgg tot;
    block_scan_pair(sum, cnt, &tot, s_sum, s_cnt);
    if (threadIdx.x == 0) tiles[blockIdx.x] = tot;
}

// single workgroup: exclusive scan of the tile aggregates in place; grand total -> totals.  Every thread adds up a
// contiguous share of the tiles, one workgroup scan joins the shares, every thread rewrites its share (one pass of
// synchronisation whatever the number of tiles: 18 k tiles of a 37 MB Mask section took 0.43 ms tile row by tile row).
__global__ __launch_bounds__(256) void k_scan_tiles(TileAgg *tiles, uint64_t n_tiles, ScanTotals *totals) {
    __shared__ uint64_t s_sum[kScanThreads];
    __shared__ uint32_t s_cnt[kScanThreads];
    const uint64_t per = (n_tiles + kScanThreads - 1) / kScanThreads;
    const uint64_t lo = threadIdx.x * per < n_tiles ? threadIdx.x * per : n_tiles;
    const uint64_t hi = lo + per < n_tiles ? lo + per : n_tiles;
    uint64_t sum = 0, cnt64 = 0;
    for (uint64_t i = lo; i < hi; i++) {
        const TileAgg a = tiles[i];
        sum += a.sum;
        cnt64 += a.cnt;
    }
    // (a share may hold more than 2^32 terminators: the counts go through the 64-bit channel of a second scan)
    uint32_t z = 0;
    TileAgg tot;
    block_scan_pair(sum, z, &tot, s_sum, s_cnt);
    z = 0;
    block_scan_pair(cnt64, z, &tot, s_sum, s_cnt);
    uint64_t run_sum = sum, run_cnt = cnt64;
    for (uint64_t i = lo; i < hi; i++) {
        const TileAgg a = tiles[i];
        tiles[i] = TileAgg{run_sum, run_cnt};
        run_sum += a.sum;
        run_cnt += a.cnt;
    }
    if (threadIdx.x == kScanThreads - 1) {
        totals->sum = run_sum;
        totals->count = run_cnt;
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_scan_emit(const uint8_t *in, uint64_t n, const TileAgg *tiles, uint64_t *out,
                                                   uint64_t cap, uint32_t *status) {
    __shared__ uint64_t s_sum[kScanThreads];
    __shared__ uint32_t s_cnt[kScanThreads];
    const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kScanTile + threadIdx.x * kScanItems;
    uint64_t v[kScanItems];
    uint32_t t[kScanItems];
    uint64_t sum = 0;
    uint32_t cnt = 0;
#pragma unroll
    for (uint32_t k = 0; k < kScanItems; k++) {
        scan_item<MODE>(in, base + k, n, &v[k], &t[k]);
        sum += v[k];
        cnt += t[k];
    }
    TileAgg tot;
    block_scan_pair(sum, cnt, &tot, s_sum, s_cnt);
    const TileAgg tile = tiles[blockIdx.x];
    uint64_t run = tile.sum + sum;          // exclusive prefix of this thread's first item
    uint64_t idx = tile.cnt + cnt;
#pragma unroll
    for (uint32_t k = 0; k < kScanItems; k++) {
        if (base + k >= n) break;
        if (MODE == kModeExcl || MODE == kModeExclU64) {
            out[base + k] = run;
            run += v[k];
        } else {
            run += v[k];
            if (t[k]) {
                if (idx < cap)
                    out[idx] = run;
                else if (MODE != kModeNul)
                    flag_error(status, kStRunsOverflow, 0);
                idx++;
            }
        }
    }
}

__global__ void k_scan_finish_blocks(uint64_t *blk_base, uint64_t n, const ScanTotals *totals, uint64_t expect,
                                     uint32_t *status) {
    if (status[0] != 0) return;
    blk_base[n] = totals->sum;
    // More than the archive announces would not fit the output; LESS is not an error in the reference -- its zstd reader is
    // never told the size, the record that needs the missing bytes fails (Io(UnexpectedEof)) -- and so not here: the host
    // reads the total back (SectionJob::check).  (~0: a tile of a section with sequences -- the host adds the tiles up)
    if (expect != ~0ull && totals->sum > expect) flag_error(status, kStSizeMismatch, 0xFFFFFFFFu);
}

// ======================================================================================
// UTF-8 validation (into_string().expect at mod.rs:362,368; std::str::from_utf8 at reader.rs:108-109)
// ======================================================================================
// Byte-parallel: byte i must be a continuation byte exactly when one of the three bytes before it is
// a lead byte that reaches it; lead bytes C0, C1, F5..FF are invalid; the byte after E0 / ED / F0 / F4
// has a narrower range (no overlong forms, no surrogates, nothing above U+10FFFF); a sequence must
// not run past the end.  Sets bit `bit` of *flags if the buffer is not valid UTF-8.
__device__ inline uint32_t utf8_lead_len(uint32_t b) {     // bytes a lead byte announces (0: not a lead byte)
    if (b < 0x80u) return 1;
    if (b < 0xC0u) return 0;
    if (b < 0xE0u) return 2;
    if (b < 0xF0u) return 3;
    return 4;
}

__device__ inline bool utf8_bad_at(const uint8_t *__restrict__ p, uint64_t n, uint64_t i) {   // the rules above for byte i
    const uint32_t b = p[i];
    const uint32_t b1 = i >= 1 ? p[i - 1] : 0u, b2 = i >= 2 ? p[i - 2] : 0u, b3 = i >= 3 ? p[i - 3] : 0u;
    const bool need_cont = utf8_lead_len(b1) >= 2 || utf8_lead_len(b2) >= 3 || (b3 >= 0xF0u && utf8_lead_len(b3) == 4);
    const bool is_cont = (b & 0xC0u) == 0x80u;
    bool bad = need_cont != is_cont;
    bad = bad || b == 0xC0u || b == 0xC1u || b >= 0xF5u;
    bad = bad || (b1 == 0xE0u && b < 0xA0u);                // overlong 3-byte form
    bad = bad || (b1 == 0xEDu && b > 0x9Fu);                // surrogates
    bad = bad || (b1 == 0xF0u && b < 0x90u);                // overlong 4-byte form
    bad = bad || (b1 == 0xF4u && b > 0x8Fu);                // above U+10FFFF
    bad = bad || (b >= 0xC0u && i + utf8_lead_len(b) > n);  // truncated at the end
    return bad;
}

__global__ __launch_bounds__(256) void k_utf8_check(const uint8_t *__restrict__ p, uint64_t n, uint32_t *flags, uint32_t bit) {
    // Sixteen bytes per thread.  Names and quality strings are ASCII almost everywhere: a chunk whose bytes -- and the
    // four in front of it -- all have bit 7 clear cannot break a rule, and costs two loads and five tests (byte by byte
    // the quality section of 10 M reads took 2.8 ms); any other chunk is checked byte by byte.
    bool bad = false;
    const uint64_t head = (16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15;      // bytes in front of the first aligned chunk
    const uint64_t first = head < n ? head : n;
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (uint64_t i = 0; i < first; i++) bad = bad || utf8_bad_at(p, n, i);
    const uint64_t n_chunks = (n - first) / 16;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * 256;
    for (uint64_t c = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x; c < n_chunks; c += stride) {
        const uint64_t i0 = first + 16 * c;
        const uint4 w = *reinterpret_cast<const uint4 *>(p + i0);
        uint32_t prev = 0;
        if (i0 >= 4) __builtin_memcpy(&prev, p + i0 - 4, 4);
        if (((w.x | w.y | w.z | w.w | prev) & 0x80808080u) == 0) continue;
        for (uint64_t i = i0; i < i0 + 16; i++) bad = bad || utf8_bad_at(p, n, i);
    }
    if (blockIdx.x == 0 && threadIdx.x == 1)                                      // the bytes behind the last whole chunk
        for (uint64_t i = first + 16 * n_chunks; i < n; i++) bad = bad || utf8_bad_at(p, n, i);
    if (bad) atomicOr(flags, 1u << bit);
}

// ======================================================================================
// FASTA / FASTQ text from the decoded buffers (what every consumer of the iterator does
// next, cf. unnaf): record k becomes
//   FASTA  '>' id [sep comment] '\n'  then the sequence in lines of `line_length` characters
//   FASTQ  '@' id [sep comment] '\n' sequence '\n' '+' '\n' quality '\n'
// (the comment, with its separator, only when it is not empty).  k_fmt_sizes computes every record's
// text size, an exclusive scan turns sizes into offsets, k_fmt_write fills the text.
// ======================================================================================
struct FmtArgs {
    const uint8_t *seq;          // ASCII, mask applied
    const uint8_t *qual;         // null: FASTA
    const uint64_t *rec_end;     // inclusive prefix sums of the record lengths
    const uint8_t *ids;          // null: no names
    const uint64_t *id_end;      // offset just past the k-th NUL
    uint64_t n_ids;
    const uint8_t *com;
    const uint64_t *com_end;
    uint64_t n_com;
    uint64_t n_rec;
    uint64_t line_length;        // 0: one line per sequence
    uint32_t sep;
    uint32_t pad;
};

struct FmtRec {                  // where the pieces of one record are
    uint64_t id0, id_len, com0, com_len, s0, n, hdr, body;
};

__device__ inline FmtRec fmt_record(const FmtArgs &a, uint64_t k) {
    FmtRec r;
    r.id0 = r.id_len = r.com0 = r.com_len = 0;
    if (a.ids && k < a.n_ids) {
        r.id0 = k ? a.id_end[k - 1] : 0;
        r.id_len = a.id_end[k] - r.id0 - 1;
    }
    if (a.com && k < a.n_com) {
        r.com0 = k ? a.com_end[k - 1] : 0;
        r.com_len = a.com_end[k] - r.com0 - 1;
    }
    r.s0 = k ? a.rec_end[k - 1] : 0;
    r.n = a.rec_end[k] - r.s0;
    r.hdr = 1 + r.id_len + (r.com_len ? 1 + r.com_len : 0) + 1;
    if (a.qual)
        r.body = 2 * r.n + 4;
    else if (r.n == 0)
        r.body = 0;
    else
        r.body = r.n + (a.line_length ? (r.n + a.line_length - 1) / a.line_length : 1);
    return r;
}

__global__ __launch_bounds__(256) void k_fmt_sizes(FmtArgs a, uint64_t *sizes) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * 256;
    for (uint64_t k = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x; k < a.n_rec; k += stride) {
        const FmtRec r = fmt_record(a, k);
        sizes[k] = r.hdr + r.body;
    }
}

constexpr uint32_t kFmtChunk = 16u << 10;    // text bytes per workgroup step

__device__ inline uint4 load16u(const uint8_t *p) {       // one global_load_dwordx4 at any alignment
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}

// one record's share [t0, t1) of a chunk, written by `nthr` threads (this one is number `me`):
// 16-byte aligned granules as one uint4 store each -- a plain unaligned 16-byte copy inside a
// sequence / quality line, two such loads merged around the line feed where a FASTA line ends --
// and single characters for the header, the record's last line and the unaligned edges.
__device__ inline void fmt_record_part(const FmtArgs &a, const FmtRec &r, uint8_t *dst, uint64_t t0, uint64_t t1, uint32_t me,
                                       uint32_t nthr) {
    const uint64_t size = r.hdr + r.body;
    const uint64_t L = a.line_length ? a.line_length : r.n;        // characters per full line
    const uint64_t Lp = L + 1;
    auto char_at = [&](uint64_t t) -> uint32_t {
        if (t < r.hdr) {
            if (t == 0) return a.qual ? '@' : '>';
            if (t <= r.id_len) return a.ids[r.id0 + t - 1];
            if (t == r.hdr - 1) return '\n';
            if (t == r.id_len + 1) return a.sep;
            return a.com[r.com0 + t - r.id_len - 2];
        }
        const uint64_t q = t - r.hdr;
        if (a.qual) {
            if (q < r.n) return a.seq[r.s0 + q];
            if (q == r.n) return '\n';
            if (q == r.n + 1) return '+';
            if (q == r.n + 2) return '\n';
            if (q < 2 * r.n + 3) return a.qual[r.s0 + q - r.n - 3];
            return '\n';
        }
        const uint64_t line = q / Lp, col = q - line * Lp;
        if (col == L || q == r.body - 1) return '\n';
        return a.seq[r.s0 + line * L + col];
    };
    const uint64_t mis = (16 - (reinterpret_cast<uintptr_t>(dst + t0) & 15)) & 15;
    const uint64_t a0 = t0 + (mis < t1 - t0 ? mis : t1 - t0);
    const uint64_t n_g = (t1 - a0) >> 4, tail0 = a0 + 16 * n_g;
    for (uint64_t x = t0 + me; x < a0; x += nthr) dst[x] = static_cast<uint8_t>(char_at(x));
    for (uint64_t x = tail0 + me; x < t1; x += nthr) dst[x] = static_cast<uint8_t>(char_at(x));
    // line / column of the first body granule, once; the granules then work with 32-bit offsets from it
    const uint64_t g_body = a0 >= r.hdr ? 0 : (r.hdr - a0 + 15) >> 4;     // first granule that starts inside the body
    const uint64_t q0 = a0 + 16 * g_body - r.hdr;
    const bool fasta_fast = !a.qual && L >= 16 && Lp < (1ull << 31);
    const uint64_t line0 = fasta_fast ? q0 / Lp : 0;
    const uint32_t col0 = fasta_fast ? static_cast<uint32_t>(q0 - line0 * Lp) : 0, Lp32 = static_cast<uint32_t>(Lp);
    for (uint64_t g = me; g < n_g; g += nthr) {
        const uint64_t t = a0 + 16 * g;
        uint4 w;
        bool done = false;
        if (g >= g_body && t + 16 <= size) {
            const uint64_t q = t - r.hdr;
            if (a.qual) {
                if (q + 16 <= r.n) {
                    w = load16u(a.seq + r.s0 + q);
                    done = true;
                } else if (q >= r.n + 3 && q + 16 <= 2 * r.n + 3) {
                    w = load16u(a.qual + r.s0 + q - r.n - 3);
                    done = true;
                }
            } else if (fasta_fast && q + 16 <= r.body - 1) {     // at most one line feed inside, and not the record's last
                const uint32_t cq = col0 + static_cast<uint32_t>(16 * (g - g_body));
                const uint32_t dl = cq / Lp32, col = cq - dl * Lp32;
                const uint8_t *src = a.seq + r.s0 + (line0 + dl) * L + col;
                const uint32_t k = static_cast<uint32_t>(L) - col;       // the line feed is byte k of this granule (if k < 16)
                if (k >= 16) {
                    w = load16u(src);
                } else {
                    const uint4 p0 = load16u(src), p1 = load16u(src - 1);  // bytes after the line feed come from one byte earlier
                    const uint32_t a0w[4] = {p0.x, p0.y, p0.z, p0.w}, a1w[4] = {p1.x, p1.y, p1.z, p1.w};
                    uint32_t o[4];
#pragma unroll
                    for (uint32_t d = 0; d < 4; d++) {
                        const uint32_t below = k <= 4 * d ? 0u : (k >= 4 * d + 4 ? 0xFFFFFFFFu : (1u << (8 * (k - 4 * d))) - 1u);
                        const uint32_t at = (k >= 4 * d && k < 4 * d + 4) ? 0xFFu << (8 * (k - 4 * d)) : 0u;
                        o[d] = (a0w[d] & below) | (0x0A0A0A0Au & at) | (a1w[d] & ~(below | at));
                    }
                    w = make_uint4(o[0], o[1], o[2], o[3]);
                }
                done = true;
            }
        }
        if (!done) {
            uint32_t o[4] = {0, 0, 0, 0};
            for (uint32_t i = 0; i < 16; i++) o[i >> 2] |= char_at(t + i) << (8 * (i & 3));
            w = make_uint4(o[0], o[1], o[2], o[3]);
        }
        *reinterpret_cast<uint4 *>(dst + t) = w;
    }
}

__global__ __launch_bounds__(256) void k_fmt_write(FmtArgs a, const uint64_t *__restrict__ off, uint64_t n_text, uint8_t *text) {
    // A workgroup takes 16 KiB of text at a time: one binary search for the record its first byte
    // belongs to.  A chunk inside one long record is written by the whole workgroup; a chunk holding
    // many short records (reads) is shared out record by record over the four waves.
    const uint32_t tid = threadIdx.x;
    const uint64_t n_chunks = (n_text + kFmtChunk - 1) / kFmtChunk;
    for (uint64_t ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
        const uint64_t c0 = ch * kFmtChunk, c1 = c0 + kFmtChunk < n_text ? c0 + kFmtChunk : n_text;
        uint64_t lo = 0, hi = a.n_rec;                     // last record with off[k] <= c0
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (off[mid] <= c0)
                lo = mid;
            else
                hi = mid;
        }
        const bool one = lo + 1 >= a.n_rec || off[lo + 1] >= c1;      // uniform
        const uint32_t me = one ? tid : tid & 63, nthr = one ? 256 : 64;
        for (uint64_t k = lo + (one ? 0 : tid >> 6); k < a.n_rec && off[k] < c1; k += one ? 1 : 4) {
            const FmtRec r = fmt_record(a, k);
            const uint64_t base = off[k], size = r.hdr + r.body;
            const uint64_t t0 = c0 > base ? c0 - base : 0, t1 = c1 - base < size ? c1 - base : size;
            if (t1 > t0) fmt_record_part(a, r, text + base, t0, t1, me, nthr);
        }
    }
}

// ======================================================================================
// Raw / RLE blocks and literal sections
// ======================================================================================
constexpr uint32_t kCopySlice = 8u << 10;

template <bool ASCII>
__global__ __launch_bounds__(256) void k_copy_fill(const uint8_t *__restrict__ src, const CopyTask *__restrict__ tasks,
                                                   const uint64_t *__restrict__ blk_base, uint8_t *out, uint8_t *lit,
                                                   uint32_t t_char, const uint32_t *status) {
    if (status[0] != 0) return;
    const CopyTask t = tasks[blockIdx.x];
    const bool fill = (t.flags & 2) != 0;
    const uint8_t fv = static_cast<uint8_t>(t.src_off);
    const uint8_t *s = src + (fill ? 0 : t.src_off);
    // a task (<= 128 KiB) is cut into kCopySlice-byte slices, one workgroup each (blockIdx.y)
    const uint32_t lo = blockIdx.y * kCopySlice;
    const uint32_t hi = lo + kCopySlice < t.len ? lo + kCopySlice : t.len;
    if (ASCII && !(t.flags & 1)) {                       // nucleotide section: expand while copying
        uint16_t *d = reinterpret_cast<uint16_t *>(out) + blk_base[t.blk] + t.dst;
        for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x)
            d[i] = static_cast<uint16_t>(byte_chars(fill ? fv : s[i], t_char));
    } else {
        uint8_t *d = (t.flags & 1) ? lit + t.dst : out + blk_base[t.blk] + t.dst;
        for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) d[i] = fill ? fv : s[i];
    }
}

// nibble -> character through v_perm_b32 (shared by K4's literal copy and K5): four packed bytes -> eight characters
__device__ inline uint32_t lut4(uint32_t nib, uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3) {
    const uint32_t sel = nib & 0x07070707u;
    const uint32_t lo = __builtin_amdgcn_perm(t1, t0, sel);
    const uint32_t hi = __builtin_amdgcn_perm(t3, t2, sel);
    const uint32_t m = ((nib >> 3) & 0x01010101u) * 0xFFu;
    return (hi & m) | (lo & ~m);
}

__device__ inline void unpack_dword(uint32_t w, uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3, uint32_t *o0,
                                    uint32_t *o1) {
    const uint32_t L = lut4(w & 0x0F0F0F0Fu, t0, t1, t2, t3);
    const uint32_t H = lut4((w >> 4) & 0x0F0F0F0Fu, t0, t1, t2, t3);
    *o0 = __builtin_amdgcn_perm(H, L, 0x05010400u);   // L0 H0 L1 H1
    *o1 = __builtin_amdgcn_perm(H, L, 0x07030602u);   // L2 H2 L3 H3
}

// ======================================================================================
// K4  LZ77 execution (App. B "Repeat offsets" + "Execute")
// ======================================================================================
//   k_rep_partial / k_rep_scan / k_rep_apply   block b's initial repeat offsets from its predecessors' final tokens
//   k_lz_index       first sequence at or after every 128th output element (starting point of lz_range_final)
//   k_lz_literals    every literal run to its final position; no dependencies; workgroup per block
//   k_lz_match_pass  run several times: every pending match whose source bytes are all final
//                    (literals, or matches completed in an EARLIER pass) is copied; matches are
//                    independent of block boundaries, so all blocks progress together and the
//                    number of passes is the depth of the match-on-match dependency chain
//   k_lz_matches_ordered   whatever is still pending after the fixed number of passes (long chains,
//                    e.g. tandem repeats): one workgroup, frame order -- always terminates
// ASCII = true: the output is the expanded base stream (two characters per packed byte), so a
// match of `ml` packed bytes at distance `off` copies ml 16-bit elements at distance off, and
// literals are expanded while they are scattered.  The ASCII buffer itself is the LZ window.
// the section's counter words (SectionJob::d_counters_, 32 x u64): [0] matches still pending (sparse), [1] matches the launched
// passes left to the one-workgroup stage, [4..6] rotating counters of the stage in use, bytes 64..75 the repeat offsets behind
// the last block, [10..13] the pending lists of the sweeps, and for the shard protocol:
constexpr uint32_t kCtrLeft = 16, kCtrWhich = 17, kCtrPass = 18;   // sparse: survivors of the last pass -- how many, in which list, the next pass number
constexpr uint32_t kCtrTail = 19;                                  // 1: something still pending lies in the tail the next shard waits for
constexpr uint32_t kLzShort = 48;        // runs up to this many elements are copied by their own thread
constexpr uint32_t kLzFewSequences = 8192;   // sections with no more sequences than this: two launched passes, then k_lz_finish_small
constexpr uint32_t kLzPasses = 24;       // launched passes: the first walks every block, the others the list of what is still pending;
                                         // what they leave goes to ONE workgroup (k_lz_finish_small), then to the frame-order walk

__device__ inline uint32_t rep_resolve(uint32_t tok, const uint32_t *init, bool *bad) {
    if (!(tok & kRepToken)) return tok;
    const uint32_t base = init[(tok >> 24) & 3u], d = tok & 0xFFFFFFu;
    if (base <= d) {
        *bad = true;
        return 1;
    }
    return base - d;
}

// A block maps the three repeat offsets it inherits to the three it leaves behind; k_seq_values
// recorded that map symbolically: each outgoing offset is a value, or "incoming rep[slot] - d".
// Such maps compose (rep_resolve is "apply entry `tok` after the map `f`", whether f is concrete or
// symbolic), so the chain over all blocks is done in three steps instead of one serial walk:
//   k_rep_partial  thread t composes the maps of blocks [C t, C t + C)          (C = kRepChunk)
//   k_rep_scan     one thread applies the chunk maps in order: the triple each chunk starts from
//   k_rep_apply    thread t walks its C blocks from that triple and writes every block's initial triple
// Repeat offsets restart at {1, 4, 8} with each frame.
constexpr uint32_t kRepChunk = 64;

// (continues: the first block goes on with a frame begun in front of this tile -- its repeat offsets are the
//  ones carried over, not {1, 4, 8})
__global__ __launch_bounds__(256) void k_rep_partial(const SeqBlock *__restrict__ blocks, uint32_t n_blocks,
                                                     const uint32_t *__restrict__ rep_final, uint32_t *partial, uint32_t continues,
                                                     uint32_t *status) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    const uint32_t b0 = t * kRepChunk;
    if (status[0] != 0 || b0 >= n_blocks) return;
    const uint32_t b1 = b0 + kRepChunk < n_blocks ? b0 + kRepChunk : n_blocks;
    uint32_t cur[3] = {kRepToken | (0u << 24), kRepToken | (1u << 24), kRepToken | (2u << 24)};   // identity
    uint32_t frame = b0 ? blocks[b0 - 1].frame_first_blk : (continues ? blocks[0].frame_first_blk : 0xFFFFFFFFu);
    bool bad = false;
    for (uint32_t b = b0; b < b1; b++) {
        if (blocks[b].frame_first_blk != frame) {
            frame = blocks[b].frame_first_blk;
            cur[0] = 1;
            cur[1] = 4;
            cur[2] = 8;
        }
        const uint32_t n0 = rep_apply_entry(rep_final[3 * b + 0], cur, &bad);
        const uint32_t n1 = rep_apply_entry(rep_final[3 * b + 1], cur, &bad);
        const uint32_t n2 = rep_apply_entry(rep_final[3 * b + 2], cur, &bad);
        cur[0] = n0;
        cur[1] = n1;
        cur[2] = n2;
    }
    partial[3 * t + 0] = cur[0];
    partial[3 * t + 1] = cur[1];
    partial[3 * t + 2] = cur[2];
    if (bad) flag_error(status, kStBadOffset, 0xFFFFFFFEu);
}

// (final_out: where the map of all chunks together goes -- with tokens for c0..c2 that is the map of the whole range,
//  which a shard hands to the ranks behind it before anybody knows a concrete triple)
__global__ void k_rep_scan(uint32_t n_chunks, const uint32_t *__restrict__ partial, uint32_t *chunk_init, uint32_t c0, uint32_t c1,
                           uint32_t c2, uint32_t *final_out, uint32_t *status) {
    if (status[0] != 0 || blockIdx.x != 0 || threadIdx.x != 0) return;
    uint32_t cur[3] = {c0, c1, c2};                        // {1, 4, 8}, or what the tile in front left behind
    bool bad = false;
    for (uint32_t t = 0; t < n_chunks; t++) {
        chunk_init[3 * t + 0] = cur[0];
        chunk_init[3 * t + 1] = cur[1];
        chunk_init[3 * t + 2] = cur[2];
        const uint32_t n0 = rep_apply_entry(partial[3 * t + 0], cur, &bad);
        const uint32_t n1 = rep_apply_entry(partial[3 * t + 1], cur, &bad);
        const uint32_t n2 = rep_apply_entry(partial[3 * t + 2], cur, &bad);
        cur[0] = n0;
        cur[1] = n1;
        cur[2] = n2;
    }
    if (final_out) {
        final_out[0] = cur[0];
        final_out[1] = cur[1];
        final_out[2] = cur[2];
    }
    if (bad) flag_error(status, kStBadOffset, 0xFFFFFFFEu);
}

__global__ __launch_bounds__(256) void k_rep_apply(const SeqBlock *__restrict__ blocks, uint32_t n_blocks,
                                                   const uint32_t *__restrict__ rep_final, const uint32_t *__restrict__ chunk_init,
                                                   uint32_t *rep_init, uint32_t continues, uint32_t *rep_out, uint32_t *status) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    const uint32_t b0 = t * kRepChunk;
    if (status[0] != 0 || b0 >= n_blocks) return;
    const uint32_t b1 = b0 + kRepChunk < n_blocks ? b0 + kRepChunk : n_blocks;
    uint32_t cur[3] = {chunk_init[3 * t], chunk_init[3 * t + 1], chunk_init[3 * t + 2]};
    uint32_t frame = b0 ? blocks[b0 - 1].frame_first_blk : (continues ? blocks[0].frame_first_blk : 0xFFFFFFFFu);
    bool bad = false;
    for (uint32_t b = b0; b < b1; b++) {
        if (blocks[b].frame_first_blk != frame) {        // repeat offsets restart with each frame
            frame = blocks[b].frame_first_blk;
            cur[0] = 1;
            cur[1] = 4;
            cur[2] = 8;
        }
        rep_init[3 * b + 0] = cur[0];
        rep_init[3 * b + 1] = cur[1];
        rep_init[3 * b + 2] = cur[2];
        const uint32_t n0 = rep_resolve(rep_final[3 * b + 0], cur, &bad);
        const uint32_t n1 = rep_resolve(rep_final[3 * b + 1], cur, &bad);
        const uint32_t n2 = rep_resolve(rep_final[3 * b + 2], cur, &bad);
        cur[0] = n0;
        cur[1] = n1;
        cur[2] = n2;
    }
    if (b1 == n_blocks) {                                  // what the next tile starts from
        rep_out[0] = cur[0];
        rep_out[1] = cur[1];
        rep_out[2] = cur[2];
    }
    if (bad) flag_error(status, kStBadOffset, 0xFFFFFFFEu);
}

// A long literal run (n packed bytes -> n output elements) copied by the whole workgroup: 16-byte aligned
// stores -- eight source bytes expanded to sixteen characters, or sixteen bytes copied -- between a
// scalar head and tail.  Blocks with only a few sequences (what real genomes give zstd level 1) are
// almost all trailing literals, so this is a streaming copy.
constexpr uint32_t kLzWide = 512;        // runs at least this long take the wide path

template <bool ASCII>
__device__ inline void lz_copy_wide(typename std::conditional<ASCII, uint16_t, uint8_t>::type *dst, const uint8_t *src, uint32_t n,
                                    uint32_t me, uint32_t t_char) {
    using Elem = typename std::conditional<ASCII, uint16_t, uint8_t>::type;
    constexpr uint32_t kPer = ASCII ? 8 : 16;              // elements per 16-byte store
    const uint32_t t0 = 0x4B47002Du | (t_char << 8), t1 = 0x42535943u, t2 = 0x44525741u, t3 = 0x4E56484Du;   // "-TGKCYSBAWRDMHVN"
    const uint32_t mis = static_cast<uint32_t>((16 - (reinterpret_cast<uintptr_t>(dst) & 15)) & 15) / sizeof(Elem);
    const uint32_t head = mis < n ? mis : n;
    const uint32_t n_vec = (n - head) / kPer, tail0 = head + n_vec * kPer;
    auto one = [&](uint32_t k) { dst[k] = ASCII ? static_cast<Elem>(byte_chars(src[k], t_char)) : static_cast<Elem>(src[k]); };
    if (me < head) one(me);
    if (tail0 + me < n) one(tail0 + me);
    for (uint32_t v = me; v < n_vec; v += 256) {
        const uint8_t *sp = src + head + v * kPer;
        uint4 w;
        if (ASCII) {
            uint2 in;
            __builtin_memcpy(&in, sp, 8);                  // one unaligned 8-byte load
            unpack_dword(in.x, t0, t1, t2, t3, &w.x, &w.y);
            unpack_dword(in.y, t0, t1, t2, t3, &w.z, &w.w);
        } else {
            __builtin_memcpy(&w, sp, 16);
        }
        *reinterpret_cast<uint4 *>(dst + head + v * kPer) = w;
    }
}

template <bool ASCII>
__global__ __launch_bounds__(256) void k_lz_literals(const SeqBlock *__restrict__ blocks, const Seq *__restrict__ seqs,
                                                     const uint8_t *__restrict__ lit, const uint64_t *__restrict__ blk_base,
                                                     SeqMeta *meta, uint32_t *blk_pending, uint8_t *out_bytes, uint32_t t_char,
                                                     const uint32_t *status) {
    using Elem = typename std::conditional<ASCII, uint16_t, uint8_t>::type;
    __shared__ uint32_t s_long[256];
    __shared__ uint32_t s_nlong, s_abort;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) s_abort = status[0];
    __syncthreads();
    if (s_abort) return;
    const SeqBlock sb = blocks[blockIdx.x];
    const uint64_t obase = blk_base[sb.blk];
    Elem *out = reinterpret_cast<Elem *>(out_bytes) + obase;
    const uint8_t *blit = lit + sb.lit_off;
    const Seq *sq = seqs + sb.seq_first;
    if (tid == 0) blk_pending[blockIdx.x] = sb.n_seq;
    auto put = [&](Elem *d, uint8_t b) { *d = ASCII ? static_cast<Elem>(byte_chars(b, t_char)) : static_cast<Elem>(b); };
    if (sb.direct) {                                       // k_huf_decode put the literals in place: only the match table is left to do
        if (!meta) return;
        for (uint32_t s = tid; s < sb.n_seq; s += 256) {
            const Seq q = sq[s];
            if (meta) meta[sb.seq_first + s] = SeqMeta{obase + q.opos + q.ll, q.ml, 0u};
        }
        return;
    }
    for (uint32_t s0 = 0; s0 < sb.n_seq; s0 += 256) {
        if (tid == 0) s_nlong = 0;
        __syncthreads();
        if (s0 + tid < sb.n_seq) {
            const Seq q = sq[s0 + tid];
            if (meta) meta[sb.seq_first + s0 + tid] = SeqMeta{obase + q.opos + q.ll, q.ml, 0u};   // where the match of this sequence starts; pending (swept sections keep no such records)
            if (q.ll <= kLzShort) {
                for (uint32_t k = 0; k < q.ll; k++) put(out + q.opos + k, blit[q.lpos + k]);
            } else {
                s_long[atomicAdd(&s_nlong, 1u)] = s0 + tid;
            }
        }
        __syncthreads();
        const uint32_t nl = s_nlong;
        for (uint32_t j = 0; j < nl; j++) {
            const Seq q = sq[s_long[j]];
            if (q.ll >= kLzWide)
                lz_copy_wide<ASCII>(out + q.opos, blit + q.lpos, q.ll, tid, t_char);
            else
                for (uint32_t k = tid; k < q.ll; k += 256) put(out + q.opos + k, blit[q.lpos + k]);
        }
        __syncthreads();
    }
    // literals after the last sequence run to the end of the block
    const Seq last = sq[sb.n_seq - 1];
    const uint32_t lused = last.lpos + last.ll, oend = last.opos + last.ll + last.ml;
    if (sb.lit_size - lused >= kLzWide)
        lz_copy_wide<ASCII>(out + oend, blit + lused, sb.lit_size - lused, tid, t_char);
    else
        for (uint32_t k = tid; k < sb.lit_size - lused; k += 256) put(out + oend + k, blit[lused + k]);
}

// One match copied by `nthr` threads (this one is number `me`): d[k] = d[k - off] for k < ml.  A match that does not
// reach into itself (off >= ml) is a plain copy: 16-byte aligned stores from unaligned 16-byte loads between a scalar
// head and tail; one that does is periodic with period off (a run, in quality strings and homopolymers).
template <class Elem>
__device__ inline void lz_copy_match(Elem *d, uint32_t ml, uint32_t off, uint32_t me, uint32_t nthr) {
    const Elem *s = d - off;
    constexpr uint32_t kPer = 16 / sizeof(Elem);
    if (off >= ml && ml >= 4 * kPer) {
        const uint32_t mis = static_cast<uint32_t>((16 - (reinterpret_cast<uintptr_t>(d) & 15)) & 15) / sizeof(Elem);
        const uint32_t head = mis < ml ? mis : ml;
        const uint32_t n_vec = (ml - head) / kPer, tail0 = head + n_vec * kPer;
        for (uint32_t k = me; k < head; k += nthr) d[k] = s[k];
        for (uint32_t k = tail0 + me; k < ml; k += nthr) d[k] = s[k];
        for (uint32_t v = me; v < n_vec; v += nthr) {
            uint4 w;
            __builtin_memcpy(&w, s + head + v * kPer, 16);
            *reinterpret_cast<uint4 *>(d + head + v * kPer) = w;
        }
    } else if (off >= ml) {
        for (uint32_t k = me; k < ml; k += nthr) d[k] = s[k];
    } else {
        for (uint32_t k = me; k < ml; k += nthr) d[k] = s[k % off];
    }
}

// cidx[c] = first sequence whose match starts at or after output element c << kLzIdxShift: where the
// search for "who wrote this source range" starts (one load instead of a binary search over all
// sequences per pending match and pass)
constexpr uint32_t kLzIdxShift = 7;

__global__ __launch_bounds__(256) void k_lz_index(const SeqMeta *__restrict__ meta, uint64_t n_seq, uint64_t n_chunks, uint32_t *cidx,
                                                  const uint32_t *status) {
    if (status[0] != 0) return;
    for (uint64_t c = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x; c < n_chunks; c += static_cast<uint64_t>(gridDim.x) * 256) {
        const uint64_t target = c << kLzIdxShift;
        uint64_t a = 0, b = n_seq;                        // first g with meta[g].pos >= target
        while (a < b) {
            const uint64_t mid = (a + b) >> 1;
            if (meta[mid].pos < target)
                a = mid + 1;
            else
                b = mid;
        }
        cidx[c] = static_cast<uint32_t>(a);
    }
}

// Are all output bytes in [lo, hi) final for a reader in pass `pass`?  Bytes that no match writes
// are literals (final since k_lz_literals / K1); bytes of match g are final once meta[g].flag holds an
// earlier pass number.  mdst[] (match start positions) is sorted: sequences are stored in frame order.
__device__ inline bool lz_range_final(uint64_t lo, uint64_t hi, const SeqMeta *meta, const uint32_t *cidx, uint64_t g_self,
                                      uint32_t pass, uint64_t halo_end = 0) {
    if (hi <= lo) return true;
    if (lo < halo_end) return false;                     // reaches into the window in front of a shard, which has not arrived yet
    uint64_t g;
    if (cidx) {                                          // the sequence before the first one of lo's chunk may reach into the range
        const uint64_t f = cidx[lo >> kLzIdxShift];
        g = f ? f - 1 : 0;
    } else {
        uint64_t a = 0, b = g_self;                      // last g < g_self with meta[g].pos <= lo
        while (a < b) {
            const uint64_t mid = (a + b) >> 1;
            if (meta[mid].pos <= lo)
                a = mid + 1;
            else
                b = mid;
        }
        g = a ? a - 1 : 0;
    }
    for (; g < g_self; g++) {
        const SeqMeta m = meta[g];                           // one 16-byte load: position, length and pass stamp of a producer
        if (m.pos >= hi) break;
        if (m.pos + m.ml <= lo) continue;                    // ends before the range
        if (m.flag == 0 || m.flag >= pass) return false;
    }
    return true;
}

template <bool ASCII>
__global__ __launch_bounds__(256) void k_lz_match_pass(const SeqBlock *__restrict__ blocks, uint32_t n_blocks,
                                                       const Seq *__restrict__ seqs, SeqMeta *meta,
                                                       const uint32_t *__restrict__ cidx, uint32_t *blk_pending,
                                                       uint32_t *roff, unsigned long long *remaining, const uint32_t *__restrict__ rep_init,
                                                       const uint64_t *__restrict__ blk_base, uint8_t *out_bytes, uint32_t pass,
                                                       uint64_t *plist, unsigned long long *pcount, uint64_t halo_end, uint32_t *status) {
    using Elem = typename std::conditional<ASCII, uint16_t, uint8_t>::type;
    Elem *out = reinterpret_cast<Elem *>(out_bytes);
    __shared__ uint32_t s_long[256];
    __shared__ uint32_t s_nlong, s_ndone, s_abort, s_pending, s_npend;
    __shared__ unsigned long long s_pbase;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) s_abort = status[0];
    __syncthreads();
    if (s_abort) return;
    for (uint32_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        if (tid == 0) s_pending = blk_pending[b];
        __syncthreads();
        const uint32_t pend = s_pending;
        __syncthreads();
        if (pend == 0) continue;
        const SeqBlock sb = blocks[b];
        const uint64_t fstart = blk_base[sb.frame_first_blk];
        const uint32_t init[3] = {rep_init[3 * b], rep_init[3 * b + 1], rep_init[3 * b + 2]};
        for (uint32_t s0 = 0; s0 < sb.n_seq; s0 += 256) {
            if (tid == 0) {
                s_nlong = 0;
                s_ndone = 0;
                s_npend = 0;
            }
            __syncthreads();
            const uint64_t g = sb.seq_first + s0 + tid;
            uint32_t my_slot = 0xFFFFFFFFu;                      // position in this step's part of the pending list
            if (s0 + tid < sb.n_seq && meta[g].flag == 0) {
                const Seq q = seqs[g];
                bool bad = false;
                const uint32_t off = rep_resolve(q.off, init, &bad);
                const uint64_t mpos = meta[g].pos;
                if (bad || off > mpos - fstart) {                // reaches before the frame: corrupt
                    flag_error(status, kStBadOffset, sb.blk);
                    meta[g].flag = pass;
                    atomicAdd(&s_ndone, 1u);
                } else {
                    const uint64_t src = mpos - off;
                    const uint64_t need_hi = src + q.ml < mpos ? src + q.ml : mpos;   // the rest is the match itself
                    roff[g] = off;                               // kept for the pointer-jumping stage
                    if (lz_range_final(src, need_hi, meta, cidx, g, pass, halo_end)) {
                        if (q.ml <= kLzShort) {
                            Elem *d = out + mpos;
                            const Elem *s = out + src;
                            for (uint32_t k = 0; k < q.ml; k++) d[k] = s[k];   // byte-serial: overlap allowed
                            meta[g].flag = pass;
                            atomicAdd(&s_ndone, 1u);
                        } else {
                            s_long[atomicAdd(&s_nlong, 1u)] = s0 + tid;
                        }
                    } else if (plist) {
                        my_slot = atomicAdd(&s_npend, 1u);       // stays pending: goes on the list the later passes work from
                    }
                }
            }
            __syncthreads();
            const uint32_t nl = s_nlong;
            for (uint32_t j = 0; j < nl; j++) {                  // long matches: the whole workgroup on each
                const uint64_t gi = sb.seq_first + s_long[j];
                const Seq q = seqs[gi];
                bool bad = false;
                const uint32_t off = rep_resolve(q.off, init, &bad);
                lz_copy_match<Elem>(out + meta[gi].pos, q.ml, off, tid, 256);
                if (tid == 0) meta[gi].flag = pass;
            }
            __syncthreads();
            if (tid == 0 && (s_ndone + nl)) {
                atomicSub(&blk_pending[b], s_ndone + nl);
                atomicAdd(remaining, ~static_cast<unsigned long long>(s_ndone + nl) + 1ull);   // -= done
            }
            if (tid == 0 && s_npend) s_pbase = atomicAdd(pcount, static_cast<unsigned long long>(s_npend));
            __syncthreads();
            if (my_slot != 0xFFFFFFFFu) plist[s_pbase + my_slot] = (static_cast<uint64_t>(b) << 40) | g;
        }
    }
}

// Passes after the first work from the list of matches that are still pending (entry = seq-block
// index << 40 | sequence index) instead of re-reading every sequence of every unfinished block: a
// pass costs what is left, not what there was.  Survivors go to the next pass's list.  No kernel
// in between: pass k reads list k & 1 (its length in lcount[k % 3]), appends to the other list
// (length lcount[(k + 1) % 3]) and clears lcount[(k + 2) % 3], which pass k + 1 appends to; a pass
// whose input is empty returns at once, so the host enqueues a fixed number of them.
template <bool ASCII>
__global__ __launch_bounds__(256) void k_lz_match_list(const uint64_t *__restrict__ lin, uint64_t *lout, unsigned long long *lcount,
                                                       const Seq *__restrict__ seqs,
                                                       SeqMeta *meta, const uint32_t *__restrict__ cidx,
                                                       uint32_t *blk_pending, const uint32_t *__restrict__ roff,
                                                       unsigned long long *remaining,
                                                       uint8_t *out_bytes, uint32_t pass, uint64_t halo_end, const uint32_t *status) {
    const unsigned long long *nin = lcount + pass % 3u;
    unsigned long long *nout = lcount + (pass + 1u) % 3u;
    if (blockIdx.x == 0 && threadIdx.x == 0) lcount[(pass + 2u) % 3u] = 0;
    using Elem = typename std::conditional<ASCII, uint16_t, uint8_t>::type;
    Elem *out = reinterpret_cast<Elem *>(out_bytes);
    __shared__ uint32_t s_long[256];
    __shared__ uint64_t s_ent[256];
    __shared__ uint32_t s_nlong, s_ndone, s_abort, s_npend;
    __shared__ unsigned long long s_pbase;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) s_abort = status[0];
    __syncthreads();
    if (s_abort) return;
    const unsigned long long n = *nin;
    for (unsigned long long base = static_cast<unsigned long long>(blockIdx.x) * 256; base < n;
         base += static_cast<unsigned long long>(gridDim.x) * 256) {
        if (tid == 0) {
            s_nlong = 0;
            s_ndone = 0;
            s_npend = 0;
        }
        __syncthreads();
        uint32_t my_slot = 0xFFFFFFFFu;
        uint64_t ent = 0;
        if (base + tid < n) {
            ent = lin[base + tid];
            s_ent[tid] = ent;
            const uint64_t g = ent & ((1ull << 40) - 1ull);
            const Seq q = seqs[g];
            const uint32_t off = roff[g];                        // resolved in the first pass
            const uint64_t mpos = meta[g].pos;
            const uint64_t src = mpos - off;
            const uint64_t need_hi = src + q.ml < mpos ? src + q.ml : mpos;
            if (lz_range_final(src, need_hi, meta, cidx, g, pass, halo_end)) {
                if (q.ml <= kLzShort) {
                    Elem *d = out + mpos;
                    const Elem *sp = out + src;
                    for (uint32_t k = 0; k < q.ml; k++) d[k] = sp[k];   // element-serial: overlap allowed
                    meta[g].flag = pass;
                    atomicSub(&blk_pending[ent >> 40], 1u);
                    atomicAdd(&s_ndone, 1u);
                } else {
                    s_long[atomicAdd(&s_nlong, 1u)] = tid;
                }
            } else {
                my_slot = atomicAdd(&s_npend, 1u);
            }
        }
        __syncthreads();
        const uint32_t nl = s_nlong;
        for (uint32_t j = 0; j < nl; j++) {                      // long matches: the whole workgroup on each
            const uint64_t e = s_ent[s_long[j]];
            const uint64_t gi = e & ((1ull << 40) - 1ull);
            const Seq q = seqs[gi];
            const uint32_t off = roff[gi];
            lz_copy_match<Elem>(out + meta[gi].pos, q.ml, off, tid, 256);
            if (tid == 0) {
                meta[gi].flag = pass;
                atomicSub(&blk_pending[e >> 40], 1u);
            }
        }
        __syncthreads();
        if (tid == 0) {
            if (s_ndone + nl) atomicAdd(remaining, ~static_cast<unsigned long long>(s_ndone + nl) + 1ull);   // -= done
            if (s_npend) s_pbase = atomicAdd(nout, static_cast<unsigned long long>(s_npend));
        }
        __syncthreads();
        if (my_slot != 0xFFFFFFFFu) lout[s_pbase + my_slot] = ent;
    }
}

// ---- what the list passes leave behind, sparse sections: one workgroup walks on ---------------
// A handful of pending matches (a real genome's few far repeats; the Length section of equal reads:
// one whole-block run per block, each copying from the block before) needs no launch per pass: ONE
// workgroup keeps passing over the list until it is empty, or until a pass resolves less than an
// eighth of it -- chains that deep are left to the frame-order walk behind it (k_lz_matches_ordered,
// which returns at once when nothing is pending).
template <bool ASCII>
__global__ __launch_bounds__(1024) void k_lz_finish_small(uint64_t *l0, uint64_t *l1, unsigned long long *lcount, uint32_t first_pass,
                                                          const Seq *__restrict__ seqs, SeqMeta *meta, const uint32_t *__restrict__ cidx,
                                                          uint32_t *blk_pending, const uint32_t *__restrict__ roff,
                                                          unsigned long long *counters, uint8_t *out_bytes, uint64_t halo_end, uint32_t resume,
                                                          const uint64_t *__restrict__ n_total, uint64_t tail_elems, const uint32_t *status) {
    // halo_end / resume / n_total / tail_elems: the shard protocol.  With the window in front of the shard still unknown
    // (halo_end > 0) whatever reaches into it survives; the survivors' list, its length and the next pass number are left in
    // counters[kCtrLeft], [kCtrWhich] (which list) and [kCtrPass], and counters[kCtrTail] says whether one of them lies in the last tail_elems elements
    // (which the next shard waits for).  resume = 1 picks that list up again once the window has arrived.
    using Elem = typename std::conditional<ASCII, uint16_t, uint8_t>::type;
    Elem *out = reinterpret_cast<Elem *>(out_bytes);
    __shared__ unsigned long long s_n, s_nout, s_left;
    __shared__ uint64_t s_long[1024];
    __shared__ uint32_t s_done, s_abort, s_nlong, s_pass, s_which, s_tail;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) {
        s_abort = status[0];
        if (resume) {
            s_n = counters[kCtrLeft];
            s_which = static_cast<uint32_t>(counters[kCtrWhich]);
            s_pass = static_cast<uint32_t>(counters[kCtrPass]);
        } else {
            s_n = lcount[first_pass % 3u];
            s_which = first_pass & 1u;
            s_pass = first_pass;
            counters[1] = s_n;                             // statistics: matches the launched passes left over
        }
        s_left = s_n;
        s_tail = 0;
    }
    __syncthreads();
    if (s_abort) return;
    uint64_t *lin = s_which ? l1 : l0, *lout = s_which ? l0 : l1;
    uint32_t pass = s_pass;
    for (; s_n != 0; pass++) {
        const unsigned long long n = s_n;
        if (tid == 0) {
            s_nout = 0;
            s_done = 0;
        }
        __syncthreads();
        for (unsigned long long base = 0; base < n; base += 1024) {
            const unsigned long long i = base + tid;
            if (tid == 0) s_nlong = 0;
            __syncthreads();
            if (i < n) {
                const uint64_t ent = lin[i];
                const uint64_t g = ent & ((1ull << 40) - 1ull);
                const Seq q = seqs[g];
                const uint32_t off = roff[g];
                const uint64_t mpos = meta[g].pos;
                const uint64_t src = mpos - off;
                const uint64_t need_hi = src + q.ml < mpos ? src + q.ml : mpos;
                if (lz_range_final(src, need_hi, meta, cidx, g, pass, halo_end)) {
                    if (q.ml <= kLzShort) {
                        Elem *d = out + mpos;
                        const Elem *sp = out + src;
                        for (uint32_t k = 0; k < q.ml; k++) d[k] = sp[k];   // element-serial: overlap allowed
                    } else {
                        s_long[atomicAdd(&s_nlong, 1u)] = ent;   // long: the whole workgroup copies it below
                    }
                    meta[g].flag = pass;
                    atomicSub(&blk_pending[ent >> 40], 1u);
                    atomicAdd(&s_done, 1u);
                } else {
                    lout[atomicAdd(&s_nout, 1ull)] = ent;
                }
            }
            __syncthreads();
            const uint32_t nl = s_nlong;
            for (uint32_t j = 0; j < nl; j++) {
                const uint64_t g = s_long[j] & ((1ull << 40) - 1ull);
                lz_copy_match<Elem>(out + meta[g].pos, seqs[g].ml, roff[g], tid, 1024);
            }
            __syncthreads();
        }
        __threadfence();                                   // this pass's bytes and stamps before the next pass reads them
        __syncthreads();
        const uint32_t done = s_done;
        const unsigned long long left = s_nout;
        __syncthreads();
        if (tid == 0) {
            atomicAdd(counters, ~static_cast<unsigned long long>(done) + 1ull);   // remaining -= done
            s_left = left;
            s_n = done * 8ull >= n ? left : 0;             // slow progress: stop here, the frame-order walk finishes
        }
        uint64_t *t = lin;
        lin = lout;
        lout = t;
        __syncthreads();
    }
    // what is left (in `lin`), for a later resume; and whether any of it lies in the tail the next shard waits for
    const unsigned long long left = s_left;
    if (tail_elems) {
        const uint64_t total = *n_total, tail_lo = total > tail_elems ? total - tail_elems : 0;
        for (unsigned long long i = tid; i < left; i += 1024) {
            const uint64_t g = lin[i] & ((1ull << 40) - 1ull);
            if (meta[g].pos + seqs[g].ml > tail_lo) s_tail = 1;
        }
    }
    __syncthreads();
    if (tid == 0) {
        counters[kCtrLeft] = left;
        counters[kCtrWhich] = lin == l1 ? 1ull : 0ull;
        counters[kCtrPass] = pass;
        counters[kCtrTail] = s_tail;
    }
}

// ---- pointer jumping, dense sections -------------------------------------------------------
// Where matches make up a good part of the output (level-3 DNA; quality strings, whose dense
// short-offset matches chain through the whole frame) the matches are not visited one by one at
// all.  Every output element p gets D[p] = distance to an element it is a copy of (k_pj_fill: the
// match offset; 0 for literals, which are final), and the frame is SWEPT: an element whose source
// is final takes its byte and becomes final; otherwise it points past its source,
// D[p] += D[p - D[p]] -- the chain above it halves.  A sweep reads D sequentially, its gathers stay
// inside the window, nothing but D and the output is touched (no per-match records), tiles without
// pending elements cost one word, and after ~log2(longest chain) sweeps everything is final.  The
// host enqueues a fixed number of sweeps; each returns at once when the one before left nothing.
// The word of an element that became final holds the element itself beside the mark (one 32-bit store):
// whoever finds its source final has the value in the same look-up, whenever that source was written.
constexpr uint32_t kPjTile = 2048;           // elements per tile (256 threads x 8)
constexpr uint32_t kPjFinal = 0xFFFF0000u;   // D >= kPjFinal: final, and the low 16 bits ARE the element (byte / two characters)
constexpr uint32_t kPjWait = kPjFinal - 1u;  // an element of the window in front of a shard whose value has not arrived yet (shard protocol):
                                             // not pending, not final -- whoever copies from it stays pending, and no distance reaches this value
__device__ inline bool pj_pending(uint32_t v) { return v - 1u < kPjWait - 1u; }   // 1 <= v < kPjWait: a distance
constexpr uint32_t kPjLocal = 4;             // jumps inside the tile (LDS) before a sweep looks into memory
constexpr uint32_t kPjStripSweeps = 1;       // ... for the first this-many sweeps; what they leave is scattered: tile-wise from there
constexpr uint32_t kPjWin = 1;               // tiles a strip-wise sweep keeps in LDS behind the current one (k_pj_sweep).  (Round 4: the held tiles as
                                             // 16-bit words -- final 0xFF00 | byte, distances below 0xFE00, else "ask memory" -- and THREE of them in the
                                             // same LDS: 29.1-29.3 against 28.3 ms on the FASTQ-like probe, profiles/r04_narrow_lds_ab.log: dropped.)
constexpr uint32_t kPjHops = 2;              // look-ups per element and pass (a pending source hands over its distance: look again) ...
constexpr uint32_t kPjHopsDeep = 4;          // ... where chains are deep (few literals: no strip-wise sweep).  Same box, hops 1 / 2 / 3 / 4
                                             // (profiles/r04_pj_hops_probe.log): FASTQ-like level 1 29.85 / 28.41 / 28.81 / 29.27 ms, level 3
                                             // 79.6 / 72.0 / 67.1 / 64.9, level-3 DNA 6.27 / 6.12 / 6.19 / 6.26
constexpr uint32_t kPjFinishSweeps = 4;      // a shard's sweeps after the window in front of it has arrived
constexpr uint32_t kPjSweeps = 24;           // a sweep at least halves every chain: 2^24 matches deep; what is left after them goes to the
                                             // frame-order walk (every launch that finds nothing left still costs its 4-5 us: 40 of them
                                             // plus as many k_pj_list were 0.4 ms per section)

// Output-ordered (round 3): the threads of a workgroup take CONSECUTIVE elements of the block's output, four each, find the
// sequence an element belongs to with one binary search over the batch's output positions (LDS) and write its word of D --
// a literal: the FINAL word, its value taken from the literal buffer (the output itself gets it from k_pj_emit, in order,
// with everything else); an element of a match: the distance.  Every word of the block's range is written, in 16-byte
// stores: no memset of D in front, no scattered runs (the first version wrote 4-byte runs at every match and the literal
// runs into the output: 2.5 GB of write granules for 0.5 GB of words on level-3 DNA), and the sweeps' first look at a
// literal source finds its value in the same word.  Blocks whose literals k_huf_decode put in place (sb.direct) write 0
// for them ("literal, value in the output").  What lies between the blocks with sequences (raw / RLE / literal-only
// blocks, the window in front of a tile) gets zeros from the block behind it; the last block also zeroes the tail up to
// n_elems.  skip_lo: leading elements that are somebody else's (a shard's window that has not arrived: kPjWait).
// (Round 4, measured and dropped: sequence-ordered once more, but through LDS -- every thread writes the elements of ITS
//  sequence into a 4 096-element window that then leaves as 16-byte stores, long sequences by the whole workgroup: no search, no
//  selects -- and 2.5 ms against 1.57 on level-3 DNA, 31.4 against 28.8 ms on the FASTQ-like probe
//  (profiles/r04_fill_ab.log): a wave's loop is as long as its longest sequence, three to five times the mean.)
constexpr uint32_t kFillPer = 8;              // elements a thread of k_pj_fill takes per step
template <bool ASCII>
__global__ __launch_bounds__(256) void k_pj_fill(const SeqBlock *__restrict__ blocks, uint32_t n_blocks, const Seq *__restrict__ seqs,
                                                 const uint32_t *__restrict__ rep_init, const uint64_t *__restrict__ blk_base,
                                                 uint32_t *D, const uint8_t *__restrict__ lit, uint32_t *blk_pending,
                                                 uint32_t t_char, uint32_t n_sel_blocks, uint64_t n_elems, uint64_t skip_lo, uint32_t *status) {
    __shared__ uint32_t s_opos[257], s_ll[256], s_off[256], s_lpos[256];
    __shared__ uint16_t s_chars[ASCII ? 256 : 2];          // the two characters of a packed byte (byte_chars costs twenty instructions a literal)
    __shared__ uint32_t s_abort;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) s_abort = status[0];
    if (ASCII) s_chars[tid] = static_cast<uint16_t>(byte_chars(tid, t_char));
    __syncthreads();
    if (s_abort) return;
    auto zero_range = [&](uint64_t lo, uint64_t hi) {      // D[lo, hi) = 0, by the whole workgroup (lo, hi uniform)
        if (lo < skip_lo) lo = skip_lo;
        if (hi <= lo) return;
        const uint64_t a0 = (lo + 3) & ~uint64_t(3), a1 = hi & ~uint64_t(3);
        if (a0 >= a1) {
            for (uint64_t e = lo + tid; e < hi; e += 256) D[e] = 0;
            return;
        }
        for (uint64_t e = lo + tid; e < a0; e += 256) D[e] = 0;
        for (uint64_t e = a0 + 4ull * tid; e < a1; e += 1024) *reinterpret_cast<uint4 *>(D + e) = make_uint4(0, 0, 0, 0);
        for (uint64_t e = a1 + tid; e < hi; e += 256) D[e] = 0;
    };
    auto lit_word = [&](uint8_t c) -> uint32_t { return kPjFinal | (ASCII ? static_cast<uint32_t>(s_chars[c]) : static_cast<uint32_t>(c)); };
    for (uint32_t b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        const SeqBlock sb = blocks[b];
        const uint64_t obase = blk_base[sb.blk], fstart = blk_base[sb.frame_first_blk];
        const uint32_t init[3] = {rep_init[3 * b], rep_init[3 * b + 1], rep_init[3 * b + 2]};
        const uint8_t *blit = lit + sb.lit_off;
        const bool has_lit = !sb.direct;                   // (else k_huf_decode put the literals in place: their words are 0)
        if (tid == 0) blk_pending[b] = sb.n_seq;           // (for the frame-order walk, should it have to run)
        // what lies between the block with sequences in front of this one (or the start) and this block
        zero_range(b ? blk_base[blocks[b - 1].blk + 1] : 0, obase);
        uint32_t out_end = 0;                              // elements of the block written so far
        for (uint32_t s0 = 0; s0 < sb.n_seq; s0 += 256) {
            __syncthreads();                               // the batch before is done with the arrays
            uint32_t span_end = 0;
            if (s0 + tid < sb.n_seq) {
                const Seq q = seqs[sb.seq_first + s0 + tid];
                bool bad = false;
                uint32_t off = rep_resolve(q.off, init, &bad);
                const uint64_t mpos = obase + q.opos + q.ll;
                if (bad || off > mpos - fstart || off >= kPjWait) {   // reaches before the frame (corrupt) / beyond any legal window
                    flag_error(status, kStBadOffset, sb.blk);
                    off = 1;
                }
                s_opos[tid] = q.opos;
                s_ll[tid] = q.ll;
                s_off[tid] = off;
                s_lpos[tid] = q.lpos;
                span_end = q.opos + q.ll + q.ml;
            }
            const uint32_t n_here = sb.n_seq - s0 < 256 ? sb.n_seq - s0 : 256;
            if (tid == n_here - 1) s_opos[n_here] = span_end;   // one past the batch's last element
            __syncthreads();
            const uint32_t e0 = s_opos[0], e1 = s_opos[n_here];
            if (e1 > kBlockMax || e0 != out_end) {         // (k_seq_values checked the sums: cannot happen)
                flag_error(status, kStSizeMismatch, sb.blk);
                break;
            }
            // kFillPer consecutive elements per thread and step, aligned to 16 bytes of D where the range allows.  The kernel is
            // bound by latency (PMC: waves wait 71 % of their cycles; every step is search -> addresses -> literal bytes ->
            // store), so a step carries as much independent work as registers allow: ONE binary search for the first
            // element, the next sequences' boundaries and fields fetched together, then all literal bytes at once.
            const uint64_t g0 = obase + e0, g1 = obase + e1;
            const uint64_t a0 = (g0 + 3) & ~uint64_t(3);
            for (uint64_t g = (g0 & ~uint64_t(kFillPer - 1)) + static_cast<uint64_t>(kFillPer) * tid; g < g1; g += static_cast<uint64_t>(kFillPer) * 256) {
                uint32_t w[kFillPer];
                uint32_t lo = 0;
                const uint64_t first = g < g0 ? g0 : g, last = g + kFillPer < g1 ? g + kFillPer : g1;
                {   // the sequence that holds the first element: largest j with opos[j] <= e
                    const uint32_t e = static_cast<uint32_t>(first - obase);
                    uint32_t hi = n_here;
                    while (hi - lo > 1) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (s_opos[mid] <= e)
                            lo = mid;
                        else
                            hi = mid;
                    }
                }
                // that sequence and the three behind it, in registers (indices past the batch repeat its last sequence: the
                // boundary in front of them is the batch's end, which no element of the step reaches)
                uint32_t so[4], sl[4], sf[4], sp[4], nb[4];
#pragma unroll
                for (uint32_t i = 0; i < 4; i++) {
                    const uint32_t j = lo + i < n_here ? lo + i : n_here - 1;
                    so[i] = s_opos[j];
                    sl[i] = s_ll[j];
                    sf[i] = s_off[j];
                    sp[i] = s_lpos[j];
                    nb[i] = s_opos[lo + i + 1 < n_here ? lo + i + 1 : n_here];
                }
                const bool simple = static_cast<uint32_t>(last - 1 - obase) < nb[3] || lo + 4 >= n_here;   // the step's elements lie in those four
                uint32_t litpos[kFillPer];                                                                 // literal-buffer index of an element, ~0: not a literal
#pragma unroll
                for (uint32_t k = 0; k < kFillPer; k++) {
                    const uint64_t ge = g + k;
                    w[k] = 0;
                    litpos[k] = 0xFFFFFFFFu;
                    if (ge < first || ge >= last) continue;
                    const uint32_t e = static_cast<uint32_t>(ge - obase);
                    uint32_t o, ll, off, lp;
                    if (simple) {
                        const uint32_t i = (e >= nb[0] ? 1u : 0u) + (e >= nb[1] ? 1u : 0u) + (e >= nb[2] ? 1u : 0u);
                        o = i == 0 ? so[0] : (i == 1 ? so[1] : (i == 2 ? so[2] : so[3]));
                        ll = i == 0 ? sl[0] : (i == 1 ? sl[1] : (i == 2 ? sl[2] : sl[3]));
                        off = i == 0 ? sf[0] : (i == 1 ? sf[1] : (i == 2 ? sf[2] : sf[3]));
                        lp = i == 0 ? sp[0] : (i == 1 ? sp[1] : (i == 2 ? sp[2] : sp[3]));
                    } else {                               // (more than four sequences in the step: very short ones)
                        while (lo + 1 < n_here && s_opos[lo + 1] <= e) lo++;
                        o = s_opos[lo];
                        ll = s_ll[lo];
                        off = s_off[lo];
                        lp = s_lpos[lo];
                    }
                    const uint32_t r = e - o;
                    if (r < ll) {
                        litpos[k] = lp + r;
                    } else {
                        const uint32_t kk = r - ll;
                        // element kk of a match at distance off copies element kk - off; where the match reaches into itself
                        // (off <= kk: a run) that is an element of the same match, and so on down to the `off` elements in
                        // front of it -- point there at once instead of leaving kk / off hops to the sweeps
                        w[k] = off;
                        if (kk >= off) {
                            // kk / off without the twenty instructions of an integer division, eight times a step: both are
                            // below 2^24, the quotient of the float reciprocal is off by one at most
                            uint32_t q = static_cast<uint32_t>(static_cast<float>(kk) * __frcp_rn(static_cast<float>(off)));
                            const int32_t rem = static_cast<int32_t>(kk - q * off);
                            q += rem < 0 ? 0xFFFFFFFFu : (rem >= static_cast<int32_t>(off) ? 1u : 0u);
                            w[k] = off * (q + 1u);
                        }
                    }
                }
                if (has_lit) {                             // all literal bytes of the step on their way together
                    uint8_t c[kFillPer];
#pragma unroll
                    for (uint32_t k = 0; k < kFillPer; k++) c[k] = litpos[k] != 0xFFFFFFFFu ? blit[litpos[k]] : 0;
#pragma unroll
                    for (uint32_t k = 0; k < kFillPer; k++)
                        if (litpos[k] != 0xFFFFFFFFu) w[k] = lit_word(c[k]);
                }
#pragma unroll
                for (uint32_t q = 0; q < kFillPer / 4; q++) {
                    const uint64_t gq = g + 4 * q;
                    if (gq >= a0 && gq + 4 <= g1) {
                        *reinterpret_cast<uint4 *>(D + gq) = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
                    } else {
#pragma unroll
                        for (uint32_t k = 0; k < 4; k++)
                            if (gq + k >= g0 && gq + k < g1) D[gq + k] = w[4 * q + k];
                    }
                }
            }
            out_end = e1;
        }
        __syncthreads();
        // literals after the last sequence run to the end of the block
        {
            const Seq last = seqs[sb.seq_first + sb.n_seq - 1];
            const uint32_t lused = last.lpos + last.ll, oend = last.opos + last.ll + last.ml;
            const uint32_t n_tail = sb.lit_size > lused ? sb.lit_size - lused : 0;
            if (has_lit) {
                for (uint32_t k = tid; k < n_tail; k += 256) D[obase + oend + k] = lit_word(blit[lused + k]);
            } else {
                zero_range(obase + oend, obase + oend + n_tail);
            }
            // ... and behind the last block with sequences: the rest of the selection, and of D
            if (b + 1 == n_blocks) zero_range(obase + oend + n_tail, n_elems);
        }
    }
}

// Once fewer than a third of the elements are pending, a sweep over all of D mostly reads words that are final.  The first
// sweep that starts below that mark also LISTS what it leaves pending (indices, 32 bits: sections of 2^32 elements or more
// keep sweeping), and from then on k_pj_list takes over: it walks the list, writes the survivors to a second list, and so
// on -- a pass costs what is pending, not what there is.  Both kernels are enqueued for every sweep number; `lstate[3]`
// (0: sweeping; s: listing since sweep s) tells each whether it is its turn.  lstate[s % 3] = entries listed by sweep s.
// Survivors are gathered in LDS and handed to the list 2 048 at a time: one addition to the list's length per batch.
constexpr uint32_t kPjBatch = 2048;
template <uint32_t N>
struct PjLister {                                          // (LDS state of one workgroup)
    uint32_t buf[N];
    uint32_t n, base;
};
// hands the gathered entries to the list when there are `limit` of them or more (limit 1: whatever there is)
template <uint32_t N>
__device__ inline void pj_list_flush(PjLister<N> *L, uint32_t *list_out, unsigned long long *len, uint64_t cap, uint32_t limit,
                                     unsigned long long *overflow = nullptr) {
    // called by every thread of the workgroup with uniform arguments, at a point where L->n is stable
    // (overflow: set when the list does not hold what is handed to it -- a list begun by the FIRST sweep, which cannot know
    //  how much it will leave pending; whoever reads the list afterwards then goes on sweeping instead)
    __syncthreads();
    const uint32_t n = L->n;
    if (n >= limit && n) {
        if (threadIdx.x == 0) {
            const unsigned long long at = atomicAdd(len, static_cast<unsigned long long>(n));
            L->base = static_cast<uint32_t>(at);
            if (overflow && at + n > cap) *overflow = 1;
        }
        __syncthreads();
        const uint64_t base = L->base;
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x)
            if (base + i < cap) list_out[base + i] = L->buf[i];
        __syncthreads();
        if (threadIdx.x == 0) L->n = 0;
    }
    __syncthreads();
}

// WIN > 0 (text and quality sections whose chains are shallow: LzArgs.strips): the workgroup takes a STRIP of consecutive
// tiles, front to back, and keeps the words of the last WIN tiles it swept (as it left them: final wherever they got
// final) in LDS beside the current one; the next tile's words are on their way while this one's are worked on.  In a
// strip everything in front of an element has been swept already -- as in a serial decoder -- so with a fifth of the
// elements literals (level-1 quality strings: chains a few links long) nearly everything resolves in the first pass,
// sources a few thousand elements back out of LDS, the others by the gathers below: 2 sweeps instead of 4-5.
// Where chains are deep (level 3: 2 % literals, forty links) a pass resolves no more than one of the tile-wise sweeps
// (WIN = 0: the tile alone, tiles dealt round robin, twice the workgroups per CU for the gathers) -- those keep them.
template <bool ASCII, uint32_t WIN>
__global__ __launch_bounds__(256, WIN ? 5 : 6) void k_pj_sweep(uint32_t *D, uint8_t *out_bytes, uint32_t *tile_pending, unsigned long long *pcount,
                                                  uint64_t n_elems, uint32_t sweep, uint32_t max_dist, uint32_t *list_out,
                                                  uint64_t list_cap, unsigned long long *lstate, uint32_t first_list, uint32_t hops,
                                                  uint32_t emit_now, uint32_t xcd_map, const uint32_t *status) {   // emit_now: 0 no, 1 every final element, 2 what this sweep made final
    using Elem = typename std::conditional<ASCII, uint16_t, uint8_t>::type;
    Elem *out = reinterpret_cast<Elem *>(out_bytes);
    constexpr uint32_t kSlots = WIN + 1;
    __shared__ uint32_t s_cnt[3];                          // pending elements of the tile, three counters in rotation
    __shared__ uint32_t s_D[kSlots * kPjTile];             // the tile's words (and the WIN tiles before it): the first few jumps of a sweep stay in LDS
    __shared__ uint64_t s_held[kSlots];                    // which tile each slot holds (~0: none)
    __shared__ PjLister<kPjBatch> s_list;                  // (a tile's survivors at most: flushed before a tile that would not fit)
    const uint32_t tid = threadIdx.x;
    // pcount[s % 3] = elements still pending after sweep s (pcount[0] != 0 before the first one)
    const unsigned long long before = pcount[(sweep + 2u) % 3u];
    // lstate[3]: listing since sweep ...; lstate[4]: that list overflowed (it was begun by the first sweep) -- then nobody
    // lists, the sweeps go on, and a later one begins a new list once it knows that what is pending fits
    const unsigned long long listing = lstate[4] != 0 ? 0ull : lstate[3];
    if (listing != 0 && listing < sweep) return;           // k_pj_list's turn (it keeps the counters from here on)
    // (listing == sweep: workgroup 0 of THIS launch has set it already.  Sweep 1 starts from a placeholder count: it
    //  lists only when the host expects little to be left -- first_list -- and the overflow flag covers the rest.)
    const bool build = list_out != nullptr && status[0] == 0 &&
                       (sweep >= 2 ? (before != 0 && before * 3 < n_elems && before <= list_cap) : first_list != 0);
    if (blockIdx.x == 0 && tid == 0) {
        pcount[(sweep + 1u) % 3u] = 0;
        lstate[(sweep + 1u) % 3u] = 0;
        if (before == 0 || status[0] != 0) pcount[sweep % 3u] = 0;
        if (lstate[4] != 0) {                              // (the order matters to the workgroups that read both words meanwhile)
            lstate[3] = 0;
            __threadfence();
            lstate[4] = 0;
        }
        if (build) lstate[3] = sweep;                      // (read by the kernels launched after this one)
    }
    if (before == 0 || status[0] != 0) return;
    // (a list begun by the first sweep is worth having only when it is short -- an eighth of the elements: a pass over a list
    //  is two gathers per entry, a tile-wise sweep reads its tiles in order -- beyond that it counts as overflowed)
    const uint64_t cap_now = sweep >= 2 ? list_cap : (n_elems / 8 < list_cap ? n_elems / 8 : list_cap);
    if (tid == 0) s_list.n = 0;
    if (tid < 3) s_cnt[tid] = 0;
    if (tid < kSlots) s_held[tid] = ~0ull;
    __syncthreads();
    const uint64_t n_tiles = (n_elems + kPjTile - 1) / kPjTile;
    // WIN: a strip of consecutive tiles per workgroup; else tiles dealt round robin
    const uint64_t per_wg = (n_tiles + gridDim.x - 1) / gridDim.x;
    // Which tiles a workgroup takes follows the chip: workgroups b and b + 8 run on the same XCD -- one L2 -- and an element's
    // source lies within a window (a few hundred tiles) in front of it, so the workgroups of ONE XCD take NEIGHBOURING strips
    // (tiles): what they gather was read or written a moment ago by a neighbour on the same L2.  Dealt in launch order,
    // neighbouring strips sit on eight different L2s and every gather goes to memory.  (xcd_map 0: launch order.)
    uint32_t slot = blockIdx.x;                             // position of this workgroup in tile order
    if (xcd_map && gridDim.x >= 16u) {
        const uint32_t per_xcd = gridDim.x / 8u, body = per_xcd * 8u;   // (the last gridDim.x % 8 workgroups keep their places)
        if (blockIdx.x < body) slot = (blockIdx.x % 8u) * per_xcd + blockIdx.x / 8u;
    }
    const uint64_t t_first = WIN ? slot * per_wg : slot;
    const uint64_t t_last = WIN ? (t_first + per_wg < n_tiles ? t_first + per_wg : n_tiles) : n_tiles;
    const uint64_t t_step = WIN ? 1 : gridDim.x;
    uint32_t flip = 0;
    unsigned long long wg_pending = 0;                     // (thread 0)
    auto load_words = [&](uint64_t at, uint32_t *x) {      // four words of D at element `at` (a multiple of four), zeros past the end
        x[0] = x[1] = x[2] = x[3] = 0;
        if (at >= n_elems) return;
        if (n_elems - at >= 4) {
            const uint4 q = *reinterpret_cast<const uint4 *>(D + at);   // D is 16-byte aligned
            x[0] = q.x;
            x[1] = q.y;
            x[2] = q.z;
            x[3] = q.w;
        } else {
            for (uint32_t k = 0; k < static_cast<uint32_t>(n_elems - at); k++) x[k] = D[at + k];
        }
    };
    uint32_t nxt[2][4];
    bool have_next = false;
    for (uint64_t t = t_first; t < t_last; t += t_step) {
        if (sweep > 1 && tile_pending[t] == 0) {            // (the same word for every thread: uniform)
            have_next = false;
            continue;
        }
        const uint32_t slot = WIN ? static_cast<uint32_t>(t % kSlots) : 0u;
        uint32_t *s_cur = s_D + slot * kPjTile;
        const uint64_t tile0 = t * kPjTile;                // first element of the tile
        // Two runs of four consecutive elements per thread.  All the look-ups of a thread are issued before any of its
        // stores: a source read a moment too early is still a valid ancestor in the chain (its old distance points
        // further back), so the order inside a sweep does not matter -- but eight dependent round trips to memory do.
        uint64_t p[2];
        uint32_t v[2][4], w[2][4];
        uint32_t dirty = 0;                                // bit 4 half + k: the element's word differs from what memory holds
#pragma unroll
        for (uint32_t half = 0; half < 2; half++) {
            p[half] = tile0 + tid * 4 + half * (kPjTile / 2);
            if (WIN && have_next) {                         // (the words loaded while the tile in front was worked on)
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) v[half][k] = nxt[half][k];
                continue;
            }
            load_words(p[half], v[half]);
        }
        if (WIN) {                                         // the next tile's words set out now
            have_next = t + 1 < t_last;
            if (have_next) {
                load_words(tile0 + kPjTile + tid * 4, nxt[0]);
                load_words(tile0 + kPjTile + tid * 4 + kPjTile / 2, nxt[1]);
            }
        }
        // Jumps whose source lies in the same tile (or, WIN, in one of the tiles of the strip still held) are taken in LDS
        // first, kPjLocal times over: quality strings and tandem repeats copy from a few hundred elements back, so a sweep
        // advances a chain by up to 2^kPjLocal doublings for one pass over D in memory.  No barriers between the rounds: a
        // thread writes its own eight words only, and a word read a moment early or late is a valid ancestor (or the final
        // value) either way.
#pragma unroll
        for (uint32_t half = 0; half < 2; half++) {
            // literals (0: final, element in the output) become final WORDS the first time a sweep sees them: from then on
            // whoever copies from them -- in this tile's LDS rounds right away -- needs no second look-up
            if ((v[half][0] == 0 || v[half][1] == 0 || v[half][2] == 0 || v[half][3] == 0) && p[half] < n_elems) {
                Elem el[4] = {0, 0, 0, 0};
                if (n_elems - p[half] >= 4)
                    __builtin_memcpy(el, out + p[half], 4 * sizeof(Elem));
                else
                    for (uint32_t k = 0; k < static_cast<uint32_t>(n_elems - p[half]); k++) el[k] = out[p[half] + k];
#pragma unroll
                for (uint32_t k = 0; k < 4; k++)
                    if (v[half][k] == 0 && p[half] + k < n_elems) {
                        v[half][k] = kPjFinal | el[k];
                        dirty |= 1u << (4 * half + k);
                    }
            }
            *reinterpret_cast<uint4 *>(&s_cur[tid * 4 + half * (kPjTile / 2)]) = make_uint4(v[half][0], v[half][1], v[half][2], v[half][3]);
        }
        if (WIN && tid == 0) s_held[slot] = t;
        __syncthreads();
        // how far back LDS reaches: the held tiles directly in front of this one
        uint32_t reach = 0;                                // elements in front of the tile that are in LDS
        if (WIN) {
            for (uint32_t k = 1; k <= WIN; k++) {
                if (t < k || s_held[(t - k) % kSlots] != t - k) break;
                reach = k * kPjTile;
            }
        }
#pragma unroll 1
        for (uint32_t it = 0; it < kPjLocal; it++) {
#pragma unroll
            for (uint32_t half = 0; half < 2; half++)
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    const uint32_t li = tid * 4 + half * (kPjTile / 2) + k, d = v[half][k];
                    if (!pj_pending(d) || d > li + reach) continue;             // literal / final / source not in LDS
                    // (element index of the source modulo what the slots hold: consecutive tiles sit in consecutive slots, cyclically)
                    const uint32_t ws = WIN ? s_D[static_cast<uint32_t>((tile0 + li - d) % (kSlots * kPjTile))] : s_cur[li - d];
                    if (ws >= kPjFinal) {
                        v[half][k] = ws;                                        // (mark and element of the source: ours too)
                        s_cur[li] = ws;
                        dirty |= 1u << (4 * half + k);
                    } else if (ws != 0 && static_cast<uint64_t>(d) + ws < max_dist) {
                        v[half][k] = d + ws;
                        s_cur[li] = d + ws;
                        dirty |= 1u << (4 * half + k);
                    }                                                           // (a literal in the tile: its element is in the output -- below)
                }
        }
        // Up to `hops` look-ups per element, one behind the other: a source that is itself pending hands over its distance
        // (the chain above the element halves, as in every sweep) and the element looks again at once, instead of waiting
        // for the next pass over all of D -- a pass is 4 bytes per element read and written, a look-up one sector.  All
        // the look-ups of one round are issued together; an element whose distance cannot grow (max_dist, or a source
        // that waits for a shard's window) is `stuck` and stops looking.
        uint32_t stuck = 0;
#pragma unroll 1
        for (uint32_t hop = 0; hop < hops; hop++) {
            bool more = false;
#pragma unroll
            for (uint32_t half = 0; half < 2; half++)
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    const bool pending = pj_pending(v[half][k]) && !(stuck & (1u << (4 * half + k)));   // else: literal / final already
                    w[half][k] = pending ? D[p[half] + k - v[half][k]] : 1u;
                }
#pragma unroll
            for (uint32_t half = 0; half < 2; half++)
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {         // a literal as the source (D == 0): its element is in the output only
                    const bool pending = pj_pending(v[half][k]) && !(stuck & (1u << (4 * half + k)));
                    if (pending && w[half][k] == 0) w[half][k] = kPjFinal | out[p[half] + k - v[half][k]];
                }
#pragma unroll
            for (uint32_t half = 0; half < 2; half++)
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    const uint32_t bit = 1u << (4 * half + k);
                    if (!pj_pending(v[half][k]) || (stuck & bit)) continue;
                    const uint32_t ws = w[half][k];
                    if (ws >= kPjFinal) {
                        // final: the word of a final element carries the element itself, so ONE look-up both tells that the
                        // source is final and fetches it (and there is no "became final a moment ago, is its byte visible?")
                        v[half][k] = ws;                   // (the output gets it from here when the sweeps are over: k_pj_emit)
                        dirty |= bit;
                    } else if (static_cast<uint64_t>(v[half][k]) + ws < max_dist) {
                        v[half][k] += ws;
                        dirty |= bit;
                        more = true;
                    } else {
                        stuck |= bit;                      // (a distance that cannot grow waits for its source)
                    }
                }
            if (!__any(more ? 1 : 0)) break;               // (per wave: nobody in it has anything to look at again)
        }
        uint32_t remaining = 0, survivors = 0;
#pragma unroll
        for (uint32_t half = 0; half < 2; half++) {
            const bool changed = ((dirty >> (4 * half)) & 15u) != 0;
#pragma unroll
            for (uint32_t k = 0; k < 4; k++)
                if (pj_pending(v[half][k])) {
                    remaining++;
                    survivors |= 1u << (4 * half + k);
                }
            // the thread's four words go back as one store (nobody else writes them; readers take the old or the new value)
            if (changed) {
                if (n_elems - p[half] >= 4) {
                    *reinterpret_cast<uint4 *>(D + p[half]) = make_uint4(v[half][0], v[half][1], v[half][2], v[half][3]);
                } else {
                    for (uint32_t k = 0; k < static_cast<uint32_t>(n_elems - p[half]); k++) D[p[half] + k] = v[half][k];
                }
            }
            // ... and the elements that are final go to the output: in the first sweep all of them (the literals' words are
            // final since k_pj_fill), later what this sweep made final.  A thread's four elements are consecutive and so are the
            // threads: the wave writes 256 (512) contiguous bytes where everything is final -- no pass over D afterwards
            // (k_pj_emit read all of it again: 6 GB for the qualities of 10 M reads).
            if (emit_now) {
                const uint32_t sel = emit_now == 1u ? 15u : (dirty >> (4 * half)) & 15u;
                uint32_t fin = 0;
#pragma unroll
                for (uint32_t k = 0; k < 4; k++)
                    if (v[half][k] >= kPjFinal && ((sel >> k) & 1u) && p[half] + k < n_elems) fin |= 1u << k;
                if (fin == 15u) {
                    if (ASCII) {                           // (the output of a tile need not be aligned: memcpy)
                        const uint32_t x[2] = {(v[half][0] & 0xFFFFu) | (v[half][1] << 16), (v[half][2] & 0xFFFFu) | (v[half][3] << 16)};
                        __builtin_memcpy(out + p[half], x, 8);
                    } else {
                        const uint32_t x = (v[half][0] & 0xFFu) | ((v[half][1] & 0xFFu) << 8) | ((v[half][2] & 0xFFu) << 16) | (v[half][3] << 24);
                        __builtin_memcpy(out + p[half], &x, 4);
                    }
                } else if (fin) {
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++)
                        if (fin & (1u << k)) out[p[half] + k] = static_cast<Elem>(v[half][k] & 0xFFFFu);
                }
            }
            // (WIN) ... and into the slot, as the tiles behind this one will find them
            if (WIN) *reinterpret_cast<uint4 *>(&s_cur[tid * 4 + half * (kPjTile / 2)]) = make_uint4(v[half][0], v[half][1], v[half][2], v[half][3]);
        }
        if (remaining) atomicAdd(&s_cnt[flip], remaining);
        __syncthreads();
        if (tid == 0) {
            const uint32_t c = s_cnt[flip];
            tile_pending[t] = c;
            wg_pending += c;
            s_cnt[(flip + 2u) % 3u] = 0;                   // (the counter of the tile after next: nobody adds to it before the next barrier)
        }
        if (build) {                                       // (uniform) the tile's survivors join the list
            const uint32_t c = s_cnt[flip];                // (stable: thread 0 reset the OTHER counters only)
            pj_list_flush(&s_list, list_out, &lstate[sweep % 3u], cap_now, s_list.n + c > kPjBatch ? 1u : kPjBatch, &lstate[4]);
            if (remaining) {
                uint32_t slot_l = atomicAdd(&s_list.n, remaining);
#pragma unroll
                for (uint32_t j = 0; j < 8; j++)
                    if (survivors & (1u << j)) s_list.buf[slot_l++] = static_cast<uint32_t>(p[j >> 2] + (j & 3u));
            }
        }
        flip = (flip + 1u) % 3u;
    }
    if (build) pj_list_flush(&s_list, list_out, &lstate[sweep % 3u], cap_now, 1u, &lstate[4]);
    // (one addition per workgroup, not per tile: 700 k additions to one word serialise)
    if (tid == 0 && wg_pending) atomicAdd(&pcount[sweep % 3u], wg_pending);
}

// A pass over the list of pending elements (see above): sweep `sweep` reads what sweep - 1 listed and lists what it leaves.
template <bool ASCII>
__global__ __launch_bounds__(256) void k_pj_list(uint32_t *D, uint8_t *out_bytes, unsigned long long *pcount, const uint32_t *list_in,
                                                 uint32_t *list_out, uint64_t list_cap, unsigned long long *lstate, uint32_t sweep,
                                                 uint32_t max_dist, uint32_t emit, uint32_t hops, const uint32_t *status) {
    using Elem = typename std::conditional<ASCII, uint16_t, uint8_t>::type;
    Elem *out = reinterpret_cast<Elem *>(out_bytes);
    __shared__ PjLister<kPjBatch + 256> s_list;
    const uint32_t tid = threadIdx.x;
    const unsigned long long listing = lstate[4] != 0 ? 0ull : lstate[3];
    if (listing == 0 || sweep <= listing) return;          // still sweeping (or the list overflowed) / the sweep that made the first list
    const unsigned long long before = pcount[(sweep + 2u) % 3u];
    unsigned long long n_in = lstate[(sweep + 2u) % 3u];
    if (n_in > list_cap) n_in = list_cap;
    if (blockIdx.x == 0 && tid == 0) {
        pcount[(sweep + 1u) % 3u] = 0;
        lstate[(sweep + 1u) % 3u] = 0;
        if (before == 0 || status[0] != 0) pcount[sweep % 3u] = 0;
    }
    if (before == 0 || status[0] != 0) return;
    if (tid == 0) s_list.n = 0;
    __syncthreads();
    unsigned long long mine = 0;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * 256;
    const uint64_t rounds = (n_in + stride - 1) / stride;                        // (uniform: every thread takes part in every flush)
    for (uint64_t r = 0; r < rounds; r++) {
        const uint64_t i = r * stride + static_cast<uint64_t>(blockIdx.x) * 256 + tid;
        if (i < n_in) {
            const uint32_t p = list_in[i];
            const uint32_t v0 = D[p];
            if (pj_pending(v0)) {
                uint32_t v = v0;
                bool final = false;
                Elem el = 0;
                for (uint32_t hop = 0; hop < hops; hop++) {                     // (as in k_pj_sweep: a pending source hands over its distance, look again)
                    const uint32_t ws = D[p - v];
                    if (ws == 0 || ws >= kPjFinal) {
                        el = ws == 0 ? out[p - v] : static_cast<Elem>(ws & 0xFFFFu);
                        final = true;
                        break;
                    }
                    if (static_cast<uint64_t>(v) + ws >= max_dist) break;
                    v += ws;
                }
                if (final) {
                    D[p] = kPjFinal | el;
                    if (emit) out[p] = el;                                      // (after k_pj_emit has run: the finishing passes of a shard)
                } else {
                    if (v != v0) D[p] = v;
                    mine++;
                    s_list.buf[atomicAdd(&s_list.n, 1u)] = p;                    // (at most 256 a round on top of < 2 048)
                }
            }
        }
        pj_list_flush(&s_list, list_out, &lstate[sweep % 3u], list_cap, kPjBatch);
    }
    pj_list_flush(&s_list, list_out, &lstate[sweep % 3u], list_cap, 1u);
    // pending after this pass, one addition per workgroup
    __shared__ unsigned long long s_sum;
    if (tid == 0) s_sum = 0;
    __syncthreads();
    if (mine) atomicAdd(&s_sum, mine);
    __syncthreads();
    if (tid == 0 && s_sum) atomicAdd(&pcount[sweep % 3u], s_sum);
}

// After the sweeps: every element that a sweep made final goes from its word of D to the output (the sweeps themselves
// do not write the output: one sequential pass here instead of a scattered one-element store per element and sweep).
template <bool ASCII>
__global__ __launch_bounds__(256) void k_pj_emit(const uint32_t *__restrict__ D, uint8_t *out_bytes, uint64_t n_elems,
                                                 const unsigned long long *skip_if, unsigned long long skip_max, const uint32_t *status) {
    using Elem = typename std::conditional<ASCII, uint16_t, uint8_t>::type;
    Elem *out = reinterpret_cast<Elem *>(out_bytes);
    if (status[0] != 0) return;
    // (a shard's finishing passes: when they all ran from the list -- list mode since a sweep <= skip_max, before they began --
    //  they wrote the output themselves)
    if (skip_if && skip_if[0] != 0 && skip_if[1] == 0 && skip_if[0] <= skip_max) return;   // ([1]: that list overflowed -- nobody used it)
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * 256 * 4;
    for (uint64_t p = (static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x) * 4; p < n_elems; p += stride) {
        if (n_elems - p >= 4) {
            const uint4 w = *reinterpret_cast<const uint4 *>(D + p);
            const uint32_t v[4] = {w.x, w.y, w.z, w.w};
            const bool all = v[0] >= kPjFinal && v[1] >= kPjFinal && v[2] >= kPjFinal && v[3] >= kPjFinal;
            if (all) {                                     // (inside a match: the usual case) one store
                if (ASCII) {                               // (the output of a tile need not be aligned: memcpy)
                    const uint32_t x[2] = {(v[0] & 0xFFFFu) | (v[1] << 16), (v[2] & 0xFFFFu) | (v[3] << 16)};
                    __builtin_memcpy(out + p, x, 8);
                } else {
                    const uint32_t x = (v[0] & 0xFFu) | ((v[1] & 0xFFu) << 8) | ((v[2] & 0xFFu) << 16) | (v[3] << 24);
                    __builtin_memcpy(out + p, &x, 4);
                }
            } else {
#pragma unroll
                for (uint32_t k = 0; k < 4; k++)
                    if (v[k] >= kPjFinal) out[p + k] = static_cast<Elem>(v[k] & 0xFFFFu);
            }
        } else {
            for (uint32_t k = 0; k < static_cast<uint32_t>(n_elems - p); k++) {
                const uint32_t v = D[p + k];
                if (v >= kPjFinal) out[p + k] = static_cast<Elem>(v & 0xFFFFu);
            }
        }
    }
}

// ---- shard protocol: the window in front of a shard arrives after everything that does not depend on it is done ----
__global__ __launch_bounds__(256) void k_fill_u32(uint32_t *p, uint64_t n, uint32_t v) {
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<uint64_t>(gridDim.x) * 256) p[i] = v;
}

// the window has arrived (it is in the output buffer, elements [0, n_halo)): its elements become final words of D
template <bool ASCII>
__global__ __launch_bounds__(256) void k_pj_halo_words(uint32_t *D, const uint8_t *out_bytes, uint64_t n_halo) {
    using Elem = typename std::conditional<ASCII, uint16_t, uint8_t>::type;
    const Elem *out = reinterpret_cast<const Elem *>(out_bytes);
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x; i < n_halo; i += static_cast<uint64_t>(gridDim.x) * 256)
        D[i] = kPjFinal | out[i];
}

// Dense sections: does anything still pending (it waits for the window) lie in the last tail_elems elements -- what the
// next shard waits for?  From the list of pending elements when the sweeps ended in list mode, else from the tiles' counts.
__global__ __launch_bounds__(256) void k_pj_tail_check(const uint32_t *list, const unsigned long long *lstate, uint32_t last_sweep,
                                                       uint64_t list_cap, const uint32_t *tile_pending, const unsigned long long *pcount,
                                                       const uint64_t *n_total, uint64_t tail_elems, unsigned long long *flag) {
    const uint64_t total = *n_total, tail_lo = total > tail_elems ? total - tail_elems : 0;
    if (pcount[last_sweep % 3u] == 0) return;              // nothing pending at all
    bool hit = false;
    const uint64_t me = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x, stride = static_cast<uint64_t>(gridDim.x) * 256;
    if (lstate[3] != 0 && lstate[4] == 0) {
        unsigned long long n = lstate[last_sweep % 3u];
        if (n > list_cap) n = list_cap;
        for (uint64_t i = me; i < n; i += stride) hit = hit || list[i] >= tail_lo;
    } else {
        const uint64_t t0 = tail_lo / kPjTile, t1 = (total + kPjTile - 1) / kPjTile;
        for (uint64_t t = t0 + me; t < t1; t += stride) hit = hit || tile_pending[t] != 0;
    }
    if (hit) *flag = 1;
}

// the 32 counter words of a section's LZ stages ([0] = sequences, [4] = dense ? 1 : 0, everything else 0)
__global__ void k_lz_init_counters(unsigned long long *counters, unsigned long long n_sequences, unsigned long long dense) {
    const uint32_t t = threadIdx.x;
    if (t < 32) counters[t] = t == 0 ? n_sequences : (t == 4 ? dense : 0ull);
}

template <bool ASCII>
__global__ __launch_bounds__(256) void k_lz_matches_ordered(const SeqBlock *__restrict__ blocks, uint32_t n_blocks,
                                                            const Seq *__restrict__ seqs, SeqMeta *meta,
                                                            const uint32_t *__restrict__ blk_pending,
                                                            const uint32_t *__restrict__ rep_init, const uint64_t *__restrict__ blk_base,
                                                            uint8_t *out_bytes, const unsigned long long *gate, uint32_t *status) {
    using Elem = typename std::conditional<ASCII, uint16_t, uint8_t>::type;
    Elem *out = reinterpret_cast<Elem *>(out_bytes);
    __shared__ uint32_t s_abort, s_pending, s_n;
    __shared__ uint32_t s_idx[256], s_ml[256], s_off[256];
    __shared__ uint64_t s_mpos[256];
    const uint32_t tid = threadIdx.x;
    if (tid == 0) s_abort = status[0] != 0 || *gate == 0;  // gate: what the parallel stages left pending (0: nothing to do)
    __syncthreads();
    if (s_abort != 0) return;
    for (uint32_t b = 0; b < n_blocks; b++) {
        if (tid == 0) s_pending = blk_pending[b];
        __syncthreads();
        const uint32_t pend = s_pending;
        __syncthreads();
        if (pend == 0) continue;
        const SeqBlock sb = blocks[b];
        const uint64_t fstart = blk_base[sb.frame_first_blk];
        const uint32_t init[3] = {rep_init[3 * b], rep_init[3 * b + 1], rep_init[3 * b + 2]};
        for (uint32_t s0 = 0; s0 < sb.n_seq; s0 += 256) {
            // collect the batch's pending matches (flags are read 256 at a time), restore their order
            if (tid == 0) s_n = 0;
            __syncthreads();
            // (meta == nullptr: a swept section -- no per-match records, every match is redone in order)
            if (s0 + tid < sb.n_seq && (!meta || meta[sb.seq_first + s0 + tid].flag == 0)) s_idx[atomicAdd(&s_n, 1u)] = s0 + tid;
            __syncthreads();
            const uint32_t n = s_n;
            if (n > 1) {
                uint32_t mine = 0xFFFFFFFFu, rank = 0;
                if (tid < n) {
                    mine = s_idx[tid];
                    for (uint32_t j = 0; j < n; j++) rank += s_idx[j] < mine ? 1u : 0u;
                }
                __syncthreads();
                if (tid < n) s_idx[rank] = mine;
                __syncthreads();
            }
            // stage the pending records in LDS: the in-order loop below must not wait on global loads
            if (tid < n) {
                const uint64_t g = sb.seq_first + s_idx[tid];
                const Seq q = seqs[g];
                bool bad = false;
                const uint32_t off = rep_resolve(q.off, init, &bad);
                const uint64_t mpos = meta ? meta[g].pos : blk_base[sb.blk] + q.opos + q.ll;
                s_mpos[tid] = mpos;
                s_ml[tid] = q.ml;
                s_off[tid] = (bad || off > mpos - fstart) ? 0u : off;
            }
            __syncthreads();
            for (uint32_t j = 0; j < n; j++) {               // strictly in order: every source is final by now
                const uint32_t off = s_off[j], ml = s_ml[j];
                if (off == 0) {
                    flag_error(status, kStBadOffset, sb.blk);
                } else {
                    lz_copy_match<Elem>(out + s_mpos[j], ml, off, tid, 256);
                }
                __syncthreads();
            }
            __syncthreads();
        }
    }
}

// ======================================================================================
// soft mask (MaskReader + Decoder::mask_sequence, reader.rs:198-231, mod.rs:402-441)
// ======================================================================================
// Runs alternate unmasked / masked starting unmasked; run k covers [ends[k-1], ends[k]).
// The reference lower-cases a masked run only where it ENDS strictly inside the current record
// (mod.rs:410-415); the part of a run that reaches or crosses a record end stays upper case
// (SURVEY App. D-1).  In global coordinates: for a masked run [s, e) let r be the record that
// holds base e-1; it is lower-cased over [max(s, start_r), e) iff e < end_r.  spec_mask != 0
// lower-cases the whole run instead.
// make_ascii_lowercase on four bytes at once: bytes in 'A'..'Z' get bit 5 set, everything else
// (including bytes >= 0x80) stays
__device__ inline uint32_t lower4(uint32_t w) {
    const uint32_t w7 = w & 0x7F7F7F7Fu;
    const uint32_t up = (w7 + 0x3F3F3F3Fu) & ~(w7 + 0x25252525u) & ~w & 0x80808080u;
    return w | (up >> 2);
}

// masked run k (odd) -> the bases [s, e) it lower-cases, inside [lo_clamp, hi_clamp); e == s: nothing
__device__ inline void masked_run_range(uint64_t k, uint64_t n_bases, uint64_t lo_clamp, uint64_t hi_clamp,
                                        const uint64_t *__restrict__ mask_ends, const uint64_t *__restrict__ rec_ends, uint64_t n_rec,
                                        int spec_mask, uint64_t *s_out, uint64_t *e_out) {
    uint64_t s = mask_ends[k - 1], e = mask_ends[k];
    if (s >= n_bases) {
        s = e = 0;
    } else if (e > n_bases) {                          // MaskReader stops at `total`: the overshoot is never applied
        e = spec_mask ? n_bases : s;
    }
    if (e > s && !spec_mask && !(kAblate & 1024u)) {
        uint64_t lo = 0, hi = n_rec;                   // first record whose end is > e - 1
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (rec_ends[mid] > e - 1)
                hi = mid;
            else
                lo = mid + 1;
        }
        if (lo >= n_rec) {
            e = s;                                     // beyond the last record
        } else {
            const uint64_t rend = rec_ends[lo];
            const uint64_t rstart = lo ? rec_ends[lo - 1] : 0;
            if (e >= rend)
                e = s;                                 // run reaches the record end: stays upper case
            else if (s < rstart)
                s = rstart;
        }
    }
    // a shard holds bases [lo_clamp, hi_clamp) only
    if (s < lo_clamp) s = lo_clamp;
    if (e > hi_clamp) e = hi_clamp;
    if (e < s) e = s;
    *s_out = s;
    *e_out = e;
}

__global__ __launch_bounds__(256) void k_mask_apply(uint8_t *ascii, uint64_t n_bases, uint64_t lo_clamp, uint64_t hi_clamp,
                                                    const uint64_t *__restrict__ mask_ends,
                                                    const ScanTotals *mask_totals, const uint64_t *__restrict__ rec_ends,
                                                    const ScanTotals *rec_totals, int spec_mask, const uint32_t *status) {
    // A workgroup takes 256 consecutive masked runs.  Phase 1: one run per thread -- clamp it and
    // (reference behaviour) find the record that holds its last base with a binary search, 256
    // searches in flight at once.  Phase 2: the runs are cut into 16-byte aligned chunks, an exclusive
    // prefix sum of the chunk counts turns (run, chunk) into one flat index, and the 256 threads sweep
    // that index space -- consecutive threads on consecutive 16 bytes, whatever the run lengths.
    // Chunks inside a run are rewritten as one uint4; the (at most two) chunks a run shares with its
    // neighbours are done byte-wise, so no thread ever writes a byte outside its own run.
    __shared__ uint64_t s_lo[256], s_hi[256];
    __shared__ uint64_t s_pre[2][257];                      // chunk-count prefix sums (ping-pong for the scan)
    const uint32_t abort_now = status[0];                  // same word for every thread of the launch
    const uint64_t n_runs = mask_totals->count;
    const uint64_t n_rec = rec_totals->count;
    const uint32_t tid = threadIdx.x;
    const uint64_t skew = reinterpret_cast<uintptr_t>(ascii) & 15;   // a shard's base pointer need not be 16-byte aligned
    uint8_t *const abase = ascii - skew;
    for (uint64_t base = static_cast<uint64_t>(blockIdx.x) * 256; 2 * base + 1 < n_runs && !abort_now;
         base += static_cast<uint64_t>(gridDim.x) * 256) {
        const uint64_t k = 2 * (base + tid) + 1;           // odd runs are the masked ones
        uint64_t s = 0, e = 0;
        if (k < n_runs) masked_run_range(k, n_bases, lo_clamp, hi_clamp, mask_ends, rec_ends, n_rec, spec_mask, &s, &e);
        __syncthreads();                                   // previous round's readers are done
        s_lo[tid] = s;
        s_hi[tid] = e;
        s_pre[0][tid + 1] = e > s ? ((e + skew + 15) >> 4) - ((s + skew) >> 4) : 0;
        if (tid == 0) s_pre[0][0] = s_pre[1][0] = 0;
        __syncthreads();
        uint32_t cur = 0;
        for (uint32_t d = 1; d < 256; d <<= 1) {           // inclusive scan of entries 1..256
            const uint64_t v = s_pre[cur][tid + 1] + (tid >= d ? s_pre[cur][tid + 1 - d] : 0);
            s_pre[cur ^ 1][tid + 1] = v;
            cur ^= 1;
            __syncthreads();
        }
        const uint64_t *pre = s_pre[cur];                  // pre[r] = chunks of runs 0..r-1
        const uint64_t total = pre[256];
        // four chunks per thread and step: the loads of all four are in flight before the first store
        for (uint64_t c0 = tid; c0 < total; c0 += 4 * 256) {
            uint64_t addr[4], e0[4], e1[4];
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint64_t c = c0 + 256 * u;
                uint32_t lo = 0, hi = 256;                 // largest r with pre[r] <= c
                while (hi - lo > 1) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (pre[mid] <= c)
                        lo = mid;
                    else
                        hi = mid;
                }
                // in coordinates whose multiples of 16 are aligned addresses
                const uint64_t rs = s_lo[lo] + skew, re = s_hi[lo] + skew;
                const uint64_t a = ((rs >> 4) + (c - pre[lo])) << 4;
                addr[u] = a;
                e0[u] = a > rs ? a : rs;
                e1[u] = a + 16 < re ? a + 16 : re;
                if (c >= total) e0[u] = e1[u] = a;         // past the end: nothing to do
                if (e1[u] > e0[u]) v[u] = *reinterpret_cast<const uint4 *>(abase + a);   // whole aligned chunk: inside the buffer's padding at worst
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (e1[u] - e0[u] == 16) {
                    v[u].x = lower4(v[u].x);
                    v[u].y = lower4(v[u].y);
                    v[u].z = lower4(v[u].z);
                    v[u].w = lower4(v[u].w);
                    *reinterpret_cast<uint4 *>(abase + addr[u]) = v[u];
                } else if (e1[u] > e0[u]) {
                    // a chunk shared with the neighbours (unmasked bases, or another masked run that some other
                    // thread is rewriting right now): set bit 5 of OUR upper-case bytes with one atomic OR per
                    // dword -- no byte loops (a wave would wait for its slowest lane), no lost updates
                    const uint32_t in_range = ((1u << (e1[u] - addr[u])) - 1u) & ~((1u << (e0[u] - addr[u])) - 1u);   // 16 bits
                    const uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                    for (int d = 0; d < 4; d++) {
                        const uint32_t sel = (((in_range >> (4 * d)) & 0xFu) * 0x00204081u & 0x01010101u) * 0xFFu;
                        const uint32_t m = (lower4(w[d]) ^ w[d]) & sel;
                        if (m && !(kAblate & 512u)) atomicOr(reinterpret_cast<uint32_t *>(abase + addr[u]) + d, m);
                    }
                }
            }
        }
    }
}

// ======================================================================================
// frame checksums
// ======================================================================================
// Content_Checksum (RFC 8878 3.1.1): the low 32 bits of XXH64 (seed 0) of everything the frame decodes to, which
// the reference's zstd decoder verifies as it reaches the end of the frame (mod.rs:221-222 builds that decoder;
// a mismatch surfaces as Io(InvalidData)).  XXH64 is four serial multiply-rotate chains over the 32-byte stripes
// of the frame: one workgroup per frame piece, all 64 lanes stage a chunk of 32 stripes (packing the characters
// of a nucleotide section back into the bytes they came from) and pre-multiply it, lanes 0-3 run the chains.
// About 0.5 GB/s per frame: archives written by `ennaf` or the reference's encoder carry no checksum, those that
// do are verified in full, frames side by side.
constexpr uint64_t kXxhP1 = 11400714785074694791ull, kXxhP2 = 14029467366897019727ull, kXxhP3 = 1609587929392839161ull,
                   kXxhP4 = 9650029242287828579ull, kXxhP5 = 2870177450012600261ull;
__device__ inline uint64_t rotl64(uint64_t x, uint32_t r) { return (x << r) | (x >> (64u - r)); }
__device__ inline uint64_t xxh_round(uint64_t acc, uint64_t prod) { return rotl64(acc + prod, 31) * kXxhP1; }   // prod = input * P2

__global__ __launch_bounds__(64) void k_xxh64_frames(const XxhSeg *__restrict__ segs, const uint64_t *__restrict__ blk_base,
                                                      const uint8_t *__restrict__ out, uint32_t ascii, uint32_t t_char,
                                                      const XxhCarry *__restrict__ carry_in, XxhCarry *__restrict__ carry_out,
                                                      uint32_t *status) {
    __shared__ uint64_t s_prod[2][128];
    __shared__ XxhCarry s_c;
    __shared__ uint8_t s_inv[256];               // character -> 4-bit code
    const uint32_t lane = threadIdx.x;
    if (status[0] != 0) return;                  // (a stage in front failed: block bases and output are not to be trusted)
    const XxhSeg seg = segs[blockIdx.x];
    uint64_t a = blk_base[seg.blk0];
    const uint64_t b = blk_base[seg.blk1];
    for (uint32_t k = lane; k < 256; k += 64) s_inv[k] = 0;
    __syncthreads();
    if (ascii && lane < 16) s_inv[nib_char(lane, t_char)] = static_cast<uint8_t>(lane);
    if (lane == 0) {
        if (seg.flags & 1u) {
            s_c.v[0] = kXxhP1 + kXxhP2;
            s_c.v[1] = kXxhP2;
            s_c.v[2] = 0;
            s_c.v[3] = 0ull - kXxhP1;
            s_c.total = 0;
            s_c.n_mem = 0;
        } else {
            s_c = *carry_in;
        }
    }
    __syncthreads();
    auto byte_at = [&](uint64_t pos) -> uint8_t {   // decoded byte `pos` of the loaded selection
        if (!ascii) return out[pos];
        return static_cast<uint8_t>(s_inv[out[2 * pos]] | (s_inv[out[2 * pos + 1]] << 4));
    };
    uint64_t acc = s_c.v[lane & 3u];
    const uint64_t n_new = b - a;
    uint32_t n_mem = s_c.n_mem;
    if (n_mem) {                                 // finish the stripe the tile before left open
        const uint32_t take = static_cast<uint32_t>(n_new < 32u - n_mem ? n_new : 32u - n_mem);
        if (lane < take) s_c.mem[n_mem + lane] = byte_at(a + lane);
        __syncthreads();
        a += take;
        n_mem += take;
        if (n_mem == 32) {
            if (lane < 4) {
                uint64_t w;
                __builtin_memcpy(&w, s_c.mem + 8 * lane, 8);
                acc = xxh_round(acc, w * kXxhP2);
            }
            n_mem = 0;
        }
        __syncthreads();
    }
    const uint64_t n_stripes = (b - a) / 32, n_chunks = (n_stripes + 31) / 32;
    auto load_words = [&](uint64_t chunk, uint64_t *w) {     // this lane's 16 bytes of the chunk
        const uint64_t off = chunk * 1024 + lane * 16;
        w[0] = w[1] = 0;
        if (off >= n_stripes * 32) return;
        if (!ascii) {
            __builtin_memcpy(w, out + a + off, 16);
        } else {
            uint32_t cw[8];
            __builtin_memcpy(cw, out + 2 * (a + off), 32);
#pragma unroll
            for (uint32_t k = 0; k < 16; k++) {
                const uint32_t two = cw[k >> 1] >> (16 * (k & 1));
                const uint64_t byte = s_inv[two & 0xFFu] | (static_cast<uint32_t>(s_inv[(two >> 8) & 0xFFu]) << 4);
                w[k >> 3] |= byte << (8 * (k & 7));
            }
        }
    };
    uint64_t w[2];
    if (n_chunks) load_words(0, w);
    for (uint64_t chunk = 0; chunk < n_chunks; chunk++) {
        uint64_t *buf = s_prod[chunk & 1];
        buf[2 * lane] = w[0] * kXxhP2;
        buf[2 * lane + 1] = w[1] * kXxhP2;
        __syncthreads();
        if (chunk + 1 < n_chunks) load_words(chunk + 1, w);  // in flight while the chains run
        const uint64_t left = n_stripes - chunk * 32;
        if (lane < 4) {
            if (left >= 32) {
#pragma unroll 8
                for (uint32_t s = 0; s < 32; s++) acc = xxh_round(acc, buf[4 * s + lane]);
            } else {
                for (uint32_t s = 0; s < left; s++) acc = xxh_round(acc, buf[4 * s + lane]);
            }
        }
    }
    a += n_stripes * 32;
    const uint32_t tail = static_cast<uint32_t>(b - a);      // < 32 (and 0 when the lead-in took everything)
    __syncthreads();
    if (lane < tail) s_c.mem[n_mem + lane] = byte_at(a + lane);
    if (lane < 4) s_c.v[lane] = acc;
    __syncthreads();
    if (lane != 0) return;
    n_mem += tail;
    s_c.n_mem = n_mem;
    s_c.total += n_new;
    if (!(seg.flags & 2u)) {
        *carry_out = s_c;
        return;
    }
    uint64_t h;
    if (s_c.total >= 32) {
        h = rotl64(s_c.v[0], 1) + rotl64(s_c.v[1], 7) + rotl64(s_c.v[2], 12) + rotl64(s_c.v[3], 18);
        for (uint32_t k = 0; k < 4; k++) h = (h ^ xxh_round(0, s_c.v[k] * kXxhP2)) * kXxhP1 + kXxhP4;
    } else {
        h = kXxhP5;
    }
    h += s_c.total;
    uint32_t i = 0;
    for (; i + 8 <= n_mem; i += 8) {
        uint64_t x;
        __builtin_memcpy(&x, s_c.mem + i, 8);
        h ^= xxh_round(0, x * kXxhP2);
        h = rotl64(h, 27) * kXxhP1 + kXxhP4;
    }
    if (i + 4 <= n_mem) {
        uint32_t x;
        __builtin_memcpy(&x, s_c.mem + i, 4);
        h ^= static_cast<uint64_t>(x) * kXxhP1;
        h = rotl64(h, 23) * kXxhP2 + kXxhP3;
        i += 4;
    }
    for (; i < n_mem; i++) {
        h ^= static_cast<uint64_t>(s_c.mem[i]) * kXxhP5;
        h = rotl64(h, 11) * kXxhP1;
    }
    h ^= h >> 33;
    h *= kXxhP2;
    h ^= h >> 29;
    h *= kXxhP3;
    h ^= h >> 32;
    if (static_cast<uint32_t>(h) != seg.expected) flag_error(status, kStChecksum, seg.blk0);
}

// ======================================================================================
// checksum (hash64.h)
// ======================================================================================
// Bytes out of HBM into PINNED HOST memory, stored by the GPU itself over PCIe: 16-byte stores aligned in the destination, the
// source read at whatever alignment that leaves it (load16u), odd edges byte-wise.  Why not hipMemcpyAsync: the runtime's copy
// engines move 55 GB/s in one process and 29-35 in the next (which engine a process is dealt, it seems: the same binary, the same
// box, tools/iter_regime_probe.py), this kernel 54 GB/s in every one (profiles/r04_iter_regime_probe.log).
__global__ __launch_bounds__(256) void k_copy_out(uint8_t *dst, const uint8_t *__restrict__ src, uint64_t n) {
    const uint64_t head = (16u - (reinterpret_cast<uintptr_t>(dst) & 15u)) & 15u;
    const uint64_t h = head < n ? head : n;
    const uint64_t tid = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x, nthr = static_cast<uint64_t>(gridDim.x) * 256;
    if (tid < h) dst[tid] = src[tid];
    const uint64_t n16 = (n - h) / 16;
    for (uint64_t i = tid; i < n16; i += nthr) *reinterpret_cast<uint4 *>(dst + h + 16 * i) = load16u(src + h + 16 * i);
    const uint64_t done = h + 16 * n16;
    if (tid < n - done) dst[done + tid] = src[done + tid];
}

__global__ __launch_bounds__(256) void k_hash64(const uint8_t *__restrict__ p, uint64_t n, uint64_t first_chunk,
                                                unsigned long long *result) {
    // sum over the 8-byte words of hash_word(word, position): any split of the words over threads gives
    // the same value.  Two words (16 bytes, any alignment) per thread and step.
    __shared__ uint64_t s_part[256];
    const uint64_t w0 = first_chunk * (kHashChunk / 8), n_full = n / 8;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * 256 * 2;
    uint64_t acc = 0;
    for (uint64_t j = (static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x) * 2; j < n_full; j += stride) {
        if (j + 1 < n_full) {
            uint64_t w[2];
            __builtin_memcpy(w, p + 8 * j, 16);
            acc += hash_word(w[0], w0 + j) + hash_word(w[1], w0 + j + 1);
        } else {
            uint64_t w;
            __builtin_memcpy(&w, p + 8 * j, 8);
            acc += hash_word(w, w0 + j);
        }
    }
    if ((n & 7) && blockIdx.x == 0 && threadIdx.x == 0) {  // last, zero-padded word
        uint64_t w = 0;
        for (uint32_t k = 0; k < (n & 7); k++) w |= static_cast<uint64_t>(p[8 * n_full + k]) << (8 * k);
        acc += hash_word(w, w0 + n_full);
    }
    s_part[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t d = 128; d > 0; d >>= 1) {
        if (threadIdx.x < d) s_part[threadIdx.x] += s_part[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0 && s_part[0]) atomicAdd(result, static_cast<unsigned long long>(s_part[0]));
}

}  // namespace

// ======================================================================================
// launchers
// ======================================================================================
void launch_seq_decode(hipStream_t stream, const uint8_t *src, const SeqBlock *blocks, uint32_t n_blocks,
                       const SeqCell *cells, Seq *seqs, SeqMeta *meta, uint32_t *blk_size, uint32_t *rep_final, uint32_t *status,
                       uint32_t cells_cap, long long src_min) {
    if (!n_blocks) return;
    SeqRec *recs = reinterpret_cast<SeqRec *>(meta);      // (the SeqMeta records are written after these two kernels)
    // The chain out of LDS (k_seq_states_lds) when every block of the section is resident at once: tables + ring per block,
    // 160 KiB per CU, one wave per SIMD.  NAFGPU_K2_LDS=0/1 (nafgpu_test_hooks) forces one or the other.
    static const uint32_t n_cu = [] {
        hipDeviceProp_t p;
        int dev = 0;
        (void)hipGetDevice(&dev);
        return hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0 ? static_cast<uint32_t>(p.multiProcessorCount) : 1u;
    }();
    cells_cap = (cells_cap + 3u) & ~3u;                               // (the ring behind the cells takes 16-byte stores)
    const uint32_t per_block = cells_cap * 4u + kSeqRing + 16u;
    uint32_t per_cu = (160u * 1024u - 1024u) / per_block;             // blocks resident per CU
    uint32_t lds_lanes = (per_cu + 3u) / 4u;                          // ... spread over four waves, at most kSeqLdsLanes each
    if (lds_lanes > kSeqLdsLanes) lds_lanes = kSeqLdsLanes;
    if (lds_lanes * per_block > 64u * 1024u) lds_lanes = (64u * 1024u) / per_block;
    if (lds_lanes) per_cu = ((160u * 1024u - 1024u) / (lds_lanes * per_block + 256u)) * lds_lanes;
    bool use_lds = cells_cap != 0 && lds_lanes != 0 && static_cast<uint64_t>(n_blocks) <= static_cast<uint64_t>(n_cu) * per_cu;
    // ... or, for sections up to about twice that size, with 2-byte cells and a 512-byte ring (k_seq_states_lds16): about one
    // wave per SIMD again, each with as many blocks as it takes to have every block of the section resident
    const uint32_t per_block_s = cells_cap * 2u + kSeqRingS + 16u;
    uint32_t lanes16 = 0;
    uint32_t k2_waves = 1;                                            // waves per SIMD the chains are spread over
    if (const char *e = hook_env("NAFGPU_K2_WAVES")) k2_waves = static_cast<uint32_t>(std::atoi(e)) > 0 ? static_cast<uint32_t>(std::atoi(e)) : 1u;
    if (cells_cap != 0) {
        const uint32_t spread = (n_blocks + 4u * k2_waves * n_cu - 1u) / (4u * k2_waves * n_cu);
        for (uint32_t l = spread < 1 ? 1 : spread; l <= kSeqLdsLanesS && l * per_block_s <= 64u * 1024u; l++) {
            uint32_t wgs = (160u * 1024u - 1024u) / (l * per_block_s + 256u);
            if (wgs > 32u) wgs = 32u;
            if (static_cast<uint64_t>(n_cu) * wgs * l >= n_blocks) {
                lanes16 = l;
                break;
            }
        }
    }
    bool use_lds16 = !use_lds && lanes16 != 0;
    if (const char *e = hook_env("NAFGPU_K2_LDS")) {
        use_lds = cells_cap != 0 && lds_lanes != 0 && e[0] == '1';
        use_lds16 = lanes16 != 0 && e[0] == '2';
    }
    if (use_lds16) {
        hipLaunchKernelGGL(k_seq_states_lds16, dim3((n_blocks + lanes16 - 1) / lanes16), dim3(64), lanes16 * per_block_s, stream, src, blocks,
                           n_blocks, cells, recs, lanes16, cells_cap, src_min, status);
    } else if (use_lds) {
        // fewer lanes per wave than fit, when there are few blocks: every CU gets its share of the chains
        uint32_t lanes = (n_blocks + 4u * k2_waves * n_cu - 1u) / (4u * k2_waves * n_cu);
        lanes = lanes < 1 ? 1 : (lanes > lds_lanes ? lds_lanes : lanes);
        hipLaunchKernelGGL(k_seq_states_lds, dim3((n_blocks + lanes - 1) / lanes), dim3(64), lanes * per_block, stream, src, blocks, n_blocks,
                           cells, recs, lanes, cells_cap, src_min, status);
#ifdef NAFGPU_EMU
        if (std::getenv("NAFGPU_K2_CHECK")) {
            uint64_t total = 0;
            for (uint32_t b = 0; b < n_blocks; b++) total = std::max<uint64_t>(total, blocks[b].seq_first + blocks[b].n_seq);
            std::vector<SeqRec> ref(total);
            uint32_t st2[16] = {0};
            hipLaunchKernelGGL(k_seq_states, dim3((n_blocks + 15) / 16), dim3(64), 0, stream, src, blocks, n_blocks, cells, ref.data(), 16u, st2);
            std::fprintf(stderr, "K2 check: status lds %u ref %u\n", status[0], st2[0]);
            for (uint32_t b = 0; b < n_blocks; b++)
                for (uint32_t i = 0; i < blocks[b].n_seq; i++) {
                    const SeqRec &x = recs[blocks[b].seq_first + i], &y = ref[blocks[b].seq_first + i];
                    if (x.pos != y.pos || x.states != y.states) {
                        std::fprintf(stderr, "K2 check: block %u (bits_off %llu len %u n_seq %u al %u %u %u) seq %u: lds pos %d states %x, ref pos %d states %x\n", b,
                                     (unsigned long long)blocks[b].bits_off, blocks[b].bits_len, blocks[b].n_seq, blocks[b].ll_al, blocks[b].of_al, blocks[b].ml_al, i, x.pos, x.states, y.pos, y.states);
                        break;
                    }
                }
        }
#endif
    } else {
        const char *k2e = hook_env("NAFGPU_K2_LANES");        // measurements only (nafgpu_test_hooks)
        const uint32_t forced = k2e ? static_cast<uint32_t>(std::atoi(k2e)) : 0u;
        uint32_t lanes = forced ? forced : (n_blocks + 359u) / 360u;     // about one wave per CU ...
        lanes = forced ? lanes : (lanes < 16 ? 16 : lanes);              // ... of at least 16 lanes (measured: see above)
        lanes = lanes < 1 ? 1 : (lanes > 64 ? 64 : lanes);
        hipLaunchKernelGGL(k_seq_states, dim3((n_blocks + lanes - 1) / lanes), dim3(64), 0, stream, src, blocks, n_blocks, cells, recs,
                           lanes, status);
    }
    hipLaunchKernelGGL(k_seq_values, dim3(n_blocks), dim3(64), cells_cap * 8u, stream, src, blocks, n_blocks, cells, recs, seqs, blk_size,
                       rep_final, cells_cap, status);
}

size_t scan_tmp_bytes(uint64_t n) { return static_cast<size_t>((n + kScanTile - 1) / kScanTile + 1) * sizeof(TileAgg); }

template <int MODE>
static void scan_generic(hipStream_t stream, const uint8_t *in, uint64_t n, uint64_t *out, uint64_t cap, void *tile_tmp,
                         ScanTotals *totals, uint32_t *status) {
    const uint64_t n_tiles = (n + kScanTile - 1) / kScanTile;
    TileAgg *tiles = static_cast<TileAgg *>(tile_tmp);
    if (n_tiles)
        hipLaunchKernelGGL(k_scan_reduce<MODE>, dim3(static_cast<uint32_t>(n_tiles)), dim3(kScanThreads), 0, stream, in, n,
                           tiles);
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(kScanThreads), 0, stream, tiles, n_tiles, totals);
    if (n_tiles)
        hipLaunchKernelGGL(k_scan_emit<MODE>, dim3(static_cast<uint32_t>(n_tiles)), dim3(kScanThreads), 0, stream, in, n,
                           tiles, out, cap, status);
}

void launch_scan_blocks(hipStream_t stream, const uint32_t *blk_size, uint64_t n, uint64_t *blk_base, void *tile_tmp,
                        uint64_t expect_total, uint32_t *status) {
    // totals live right behind the tile aggregates
    TileAgg *tiles = static_cast<TileAgg *>(tile_tmp);
    ScanTotals *totals = reinterpret_cast<ScanTotals *>(tiles + (n + kScanTile - 1) / kScanTile);
    scan_generic<kModeExcl>(stream, reinterpret_cast<const uint8_t *>(blk_size), n, blk_base, n, tile_tmp, totals, status);
    hipLaunchKernelGGL(k_scan_finish_blocks, dim3(1), dim3(1), 0, stream, blk_base, n, totals, expect_total, status);
}

void launch_scan_runs_u32(hipStream_t stream, const uint8_t *words, uint64_t n_words, uint64_t *ends, uint64_t cap,
                          void *tile_tmp, ScanTotals *totals, uint32_t *status) {
    scan_generic<kModeRunsU32>(stream, words, n_words, ends, cap, tile_tmp, totals, status);
}

void launch_scan_runs_u8(hipStream_t stream, const uint8_t *bytes, uint64_t n_bytes, uint64_t *ends, uint64_t cap,
                         void *tile_tmp, ScanTotals *totals, uint32_t *status) {
    scan_generic<kModeRunsU8>(stream, bytes, n_bytes, ends, cap, tile_tmp, totals, status);
}

void launch_scan_nul(hipStream_t stream, const uint8_t *bytes, uint64_t n_bytes, uint64_t *ends, uint64_t cap, void *tile_tmp,
                     ScanTotals *totals, uint32_t *status) {
    scan_generic<kModeNul>(stream, bytes, n_bytes, ends, cap, tile_tmp, totals, status);
}

void launch_scan_excl_u64(hipStream_t stream, const uint64_t *items, uint64_t n, uint64_t *out, void *tile_tmp, ScanTotals *totals,
                          uint32_t *status) {
    scan_generic<kModeExclU64>(stream, reinterpret_cast<const uint8_t *>(items), n, out, n, tile_tmp, totals, status);
}

void launch_utf8_check(hipStream_t stream, const uint8_t *p, uint64_t n, uint32_t *flags, uint32_t bit) {
    if (!n) return;
    uint64_t blocks = (n / 16 + 255) / 256;
    if (blocks > 256u * 8u) blocks = 256u * 8u;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(k_utf8_check, dim3(static_cast<uint32_t>(blocks)), dim3(256), 0, stream, p, n, flags, bit);
}

void launch_fmt_sizes(hipStream_t stream, const FmtText &t, uint64_t *sizes) {
    if (!t.n_rec) return;
    FmtArgs a{t.seq, t.qual, t.rec_end, t.ids, t.id_end, t.n_ids, t.com, t.com_end, t.n_com, t.n_rec, t.line_length, t.sep, 0};
    uint64_t blocks = (t.n_rec + 255) / 256;
    if (blocks > 256u * 8u) blocks = 256u * 8u;
    hipLaunchKernelGGL(k_fmt_sizes, dim3(static_cast<uint32_t>(blocks)), dim3(256), 0, stream, a, sizes);
}

void launch_fmt_write(hipStream_t stream, const FmtText &t, const uint64_t *off, uint64_t n_text, uint8_t *text) {
    if (!t.n_rec || !n_text) return;
    FmtArgs a{t.seq, t.qual, t.rec_end, t.ids, t.id_end, t.n_ids, t.com, t.com_end, t.n_com, t.n_rec, t.line_length, t.sep, 0};
    uint64_t blocks = (n_text + kFmtChunk - 1) / kFmtChunk;
    if (blocks > 256u * 8u) blocks = 256u * 8u;
    hipLaunchKernelGGL(k_fmt_write, dim3(static_cast<uint32_t>(blocks)), dim3(256), 0, stream, a, off, n_text, text);
}

void launch_copy_fill(hipStream_t stream, const uint8_t *src, const CopyTask *tasks, uint32_t n_tasks,
                      const uint64_t *blk_base, uint8_t *out, uint8_t *lit, bool ascii, uint32_t t_char,
                      uint32_t *status) {
    if (!n_tasks) return;
    if (ascii)
        hipLaunchKernelGGL(k_copy_fill<true>, dim3(n_tasks, kBlockMax / kCopySlice), dim3(256), 0, stream, src, tasks, blk_base, out, lit, t_char,
                           status);
    else
        hipLaunchKernelGGL(k_copy_fill<false>, dim3(n_tasks, kBlockMax / kCopySlice), dim3(256), 0, stream, src, tasks, blk_base, out, lit, t_char,
                           status);
}

// Streams in parts (plan.h: HufStream::sub): where the parts begin, for ALL the tasks of the selection at once -- one launch
// each of k_huf_sync and k_huf_bounds in front of the classes' decodes (a launch per class would be a part's serial decode
// per class, one after the other).
void launch_huf_parts(hipStream_t stream, const uint8_t *src, const HufTask *tasks, uint32_t n_tasks, const HufTblCopy *copies,
                      HufStream *streams, const uint16_t *pool, HufSync *sync, uint32_t sync_lds, uint32_t *status) {
    if (!n_tasks || !sync) return;
    const uint32_t sl = (sync_lds + 15u) & ~15u;
    hipLaunchKernelGGL(k_huf_sync, dim3(n_tasks), dim3(64 * kHufSyncSpread), sl, stream, src, tasks, copies, streams, pool, sync, status);
    hipLaunchKernelGGL(k_huf_bounds, dim3(n_tasks), dim3(64), sl, stream, src, tasks, copies, streams, pool, sync, status);
}

void launch_huf_decode(hipStream_t stream, const uint8_t *src, const HufTask *tasks, const HufClass &cls,
                       const HufTblCopy *copies, const HufStream *streams, const uint16_t *pool,
                       const uint64_t *blk_base, uint8_t *out, uint8_t *lit, const SeqBlock *seq_blocks, const Seq *seqs,
                       const uint8_t *dicts, bool ascii, uint32_t t_char, uint32_t *status) {
    if (!cls.n_tasks) return;
    const uint32_t lds = (cls.lds_bytes + 15u) & ~15u;
    const bool a = ascii && !cls.to_lit;                   // the literal buffer always holds packed bytes
    const HufTask *t0 = tasks + cls.first_task;
#define NAFGPU_LAUNCH_HUF(A, T, S)                                                                                            \
    hipLaunchKernelGGL((k_huf_decode<A, T, S>), dim3(cls.n_tasks), dim3(64), lds, stream, src, t0, copies, streams, pool,     \
                       blk_base, out, lit, seq_blocks, seqs, dicts, t_char, status)
#define NAFGPU_LAUNCH_HUF_T(T)                       \
    do {                                             \
        if (a && cls.seg)                            \
            NAFGPU_LAUNCH_HUF(true, T, true);        \
        else if (a)                                  \
            NAFGPU_LAUNCH_HUF(true, T, false);       \
        else if (cls.seg)                            \
            NAFGPU_LAUNCH_HUF(false, T, true);       \
        else                                         \
            NAFGPU_LAUNCH_HUF(false, T, false);      \
    } while (0)
    if (cls.tbl == kTblBaked)
        NAFGPU_LAUNCH_HUF_T(kTblBaked);
    else if (cls.tbl == kTblCompact)
        NAFGPU_LAUNCH_HUF_T(kTblCompact);
    else
        NAFGPU_LAUNCH_HUF_T(kTblDict);
#undef NAFGPU_LAUNCH_HUF_T
#undef NAFGPU_LAUNCH_HUF
}

// largest distance a pointer-jumping element may hold (32-bit D, the top 256 values mark final elements);
// NAFGPU_PJ_MAX_DIST lowers it so that a test can reach the "chain longer than D can express" fallback on a small input
static uint32_t pj_max_dist() {
    const char *e = hook_env("NAFGPU_PJ_MAX_DIST");        // read per call: a test switches it inside one process
    const uint32_t v = e ? static_cast<uint32_t>(std::strtoul(e, nullptr, 10)) : kPjWait;
    return v < kPjWait ? v : kPjWait;
}

template <bool ASCII>
static void lz_execute(hipStream_t stream, const LzArgs &a) {
    // a.phase (the shard protocol, engine.cpp): 0 = everything.  1 = the window in front of this shard (the pseudo block's
    // a.halo_wait elements) has not arrived: everything that does not depend on it -- whatever copies from it, directly or
    // not, stays pending, the frame-order walk does not run -- and a look at whether the last a.tail_elems elements (the next
    // shard's window) are final.  2 = it has arrived (same buffers): the pending rest, then the frame-order walk.
    const bool wait = a.phase == 1 && a.halo_wait != 0, finish = a.phase == 2;
    const uint32_t grid = a.n_blocks < 256u * 8u ? a.n_blocks : 256u * 8u;
    const uint64_t *n_total = a.blk_base + a.n_sel_blocks;   // (on the device) elements of the selection, pseudo block included
    if (!finish) {
        const uint32_t n_chunks = (a.n_blocks + kRepChunk - 1) / kRepChunk;
        uint32_t *partial = a.rep_scratch, *chunk_init = a.rep_scratch + 3 * static_cast<size_t>(n_chunks);
        hipLaunchKernelGGL(k_rep_partial, dim3((n_chunks + 255) / 256), dim3(256), 0, stream, a.blocks, a.n_blocks, a.rep_final, partial,
                           a.rep_continues, a.status);
        hipLaunchKernelGGL(k_rep_scan, dim3(1), dim3(1), 0, stream, n_chunks, partial, chunk_init, a.rep_carry[0], a.rep_carry[1],
                           a.rep_carry[2], static_cast<uint32_t *>(nullptr), a.status);
        hipLaunchKernelGGL(k_rep_apply, dim3((n_chunks + 255) / 256), dim3(256), 0, stream, a.blocks, a.n_blocks, a.rep_final,
                           chunk_init, a.rep_init, a.rep_continues, a.rep_out, a.status);
        if (!a.pj_dist)
            hipLaunchKernelGGL(k_lz_literals<ASCII>, dim3(a.n_blocks), dim3(256), 0, stream, a.blocks, a.seqs, a.lit, a.blk_base,
                               a.meta, a.blk_pending, a.out, a.t_char, a.status);
    }
    if (a.pj_dist) {
        // ---- dense: every element learns its source distance (the same walk puts the literals in place), then the frame
        // is swept (see k_pj_sweep)
        const uint32_t max_dist = pj_max_dist();
        // look-ups per element and pass (k_pj_sweep); and whether the FIRST sweep already lists what it leaves pending
        uint32_t hops = a.strips ? kPjHops : kPjHopsDeep, first_list = a.shallow;
        // the sweeps write what they make final to the output themselves (no k_pj_emit pass behind them)
        bool fused_emit = true;
        if (const char *e = hook_env("NAFGPU_PJ_EMIT")) fused_emit = e[0] != '1';     // 1: the separate pass (measurements)
        // tiles follow the XCDs where most sources are final early (half the elements or more are literals: level-3 DNA 6.05 ->
        // 5.81 ms); where chains are deep it loses (FASTQ-like level 3 63.3 -> 66.4 ms), level 1 is indifferent (profiles/r04_pj_xcd_probe.log)
        uint32_t xcd_map = a.shallow ? 1u : 0u;
        if (const char *e = hook_env("NAFGPU_PJ_XCD")) xcd_map = e[0] == '0' ? 0u : 1u;
        if (const char *e = hook_env("NAFGPU_PJ_HOPS")) hops = static_cast<uint32_t>(std::atoi(e)) > 0 ? static_cast<uint32_t>(std::atoi(e)) : 1u;
        if (const char *e = hook_env("NAFGPU_PJ_FIRST_LIST")) first_list = e[0] == '1' ? 1u : 0u;
        uint64_t tiles = (a.n_elems + kPjTile - 1) / kPjTile;
        // strips: at least 32 tiles each where there are that many; three workgroups per CU, four rounds of them at most
        uint64_t strips = tiles / 32 ? tiles / 32 : 1;
        if (strips > 256u * 20u) strips = 256u * 20u;
        if (tiles > 256u * 16u) tiles = 256u * 16u;
        unsigned long long *pcount = a.counters + 4;       // [4..6]: pending elements, rotating (k_pj_sweep)
        unsigned long long *lstate = a.counters + 10;      // [10..12]: lengths of the pending lists, rotating; [13]: listing since sweep ...
        uint32_t *lists[2] = {a.pj_list[0], a.pj_list[1]};
        const bool can_list = lists[0] && lists[1] && a.n_elems < (1ull << 32);
        uint64_t lg = (a.pj_list_cap / 8 + 255) / 256;     // (a list is at most pj_list_cap long: eight entries per thread and round at that size)
        const uint32_t list_grid = static_cast<uint32_t>(lg < 1 ? 1 : (lg > 256u * 8u ? 256u * 8u : lg));
        uint64_t eg = (a.n_elems / 4 + 255) / 256;
        if (eg > 256u * 16u) eg = 256u * 16u;
        if (eg == 0) eg = 1;
        uint32_t sweep0 = 1, sweep1 = kPjSweeps;
        if (!finish) {
            if (wait)
                hipLaunchKernelGGL(k_fill_u32, dim3(static_cast<uint32_t>(std::min<uint64_t>((a.halo_wait + 255) / 256, 2048))), dim3(256), 0, stream,
                                   a.pj_dist, a.halo_wait, kPjWait);
            // (every word of D is written by k_pj_fill: no memset in front)
            hipLaunchKernelGGL(k_pj_fill<ASCII>, dim3(grid), dim3(256), 0, stream, a.blocks, a.n_blocks, a.seqs, a.rep_init, a.blk_base,
                               a.pj_dist, a.lit, a.blk_pending, a.t_char, a.n_sel_blocks, a.n_elems, wait ? a.halo_wait : uint64_t(0), a.status);
            (void)hipMemsetAsync(lstate, 0, 5 * sizeof(unsigned long long), stream);
        } else {
            // the window is in the output buffer now: its elements become final words, and a few more sweeps -- chains were
            // jumped down to their first element inside the window while it was away -- finish what waited for them
            if (a.halo_wait)
                hipLaunchKernelGGL(k_pj_halo_words<ASCII>, dim3(static_cast<uint32_t>(std::min<uint64_t>((a.halo_wait + 255) / 256, 2048))), dim3(256), 0,
                                   stream, a.pj_dist, a.out, a.halo_wait);
            sweep0 = kPjSweeps + 1;
            sweep1 = kPjSweeps + kPjFinishSweeps;
        }
        for (uint32_t sweep = sweep0; sweep <= sweep1; sweep++) {
            // (shallow chains: strip-wise, see k_pj_sweep)
            if (a.strips && sweep <= kPjStripSweeps)
                hipLaunchKernelGGL((k_pj_sweep<ASCII, kPjWin>), dim3(static_cast<uint32_t>(strips)), dim3(256), 0, stream, a.pj_dist, a.out,
                                   a.pj_tiles, pcount, a.n_elems, sweep, max_dist, can_list ? lists[sweep & 1u] : nullptr, a.pj_list_cap, lstate,
                                   first_list, hops, fused_emit ? (sweep == 1 ? 1u : 2u) : 0u, xcd_map, a.status);
            else
                hipLaunchKernelGGL((k_pj_sweep<ASCII, 0u>), dim3(static_cast<uint32_t>(tiles)), dim3(256), 0, stream, a.pj_dist, a.out,
                                   a.pj_tiles, pcount, a.n_elems, sweep, max_dist, can_list ? lists[sweep & 1u] : nullptr, a.pj_list_cap,
                                   lstate, first_list, hops, fused_emit ? (sweep == 1 ? 1u : 2u) : 0u, xcd_map, a.status);
            if (can_list && sweep >= 2)
                hipLaunchKernelGGL(k_pj_list<ASCII>, dim3(list_grid), dim3(256), 0, stream, a.pj_dist, a.out, pcount,
                                   lists[(sweep & 1u) ^ 1u], lists[sweep & 1u], a.pj_list_cap, lstate, sweep, max_dist, (finish || fused_emit) ? 1u : 0u, hops, a.status);
        }
        // (finish: the list passes wrote the output themselves; sweeps over all of D did not)
        if (!fused_emit)
        hipLaunchKernelGGL(k_pj_emit<ASCII>, dim3(static_cast<uint32_t>(eg)), dim3(256), 0, stream, a.pj_dist, a.out, a.n_elems,
                           finish ? static_cast<const unsigned long long *>(lstate + 3) : static_cast<const unsigned long long *>(nullptr),
                           static_cast<unsigned long long>(kPjSweeps), a.status);
        if (wait) {
            if (a.tail_elems)
                hipLaunchKernelGGL(k_pj_tail_check, dim3(64), dim3(256), 0, stream, can_list ? lists[kPjSweeps & 1u] : nullptr, lstate, kPjSweeps,
                                   a.pj_list_cap, a.pj_tiles, pcount, n_total, a.tail_elems, a.counters + kCtrTail);
            return;
        }
        // anything still pending (a distance that would not fit 32 bits): frame order
        hipLaunchKernelGGL(k_lz_matches_ordered<ASCII>, dim3(1), dim3(256), 0, stream, a.blocks, a.n_blocks, a.seqs,
                           static_cast<SeqMeta *>(nullptr), a.blk_pending, a.rep_init, a.blk_base, a.out, pcount + sweep1 % 3u, a.status);
        return;
    }
    // ---- sparse: matches one by one; pass 1 walks the blocks, the later ones the list of what is still pending
    const uint64_t halo_end = wait ? a.halo_wait : 0;
    const bool lists = a.plist[0] && a.plist[1];
    if (!finish) {
        if (a.cidx) {
            uint64_t ib = (a.n_idx_chunks + 255) / 256;
            if (ib > 256u * 16u) ib = 256u * 16u;
            hipLaunchKernelGGL(k_lz_index, dim3(static_cast<uint32_t>(ib)), dim3(256), 0, stream, a.meta, a.n_sequences, a.n_idx_chunks,
                               a.cidx, a.status);
        }
        if (!lists) {                                          // no memory for the pending lists: every pass walks the blocks
            for (uint32_t pass = 1; pass <= kLzPasses; pass++)
                hipLaunchKernelGGL(k_lz_match_pass<ASCII>, dim3(grid), dim3(256), 0, stream, a.blocks, a.n_blocks, a.seqs, a.meta,
                                   a.cidx, a.blk_pending, a.roff, a.counters, a.rep_init, a.blk_base, a.out, pass, nullptr,
                                   nullptr, halo_end, a.status);
            if (wait) (void)hipMemsetAsync(a.counters + kCtrTail, 0xFF, 8, stream);   // (not looked at: the next shard waits until this one is whole)
        } else {
            unsigned long long *lcount = a.counters + 4;       // [4..6]: lengths of the pending lists, rotating (k_lz_match_list)
            hipLaunchKernelGGL(k_lz_match_pass<ASCII>, dim3(grid), dim3(256), 0, stream, a.blocks, a.n_blocks, a.seqs, a.meta, a.cidx,
                               a.blk_pending, a.roff, a.counters, a.rep_init, a.blk_base, a.out, 1u, a.plist[0], lcount + 2u, halo_end, a.status);
            uint64_t lgrid = (a.n_sequences + 255) / 256;
            if (lgrid > 256u * 8u) lgrid = 256u * 8u;
            // a section of a few thousand sequences goes to the one-workgroup stage after the second pass: launches that find
            // nothing to do still cost 5 us each, and they were a third of the 350 launches a small FASTQ archive takes
            const uint32_t passes = (!wait && a.n_sequences <= kLzFewSequences) ? 2u : kLzPasses;
            // (what is still pending after three passes is a small fraction: the later passes -- grid-stride loops, most of them
            // with nothing left to do -- get an eighth of the workgroups, 5 us a launch instead of 15)
            const uint64_t lgrid_late = std::max<uint64_t>(std::min<uint64_t>(lgrid, 64), lgrid / 8);
            for (uint32_t pass = 2; pass <= passes; pass++)   // pass k reads list k & 1 (its length in lcount[k % 3]): pass 1 wrote list 0 / lcount[2]
                hipLaunchKernelGGL(k_lz_match_list<ASCII>, dim3(static_cast<uint32_t>(pass <= 3 ? lgrid : lgrid_late)), dim3(256), 0, stream, a.plist[pass & 1u],
                                   a.plist[(pass & 1u) ^ 1u], lcount, a.seqs, a.meta, a.cidx, a.blk_pending, a.roff, a.counters, a.out,
                                   pass, halo_end, a.status);
            hipLaunchKernelGGL(k_lz_finish_small<ASCII>, dim3(1), dim3(1024), 0, stream, a.plist[0], a.plist[1], lcount, passes + 1u,
                               a.seqs, a.meta, a.cidx, a.blk_pending, a.roff, a.counters, a.out, halo_end, 0u, n_total,
                               wait ? a.tail_elems : 0, a.status);
        }
        if (wait) return;
    } else if (lists) {
        // the window has arrived: the survivors' list once more, without the limit
        hipLaunchKernelGGL(k_lz_finish_small<ASCII>, dim3(1), dim3(1024), 0, stream, a.plist[0], a.plist[1], a.counters + 4, 0u,
                           a.seqs, a.meta, a.cidx, a.blk_pending, a.roff, a.counters, a.out, uint64_t(0), 1u, n_total, uint64_t(0), a.status);
    }
    // deep chains of a sparse section: frame order (returns at once when nothing is pending)
    hipLaunchKernelGGL(k_lz_matches_ordered<ASCII>, dim3(1), dim3(256), 0, stream, a.blocks, a.n_blocks, a.seqs, a.meta,
                       a.blk_pending, a.rep_init, a.blk_base, a.out, a.counters, a.status);
}

void launch_rep_map(hipStream_t stream, const SeqBlock *blocks, uint32_t n_blocks, const uint32_t *rep_final, uint32_t *rep_scratch,
                    uint32_t continues, uint32_t *map_out, uint32_t *status) {
    if (!n_blocks) return;
    const uint32_t n_chunks = (n_blocks + kRepChunk - 1) / kRepChunk;
    uint32_t *partial = rep_scratch, *chunk_init = rep_scratch + 3 * static_cast<size_t>(n_chunks);
    hipLaunchKernelGGL(k_rep_partial, dim3((n_chunks + 255) / 256), dim3(256), 0, stream, blocks, n_blocks, rep_final, partial, continues, status);
    hipLaunchKernelGGL(k_rep_scan, dim3(1), dim3(1), 0, stream, n_chunks, partial, chunk_init, kRepToken | (0u << 24), kRepToken | (1u << 24),
                       kRepToken | (2u << 24), map_out, status);
}

uint64_t lz_pj_tiles(uint64_t n_elems) { return (n_elems + kPjTile - 1) / kPjTile; }

void launch_lz_execute(hipStream_t stream, const LzArgs &a, bool ascii) {
    if (!a.n_blocks) return;
    // [0] matches still pending (sparse), [1] matches the launched passes left to the one-workgroup stage,
    // [4..6] rotating counters of the stage in use (k_lz_match_list / k_pj_sweep); set on the device: per-launch
    // constants do not go through host storage that the next section's launch would overwrite
    if (a.phase != 2)
        hipLaunchKernelGGL(k_lz_init_counters, dim3(1), dim3(32), 0, stream, a.counters, static_cast<unsigned long long>(a.n_sequences),
                           a.pj_dist ? 1ull : 0ull);   // [4] = "pending before the first sweep": anything but 0
    if (ascii)
        lz_execute<true>(stream, a);
    else
        lz_execute<false>(stream, a);
}

void launch_mask_apply(hipStream_t stream, uint8_t *ascii, uint64_t n_bases, uint64_t lo_clamp, uint64_t hi_clamp,
                       const uint64_t *mask_ends, const ScanTotals *mask_totals, const uint64_t *rec_ends,
                       const ScanTotals *rec_totals, uint64_t max_runs, int spec_mask, uint32_t *status) {
    if (!n_bases || !max_runs || hi_clamp <= lo_clamp) return;
    const uint64_t masked_runs = (max_runs + 1) / 2;
    uint64_t blocks = (masked_runs + 255) / 256;
    if (blocks > 256u * 8u) blocks = 256u * 8u;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(k_mask_apply, dim3(static_cast<uint32_t>(blocks)), dim3(256), 0, stream, ascii, n_bases, lo_clamp,
                       hi_clamp, mask_ends, mask_totals, rec_ends, rec_totals, spec_mask, status);
}

void launch_xxh64_frames(hipStream_t stream, const XxhSeg *segs, uint32_t n_segs, const uint64_t *blk_base, const uint8_t *out,
                         bool ascii, uint32_t t_char, const XxhCarry *carry_in, XxhCarry *carry_out, uint32_t *status) {
    if (!n_segs) return;
    hipLaunchKernelGGL(k_xxh64_frames, dim3(n_segs), dim3(64), 0, stream, segs, blk_base, out, ascii ? 1u : 0u, t_char, carry_in,
                       carry_out, status);
}

// Device -> pinned host memory by a kernel (k_copy_out): see ArchiveJob::copy_to_pinned.
void launch_copy_out(hipStream_t stream, uint8_t *dst_pinned, const uint8_t *d_src, uint64_t n) {
    if (!n) return;
    uint64_t blocks = (n / 16 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(k_copy_out, dim3(static_cast<uint32_t>(blocks)), dim3(256), 0, stream, dst_pinned, d_src, n);
}

void launch_hash64(hipStream_t stream, const uint8_t *p, uint64_t n, uint64_t first_chunk, unsigned long long *result) {
    if (!n) return;
    uint64_t blocks = (n + kHashChunk - 1) / kHashChunk;   // 4 KiB (two words per thread) per workgroup step
    if (blocks > 256u * 32u) blocks = 256u * 32u;
    hipLaunchKernelGGL(k_hash64, dim3(static_cast<uint32_t>(blocks)), dim3(256), 0, stream, p, n, first_chunk, result);
}

}  // namespace nafgpu
