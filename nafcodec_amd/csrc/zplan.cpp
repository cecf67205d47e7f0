// zplan.cpp -- see zplan.h.  Host-side frame walk + table build, once per block.
#include "zplan.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace nafgpu {
namespace {

struct Fail {
    std::string msg;
    bool truncated = false;
};

inline int highbit(uint32_t v) { return 31 - __builtin_clz(v); }

// ---------------------------------------------------------------- bit readers
struct FwdBits {            // LSB-first, used by FSE table descriptions
    const uint8_t *p;
    size_t n;
    size_t bit = 0;
    uint32_t peek(int nb) const {
        uint64_t v = 0;
        size_t byte = bit >> 3;
        size_t avail = byte < n ? std::min<size_t>(8, n - byte) : 0;
        std::memcpy(&v, p + byte, avail);
        return static_cast<uint32_t>((v >> (bit & 7)) & ((1ull << nb) - 1));
    }
};

struct BackBits {           // MSB-first from the end; only the tiny Huffman-weight streams use it here
    const uint8_t *p = nullptr;
    size_t n = 0;
    int64_t pos = 0;        // unread bits below the cursor
    bool init(const uint8_t *src, size_t len) {
        if (len == 0 || src[len - 1] == 0) return false;
        p = src;
        n = len;
        pos = static_cast<int64_t>(len - 1) * 8 + highbit(src[len - 1]);
        return true;
    }
    uint32_t read(int nb) {
        pos -= nb;
        uint32_t v = 0;
        for (int k = nb - 1; k >= 0; k--) {         // weight streams are <= 127 bytes: bitwise is fine
            int64_t b = pos + k;
            uint32_t bit = b >= 0 ? (p[b >> 3] >> (b & 7)) & 1u : 0u;
            v = (v << 1) | bit;
        }
        return v;
    }
};

// ---------------------------------------------------------------- FSE
// App. B "FSE table description".  Returns bytes consumed, or -1.
long read_fse_dist(const uint8_t *src, size_t n, int max_al, int max_sym, int16_t *norm, int *nsym, int *al_out,
                   Fail &f) {
    if (n == 0) {
        f = {"FSE table description is empty", true};
        return -1;
    }
    FwdBits b{src, n};
    int al = static_cast<int>(b.peek(4)) + 5;
    b.bit += 4;
    if (al > max_al) {
        f.msg = "FSE accuracy log too large";
        return -1;
    }
    int remaining = 1 << al, sym = 0;
    while (remaining > 0 && sym <= max_sym) {
        int max = remaining + 1;
        int bits = highbit(static_cast<uint32_t>(max)) + 1;
        uint32_t v = b.peek(bits);
        uint32_t low = (1u << (bits - 1)) - 1;
        uint32_t thr = (1u << bits) - 1 - static_cast<uint32_t>(max);
        if ((v & low) < thr) {
            b.bit += bits - 1;
            v &= low;
        } else {
            b.bit += bits;
            if (v > low) v -= thr;
        }
        int p = static_cast<int>(v) - 1;
        norm[sym++] = static_cast<int16_t>(p);
        remaining -= p < 0 ? -p : p;
        if (p == 0) {
            for (;;) {
                uint32_t rep = b.peek(2);
                b.bit += 2;
                for (uint32_t k = 0; k < rep && sym <= max_sym; k++) norm[sym++] = 0;
                if (rep != 3) break;
            }
        }
        if ((b.bit >> 3) > n) {
            f = {"FSE table description runs past its section", true};
            return -1;
        }
    }
    if (remaining != 0) {
        f.msg = "FSE probabilities do not sum to the table size";
        return -1;
    }
    size_t used = (b.bit + 7) >> 3;
    if (used > n) {
        f = {"FSE table description runs past its section", true};
        return -1;
    }
    *nsym = sym;
    *al_out = al;
    return static_cast<long>(used);
}

struct FseStates {          // generic decode table: parallel arrays of 1 << al states
    int al = 0;
    uint8_t sym[512];
    uint8_t nb[512];
    uint16_t base[512];
};

// App. B "FSE table build"
bool build_fse(const int16_t *norm, int nsym, int al, FseStates *t) {
    const int S = 1 << al;
    uint16_t next[256];
    int high = S - 1;
    t->al = al;
    for (int s = 0; s < nsym; s++) {
        if (norm[s] == -1) {
            t->sym[high--] = static_cast<uint8_t>(s);
            next[s] = 1;
        } else {
            next[s] = static_cast<uint16_t>(norm[s]);
        }
    }
    const int step = (S >> 1) + (S >> 3) + 3, mask = S - 1;
    int pos = 0;
    for (int s = 0; s < nsym; s++)
        for (int k = 0; k < norm[s]; k++) {
            t->sym[pos] = static_cast<uint8_t>(s);
            do pos = (pos + step) & mask;
            while (pos > high);
        }
    if (pos != 0) return false;
    for (int i = 0; i < S; i++) {
        uint16_t d = next[t->sym[i]]++;
        int nb = al - highbit(d);
        t->nb[i] = static_cast<uint8_t>(nb);
        t->base[i] = static_cast<uint16_t>((d << nb) - S);
    }
    return true;
}

// ---------------------------------------------------------------- sequence code tables (App. B)
const int16_t kLLDefault[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2,
                                2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
const int16_t kMLDefault[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
const int16_t kOFDefault[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
const uint32_t kLLBase[36] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  9,   10,  11,  12,   13,   14,   15,   16,    18,
                              20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536};
const uint8_t kLLBits[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1,
                             1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
const uint32_t kMLBase[53] = {3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15,  16,  17,  18,   19,   20,
                              21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33,  34,  35,  37,   39,   41,
                              43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539};
const uint8_t kMLBits[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                             0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};

enum SeqKind { kLL = 0, kOF = 1, kML = 2 };
struct KindInfo {
    const int16_t *def;
    int def_n, def_al, max_al, max_sym;
};
const KindInfo kKinds[3] = {{kLLDefault, 36, 6, 9, 35}, {kOFDefault, 29, 5, 8, 31}, {kMLDefault, 53, 6, 9, 52}};

bool bake_cell(SeqKind kind, uint8_t code, SeqCell *c) {
    switch (kind) {
    case kLL:
        if (code > 35) return false;
        c->base_value = kLLBase[code];
        c->extra_bits = kLLBits[code];
        return true;
    case kML:
        if (code > 52) return false;
        c->base_value = kMLBase[code];
        c->extra_bits = kMLBits[code];
        return true;
    default:
        if (code > 31) return false;
        c->base_value = 1u << code;
        c->extra_bits = code;
        return true;
    }
}

struct TableRef {
    uint32_t off = 0;
    uint8_t al = 0;
    bool valid = false;
};

bool append_seq_table(ZPlan *plan, SeqKind kind, const FseStates &t, TableRef *ref) {
    const int S = 1 << t.al;
    ref->off = static_cast<uint32_t>(plan->fse_pool.size());
    ref->al = static_cast<uint8_t>(t.al);
    for (int i = 0; i < S; i++) {
        SeqCell c{};
        c.next_base = t.base[i];
        c.nb = t.nb[i];
        if (!bake_cell(kind, t.sym[i], &c)) return false;
        plan->fse_pool.push_back(c);
    }
    ref->valid = true;
    return true;
}

// ---------------------------------------------------------------- Huffman (App. B "Huffman tree description")
using HufRef = HufTableRef;

// Parses the tree description at src, appends the 2^max_bits decode table to the pool.
long read_huf_table(const uint8_t *src, size_t n, ZPlan *plan, HufRef *ref, Fail &f) {
    uint8_t w[258];
    int nw = 0;
    if (n < 1) {
        f = {"Huffman tree description is empty", true};
        return -1;
    }
    const int hb = src[0];
    size_t used;
    if (hb >= 128) {                                  // direct 4-bit weights, high nibble first
        nw = hb - 127;
        size_t bytes = static_cast<size_t>(nw + 1) / 2;
        if (1 + bytes > n) {
            f = {"Huffman weights run past the literals section", true};
            return -1;
        }
        for (int i = 0; i < nw; i++) {
            uint8_t b = src[1 + i / 2];
            w[i] = (i & 1) ? (b & 0xF) : (b >> 4);
        }
        used = 1 + bytes;
    } else {                                          // FSE-compressed weights, two interleaved states
        if (hb == 0 || static_cast<size_t>(hb) + 1 > n) {
            f = {"Huffman weights run past the literals section", hb != 0};
            return -1;
        }
        int16_t norm[256];
        int nsym = 0, al = 0;
        long r = read_fse_dist(src + 1, static_cast<size_t>(hb), 6, 255, norm, &nsym, &al, f);
        if (r < 0) return -1;
        FseStates t;
        if (!build_fse(norm, nsym, al, &t)) {
            f.msg = "bad FSE table for Huffman weights";
            return -1;
        }
        BackBits b;
        if (!b.init(src + 1 + r, static_cast<size_t>(hb) - static_cast<size_t>(r))) {
            f.msg = "Huffman weight stream has no end mark";
            return -1;
        }
        uint32_t s1 = b.read(al), s2 = b.read(al);
        for (;;) {
            if (nw >= 255) break;
            w[nw++] = t.sym[s1];
            if (b.pos < static_cast<int64_t>(t.nb[s1])) {       // stream exhausted: flush the other state
                w[nw++] = t.sym[s2];
                break;
            }
            s1 = t.base[s1] + b.read(t.nb[s1]);
            if (nw >= 255) break;
            w[nw++] = t.sym[s2];
            if (b.pos < static_cast<int64_t>(t.nb[s2])) {
                w[nw++] = t.sym[s1];
                break;
            }
            s2 = t.base[s2] + b.read(t.nb[s2]);
        }
        used = 1 + static_cast<size_t>(hb);
    }
    if (nw > 255) {
        f.msg = "too many Huffman weights";
        return -1;
    }
    uint32_t total = 0;
    for (int i = 0; i < nw; i++) {
        if (w[i] > 11) {
            f.msg = "Huffman weight above 11";
            return -1;
        }
        if (w[i]) total += 1u << (w[i] - 1);
    }
    if (total == 0) {
        f.msg = "all Huffman weights are zero";
        return -1;
    }
    const int max_bits = highbit(total) + 1;
    const uint32_t left = (1u << max_bits) - total;
    if (max_bits > 11 || (left & (left - 1))) {
        f.msg = "Huffman weights do not complete a tree";
        return -1;
    }
    w[nw++] = static_cast<uint8_t>(highbit(left) + 1);
    // table fill: weight 1 (longest codes) first, ascending symbol within a weight
    const uint32_t size = 1u << max_bits;
    ref->pool_off = static_cast<uint32_t>(plan->huf_pool.size());
    ref->max_bits = static_cast<uint8_t>(max_bits);
    ref->valid = true;
    ref->n_syms = 0;
    for (int i = 0; i < nw; i++) ref->n_syms += w[i] != 0;
    plan->huf_pool.resize(plan->huf_pool.size() + size);
    uint16_t *tbl = plan->huf_pool.data() + ref->pool_off;
    uint32_t pos = 0;
    for (int wt = 1; wt <= max_bits; wt++) {
        const uint32_t span = 1u << (wt - 1);
        const uint16_t len = static_cast<uint16_t>((max_bits + 1 - wt) << 8);
        for (int s = 0; s < nw; s++) {
            if (w[s] != wt) continue;
            if (pos + span > size) {
                f.msg = "Huffman table overflow";
                return -1;
            }
            for (uint32_t k = 0; k < span; k++) tbl[pos + k] = static_cast<uint16_t>(len | s);
            pos += span;
        }
    }
    if (pos != size) {
        f.msg = "Huffman table underflow";
        return -1;
    }
    plan->n_huf_tables++;
    return static_cast<long>(used);
}

// ---------------------------------------------------------------- the walk
struct Walker {
    const uint8_t *p;
    size_t n;
    ZPlan *plan;
    Fail fail;
    TableRef predefined[3];

    bool need(size_t at, size_t k, const char *what) {
        if (at > n || k > n - at) {
            fail = {std::string("payload ends inside ") + what, true};
            return false;
        }
        return true;
    }
    bool bad(const char *what) {
        fail.msg = what;
        return false;
    }

    bool seq_table(SeqKind kind, int mode, size_t &i, size_t end, TableRef *cur) {
        const KindInfo &ki = kKinds[kind];
        switch (mode) {
        case 0: {                                                // Predefined
            if (!predefined[kind].valid) {
                FseStates t;
                if (!build_fse(ki.def, ki.def_n, ki.def_al, &t)) return bad("internal: predefined table");
                if (!append_seq_table(plan, kind, t, &predefined[kind])) return bad("internal: predefined table");
            }
            *cur = predefined[kind];
            return true;
        }
        case 1: {                                                // RLE: one state, zero bits
            if (i >= end) {
                fail = {"payload ends inside a sequences header", false};
                return false;
            }
            SeqCell c{};
            if (!bake_cell(kind, p[i], &c)) return bad("RLE sequence code out of range");
            cur->off = static_cast<uint32_t>(plan->fse_pool.size());
            cur->al = 0;
            cur->valid = true;
            plan->fse_pool.push_back(c);
            i += 1;
            return true;
        }
        case 2: {                                                // FSE description
            int16_t norm[256];
            int nsym = 0, al = 0;
            long r = read_fse_dist(p + i, end - i, ki.max_al, ki.max_sym, norm, &nsym, &al, fail);
            if (r < 0) {
                fail.truncated = false;                          // inside a complete block: corrupt, not short
                return false;
            }
            FseStates t;
            if (!build_fse(norm, nsym, al, &t)) return bad("bad FSE table in a sequences section");
            if (!append_seq_table(plan, kind, t, cur)) return bad("sequence code out of range");
            i += static_cast<size_t>(r);
            return true;
        }
        default:                                                 // Repeat
            if (!cur->valid) return bad("repeat mode without a previous table");
            return true;
        }
    }

    // one Compressed block occupying [i, end)
    bool compressed_block(size_t i, size_t end, uint32_t blk, uint32_t frame_first_blk, HufRef *huf,
                          TableRef seqtbl[3]) {
        if (i >= end) return bad("empty compressed block");
        // ---- literals section header
        const int type = p[i] & 3, sf = (p[i] >> 2) & 3;
        size_t regen = 0, comp = 0, hdr = 0;
        int nstreams = 0;
        if (type <= 1) {
            if (sf == 0 || sf == 2) {
                regen = p[i] >> 3;
                hdr = 1;
            } else if (sf == 1) {
                if (end - i < 2) return bad("literals header cut short");
                regen = (p[i] >> 4) + (static_cast<size_t>(p[i + 1]) << 4);
                hdr = 2;
            } else {
                if (end - i < 3) return bad("literals header cut short");
                regen = (p[i] >> 4) + (static_cast<size_t>(p[i + 1]) << 4) + (static_cast<size_t>(p[i + 2]) << 12);
                hdr = 3;
            }
            comp = type == 0 ? regen : 1;
        } else {
            if (sf <= 1) {
                if (end - i < 3) return bad("literals header cut short");
                uint32_t v = p[i] | (uint32_t(p[i + 1]) << 8) | (uint32_t(p[i + 2]) << 16);
                regen = (v >> 4) & 0x3FF;
                comp = (v >> 14) & 0x3FF;
                hdr = 3;
                nstreams = sf == 0 ? 1 : 4;
            } else if (sf == 2) {
                if (end - i < 4) return bad("literals header cut short");
                uint32_t v = p[i] | (uint32_t(p[i + 1]) << 8) | (uint32_t(p[i + 2]) << 16) | (uint32_t(p[i + 3]) << 24);
                regen = (v >> 4) & 0x3FFF;
                comp = (v >> 18) & 0x3FFF;
                hdr = 4;
                nstreams = 4;
            } else {
                if (end - i < 5) return bad("literals header cut short");
                uint64_t v = p[i] | (uint64_t(p[i + 1]) << 8) | (uint64_t(p[i + 2]) << 16) |
                             (uint64_t(p[i + 3]) << 24) | (uint64_t(p[i + 4]) << 32);
                regen = static_cast<size_t>((v >> 4) & 0x3FFFF);
                comp = static_cast<size_t>((v >> 22) & 0x3FFFF);
                hdr = 5;
                nstreams = 4;
            }
        }
        if (regen > kBlockMax) return bad("literals larger than a block");
        if (hdr + comp > end - i) return bad("literals section runs past its block");
        const size_t lit_at = i + hdr;               // first byte after the literals header
        size_t j = lit_at + comp;                    // sequences section
        // ---- sequences section header (needed first: it decides where literals go)
        if (j >= end) return bad("block has no sequences header");
        size_t nseq;
        if (p[j] == 0) {
            nseq = 0;
            j += 1;
        } else if (p[j] < 128) {
            nseq = p[j];
            j += 1;
        } else if (p[j] < 255) {
            if (end - j < 2) return bad("sequences header cut short");
            nseq = (static_cast<size_t>(p[j] - 128) << 8) + p[j + 1];
            j += 2;
        } else {
            if (end - j < 3) return bad("sequences header cut short");
            nseq = static_cast<size_t>(p[j + 1]) + (static_cast<size_t>(p[j + 2]) << 8) + 0x7F00;
            j += 3;
        }
        // A block with a few sequences and Huffman-coded literals (what real genomes give zstd level 1): the literal
        // streams go straight to their final positions, segment by segment (k_huf_decode reads the block's decoded
        // sequences); with many sequences the literals take the literal buffer and K4 scatters them.
        const bool seg = nseq > 0 && nseq <= kDirectSeqMax && type >= 2;
        const bool to_lit = nseq > 0 && !seg;
        const uint64_t lit_base = plan->lit_bytes;  // literal-buffer offset when to_lit
        const uint8_t lflag = to_lit ? 1 : (seg ? 2 : 0);
        const uint64_t seg_tag = seg ? (static_cast<uint64_t>(plan->seq_blocks.size()) + 1) << 32 : 0;
        // ---- literals tasks
        if (type == 0) {
            if (regen) plan->copies.push_back(CopyTask{lit_at, to_lit ? lit_base : 0, uint32_t(regen), blk, lflag, 0});
        } else if (type == 1) {
            if (regen) plan->copies.push_back(CopyTask{p[lit_at], to_lit ? lit_base : 0, uint32_t(regen), blk, uint32_t(lflag | 2), 0});
        } else {
            size_t t = lit_at, tend = lit_at + comp;
            if (type == 2) {
                long r = read_huf_table(p + t, tend - t, plan, huf, fail);
                if (r < 0) {
                    fail.truncated = false;
                    return false;
                }
                t += static_cast<size_t>(r);
            } else if (!huf->valid) {
                return bad("treeless literals without a previous Huffman table");
            }
            size_t sizes[4], counts[4];
            if (nstreams == 1) {
                sizes[0] = tend - t;
                counts[0] = regen;
            } else {
                if (tend - t < 6) return bad("jump table cut short");
                sizes[0] = p[t] | (size_t(p[t + 1]) << 8);
                sizes[1] = p[t + 2] | (size_t(p[t + 3]) << 8);
                sizes[2] = p[t + 4] | (size_t(p[t + 5]) << 8);
                t += 6;
                if (sizes[0] + sizes[1] + sizes[2] > tend - t) return bad("jump table exceeds the literals section");
                sizes[3] = tend - t - sizes[0] - sizes[1] - sizes[2];
                const size_t q = (regen + 3) / 4;
                if (3 * q > regen) return bad("too few literals for four streams");
                counts[0] = counts[1] = counts[2] = q;
                counts[3] = regen - 3 * q;
            }
            uint64_t dst = to_lit ? lit_base : seg_tag;
            for (int s = 0; s < nstreams; s++) {
                if (sizes[s] == 0) return bad("empty Huffman stream");
                t += sizes[s];
                HufStream hs{};
                hs.src_end = t;
                hs.dst = dst;
                hs.src_len = static_cast<uint32_t>(sizes[s]);
                hs.n_syms = static_cast<uint32_t>(counts[s]);
                hs.blk = blk;
                hs.max_bits = huf->max_bits;
                hs.flags = lflag;
                plan->streams.push_back(hs);
                plan->stream_ref.push_back(*huf);
                dst += counts[s];
            }
        }
        // ---- sequences
        if (nseq == 0) {
            if (j != end) return bad("bytes after an empty sequences section");
            plan->blk_size.push_back(static_cast<uint32_t>(regen));
            plan->known_out += regen;
            return true;
        }
        if (j >= end) return bad("sequences header cut short");
        const int modes = p[j++];
        if (modes & 3) return bad("reserved bits set in the sequences header");
        if (!seq_table(kLL, (modes >> 6) & 3, j, end, &seqtbl[kLL])) return false;
        if (!seq_table(kOF, (modes >> 4) & 3, j, end, &seqtbl[kOF])) return false;
        if (!seq_table(kML, (modes >> 2) & 3, j, end, &seqtbl[kML])) return false;
        if (j >= end) return bad("sequence bitstream is empty");
        if (p[end - 1] == 0) return bad("sequence bitstream has no end mark");
        SeqBlock sb{};
        sb.bits_off = j;
        sb.bits_len = static_cast<uint32_t>(end - j);
        sb.n_seq = static_cast<uint32_t>(nseq);
        sb.ll_tbl = seqtbl[kLL].off;
        sb.of_tbl = seqtbl[kOF].off;
        sb.ml_tbl = seqtbl[kML].off;
        sb.ll_al = seqtbl[kLL].al;
        sb.of_al = seqtbl[kOF].al;
        sb.ml_al = seqtbl[kML].al;
        sb.blk = blk;
        sb.lit_size = static_cast<uint32_t>(regen);
        sb.lit_off = lit_base;
        sb.direct = seg ? 1 : 0;
        sb.seq_first = plan->n_sequences;
        sb.frame_first_blk = frame_first_blk;
        plan->seq_blocks.push_back(sb);
        plan->n_sequences += nseq;
        if (to_lit) plan->lit_bytes += (regen + 15) & ~size_t(15);       // keep each block's literals 16-B aligned
        plan->blk_size.push_back(static_cast<uint32_t>(regen));   // + match bytes, added on the device
        plan->known_out += regen;
        return true;
    }

    bool frame(size_t &i) {
        if (!need(i, 1, "a frame header")) return false;
        const uint8_t fhd = p[i++];
        const int fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, checksum = (fhd >> 2) & 1, dict_flag = fhd & 3;
        if (fhd & 0x08) return bad("reserved bit set in the frame header");
        uint64_t window = 0;
        if (!single) {
            if (!need(i, 1, "a frame header")) return false;
            const uint8_t wd = p[i++];
            const uint64_t base = 1ull << (10 + (wd >> 3));
            window = base + (base / 8) * (wd & 7);
        }
        static const int kDictBytes[4] = {0, 1, 2, 4};
        if (!need(i, size_t(kDictBytes[dict_flag]), "a frame header")) return false;
        uint32_t dict_id = 0;
        for (int k = 0; k < kDictBytes[dict_flag]; k++) dict_id |= uint32_t(p[i + size_t(k)]) << (8 * k);
        i += size_t(kDictBytes[dict_flag]);
        if (dict_id != 0) return bad("frame needs a dictionary");
        const int fcs_bytes = fcs_flag == 0 ? (single ? 1 : 0) : (fcs_flag == 1 ? 2 : (fcs_flag == 2 ? 4 : 8));
        if (!need(i, size_t(fcs_bytes), "a frame header")) return false;
        uint64_t fcs = 0;
        for (int k = 0; k < fcs_bytes; k++) fcs |= uint64_t(p[i + size_t(k)]) << (8 * k);
        if (fcs_bytes == 2) fcs += 256;
        i += size_t(fcs_bytes);
        if (single) window = fcs;
        plan->window_max = std::max(plan->window_max, window);
        plan->n_frames++;
        plan->has_checksum = plan->has_checksum || checksum;

        const uint32_t frame_first_blk = static_cast<uint32_t>(plan->blk_size.size());
        HufRef huf;                 // entropy tables live for one frame
        TableRef seqtbl[3];
        for (;;) {
            if (!need(i, 3, "a block header")) return false;
            const uint32_t bh = p[i] | (uint32_t(p[i + 1]) << 8) | (uint32_t(p[i + 2]) << 16);
            i += 3;
            const bool last = bh & 1;
            const int type = (bh >> 1) & 3;
            const size_t bsize = bh >> 3;
            const uint32_t blk = static_cast<uint32_t>(plan->blk_size.size());
            plan->blk_off.push_back(i - 3);
            if (type == 0) {                                   // Raw
                if (!need(i, bsize, "a raw block")) return false;
                if (bsize > kBlockMax) return bad("raw block larger than 128 KiB");
                if (bsize) plan->copies.push_back(CopyTask{i, 0, uint32_t(bsize), blk, 0, 0});
                plan->blk_size.push_back(uint32_t(bsize));
                plan->known_out += bsize;
                i += bsize;
            } else if (type == 1) {                            // RLE
                if (!need(i, 1, "an RLE block")) return false;
                if (bsize > kBlockMax) return bad("RLE block larger than 128 KiB");
                if (bsize) plan->copies.push_back(CopyTask{p[i], 0, uint32_t(bsize), blk, 2, 0});
                plan->blk_size.push_back(uint32_t(bsize));
                plan->known_out += bsize;
                i += 1;
            } else if (type == 2) {                            // Compressed
                if (!need(i, bsize, "a compressed block")) return false;
                if (bsize > kBlockMax) return bad("compressed block larger than 128 KiB");
                if (!compressed_block(i, i + bsize, blk, frame_first_blk, &huf, seqtbl)) return false;
                i += bsize;
            } else {
                return bad("reserved block type");
            }
            if (last) break;
        }
        ZPlan::Frame fr{frame_first_blk, static_cast<uint32_t>(plan->blk_size.size()), 0, checksum != 0};
        if (checksum) {                                        // low 32 bits of XXH64 of the decoded frame, verified on the device
            if (!need(i, 4, "the frame checksum")) return false;
            fr.checksum = uint32_t(p[i]) | (uint32_t(p[i + 1]) << 8) | (uint32_t(p[i + 2]) << 16) | (uint32_t(p[i + 3]) << 24);
            i += 4;
        }
        plan->frames.push_back(fr);
        return true;
    }

};

// Greedy packing of streams into wave tasks: <= 64 lanes whose decode tables fit the LDS budget of
// their table format.  Streams that write the section output (directly, or segment by segment
// around the matches of a block with a few sequences) come first, those that feed the literal
// buffer of a block with many sequences second: the two groups never share a task (the first
// may expand 4-bit codes to ASCII on the fly, the second never does).  Tasks are then grouped
// into launch classes (table format x destination x segment-aware kernel).
static uint32_t g_task_lanes = kHufWave;
static uint32_t g_dict_slots = kHufLdsSlots2;
// Sub-streams (plan.h: HufStream::sub).  Lane = stream leaves a section of a few thousand streams -- a bacterial genome is
// 84 of them -- with most of the chip idle while every lane walks its 32 K symbols alone: 2.7 ms whatever the size.  Below
// `g_split_target` / 8 streams a stream is cut into S parts (a power of two, so that 64 lanes hold whole streams), S the
// largest that keeps the lanes at or under the target; the parts' first bits are found on the device by two passes
// (k_huf_sync, k_huf_bounds).  Measured on the real-genome archive (profiles/r04_split_probe.log): the walk of k_huf_sync
// costs 0.25 us per symbol of a part -- 0.5 ms for parts of 2 K symbols (S = 16), 1.75 ms at 8 K (S = 4) -- against 0.1 us per
// symbol for k_huf_decode itself, so S = 16 and 8 pay (84 ... 2 516 streams: 2.7-3.2 -> 1.0-1.7 ms) and S = 4 and 2 do not
// (8 376 streams: 2.8 -> 4.4 ms): below kHufSplitMin parts a section is left alone.  `g_split_force`: tests.
constexpr uint32_t kHufSplitMin = 8;
static uint32_t g_split_target = 0, g_split_force = 0;

void pack_tasks(ZPlan *plan) {
    std::vector<HufRef> &stream_tbl = plan->stream_ref;
    std::vector<uint32_t> class_key;
    uint32_t split = 1;
    if (g_split_force) {
        split = g_split_force;
    } else if (g_split_target && !plan->streams.empty()) {
        while (split < kHufSplitMax && plan->streams.size() * split * 2 <= g_split_target) split *= 2;
        if (split < kHufSplitMin) split = 1;            // (fewer, longer parts: the walk in front costs more than the decode saves)
    }
    if (split > 1) {                                    // every stream becomes `split` records in a row
        std::vector<HufStream> parts;
        std::vector<HufRef> parts_tbl;
        parts.reserve(plan->streams.size() * split);
        parts_tbl.reserve(plan->streams.size() * split);
        for (size_t s = 0; s < plan->streams.size(); s++)
            for (uint32_t k = 0; k < split; k++) {
                HufStream hs = plan->streams[s];
                hs.sub = k | (split << 8);
                parts.push_back(hs);
                parts_tbl.push_back(stream_tbl[s]);
            }
        plan->streams.swap(parts);
        stream_tbl.swap(parts_tbl);
    }
    {   // stable partition by destination
        std::vector<HufStream> ordered;
        std::vector<HufRef> ordered_tbl;
        ordered.reserve(plan->streams.size());
        ordered_tbl.reserve(plan->streams.size());
        for (int pass = 0; pass < 2; pass++)
            for (size_t s = 0; s < plan->streams.size(); s++)
                if ((plan->streams[s].flags & 1) == pass) {
                    ordered.push_back(plan->streams[s]);
                    ordered_tbl.push_back(stream_tbl[s]);
                }
        plan->streams.swap(ordered);
        stream_tbl.swap(ordered_tbl);
    }
    // Every table of a task is staged with the same index width W (8, 7 or 6 bits): 2^W
    // two-symbol entries plus, for codes longer than W bits, one 2^(max_bits - W) entry
    // sub-table per escaping W-bit prefix.  The widest W whose tables fit the budget wins,
    // so archives where every block brings its own (possibly 11-bit) tree still fill 64 lanes.
    auto staged_entries = [&](const HufRef &t, uint32_t W) -> uint32_t {
        if (t.max_bits <= W) return 1u << W;
        const uint16_t *x1 = plan->huf_pool.data() + t.pool_off;
        uint32_t esc = 0;
        for (uint32_t p = 0; p < (1u << W); p++)
            if ((x1[p << (t.max_bits - W)] >> 8) > W) esc++;
        return (1u << W) + (esc << (t.max_bits - W));
    };
    struct Packed {
        HufTask task;
        uint32_t key;        // to_lit << 3 | tbl << 1 | seg
        uint32_t lds_bytes;
        uint32_t sync_lds;   // one length byte per entry of the task's trees (k_huf_sync / k_huf_bounds)
    };
    std::vector<Packed> packed;
    auto pack_group = [&](size_t g0, size_t g1) {
        size_t s = g0;
        while (s < g1) {
            size_t e = std::min(g1, s + g_task_lanes);
            // k_huf_decode addresses a task's input and output with 32-bit offsets from the lowest
            // address of the task: keep both spans within kHufTaskSpan (output of a block with
            // sequences is not known here: bound it by the block maximum, x2 for the ASCII expansion)
            for (size_t k = s + 1; k < e; k++) {
                const HufStream &a = plan->streams[s], &b = plan->streams[k];
                const uint64_t src_span = b.src_end - (a.src_end - a.src_len);
                const uint64_t dst_span = (b.flags & 1) ? b.dst + b.n_syms - a.dst
                                                        : (uint64_t(b.blk) - a.blk + 1) * kBlockMax * 2;
                if (b.src_end < a.src_end || b.blk < a.blk || src_span > kHufTaskSpan || dst_span > kHufTaskSpan) {
                    e = k;
                    break;
                }
            }
            if (split > 1 && e - s > split) e = s + (e - s) / split * split;   // (the parts of a stream stay in one task)
            uint32_t W = 8, kind = kTblBaked;
            std::vector<HufRef> distinct;
            uint8_t min_len[256];                      // shortest code of every symbol over the task's trees (0xFF: unused)
            uint32_t n_union = 0;
            for (;;) {
                distinct.clear();
                for (size_t k = s; k < e; k++) {
                    bool seen = false;
                    for (const HufRef &d : distinct) seen = seen || d.pool_off == stream_tbl[k].pool_off;
                    if (!seen) distinct.push_back(stream_tbl[k]);
                }
                // the symbols the trees use between them: few enough for one shared dictionary?
                bool small = distinct.size() > 1;
                std::memset(min_len, 0xFF, sizeof min_len);
                n_union = 0;
                for (size_t d = 0; small && d < distinct.size(); d++) {
                    if (distinct[d].n_syms > kHufDictSyms) {
                        small = false;
                        break;
                    }
                    const uint16_t *x1 = plan->huf_pool.data() + distinct[d].pool_off;
                    for (uint32_t i = 0; i < (1u << distinct[d].max_bits);) {
                        const uint32_t sym = x1[i] & 0xFFu, len = x1[i] >> 8;
                        if (min_len[sym] == 0xFF) n_union++;
                        if (len < min_len[sym]) min_len[sym] = static_cast<uint8_t>(len);
                        i += 1u << (distinct[d].max_bits - len);       // a code of `len` bits fills that many entries
                    }
                    small = n_union <= kHufDictSyms;
                }
                // several trees: 2-byte dictionary entries (small alphabets) or compact 4-byte entries, and a
                // larger budget, so that the task keeps 64 lanes
                kind = distinct.size() == 1 ? kTblBaked : (small ? kTblDict : kTblCompact);
                const uint32_t budget = kind == kTblBaked ? kHufLdsEntries : (kind == kTblDict ? g_dict_slots : kHufLdsEntries4);
                const uint32_t per_tree = kind == kTblDict ? kHufDictSlots : 0;
                bool fits = false;
                for (W = 8; W >= 6; W--) {
                    uint32_t total = 0;
                    for (const HufRef &d : distinct) total += per_tree + staged_entries(d, W);
                    if (total <= budget) {
                        fits = true;
                        break;
                    }
                }
                if (fits || e <= s + split) break;
                e = s + (e - s + 1) / 2;                 // too many distinct deep trees: take fewer streams
                if (split > 1) e = s + std::max<size_t>(split, (e - s) / split * split);
            }
            if (W < 6) W = 6;
            const uint32_t per_tree = kind == kTblDict ? kHufDictSlots : 0;
            HufTask task{static_cast<uint32_t>(s), static_cast<uint32_t>(e - s),
                         static_cast<uint32_t>(plan->tbl_copies.size()), static_cast<uint32_t>(distinct.size()), 0, 0, {0, 0}};
            if (kind == kTblDict) {                    // the task's dictionary: symbols by shortest code, then by value
                task.dict_off = static_cast<uint32_t>(plan->dict_pool.size());
                task.n_dict = n_union;
                std::vector<uint32_t> order;
                for (uint32_t sym = 0; sym < 256; sym++)
                    if (min_len[sym] != 0xFF) order.push_back((static_cast<uint32_t>(min_len[sym]) << 8) | sym);
                std::sort(order.begin(), order.end());
                for (uint32_t k = 0; k < kHufDictSyms; k++)
                    plan->dict_pool.push_back(k < order.size() ? static_cast<uint8_t>(order[k] & 0xFFu) : 0);
            }
            uint32_t lds_used = 0;
            std::vector<uint32_t> lds_of(distinct.size());
            for (size_t d = 0; d < distinct.size(); d++) {
                const uint32_t n = per_tree + staged_entries(distinct[d], W);
                lds_of[d] = lds_used;
                plan->tbl_copies.push_back(HufTblCopy{distinct[d].pool_off, lds_used, n, distinct[d].max_bits | (W << 8)});
                lds_used += n;
            }
            bool seg = false;
            for (size_t k = s; k < e; k++) {
                size_t d = 0;
                while (distinct[d].pool_off != stream_tbl[k].pool_off) d++;
                HufStream &hs = plan->streams[k];
                hs.tbl_lds = static_cast<uint16_t>(lds_of[d]);
                hs.max_bits = static_cast<uint8_t>(W);
                const uint32_t esc_bits = stream_tbl[k].max_bits > W ? stream_tbl[k].max_bits - W : 0;
                hs.flags = static_cast<uint8_t>((hs.flags & 0x0F) | (esc_bits << 4));
                seg = seg || (hs.flags & 2);
            }
            const uint32_t to_lit = plan->streams[s].flags & 1u;
            const uint32_t entry_bytes = kind == kTblBaked ? 8u : (kind == kTblDict ? 2u : 4u);
            uint32_t sync_lds = 0;
            for (const HufRef &d : distinct) sync_lds += std::max<uint32_t>(16u, 1u << d.max_bits);
            packed.push_back(Packed{task, (to_lit << 3) | (kind << 1) | (seg ? 1u : 0u), lds_used * entry_bytes, sync_lds});
            s = e;
        }
    };
    size_t first_lit = 0;
    while (first_lit < plan->streams.size() && !(plan->streams[first_lit].flags & 1)) first_lit++;
    pack_group(0, first_lit);
    pack_group(first_lit, plan->streams.size());
    // A few tasks without segments beside many with (the last blocks of a real genome's section): they join the segment-aware
    // class -- its kernel takes a stream of a block without sequences as one segment -- instead of forming a class of their own:
    // a class of ONE task still takes a whole task's time (64 streams, one lane each: 2.5 ms), and behind the classes that
    // started early on the second stream that was 1.9 of the 5.9 ms of a human-genome-sized archive.
    for (uint32_t group = 0; group < 8; group++) {            // (to_lit << 2) | kind
        size_t n_seg = 0, n_plain = 0;
        for (const Packed &pk : packed)
            if ((pk.key >> 1) == group) (pk.key & 1u ? n_seg : n_plain)++;
        if (n_plain && n_plain * 8 <= n_seg)
            for (Packed &pk : packed)
                if ((pk.key >> 1) == group) pk.key |= 1u;
    }
    // launch classes: tasks of one key together, keys in ascending order (direct before literal buffer)
    std::stable_sort(packed.begin(), packed.end(), [](const Packed &a, const Packed &b) { return a.key < b.key; });
    for (const Packed &pk : packed) {
        if (class_key.empty() || class_key.back() != pk.key) {
            plan->classes.push_back(HufClass{static_cast<uint32_t>(plan->tasks.size()), 0, (pk.key >> 1) & 3u, pk.key >> 3, pk.key & 1u, 0, split, 0});
            class_key.push_back(pk.key);
        }
        HufClass &c = plan->classes.back();
        c.n_tasks++;
        c.lds_bytes = std::max(c.lds_bytes, pk.lds_bytes);
        c.sync_lds = std::max(c.sync_lds, pk.sync_lds);
        plan->tasks.push_back(pk.task);
    }
}

}  // namespace

void pack_tasks_public(ZPlan *plan) { pack_tasks(plan); }
void set_dict_slots(uint32_t slots) { g_dict_slots = slots >= 512 && slots <= kHufLdsSlots2 ? slots : kHufLdsSlots2; }
void set_huf_split(uint32_t target_lanes, uint32_t force) {
    g_split_target = target_lanes;
    g_split_force = force >= 2 && force <= kHufSplitMax && (force & (force - 1)) == 0 ? force : 0;
}
void set_task_lanes(uint32_t lanes) { g_task_lanes = lanes >= 4 && lanes <= static_cast<uint32_t>(kHufWave) ? lanes : kHufWave; }

std::string walk_zstd(const uint8_t *payload, size_t n, ZPlan *master, bool *truncated) {
    Walker w{payload, n, master, {}, {}};
    *truncated = false;
    if (n == 0) {
        *truncated = true;
        return "empty section payload";
    }
    size_t i = 0;
    while (i < n) {                 // a section may hold several frames back to back
        if (!w.frame(i)) {
            *truncated = w.fail.truncated;
            return w.fail.msg.empty() ? std::string("malformed zstd frame") : w.fail.msg;
        }
    }
    master->blk_off.push_back(i);
    if (!master->seq_blocks.empty()) {
        master->first_seq_frame = master->seq_blocks.front().frame_first_blk;
        master->last_seq_frame = master->seq_blocks.back().frame_first_blk;
    }
    master->sel_blk0 = 0;
    master->sel_blk1 = static_cast<uint32_t>(master->blk_size.size());
    master->src_lo = 0;
    master->src_hi = n;
    return std::string();
}

bool shard_range(const ZPlan &master, uint32_t shard_rank, uint32_t shard_count, uint32_t *b0_out, uint32_t *b1_out, bool with_lz) {
    if (shard_count <= 1 || master.blk_size.empty()) return false;
    if (!master.seq_blocks.empty() && !with_lz) return false;
    // contiguous block ranges balanced by decoded bytes; block boundaries only.  The decoded size of a block with
    // sequences is literals + match bytes, the second of which only the device learns: it counts as a full block.
    const uint32_t nb = static_cast<uint32_t>(master.blk_size.size());
    size_t sb = 0;
    auto weight = [&](uint32_t b) -> uint64_t {
        while (sb < master.seq_blocks.size() && master.seq_blocks[sb].blk < b) sb++;
        return sb < master.seq_blocks.size() && master.seq_blocks[sb].blk == b ? kBlockMax : master.blk_size[b];
    };
    uint64_t total = 0;
    for (uint32_t b = 0; b < nb; b++) total += weight(b);
    sb = 0;
    const uint64_t lo_target = total / shard_count * shard_rank + std::min<uint64_t>(shard_rank, total % shard_count);
    const uint64_t hi_target = total / shard_count * (shard_rank + 1) + std::min<uint64_t>(shard_rank + 1, total % shard_count);
    uint64_t pos = 0;
    uint32_t b0 = 0, b1 = 0;
    bool have0 = false;
    for (uint32_t b = 0; b <= nb; b++) {                 // a block belongs to the shard its first byte falls in
        if (!have0 && (pos >= lo_target || b == nb)) {
            b0 = b;
            have0 = true;
        }
        if (have0 && (pos >= hi_target || b == nb)) {
            b1 = b;
            break;
        }
        if (b < nb) pos += weight(b);
    }
    if (shard_rank + 1 == shard_count) b1 = nb;
    *b0_out = b0;
    *b1_out = b1;
    return true;
}

void select_zplan(const ZPlan &m, uint32_t b0, uint32_t b1, uint64_t halo_elems, ZPlan *out, bool force_halo) {
    ZPlan &p = *out;
    p = ZPlan();
    const uint32_t nb = static_cast<uint32_t>(m.blk_size.size());
    if (b1 > nb) b1 = nb;
    if (b0 > b1) b0 = b1;
    const uint32_t halo = (halo_elems || force_halo) ? 1u : 0u;
    p.sel_blk0 = b0;
    p.sel_blk1 = b1;
    p.halo = halo;
    p.src_lo = m.blk_off[b0];
    p.src_hi = m.blk_off[b1];
    p.n_frames = m.n_frames;
    p.window_max = m.window_max;
    p.has_checksum = m.has_checksum;
    p.frames = m.frames;
    if (halo) p.blk_size.push_back(static_cast<uint32_t>(halo_elems));      // (a window is far below 4 GiB)
    p.blk_size.insert(p.blk_size.end(), m.blk_size.begin() + b0, m.blk_size.begin() + b1);
    for (uint32_t b = b0; b < b1; b++) p.known_out += m.blk_size[b];
    if (m.seq_blocks.empty()) {                                             // decoded-byte range of the selection
        for (uint32_t b = 0; b < b0; b++) p.shard_out0 += m.blk_size[b];
        p.shard_out1 = p.shard_out0 + p.known_out;
    }
    auto in_range = [&](uint32_t blk) { return blk >= b0 && blk < b1; };
    auto reblk = [&](uint32_t blk) { return blk - b0 + halo; };
    // tables: only what the selection uses, in first-use order
    std::vector<std::pair<uint32_t, uint32_t>> huf_map, fse_map;            // (old offset, new offset), few entries per selection: linear search is fine for tiles,
    auto map_get = [](std::vector<std::pair<uint32_t, uint32_t>> &mp, uint32_t old) -> int64_t {   // sorted insert + binary search for the big ones
        auto it = std::lower_bound(mp.begin(), mp.end(), std::make_pair(old, 0u));
        return (it != mp.end() && it->first == old) ? static_cast<int64_t>(it->second) : -1;
    };
    auto map_put = [](std::vector<std::pair<uint32_t, uint32_t>> &mp, uint32_t old, uint32_t neu) {
        mp.insert(std::lower_bound(mp.begin(), mp.end(), std::make_pair(old, 0u)), std::make_pair(old, neu));
    };
    auto huf_remap = [&](HufTableRef r) -> HufTableRef {
        int64_t k = map_get(huf_map, r.pool_off);
        if (k < 0) {
            k = static_cast<int64_t>(p.huf_pool.size());
            p.huf_pool.insert(p.huf_pool.end(), m.huf_pool.begin() + r.pool_off, m.huf_pool.begin() + r.pool_off + (size_t(1) << r.max_bits));
            map_put(huf_map, r.pool_off, static_cast<uint32_t>(k));
            p.n_huf_tables++;
        }
        r.pool_off = static_cast<uint32_t>(k);
        return r;
    };
    auto fse_remap = [&](uint32_t off, uint32_t al) -> uint32_t {
        int64_t k = map_get(fse_map, off);
        if (k < 0) {
            k = static_cast<int64_t>(p.fse_pool.size());
            p.fse_pool.insert(p.fse_pool.end(), m.fse_pool.begin() + off, m.fse_pool.begin() + off + (size_t(1) << al));
            map_put(fse_map, off, static_cast<uint32_t>(k));
        }
        return static_cast<uint32_t>(k);
    };
    // blocks with sequences: new literal-buffer and sequence offsets
    std::vector<std::pair<uint32_t, uint64_t>> lit_delta;                   // (master block, old literal offset - new one) of blocks that use the literal buffer
    size_t sb0 = 0;
    while (sb0 < m.seq_blocks.size() && m.seq_blocks[sb0].blk < b0) sb0++;
    size_t sb1 = sb0;
    while (sb1 < m.seq_blocks.size() && m.seq_blocks[sb1].blk < b1) sb1++;
    for (size_t k = sb0; k < sb1; k++) {
        SeqBlock sb = m.seq_blocks[k];
        if (!sb.direct) {
            lit_delta.push_back({sb.blk, sb.lit_off - p.lit_bytes});
            sb.lit_off = p.lit_bytes;
            p.lit_bytes += (static_cast<uint64_t>(sb.lit_size) + 15) & ~uint64_t(15);
        }
        if (k == sb0) {
            p.first_frame_continues = sb.frame_first_blk < b0;
            p.first_seq_frame = sb.frame_first_blk;
        }
        p.last_seq_frame = sb.frame_first_blk;
        sb.frame_first_blk = sb.frame_first_blk < b0 ? 0u : reblk(sb.frame_first_blk);   // a frame begun in front starts at the pseudo block (or at block 0)
        sb.blk = reblk(sb.blk);
        sb.seq_first = p.n_sequences;
        p.n_sequences += sb.n_seq;
        sb.ll_tbl = fse_remap(sb.ll_tbl, sb.ll_al);
        sb.of_tbl = fse_remap(sb.of_tbl, sb.of_al);
        sb.ml_tbl = fse_remap(sb.ml_tbl, sb.ml_al);
        p.seq_blocks.push_back(sb);
    }
    auto lit_rebase = [&](uint32_t blk, uint64_t old) -> uint64_t {
        auto it = std::lower_bound(lit_delta.begin(), lit_delta.end(), std::make_pair(blk, uint64_t(0)));
        return old - it->second;
    };
    // the master's seq block index of a stream tagged "segmented" -> the selection's
    auto seg_rebase = [&](uint64_t dst) -> uint64_t {
        const uint64_t idx1 = dst >> 32;
        return idx1 ? ((idx1 - sb0) << 32) | (dst & 0xFFFFFFFFull) : dst;
    };
    {
        size_t s0 = 0;                                                      // streams are in block order in the master
        size_t lo = 0, hi = m.streams.size();
        while (lo < hi) {
            const size_t mid = (lo + hi) / 2;
            if (m.streams[mid].blk < b0)
                lo = mid + 1;
            else
                hi = mid;
        }
        s0 = lo;
        for (size_t k = s0; k < m.streams.size() && m.streams[k].blk < b1; k++) {
            HufStream hs = m.streams[k];
            if (hs.flags & 1)
                hs.dst = lit_rebase(hs.blk, hs.dst);
            else if (hs.flags & 2)
                hs.dst = seg_rebase(hs.dst);
            hs.blk = reblk(hs.blk);
            p.streams.push_back(hs);
            p.stream_ref.push_back(huf_remap(m.stream_ref[k]));
        }
    }
    for (const CopyTask &t : m.copies) {
        if (!in_range(t.blk)) continue;
        CopyTask c = t;
        if (c.flags & 1) c.dst = lit_rebase(c.blk, c.dst);
        c.blk = reblk(c.blk);
        p.copies.push_back(c);
    }
    pack_tasks(&p);
}

std::string build_zplan(const uint8_t *payload, size_t n, ZPlan *plan, bool *truncated, uint32_t shard_rank,
                        uint32_t shard_count) {
    ZPlan master;
    std::string err = walk_zstd(payload, n, &master, truncated);
    if (!err.empty()) {
        *plan = ZPlan();
        return err;
    }
    uint32_t b0 = 0, b1 = static_cast<uint32_t>(master.blk_size.size());
    const bool sharded = shard_range(master, shard_rank, shard_count, &b0, &b1);
    if (!sharded) {                                       // the whole section: no need to copy anything
        *plan = std::move(master);
        pack_tasks(plan);
        return std::string();
    }
    select_zplan(master, b0, b1, 0, plan);
    plan->sharded = true;
    return std::string();
}

}  // namespace nafgpu
