// synth.cpp -- synthetic NAF archive writer (host only).
//
// SURVEY.md section 8d needs 10-80 GB DNA archives that the reference's own Encoder cannot
// produce fast enough (single-threaded, flushes per record, never writes a Mask section --
// nafcodec/src/encoder/mod.rs:240,271,298,319).  This writer emits the same container
// (layout per encoder/mod.rs:334-384: header, then (original_size, compressed_size, payload)
// per section) with the sequence section as ONE magicless Zstandard frame of 128 KiB blocks
// whose literals are 4-stream Huffman coded -- the shape `ennaf` / zstd level 1 give DNA
// (1 Huffman + 127 treeless blocks, ~0 sequences).  Blocks are encoded in
// parallel; the result depends only on (seed, n_bases, options), never on the thread count.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "../../include/nafgpu.h"
#include "hash64.h"
#include "plan.h"

namespace {

using nafgpu::kBlockMax;

// ---------------------------------------------------------------- rng
struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next() {                       // splitmix64
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uniform() { return (next() >> 11) * (1.0 / 9007199254740992.0); }
};

void put_varint(std::vector<uint8_t> &out, uint64_t v) {   // encoder/mod.rs:22-35
    uint8_t tmp[10];
    int k = 0;
    tmp[k++] = v & 0x7F;
    v >>= 7;
    while (v) {
        tmp[k++] = 0x80 | (v & 0x7F);
        v >>= 7;
    }
    while (k) out.push_back(tmp[--k]);
}

// a section of raw zstd blocks (what masked.naf uses for its small sections)
void raw_frame(const std::vector<uint8_t> &data, std::vector<uint8_t> &out) {
    out.push_back(0x00);   // FHD: no content size, no checksum, no dictionary
    out.push_back(0x48);   // window 512 KiB
    size_t pos = 0;
    do {
        const size_t n = std::min<size_t>(kBlockMax, data.size() - pos);
        const bool last = pos + n == data.size();
        const uint32_t bh = static_cast<uint32_t>(n << 3) | (last ? 1u : 0u);
        out.push_back(bh & 0xFF);
        out.push_back((bh >> 8) & 0xFF);
        out.push_back((bh >> 16) & 0xFF);
        out.insert(out.end(), data.begin() + static_cast<long>(pos), data.begin() + static_cast<long>(pos + n));
        pos += n;
    } while (pos < data.size());
}

// ---------------------------------------------------------------- Huffman code construction
inline int highbit(uint32_t v) { return 31 - __builtin_clz(v); }

struct HufCode {
    uint8_t len[256];       // 0 = symbol absent
    uint16_t code[256];
    uint8_t weight[256];
    int max_bits = 0;
    int max_sym = -1;
    bool valid = false;
};

// code lengths limited to 11 bits (counts are halved until the tree is shallow enough)
bool build_lengths(const uint32_t *count, uint8_t *len) {
    uint32_t c[256];
    int nsym = 0;
    for (int s = 0; s < 256; s++) {
        c[s] = count[s];
        nsym += count[s] != 0;
    }
    if (nsym < 2) return false;
    for (;;) {
        // O(n^2) two-smallest merge: n <= 256 and this runs once per 128 KiB block
        struct Node {
            uint64_t w;
            int parent;
        };
        Node nodes[512];
        int alive[256], n_alive = 0, n_nodes = 0, leaf_of[256];
        for (int s = 0; s < 256; s++)
            if (c[s]) {
                nodes[n_nodes] = {c[s], -1};
                leaf_of[s] = n_nodes;
                alive[n_alive++] = n_nodes++;
            }
        while (n_alive > 1) {
            int a = 0, b = 1;
            if (nodes[alive[b]].w < nodes[alive[a]].w) std::swap(a, b);
            for (int k = 2; k < n_alive; k++) {
                if (nodes[alive[k]].w < nodes[alive[a]].w) {
                    b = a;
                    a = k;
                } else if (nodes[alive[k]].w < nodes[alive[b]].w) {
                    b = k;
                }
            }
            nodes[n_nodes] = {nodes[alive[a]].w + nodes[alive[b]].w, -1};
            nodes[alive[a]].parent = n_nodes;
            nodes[alive[b]].parent = n_nodes;
            const int lo = std::min(a, b), hi = std::max(a, b);
            alive[lo] = n_nodes++;
            alive[hi] = alive[--n_alive];
        }
        int maxlen = 0;
        for (int s = 0; s < 256; s++) {
            len[s] = 0;
            if (!c[s]) continue;
            int d = 0;
            for (int v = leaf_of[s]; nodes[v].parent >= 0; v = nodes[v].parent) d++;
            len[s] = static_cast<uint8_t>(d);
            maxlen = std::max(maxlen, d);
        }
        if (maxlen <= 11) return true;
        for (int s = 0; s < 256; s++)
            if (c[s]) c[s] = (c[s] + 1) / 2;
    }
}

// canonical codes in the order the Zstandard decoder fills its table (App. B): weight 1 first,
// ascending symbol inside a weight, each symbol spanning 2^(w-1) table entries
void assign_codes(HufCode *h) {
    int max_len = 0;
    h->max_sym = -1;
    for (int s = 0; s < 256; s++)
        if (h->len[s]) {
            max_len = std::max<int>(max_len, h->len[s]);
            h->max_sym = s;
        }
    h->max_bits = max_len;
    uint32_t pos = 0;
    for (int s = 0; s < 256; s++) h->weight[s] = h->len[s] ? static_cast<uint8_t>(max_len + 1 - h->len[s]) : 0;
    for (int w = 1; w <= max_len; w++)
        for (int s = 0; s < 256; s++)
            if (h->weight[s] == w) {
                h->code[s] = static_cast<uint16_t>(pos >> (w - 1));
                pos += 1u << (w - 1);
            }
    h->valid = pos == (1u << max_len);
}

// ---------------------------------------------------------------- FSE-compressed weights
// Inverse of App. B "FSE table description" + the two-state weight stream.  Returns false if
// the description does not fit the 127-byte limit (the caller then falls back to a raw block).
struct FseDec {
    int al;
    uint8_t sym[64], nb[64];
    uint16_t base[64];
};

bool fse_build_dec(const int16_t *norm, int nsym, int al, FseDec *t) {
    const int S = 1 << al;
    uint16_t next[16];
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
        const uint16_t d = next[t->sym[i]]++;
        const int nb = al - highbit(d);
        t->nb[i] = static_cast<uint8_t>(nb);
        t->base[i] = static_cast<uint16_t>((d << nb) - S);
    }
    return true;
}

struct BitSink {            // LSB-first writer
    std::vector<uint8_t> bytes;
    uint64_t acc = 0;
    int n = 0;
    void put(uint32_t v, int nb) {
        acc |= static_cast<uint64_t>(v) << n;
        n += nb;
        while (n >= 8) {
            bytes.push_back(acc & 0xFF);
            acc >>= 8;
            n -= 8;
        }
    }
    void flush() {
        if (n) bytes.push_back(acc & 0xFF);
        acc = 0;
        n = 0;
    }
};

bool write_weights(const uint8_t *w, int nw, std::vector<uint8_t> &out) {
    // nw = number of explicit weights (the last symbol's weight is implied)
    if (nw < 2 || nw > 255) return false;
    int cnt[13] = {0}, max_w = 0;
    for (int i = 0; i < nw; i++) {
        cnt[w[i]]++;
        max_w = std::max<int>(max_w, w[i]);
    }
    for (int s = 0; s <= max_w; s++)
        if (cnt[s] == nw) return false;              // a single distinct weight: FSE cannot code it
    const int al = 6, S = 1 << al;
    // normalise: every present weight gets >= 1 slot, the rest proportionally, remainder to the largest
    int16_t norm[13];
    int total = 0, largest = 0;
    for (int s = 0; s <= max_w; s++) {
        norm[s] = 0;
        if (!cnt[s]) continue;
        int v = static_cast<int>((static_cast<int64_t>(cnt[s]) * S) / nw);
        if (v < 1) v = 1;
        norm[s] = static_cast<int16_t>(v);
        total += v;
        if (cnt[s] > cnt[largest] || !cnt[largest]) largest = s;
    }
    norm[largest] = static_cast<int16_t>(norm[largest] + (S - total));
    if (norm[largest] < 1) return false;
    if (norm[largest] >= S) return false;            // a probability of 1 leaves zero-bit states everywhere
    const int nsym = max_w + 1;
    FseDec t;
    if (!fse_build_dec(norm, nsym, al, &t)) return false;
    // ---- table description
    BitSink hdr;
    hdr.put(static_cast<uint32_t>(al - 5), 4);
    int remaining = S;
    for (int s = 0; s < nsym && remaining > 0;) {
        const int max = remaining + 1;
        const int bits = highbit(static_cast<uint32_t>(max)) + 1;
        const uint32_t low = (1u << (bits - 1)) - 1;
        const uint32_t thr = (1u << bits) - 1 - static_cast<uint32_t>(max);
        const uint32_t v = static_cast<uint32_t>(norm[s] + 1);
        if (v < thr)
            hdr.put(v, bits - 1);
        else
            hdr.put(v <= low ? v : v + thr, bits);
        remaining -= norm[s];
        const bool zero = norm[s] == 0;
        s++;
        if (zero) {
            int z = 0;
            while (s + z < nsym && norm[s + z] == 0) z++;
            s += z;
            while (z >= 3) {
                hdr.put(3, 2);
                z -= 3;
            }
            hdr.put(static_cast<uint32_t>(z), 2);
        }
    }
    if (remaining != 0) return false;
    hdr.flush();
    // ---- weight stream, two interleaved states (even indices on state 1, odd on state 2)
    auto pick = [&](int sym, int want_state) -> int {        // a state of `sym`; want_state<0: most bits
        int best = -1;
        for (int i = 0; i < S; i++) {
            if (t.sym[i] != sym) continue;
            if (want_state >= 0) {
                if (t.base[i] <= want_state && want_state < t.base[i] + (1 << t.nb[i])) return i;
            } else if (best < 0 || t.nb[i] > t.nb[best]) {
                best = i;
            }
        }
        return best;
    };
    int state[2];
    state[(nw - 1) & 1] = pick(w[nw - 1], -1);
    state[(nw - 2) & 1] = pick(w[nw - 2], -1);
    if (state[0] < 0 || state[1] < 0 || t.nb[state[(nw - 2) & 1]] == 0) return false;
    std::vector<std::pair<uint32_t, int>> upd(static_cast<size_t>(std::max(0, nw - 2)));   // update bits of symbol k
    for (int k = nw - 3; k >= 0; k--) {
        const int p = k & 1;
        const int i = pick(w[k], state[p]);
        if (i < 0) return false;
        upd[static_cast<size_t>(k)] = {static_cast<uint32_t>(state[p] - t.base[i]), t.nb[i]};
        state[p] = i;
    }
    // read order: s1, s2, upd[0], upd[1], ... ; first-read field sits right below the end mark
    int total_bits = 2 * al;
    for (auto &u : upd) total_bits += u.second;
    std::vector<uint8_t> stream(static_cast<size_t>(total_bits / 8 + 1), 0);
    int pos = total_bits;
    auto place = [&](uint32_t v, int nb) {
        pos -= nb;
        for (int b = 0; b < nb; b++)
            if ((v >> b) & 1) stream[static_cast<size_t>((pos + b) >> 3)] |= static_cast<uint8_t>(1u << ((pos + b) & 7));
    };
    place(static_cast<uint32_t>(state[0]), al);
    place(static_cast<uint32_t>(state[1]), al);
    for (auto &u : upd) place(u.first, u.second);
    stream[static_cast<size_t>(total_bits >> 3)] |= static_cast<uint8_t>(1u << (total_bits & 7));   // end mark
    const size_t csize = hdr.bytes.size() + stream.size();
    if (csize >= 128) return false;
    out.push_back(static_cast<uint8_t>(csize));
    out.insert(out.end(), hdr.bytes.begin(), hdr.bytes.end());
    out.insert(out.end(), stream.begin(), stream.end());
    return true;
}

// ---------------------------------------------------------------- one compressed block
void encode_stream(const HufCode &h, const uint8_t *sym, size_t n, std::vector<uint8_t> &out) {
    uint64_t acc = 0;
    int nbits = 0;
    for (size_t i = n; i-- > 0;) {             // last symbol first: it is read last (backward stream)
        acc |= static_cast<uint64_t>(h.code[sym[i]]) << nbits;
        nbits += h.len[sym[i]];
        while (nbits >= 8) {
            out.push_back(acc & 0xFF);
            acc >>= 8;
            nbits -= 8;
        }
    }
    acc |= 1ull << nbits;                      // end mark
    out.push_back(acc & 0xFF);
}

// Appends one zstd block (header included) holding `n` literal bytes and no sequences.
// `prev` is the table of the previous block in the same chunk (treeless reuse) and is updated.
void encode_block(const uint8_t *data, size_t n, bool last, HufCode *prev, std::vector<uint8_t> &out) {
    auto raw_block = [&]() {
        const uint32_t bh = static_cast<uint32_t>(n << 3) | (last ? 1u : 0u);
        out.push_back(bh & 0xFF);
        out.push_back((bh >> 8) & 0xFF);
        out.push_back((bh >> 16) & 0xFF);
        out.insert(out.end(), data, data + n);
    };
    if (n < 64) return raw_block();
    uint32_t count[256] = {0};
    for (size_t i = 0; i < n; i++) count[data[i]]++;
    if (count[data[0]] == n) {                                   // one byte value: an RLE block (Huffman needs two symbols)
        const uint32_t bh = static_cast<uint32_t>(n << 3) | (1u << 1) | (last ? 1u : 0u);
        out.push_back(bh & 0xFF);
        out.push_back((bh >> 8) & 0xFF);
        out.push_back((bh >> 16) & 0xFF);
        out.push_back(data[0]);
        return;
    }
    HufCode cur{};
    std::vector<uint8_t> tree;
    bool have_new = build_lengths(count, cur.len);
    if (have_new) {
        assign_codes(&cur);
        have_new = cur.valid && write_weights(cur.weight, cur.max_sym, tree);   // weights of symbols 0..max_sym-1
    }
    uint64_t cost_new = UINT64_MAX, cost_old = UINT64_MAX;
    if (have_new) {
        cost_new = tree.size() * 8;
        for (int s = 0; s < 256; s++) cost_new += static_cast<uint64_t>(count[s]) * cur.len[s];
    }
    if (prev->valid) {
        cost_old = 0;
        for (int s = 0; s < 256; s++) {
            if (!count[s]) continue;
            if (!prev->len[s]) {
                cost_old = UINT64_MAX;
                break;
            }
            cost_old += static_cast<uint64_t>(count[s]) * prev->len[s];
        }
    }
    if (cost_new == UINT64_MAX && cost_old == UINT64_MAX) return raw_block();
    const bool treeless = cost_old <= cost_new;
    const HufCode &h = treeless ? *prev : cur;
    std::vector<uint8_t> body;                                   // tree + jump table + streams
    if (!treeless) body = tree;
    const size_t q = (n + 3) / 4;
    std::vector<uint8_t> st[4];
    encode_stream(h, data, q, st[0]);
    encode_stream(h, data + q, q, st[1]);
    encode_stream(h, data + 2 * q, q, st[2]);
    encode_stream(h, data + 3 * q, n - 3 * q, st[3]);
    for (int k = 0; k < 3; k++) {
        if (st[k].size() > 0xFFFF) return raw_block();
        body.push_back(st[k].size() & 0xFF);
        body.push_back(static_cast<uint8_t>(st[k].size() >> 8));
    }
    for (int k = 0; k < 4; k++) body.insert(body.end(), st[k].begin(), st[k].end());
    const size_t comp = body.size();
    // literals header: 4 streams, size format by magnitude
    uint8_t lh[5];
    size_t lhn;
    const uint32_t type = treeless ? 3u : 2u;
    if (n <= 1023 && comp <= 1023) {
        const uint32_t v = type | (1u << 2) | (static_cast<uint32_t>(n) << 4) | (static_cast<uint32_t>(comp) << 14);
        lh[0] = v & 0xFF; lh[1] = (v >> 8) & 0xFF; lh[2] = (v >> 16) & 0xFF;
        lhn = 3;
    } else if (n <= 16383 && comp <= 16383) {
        const uint32_t v = type | (2u << 2) | (static_cast<uint32_t>(n) << 4) | (static_cast<uint32_t>(comp) << 18);
        lh[0] = v & 0xFF; lh[1] = (v >> 8) & 0xFF; lh[2] = (v >> 16) & 0xFF; lh[3] = (v >> 24) & 0xFF;
        lhn = 4;
    } else {
        const uint64_t v = type | (3u << 2) | (static_cast<uint64_t>(n) << 4) | (static_cast<uint64_t>(comp) << 22);
        for (int k = 0; k < 5; k++) lh[k] = (v >> (8 * k)) & 0xFF;
        lhn = 5;
    }
    const size_t bsize = lhn + comp + 1;                         // + "0 sequences" byte
    if (bsize >= n || bsize > kBlockMax) return raw_block();
    const uint32_t bh = static_cast<uint32_t>(bsize << 3) | (2u << 1) | (last ? 1u : 0u);
    out.push_back(bh & 0xFF);
    out.push_back((bh >> 8) & 0xFF);
    out.push_back((bh >> 16) & 0xFF);
    out.insert(out.end(), lh, lh + lhn);
    out.insert(out.end(), body.begin(), body.end());
    out.push_back(0x00);                                         // Number_of_Sequences = 0
    if (!treeless) *prev = cur;
}

// ---------------------------------------------------------------- the archive
constexpr size_t kChunkBlocks = 64;     // blocks encoded as one unit (first block carries a fresh table)

struct MaskRun {
    uint64_t start, end;                // masked interval in base coordinates
};

void fill_block(uint8_t *dst, size_t n, uint64_t seed, uint64_t block_index, uint32_t iupac_permille) {
    Rng r(seed ^ (0xD1B54A32D192ED03ull * (block_index + 1)));
    size_t i = 0;
    while (i < n) {
        uint64_t x = r.next();
        for (int k = 0; k < 16 && i < n; k++, x >>= 4) {
            const uint32_t lo = 1u << (x & 3), hi = 1u << ((x >> 2) & 3);     // A=8 C=4 G=2 T=1
            dst[i++] = static_cast<uint8_t>(lo | (hi << 4));
        }
    }
    if (iupac_permille) {
        static const uint8_t kExtra[8] = {15, 15, 15, 15, 10, 5, 9, 6};      // N N N N R Y W S
        const size_t hits = (2 * n * iupac_permille) / 1000;
        for (size_t h = 0; h < hits; h++) {
            const uint64_t x = r.next();
            const size_t nib = static_cast<size_t>(x % (2 * n));
            const uint8_t code = kExtra[(x >> 40) & 7];
            uint8_t &b = dst[nib >> 1];
            b = (nib & 1) ? static_cast<uint8_t>((b & 0x0F) | (code << 4)) : static_cast<uint8_t>((b & 0xF0) | code);
        }
    }
}

const char kLut[17] = "-TGKCYSBAWRDMHVN";

}  // namespace

// head_only: build what precedes the sequence blocks, given their total size (nafgpu_synth_head)
static int synth_impl(const nafgpu_synth_spec *spec, nafgpu_synth_archive *out, bool head_only, uint64_t seq_part_bytes) {
    if (!spec || !out || spec->n_bases == 0) return NAFGPU_E_INVALID_ARG;
    const uint32_t part_count = spec->part_count > 1 ? spec->part_count : 1, part_rank = spec->part_rank;
    if (part_rank >= part_count) return NAFGPU_E_INVALID_ARG;
    std::memset(out, 0, sizeof *out);
    const uint64_t n_bases = spec->n_bases;
    const uint64_t n_packed = (n_bases + 1) / 2;
    Rng rng(spec->seed * 0x9E3779B97F4A7C15ull + 0x4E4146);

    // ---- record lengths: half fixed 151, half log-uniform in [1e3, 1e7]
    std::vector<uint64_t> rec_end;
    {
        uint64_t pos = 0;
        while (pos < n_bases) {
            uint64_t len;
            if (rng.next() & 1) {
                len = 151;
            } else {
                len = static_cast<uint64_t>(std::floor(1000.0 * std::pow(10000.0, rng.uniform())));
                len = std::min<uint64_t>(std::max<uint64_t>(len, 1000), 10000000);
            }
            len = std::min(len, n_bases - pos);
            pos += len;
            rec_end.push_back(pos);
        }
    }
    const uint64_t n_rec = rec_end.size();
    std::vector<uint8_t> len_bytes;
    len_bytes.reserve(n_rec * 4);
    for (uint64_t k = 0; k < n_rec; k++) {
        const uint64_t l = rec_end[k] - (k ? rec_end[k - 1] : 0);
        const uint32_t w = static_cast<uint32_t>(l);               // < 2^32 - 1 by construction
        for (int b = 0; b < 4; b++) len_bytes.push_back((w >> (8 * b)) & 0xFF);
    }

    // ---- mask runs: unmasked ~Geom(3000), masked ~Geom(600), masked runs strictly inside records
    std::vector<MaskRun> runs;
    std::vector<uint8_t> mask_bytes;
    if (spec->with_mask) {
        auto put_run = [&](uint64_t n) {
            while (n >= 255) {
                mask_bytes.push_back(0xFF);
                n -= 255;
            }
            mask_bytes.push_back(static_cast<uint8_t>(n));
        };
        uint64_t pos = 0, rec = 0, forced = 0;
        while (pos < n_bases) {
            auto geom = [&](double mean) {
                const double u = std::max(rng.uniform(), 1e-12);
                return static_cast<uint64_t>(1 + std::floor(-std::log(u) * mean));
            };
            uint64_t un = geom(3000.0), mk = geom(600.0);
            if (forced == 3) { un = 255; }                         // run of exactly 255: FF 00
            if (forced == 5) { mk = 255; }
            if (forced == 7) { un = 70000; }                       // run longer than 65 535
            forced++;
            uint64_t s = std::min(pos + un, n_bases);
            while (rec < n_rec && rec_end[rec] <= s) rec++;        // record holding base s
            if (s >= n_bases || rec >= n_rec) {
                put_run(n_bases - pos);                            // trailing unmasked run
                pos = n_bases;
                break;
            }
            // keep [s, e) strictly inside the record: e < record end
            uint64_t e = std::min(s + mk, rec_end[rec] - 1);
            if (e <= s) {                                          // no room: extend the unmasked run past the record
                s = std::min<uint64_t>(rec_end[rec] + 1, n_bases);
                put_run(s - pos);
                put_run(0);                                        // empty masked run keeps the alternation
                pos = s;
                continue;
            }
            put_run(s - pos);
            put_run(e - s);
            runs.push_back({s, e});
            pos = e;
        }
    }

    // ---- sequence section: blocks in parallel, chunks of kChunkBlocks
    const uint64_t n_blocks = (n_packed + kBlockMax - 1) / kBlockMax;
    const uint64_t n_chunks = (n_blocks + kChunkBlocks - 1) / kChunkBlocks;
    // this process's share of the chunks (all of them unless the archive is written in parts)
    const uint64_t chunk0 = head_only ? 0 : n_chunks * part_rank / part_count;
    const uint64_t chunk1 = head_only ? 0 : n_chunks * (part_rank + 1) / part_count;
    std::vector<std::vector<uint8_t>> chunk_out(n_chunks);
    std::vector<uint64_t> chunk_hash(n_chunks, 0);
    std::atomic<uint64_t> next_chunk{chunk0};
    unsigned n_threads = spec->threads ? spec->threads : std::max(1u, std::thread::hardware_concurrency());
    n_threads = static_cast<unsigned>(std::max<uint64_t>(1, std::min<uint64_t>(n_threads, chunk1 - chunk0)));
    auto worker = [&]() {
        std::vector<uint8_t> packed(kBlockMax), ascii(2 * kBlockMax);
        for (;;) {
            const uint64_t c = next_chunk.fetch_add(1);
            if (c >= chunk1) break;
            HufCode prev{};
            std::vector<uint8_t> &o = chunk_out[c];
            o.reserve(kChunkBlocks * (kBlockMax / 2 + 64));
            uint64_t h = 0;
            for (uint64_t b = c * kChunkBlocks; b < std::min(n_blocks, (c + 1) * kChunkBlocks); b++) {
                const uint64_t p0 = b * kBlockMax;
                const size_t n = static_cast<size_t>(std::min<uint64_t>(kBlockMax, n_packed - p0));
                fill_block(packed.data(), n, spec->seed, b, spec->iupac_permille);
                if (b == n_blocks - 1 && (n_bases & 1)) packed[n - 1] &= 0x0F;   // pad nibble (writer.rs:21-28)
                encode_block(packed.data(), n, b == n_blocks - 1, &prev, o);
                // expected ASCII of this block (for the checksum): unpack, then lower-case masked runs
                const uint64_t b0 = 2 * p0, b1 = std::min<uint64_t>(n_bases, b0 + 2 * n);
                for (size_t i = 0; i < n; i++) {
                    ascii[2 * i] = static_cast<uint8_t>(kLut[packed[i] & 15]);
                    ascii[2 * i + 1] = static_cast<uint8_t>(kLut[packed[i] >> 4]);
                }
                if (!runs.empty()) {
                    auto it = std::lower_bound(runs.begin(), runs.end(), b0,
                                               [](const MaskRun &r, uint64_t v) { return r.end <= v; });
                    for (; it != runs.end() && it->start < b1; ++it) {
                        const uint64_t s = std::max(it->start, b0), e = std::min(it->end, b1);
                        for (uint64_t k = s; k < e; k++) {
                            uint8_t &ch = ascii[static_cast<size_t>(k - b0)];
                            if (ch >= 'A' && ch <= 'Z') ch |= 0x20;
                        }
                    }
                }
                h += nafgpu::hash64_host(ascii.data(), b1 - b0, b0 / nafgpu::kHashChunk);
            }
            chunk_hash[c] = h;
        }
    };
    {
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < n_threads; t++) pool.emplace_back(worker);
        worker();
        for (auto &t : pool) t.join();
    }
    uint64_t seq_csize = 2;
    for (auto &c : chunk_out) seq_csize += c.size();
    if (part_count > 1 && !head_only) {                  // a part: its blocks, nothing in front
        uint64_t part_n = 0;
        for (auto &c : chunk_out) part_n += c.size();
        uint8_t *pb = static_cast<uint8_t *>(std::malloc(part_n ? part_n : 1));
        if (!pb) return NAFGPU_E_DEVICE;
        uint64_t at = 0;
        for (auto &c : chunk_out) {
            std::memcpy(pb + at, c.data(), c.size());
            at += c.size();
        }
        out->bytes = pb;
        out->n = part_n;
        out->n_records = n_rec;
        out->n_bases = n_bases;
        for (uint64_t h : chunk_hash) out->seq_hash += h;
        out->offsets_hash = nafgpu::hash64_host(reinterpret_cast<const uint8_t *>(rec_end.data()), n_rec * 8);
        return NAFGPU_OK;
    }
    if (head_only) seq_csize = 2 + seq_part_bytes;

    // ---- assemble (encoder/mod.rs:334-384): header, then Length, [Mask], Sequence
    std::vector<uint8_t> head;
    head.insert(head.end(), {0x01, 0xF9, 0xEC, 0x01});                       // magic, format version 1 (DNA)
    head.push_back(static_cast<uint8_t>(0x08 | 0x02 | (spec->with_mask ? 0x04 : 0)));
    head.push_back(' ');
    put_varint(head, 60);
    put_varint(head, n_rec);
    std::vector<uint8_t> len_frame, mask_frame;
    raw_frame(len_bytes, len_frame);
    put_varint(head, len_bytes.size());
    put_varint(head, len_frame.size());
    head.insert(head.end(), len_frame.begin(), len_frame.end());
    if (spec->with_mask) {
        raw_frame(mask_bytes, mask_frame);
        put_varint(head, mask_bytes.size());
        put_varint(head, mask_frame.size());
        head.insert(head.end(), mask_frame.begin(), mask_frame.end());
    }
    put_varint(head, n_bases);                                               // nucleotides, not bytes (mod.rs:241)
    put_varint(head, seq_csize);
    const uint64_t total = head.size() + (head_only ? 2 : seq_csize);
    uint8_t *buf = static_cast<uint8_t *>(std::malloc(total));
    if (!buf) return NAFGPU_E_DEVICE;
    std::memcpy(buf, head.data(), head.size());
    uint64_t pos = head.size();
    buf[pos++] = 0x00;                                                       // FHD
    buf[pos++] = 0x48;                                                       // window 512 KiB
    for (auto &c : chunk_out) {
        std::memcpy(buf + pos, c.data(), c.size());
        pos += c.size();
        std::vector<uint8_t>().swap(c);
    }
    out->bytes = buf;
    out->n = total;
    out->n_records = n_rec;
    out->n_bases = n_bases;
    for (uint64_t h : chunk_hash) out->seq_hash += h;
    out->offsets_hash = nafgpu::hash64_host(reinterpret_cast<const uint8_t *>(rec_end.data()), n_rec * 8);
    return NAFGPU_OK;
}

extern "C" int nafgpu_synth_write(const nafgpu_synth_spec *spec, nafgpu_synth_archive *out) {
    return synth_impl(spec, out, false, 0);
}

extern "C" int nafgpu_synth_head(const nafgpu_synth_spec *spec, uint64_t seq_part_bytes, nafgpu_synth_archive *out) {
    return synth_impl(spec, out, true, seq_part_bytes);
}

extern "C" void nafgpu_synth_free(nafgpu_synth_archive *a) {
    if (a && a->bytes) std::free(a->bytes);
    if (a) std::memset(a, 0, sizeof *a);
}

// ======================================================================================
// Blocks with LZ sequences (the Encoder at compression levels >= 3)
// ======================================================================================
namespace {

// Literals_Section (RFC 8878 3.1.1.3.1) of `n` bytes: Huffman with four streams (new tree, or the previous block's), RLE, or raw.
// *used_new: the section carries a new tree (the caller makes it the previous one if the block is emitted).
void literals_section(const uint8_t *data, size_t n, const HufCode *prev, HufCode *cur, bool *used_new, std::vector<uint8_t> &out) {
    *used_new = false;
    auto raw = [&]() {
        if (n < 32) {
            out.push_back(static_cast<uint8_t>(n << 3));
        } else if (n < 4096) {
            const uint32_t v = (static_cast<uint32_t>(n) << 4) | (1u << 2);
            out.push_back(v & 0xFF);
            out.push_back(v >> 8);
        } else {
            const uint32_t v = (static_cast<uint32_t>(n) << 4) | (3u << 2);
            out.push_back(v & 0xFF);
            out.push_back((v >> 8) & 0xFF);
            out.push_back(v >> 16);
        }
        out.insert(out.end(), data, data + n);
    };
    if (n < 256) return raw();
    uint32_t count[256] = {0};
    for (size_t i = 0; i < n; i++) count[data[i]]++;
    if (count[data[0]] == n) {                                   // RLE literals
        const uint32_t v = (static_cast<uint32_t>(n) << 4) | (3u << 2) | 1u;
        out.push_back(v & 0xFF);
        out.push_back((v >> 8) & 0xFF);
        out.push_back(v >> 16);
        out.push_back(data[0]);
        return;
    }
    std::vector<uint8_t> tree;
    bool have_new = build_lengths(count, cur->len);
    if (have_new) {
        assign_codes(cur);
        have_new = cur->valid && write_weights(cur->weight, cur->max_sym, tree);
    }
    uint64_t cost_new = UINT64_MAX, cost_old = UINT64_MAX;
    if (have_new) {
        cost_new = tree.size() * 8;
        for (int k = 0; k < 256; k++) cost_new += static_cast<uint64_t>(count[k]) * cur->len[k];
    }
    if (prev->valid) {
        cost_old = 0;
        for (int k = 0; k < 256; k++) {
            if (!count[k]) continue;
            if (!prev->len[k]) {
                cost_old = UINT64_MAX;
                break;
            }
            cost_old += static_cast<uint64_t>(count[k]) * prev->len[k];
        }
    }
    if (cost_new == UINT64_MAX && cost_old == UINT64_MAX) return raw();
    const bool treeless = cost_old <= cost_new;
    const HufCode &h = treeless ? *prev : *cur;
    std::vector<uint8_t> body;
    if (!treeless) body = tree;
    const size_t q = (n + 3) / 4;
    std::vector<uint8_t> st[4];
    encode_stream(h, data, q, st[0]);
    encode_stream(h, data + q, q, st[1]);
    encode_stream(h, data + 2 * q, q, st[2]);
    encode_stream(h, data + 3 * q, n - 3 * q, st[3]);
    for (int k = 0; k < 3; k++) {
        if (st[k].size() > 0xFFFF) return raw();
        body.push_back(st[k].size() & 0xFF);
        body.push_back(static_cast<uint8_t>(st[k].size() >> 8));
    }
    for (int k = 0; k < 4; k++) body.insert(body.end(), st[k].begin(), st[k].end());
    const size_t comp = body.size();
    if (comp >= n) return raw();
    const uint32_t type = treeless ? 3u : 2u;
    if (n <= 1023 && comp <= 1023) {
        const uint32_t v = type | (1u << 2) | (static_cast<uint32_t>(n) << 4) | (static_cast<uint32_t>(comp) << 14);
        for (int k = 0; k < 3; k++) out.push_back((v >> (8 * k)) & 0xFF);
    } else if (n <= 16383 && comp <= 16383) {
        const uint32_t v = type | (2u << 2) | (static_cast<uint32_t>(n) << 4) | (static_cast<uint32_t>(comp) << 18);
        for (int k = 0; k < 4; k++) out.push_back((v >> (8 * k)) & 0xFF);
    } else {
        const uint64_t v = type | (3u << 2) | (static_cast<uint64_t>(n) << 4) | (static_cast<uint64_t>(comp) << 22);
        for (int k = 0; k < 5; k++) out.push_back((v >> (8 * k)) & 0xFF);
    }
    out.insert(out.end(), body.begin(), body.end());
    *used_new = !treeless;
}

// The predefined FSE tables of the sequence codes (RFC 8878 3.1.1.3.2.2), as the DECODER builds them; encoding walks them
// backwards: to encode symbol s in front of decoder state `next`, take the state of s whose range [base, base + 2^nb) holds `next`.
struct SeqTable {
    int al = 0, n_states = 0;
    uint8_t sym[64], nb[64];
    uint16_t base[64];
};
const int16_t kEncLL[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
const int16_t kEncML[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                            1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
const int16_t kEncOF[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
const uint32_t kEncLLBase[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512,
                                 1024, 2048, 4096, 8192, 16384, 32768, 65536};
const uint8_t kEncLLBits[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
const uint32_t kEncMLBase[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32,
                                 33, 34, 35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539};
const uint8_t kEncMLBits[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};

void seq_table_build(const int16_t *norm, int nsym, int al, SeqTable *t) {
    const int S = 1 << al;
    uint16_t next[64];
    int high = S - 1;
    t->al = al;
    t->n_states = S;
    for (int k = 0; k < nsym; k++) {
        if (norm[k] == -1) {
            t->sym[high--] = static_cast<uint8_t>(k);
            next[k] = 1;
        } else {
            next[k] = static_cast<uint16_t>(norm[k]);
        }
    }
    const int step = (S >> 1) + (S >> 3) + 3, mask = S - 1;
    int pos = 0;
    for (int k = 0; k < nsym; k++)
        for (int c = 0; c < norm[k]; c++) {
            t->sym[pos] = static_cast<uint8_t>(k);
            do pos = (pos + step) & mask;
            while (pos > high);
        }
    for (int i = 0; i < S; i++) {
        const uint16_t d = next[t->sym[i]]++;
        const int nb = al - highbit(d);
        t->nb[i] = static_cast<uint8_t>(nb);
        t->base[i] = static_cast<uint16_t>((d << nb) - S);
    }
}
// the state of symbol `s` that leads to decoder state `next` (next < 0: any state of s)
int seq_state_for(const SeqTable &t, int s, int next) {
    for (int i = 0; i < t.n_states; i++)
        if (t.sym[i] == s && (next < 0 || (next >= t.base[i] && next < t.base[i] + (1 << t.nb[i])))) return i;
    return -1;
}

struct LzSeq {
    uint32_t ll, ml, off;                                        // literal run, match length (>= 3), distance (>= 1)
};

int ll_code(uint32_t ll) {
    for (int c = 35; c >= 0; c--)
        if (ll >= kEncLLBase[c]) return c;
    return 0;
}
int ml_code(uint32_t ml) {
    for (int c = 52; c >= 0; c--)
        if (ml >= kEncMLBase[c]) return c;
    return 0;
}

// Sequences_Section with the three predefined tables: header, then the backward bitstream
void sequences_section(const std::vector<LzSeq> &seqs, std::vector<uint8_t> &out) {
    static const SeqTable *tabs = [] {
        static SeqTable t[3];
        seq_table_build(kEncLL, 36, 6, &t[0]);
        seq_table_build(kEncOF, 29, 5, &t[1]);
        seq_table_build(kEncML, 53, 6, &t[2]);
        return t;
    }();
    const SeqTable &TL = tabs[0], &TO = tabs[1], &TM = tabs[2];
    const size_t n = seqs.size();
    if (n < 128) {
        out.push_back(static_cast<uint8_t>(n));
    } else if (n < 0x7F00) {
        out.push_back(static_cast<uint8_t>((n >> 8) + 128));
        out.push_back(static_cast<uint8_t>(n & 0xFF));
    } else {
        out.push_back(255);
        out.push_back(static_cast<uint8_t>((n - 0x7F00) & 0xFF));
        out.push_back(static_cast<uint8_t>((n - 0x7F00) >> 8));
    }
    if (n == 0) return;
    out.push_back(0x00);                                         // Symbol_Compression_Modes: predefined x 3
    // What the decoder reads, first to last: initial states LL, OF, ML; then per sequence the extra bits OF, ML, LL and -- except
    // after the last one -- the state updates LL, ML, OF.  It reads from the end of the stream, so the fields are written last to first.
    BitSink bits;
    int s_ll = -1, s_of = -1, s_ml = -1;                         // decoder states of sequence i + 1
    for (size_t i = n; i-- > 0;) {
        const LzSeq &q = seqs[i];
        const uint32_t ofv = q.off + 3;                          // Offset_Value: always a new offset (no repeat codes)
        const int cl = ll_code(q.ll), cm = ml_code(q.ml), co = highbit(ofv);
        const int t_ll = seq_state_for(TL, cl, s_ll), t_of = seq_state_for(TO, co, s_of), t_ml = seq_state_for(TM, cm, s_ml);
        if (i + 1 < n) {                                         // the updates that lead from this sequence's states to the next one's
            bits.put(static_cast<uint32_t>(s_of - TO.base[t_of]), TO.nb[t_of]);
            bits.put(static_cast<uint32_t>(s_ml - TM.base[t_ml]), TM.nb[t_ml]);
            bits.put(static_cast<uint32_t>(s_ll - TL.base[t_ll]), TL.nb[t_ll]);
        }
        bits.put(q.ll - kEncLLBase[cl], kEncLLBits[cl]);
        bits.put(q.ml - kEncMLBase[cm], kEncMLBits[cm]);
        if (co > 24) {                                           // (BitSink takes at most 32 bits at a time safely)
            bits.put((ofv - (1u << co)) & 0xFFFFu, 16);
            bits.put((ofv - (1u << co)) >> 16, co - 16);
        } else {
            bits.put(ofv - (1u << co), co);
        }
        s_ll = t_ll;
        s_of = t_of;
        s_ml = t_ml;
    }
    bits.put(static_cast<uint32_t>(s_ml), TM.al);
    bits.put(static_cast<uint32_t>(s_of), TO.al);
    bits.put(static_cast<uint32_t>(s_ll), TL.al);
    bits.put(1, 1);                                              // end mark
    bits.flush();
    out.insert(out.end(), bits.bytes.begin(), bits.bytes.end());
}

constexpr uint32_t kLzHashBits = 17, kLzMinMatch = 6, kLzWindow = 1u << 20;
// One block of [data + b0, data + b0 + n) with greedy hash matching against everything since `chunk0` (and at most the window).
void encode_block_lz(const uint8_t *data, size_t chunk0, size_t b0, size_t n, bool last, HufCode *prev, std::vector<uint32_t> &head,
                     std::vector<uint8_t> &out) {
    std::vector<LzSeq> seqs;
    std::vector<uint8_t> lits;
    lits.reserve(n);
    auto hash = [&](size_t p) {
        uint32_t v;
        std::memcpy(&v, data + p, 4);
        return (v * 2654435761u) >> (32 - kLzHashBits);
    };
    const size_t end = b0 + n;
    size_t p = b0, lit0 = b0;
    while (p + kLzMinMatch <= end) {
        const uint32_t hsh = hash(p);
        const uint32_t cand = head[hsh];                         // position + 1 of an earlier occurrence in this chunk (0: none)
        head[hsh] = static_cast<uint32_t>(p - chunk0 + 1);
        size_t len = 0, from = 0;
        if (cand) {
            from = chunk0 + cand - 1;
            if (p - from <= kLzWindow - kBlockMax) {
                while (p + len < end && data[from + len] == data[p + len]) len++;
            }
        }
        if (len >= kLzMinMatch) {
            seqs.push_back(LzSeq{static_cast<uint32_t>(p - lit0), static_cast<uint32_t>(len), static_cast<uint32_t>(p - from)});
            lits.insert(lits.end(), data + lit0, data + p);
            for (size_t k = 1; k < len && p + k + 4 <= end; k += 3) head[hash(p + k)] = static_cast<uint32_t>(p + k - chunk0 + 1);
            p += len;
            lit0 = p;
        } else {
            p++;
        }
    }
    lits.insert(lits.end(), data + lit0, data + end);            // literals behind the last match
    if (seqs.empty()) return encode_block(data + b0, n, last, prev, out);
    std::vector<uint8_t> body;
    HufCode cur{};
    bool used_new = false;
    literals_section(lits.data(), lits.size(), prev, &cur, &used_new, body);
    sequences_section(seqs, body);
    if (body.size() >= n || body.size() > kBlockMax) {           // not worth it: without sequences (which may still be Huffman or raw)
        return encode_block(data + b0, n, last, prev, out);
    }
    const uint32_t bh = static_cast<uint32_t>(body.size() << 3) | (2u << 1) | (last ? 1u : 0u);
    out.push_back(bh & 0xFF);
    out.push_back((bh >> 8) & 0xFF);
    out.push_back((bh >> 16) & 0xFF);
    out.insert(out.end(), body.begin(), body.end());
    if (used_new) *prev = cur;
}

}  // namespace

// ======================================================================================
// Encoder (EncoderBuilder / Encoder / SequenceWriter: encoder/mod.rs:46-384, writer.rs:6-100)
// ======================================================================================
namespace {

// one section -> one magicless frame of 128 KiB Huffman-literal blocks; chunks of kChunkBlocks blocks in parallel
void compress_section(const std::vector<uint8_t> &data, unsigned n_threads, bool lz, std::vector<uint8_t> &out) {
    if (data.size() < 64) {                       // nothing to gain: raw blocks (an empty section is one empty last block)
        raw_frame(data, out);
        return;
    }
    const uint64_t n_blocks = (data.size() + kBlockMax - 1) / kBlockMax;
    const uint64_t n_chunks = (n_blocks + kChunkBlocks - 1) / kChunkBlocks;
    std::vector<std::vector<uint8_t>> chunk_out(n_chunks);
    std::atomic<uint64_t> next{0};
    auto worker = [&]() {
        for (;;) {
            const uint64_t c = next.fetch_add(1);
            if (c >= n_chunks) break;
            HufCode prev{};
            std::vector<uint32_t> head;                       // (LZ) hash heads: matches are looked for inside the chunk
            if (lz) head.assign(size_t(1) << kLzHashBits, 0);
            const size_t chunk0 = static_cast<size_t>(c * kChunkBlocks * kBlockMax);
            for (uint64_t b = c * kChunkBlocks; b < std::min(n_blocks, (c + 1) * kChunkBlocks); b++) {
                const size_t p0 = static_cast<size_t>(b * kBlockMax), bn = std::min<size_t>(kBlockMax, data.size() - p0);
                if (lz)
                    encode_block_lz(data.data(), chunk0, p0, bn, b == n_blocks - 1, &prev, head, chunk_out[c]);
                else
                    encode_block(data.data() + p0, bn, b == n_blocks - 1, &prev, chunk_out[c]);
            }
        }
    };
    if (n_threads == 0) n_threads = std::max(1u, std::thread::hardware_concurrency());
    n_threads = static_cast<unsigned>(std::min<uint64_t>(n_threads, n_chunks));
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < n_threads; t++) pool.emplace_back(worker);
    worker();
    for (auto &t : pool) t.join();
    out.push_back(0x00);   // FHD: no content size, no checksum, no dictionary
    out.push_back(lz ? 0x50 : 0x48);   // window 1 MiB when blocks refer back (kLzWindow), else 512 KiB
    for (auto &c : chunk_out) out.insert(out.end(), c.begin(), c.end());
}

// SequenceWriter::encode (writer.rs:31-56): upper-case IUPAC, '-' = 0; T for DNA, U for RNA
int nucleotide_code(uint8_t c, uint8_t sequence_type) {
    switch (c) {
    case 'A': return 0x08;
    case 'C': return 0x04;
    case 'G': return 0x02;
    case 'T': return sequence_type == 0 ? 0x01 : -1;
    case 'U': return sequence_type == 1 ? 0x01 : -1;
    case 'R': return 0x0A;
    case 'Y': return 0x05;
    case 'S': return 0x06;
    case 'W': return 0x09;
    case 'K': return 0x03;
    case 'M': return 0x0C;
    case 'B': return 0x07;
    case 'D': return 0x0B;
    case 'H': return 0x0D;
    case 'V': return 0x0E;
    case 'N': return 0x0F;
    case '-': return 0x00;
    default: return -1;
    }
}

void put_length_words(std::vector<uint8_t> &out, uint64_t l) {   // write_length, encoder/mod.rs:37-44
    auto word = [&](uint32_t w) {
        for (int k = 0; k < 4; k++) out.push_back(static_cast<uint8_t>(w >> (8 * k)));
    };
    while (l >= 0xFFFFFFFFull) {
        word(0xFFFFFFFFu);
        l -= 0xFFFFFFFFull;
    }
    word(static_cast<uint32_t>(l));
}

int enc_fail(nafgpu_error *err, int status, const char *msg) {
    if (err) {
        std::memset(err, 0, sizeof *err);
        err->status = status;
        std::snprintf(err->message, sizeof err->message, "%s", msg);
    }
    return status;
}

}  // namespace

struct nafgpu_encoder {
    nafgpu_encoder_opts opt{};
    std::vector<uint8_t> ids, coms, lens, seq, qual;   // section contents, uncompressed (Memory storage: storage.rs)
    uint64_t seq_letters = 0;                           // what the reference's WriteCounter counts for the sequence: letters
    int cache = -1;                                     // a nucleotide waiting for its partner (writer.rs:9,69-77)
    uint64_t n_records = 0;
    std::vector<uint8_t> archive;
    bool finished = false;
};

extern "C" void nafgpu_encoder_opts_default(uint8_t sequence_type, nafgpu_encoder_opts *opts) {
    if (!opts) return;
    std::memset(opts, 0, sizeof *opts);
    opts->sequence_type = sequence_type;
}

extern "C" void nafgpu_encoder_opts_from_flags(uint8_t sequence_type, uint8_t flags, nafgpu_encoder_opts *opts) {
    nafgpu_encoder_opts_default(sequence_type, opts);
    if (!opts) return;
    opts->id = (flags & 0x20) != 0;
    opts->comment = (flags & 0x10) != 0;
    opts->sequence = (flags & 0x02) != 0;
    opts->quality = (flags & 0x01) != 0;
}

extern "C" int nafgpu_encoder_new(const nafgpu_encoder_opts *opts, nafgpu_encoder **out, nafgpu_error *err) {
    if (!opts || !out || opts->sequence_type > 3) return enc_fail(err, NAFGPU_E_INVALID_ARG, "invalid encoder options");
    nafgpu_encoder *e = new (std::nothrow) nafgpu_encoder();
    if (!e) return enc_fail(err, NAFGPU_E_IO, "out of memory");
    e->opt = *opts;
    *out = e;
    return NAFGPU_OK;
}

extern "C" int nafgpu_encoder_push(nafgpu_encoder *e, const nafgpu_record *r, nafgpu_error *err) {
    if (!e || !r) return enc_fail(err, NAFGPU_E_INVALID_ARG, "null argument");
    if (e->finished) return enc_fail(err, NAFGPU_E_INVALID_ARG, "the archive has been written already");
    const bool nuc = e->opt.sequence_type <= 1;
    // ---- every check first (mod.rs:236-317 checks field by field, writing as it goes)
    if (e->opt.id && !r->id.present) return enc_fail(err, NAFGPU_E_MISSING_FIELD, "missing record field: \"id\"");
    if (e->opt.comment && !r->comment.present) return enc_fail(err, NAFGPU_E_MISSING_FIELD, "missing record field: \"comment\"");
    if (e->opt.sequence && !r->sequence.present) return enc_fail(err, NAFGPU_E_MISSING_FIELD, "missing record field: \"sequence\"");
    if (e->opt.sequence && r->has_length && r->length != r->sequence.len)
        return enc_fail(err, NAFGPU_E_INVALID_LENGTH, "inconsistent sequence length");
    if (e->opt.sequence && nuc)
        for (uint64_t i = 0; i < r->sequence.len; i++)
            if (nucleotide_code(r->sequence.ptr[i], e->opt.sequence_type) < 0)
                return enc_fail(err, NAFGPU_E_INVALID_SEQUENCE, "invalid character in sequence");
    if (e->opt.quality && !r->quality.present) return enc_fail(err, NAFGPU_E_MISSING_FIELD, "missing record field: \"quality\"");
    bool have_len = r->has_length != 0;
    uint64_t len = r->length;
    if (e->opt.sequence && !have_len) {
        have_len = true;
        len = r->sequence.len;
    }
    if (e->opt.quality && have_len && len != r->quality.len) return enc_fail(err, NAFGPU_E_INVALID_LENGTH, "inconsistent sequence length");
    // ---- commit
    if (r->has_length) put_length_words(e->lens, r->length);                         // mod.rs:239-242
    if (e->opt.id) {
        e->ids.insert(e->ids.end(), r->id.ptr, r->id.ptr + r->id.len);
        e->ids.push_back(0);
    }
    if (e->opt.comment) {
        e->coms.insert(e->coms.end(), r->comment.ptr, r->comment.ptr + r->comment.len);
        e->coms.push_back(0);
    }
    bool wrote_len = r->has_length != 0;
    if (e->opt.sequence) {
        if (!wrote_len) {
            put_length_words(e->lens, r->sequence.len);                              // mod.rs:278-282
            wrote_len = true;
        }
        const uint8_t *s = r->sequence.ptr;
        uint64_t n = r->sequence.len;
        e->seq_letters += n;
        if (!nuc) {
            e->seq.insert(e->seq.end(), s, s + n);
        } else if (n) {                                                              // writer.rs:60-93: two letters per byte, first in the low nibble
            if (e->cache >= 0) {
                e->seq.push_back(static_cast<uint8_t>((nucleotide_code(s[0], e->opt.sequence_type) << 4) | e->cache));
                e->cache = -1;
                s++;
                n--;
            }
            for (uint64_t i = 0; i + 1 < n; i += 2)
                e->seq.push_back(static_cast<uint8_t>((nucleotide_code(s[i + 1], e->opt.sequence_type) << 4) |
                                                      nucleotide_code(s[i], e->opt.sequence_type)));
            if (n & 1) e->cache = nucleotide_code(s[n - 1], e->opt.sequence_type);
        }
    }
    if (e->opt.quality) {
        if (!wrote_len) put_length_words(e->lens, r->quality.len);                   // mod.rs:308-312
        e->qual.insert(e->qual.end(), r->quality.ptr, r->quality.ptr + r->quality.len);
    }
    e->n_records++;
    return NAFGPU_OK;
}

extern "C" int nafgpu_encoder_finish(nafgpu_encoder *e, const uint8_t **bytes, uint64_t *n, nafgpu_error *err) {
    if (!e || !bytes || !n) return enc_fail(err, NAFGPU_E_INVALID_ARG, "null argument");
    if (!e->finished) {
        if (e->cache >= 0) {                                                         // SequenceWriter::into_inner, writer.rs:21-28
            e->seq.push_back(static_cast<uint8_t>(e->cache));
            e->cache = -1;
        }
        std::vector<uint8_t> &o = e->archive;
        o.insert(o.end(), {0x01, 0xF9, 0xEC});                                       // mod.rs:327
        uint8_t flags = 0;                                                           // mod.rs:176-193
        if (e->opt.id) flags |= 0x20;
        if (e->opt.comment) flags |= 0x10;
        if (e->opt.sequence) flags |= 0x02 | 0x08;
        if (e->opt.quality) flags |= 0x01 | 0x08;
        if (e->opt.sequence_type == 0) {                                             // V1 for DNA, V2 else (mod.rs:169-173, 329-342)
            o.insert(o.end(), {0x01, flags, ' '});
        } else {
            o.insert(o.end(), {0x02, e->opt.sequence_type, flags, ' '});
        }
        put_varint(o, 60);                                                           // Header::default().line_length (data.rs:246)
        put_varint(o, e->n_records);
        auto block = [&](const std::vector<uint8_t> &data, uint64_t original) {      // write_block!, mod.rs:349-367
            std::vector<uint8_t> frame;
            compress_section(data, e->opt.threads, e->opt.compression_level == 0 || e->opt.compression_level >= 3, frame);
            put_varint(o, original);
            put_varint(o, frame.size());
            o.insert(o.end(), frame.begin(), frame.end());
        };
        if (e->opt.id) block(e->ids, e->ids.size());
        if (e->opt.comment) block(e->coms, e->coms.size());
        block(e->lens, e->lens.size());                                              // always, whatever the flags say (mod.rs:371)
        if (e->opt.sequence) block(e->seq, e->seq_letters);                          // letters, not bytes (the counter wraps the SequenceWriter)
        if (e->opt.quality) block(e->qual, e->qual.size());
        e->finished = true;
        std::vector<uint8_t>().swap(e->ids);
        std::vector<uint8_t>().swap(e->coms);
        std::vector<uint8_t>().swap(e->lens);
        std::vector<uint8_t>().swap(e->seq);
        std::vector<uint8_t>().swap(e->qual);
    }
    *bytes = e->archive.data();
    *n = e->archive.size();
    return NAFGPU_OK;
}

extern "C" void nafgpu_encoder_free(nafgpu_encoder *e) { delete e; }
