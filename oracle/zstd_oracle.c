/*
 * oracle/zstd_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of Zstandard frame *decoding* (RFC 8878), written from
 * the format digest in SURVEY.md Appendix B.  It stands in for the third-party
 * dependency the reference calls for every NAF section:
 *
 *     crate `zstd` ^0.13.1 (feature "experimental") -> zstd-safe -> zstd-sys ->
 *     bundled libzstd 1.5.x     (nafcodec/Cargo.toml:16-18; Cargo.lock is
 *     git-ignored, so the exact version is unpinned)
 *     call site: nafcodec/src/decoder/mod.rs:32, 221-223
 *         zstd::stream::read::Decoder::new(slice) + include_magicbytes(false)
 *
 * Zstandard decoding is fully specified: every conforming decoder yields the
 * same bytes, so this restatement is pinned by (a) the reference's own
 * fixtures data/<name>.naf whose decoded text is in data/{LuxC.faa,masked.fna,
 * phix.fastq} and (b) a byte-for-byte cross-check against the system libzstd
 * in tests/test_oracle_zstd.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call this file.  The product (nafcodec_amd/) never links or loads it.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include "naf_oracle.h"

#define ZO_OK 0
#define ZERR(code) return -(code)

/* error codes (negative on return) */
enum {
    ZO_E_TRUNCATED = 1,
    ZO_E_CORRUPT = 2,
    ZO_E_DSTFULL = 3,
    ZO_E_UNSUPPORTED = 4,
    ZO_E_NOMEM = 5,
};

/* ------------------------------------------------------------------ */
/* forward (LSB-first) bit reader: FSE table descriptions              */
/* ------------------------------------------------------------------ */
typedef struct {
    const uint8_t *p;
    size_t n;
    size_t bit; /* absolute bit cursor */
} fwd_bits;

static uint32_t fwd_peek(const fwd_bits *b, int nb)
{
    uint64_t v = 0;
    size_t byte = b->bit >> 3;
    size_t avail = byte < b->n ? b->n - byte : 0;
    if (avail > 8)
        avail = 8;
    memcpy(&v, b->p + byte, avail);
    v >>= (b->bit & 7);
    return (uint32_t)(v & ((1ull << nb) - 1));
}

/* ------------------------------------------------------------------ */
/* backward bit reader (App. B "Backward bitstreams")                  */
/* ------------------------------------------------------------------ */
typedef struct {
    const uint8_t *p;
    size_t n;
    int64_t pos; /* number of unread bits below the cursor; may go negative */
} back_bits;

static int back_init(back_bits *b, const uint8_t *p, size_t n)
{
    if (n == 0)
        return -ZO_E_CORRUPT;
    uint8_t last = p[n - 1];
    if (last == 0)
        return -ZO_E_CORRUPT; /* no sentinel bit */
    int hb = 7;
    while (!((last >> hb) & 1))
        hb--;
    b->p = p;
    b->n = n;
    b->pos = (int64_t)(n - 1) * 8 + hb; /* bits strictly below the sentinel */
    return 0;
}

/* bits [lo, lo+nb) of the little-endian integer; bits below 0 read as zero */
static uint64_t back_slice(const back_bits *b, int64_t lo, int nb)
{
    if (nb <= 0)
        return 0;
    if (lo < 0) {
        int64_t neg = -lo;
        if (neg >= nb)
            return 0;
        return back_slice(b, 0, nb - (int)neg) << neg;
    }
    size_t byte = (size_t)(lo >> 3);
    int sh = (int)(lo & 7);
    uint64_t v = 0;
    size_t avail = byte < b->n ? b->n - byte : 0;
    if (avail > 8)
        avail = 8;
    memcpy(&v, b->p + byte, avail);
    return (v >> sh) & ((1ull << nb) - 1); /* nb <= 32 everywhere below */
}

static uint32_t back_read(back_bits *b, int nb)
{
    b->pos -= nb;
    return (uint32_t)back_slice(b, b->pos, nb);
}

static uint32_t back_peek(const back_bits *b, int nb)
{
    return (uint32_t)back_slice(b, b->pos - nb, nb);
}

/* ------------------------------------------------------------------ */
/* FSE                                                                 */
/* ------------------------------------------------------------------ */
typedef struct {
    uint8_t sym;
    uint8_t nb;
    uint16_t base;
} fse_cell;

typedef struct {
    int al; /* accuracy log */
    fse_cell cell[512];
} fse_table;

static int highbit32(uint32_t v)
{
    int r = 0;
    while (v >>= 1)
        r++;
    return r;
}

/* App. B "FSE table description"; returns bytes consumed or <0 */
static long fse_read_dist(const uint8_t *src, size_t n, int max_al, int max_sym,
                          int16_t *norm, int *nsym_out, int *al_out)
{
    fwd_bits b = {src, n, 0};
    if (n == 0)
        return -ZO_E_TRUNCATED;
    int al = (int)fwd_peek(&b, 4) + 5;
    b.bit += 4;
    if (al > max_al)
        return -ZO_E_CORRUPT;
    int remaining = 1 << al;
    int sym = 0;
    while (remaining > 0 && sym <= max_sym) {
        int max = remaining + 1;
        int bits = highbit32((uint32_t)max) + 1;
        uint32_t v = fwd_peek(&b, bits);
        uint32_t low = (1u << (bits - 1)) - 1;
        uint32_t thr = (1u << bits) - 1 - (uint32_t)max;
        if ((v & low) < thr) {
            b.bit += bits - 1;
            v &= low;
        } else {
            b.bit += bits;
            if (v > low)
                v -= thr;
        }
        int p = (int)v - 1;
        norm[sym++] = (int16_t)p;
        remaining -= p < 0 ? -p : p;
        if (p == 0) {
            for (;;) {
                uint32_t rep = fwd_peek(&b, 2);
                b.bit += 2;
                for (uint32_t i = 0; i < rep && sym <= max_sym; i++)
                    norm[sym++] = 0;
                if (rep != 3)
                    break;
            }
        }
        if ((b.bit >> 3) > n)
            return -ZO_E_TRUNCATED;
    }
    if (remaining != 0)
        return -ZO_E_CORRUPT;
    size_t used = (b.bit + 7) >> 3;
    if (used > n)
        return -ZO_E_TRUNCATED;
    *nsym_out = sym;
    *al_out = al;
    return (long)used;
}

/* App. B "FSE table build" */
static int fse_build(fse_table *t, const int16_t *norm, int nsym, int al)
{
    int S = 1 << al;
    uint16_t next[256];
    int high = S - 1;
    t->al = al;
    for (int s = 0; s < nsym; s++) {
        if (norm[s] == -1) {
            t->cell[high--].sym = (uint8_t)s;
            next[s] = 1;
        } else {
            next[s] = (uint16_t)norm[s];
        }
    }
    int step = (S >> 1) + (S >> 3) + 3;
    int mask = S - 1;
    int pos = 0;
    for (int s = 0; s < nsym; s++) {
        for (int i = 0; i < norm[s]; i++) {
            t->cell[pos].sym = (uint8_t)s;
            do {
                pos = (pos + step) & mask;
            } while (pos > high);
        }
    }
    if (pos != 0)
        return -ZO_E_CORRUPT;
    for (int i = 0; i < S; i++) {
        uint8_t s = t->cell[i].sym;
        uint16_t d = next[s]++;
        int nb = al - highbit32(d);
        t->cell[i].nb = (uint8_t)nb;
        t->cell[i].base = (uint16_t)((d << nb) - S);
    }
    return 0;
}

static void fse_build_rle(fse_table *t, uint8_t sym)
{
    t->al = 0;
    t->cell[0].sym = sym;
    t->cell[0].nb = 0;
    t->cell[0].base = 0;
}

/* App. B "Predefined distributions" */
static const int16_t LL_DEFAULT[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2,
                                       2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
static const int16_t ML_DEFAULT[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                       1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                       1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
static const int16_t OF_DEFAULT[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1,
                                       1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};

/* App. B "Code->value" */
static const uint32_t LL_BASE[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18,
                                     20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048,
                                     4096, 8192, 16384, 32768, 65536};
static const uint8_t LL_BITS[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1,
                                    1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static const uint32_t ML_BASE[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20,
                                     21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 37,
                                     39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051,
                                     4099, 8195, 16387, 32771, 65539};
static const uint8_t ML_BITS[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1,
                                    2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};

/* ------------------------------------------------------------------ */
/* Huffman                                                             */
/* ------------------------------------------------------------------ */
typedef struct {
    uint8_t sym;
    uint8_t len;
} huf_cell;

typedef struct {
    int max_bits; /* 0 = no table yet */
    huf_cell cell[1 << 11];
} huf_table;

/* App. B "Huffman tree description"; returns bytes consumed or <0 */
static long huf_read_table(huf_table *t, const uint8_t *src, size_t n)
{
    uint8_t w[258];
    int nw = 0;
    if (n < 1)
        return -ZO_E_TRUNCATED;
    int hb = src[0];
    size_t used;
    if (hb >= 128) {
        nw = hb - 127;
        size_t bytes = (size_t)(nw + 1) / 2;
        if (1 + bytes > n)
            return -ZO_E_TRUNCATED;
        for (int i = 0; i < nw; i++) {
            uint8_t b = src[1 + i / 2];
            w[i] = (i & 1) ? (b & 0xF) : (b >> 4);
        }
        used = 1 + bytes;
    } else {
        if ((size_t)hb + 1 > n || hb == 0)
            return -ZO_E_TRUNCATED;
        int16_t norm[256];
        int nsym, al;
        long r = fse_read_dist(src + 1, (size_t)hb, 6, 255, norm, &nsym, &al);
        if (r < 0)
            return r;
        fse_table ft;
        if (fse_build(&ft, norm, nsym, al) < 0)
            return -ZO_E_CORRUPT;
        back_bits b;
        if (back_init(&b, src + 1 + r, (size_t)hb - (size_t)r) < 0)
            return -ZO_E_CORRUPT;
        uint32_t s1 = back_read(&b, al);
        uint32_t s2 = back_read(&b, al);
        /* two interleaved states; stop when the stream is exhausted */
        for (;;) {
            if (nw >= 255)
                return -ZO_E_CORRUPT;
            w[nw++] = ft.cell[s1].sym;
            if (b.pos < (int64_t)ft.cell[s1].nb) {
                /* not enough bits to update s1: flush s2 and stop */
                w[nw++] = ft.cell[s2].sym;
                break;
            }
            s1 = ft.cell[s1].base + back_read(&b, ft.cell[s1].nb);
            if (nw >= 255)
                return -ZO_E_CORRUPT;
            w[nw++] = ft.cell[s2].sym;
            if (b.pos < (int64_t)ft.cell[s2].nb) {
                w[nw++] = ft.cell[s1].sym;
                break;
            }
            s2 = ft.cell[s2].base + back_read(&b, ft.cell[s2].nb);
        }
        used = 1 + (size_t)hb;
    }
    if (nw > 255)
        return -ZO_E_CORRUPT;
    /* implicit last weight */
    uint32_t total = 0;
    for (int i = 0; i < nw; i++) {
        if (w[i] > 11)
            return -ZO_E_CORRUPT;
        if (w[i])
            total += 1u << (w[i] - 1);
    }
    if (total == 0)
        return -ZO_E_CORRUPT;
    int max_bits = highbit32(total) + 1;
    if (max_bits > 11)
        return -ZO_E_CORRUPT;
    uint32_t left = (1u << max_bits) - total;
    if (left & (left - 1))
        return -ZO_E_CORRUPT; /* must be a power of two */
    w[nw++] = (uint8_t)(highbit32(left) + 1);
    /* table fill: weight 1 (longest codes) first, ascending symbol inside a weight */
    uint32_t pos = 0;
    for (int wt = 1; wt <= max_bits; wt++) {
        uint32_t span = 1u << (wt - 1);
        for (int s = 0; s < nw; s++) {
            if (w[s] != wt)
                continue;
            for (uint32_t k = 0; k < span; k++) {
                t->cell[pos + k].sym = (uint8_t)s;
                t->cell[pos + k].len = (uint8_t)(max_bits + 1 - wt);
            }
            pos += span;
        }
    }
    if (pos != (1u << max_bits))
        return -ZO_E_CORRUPT;
    t->max_bits = max_bits;
    return (long)used;
}

static int huf_decode_stream(const huf_table *t, const uint8_t *src, size_t n, uint8_t *dst,
                             size_t count)
{
    back_bits b;
    if (back_init(&b, src, n) < 0)
        return -ZO_E_CORRUPT;
    int mb = t->max_bits;
    uint32_t mask = (1u << mb) - 1;
    size_t i = 0;
    /* bulk: one unaligned 64-bit load serves five symbols (5 x 11 bits <= 57) */
    while (count - i >= 5 && b.pos >= 57) {
        int64_t lo = b.pos - 57;
        size_t byte = (size_t)(lo >> 3);
        if (byte + 8 > n)
            break;
        uint64_t v;
        memcpy(&v, src + byte, 8);
        v >>= (lo & 7); /* bits [0,57) of v = stream bits [lo, pos) */
        int used = 0;
        for (int k = 0; k < 5; k++) {
            huf_cell c = t->cell[(v >> (57 - used - mb)) & mask];
            dst[i++] = c.sym;
            used += c.len;
        }
        b.pos -= used;
    }
    for (; i < count; i++) {
        uint32_t v = back_peek(&b, mb);
        huf_cell c = t->cell[v];
        dst[i] = c.sym;
        b.pos -= c.len;
    }
    if (b.pos != 0)
        return -ZO_E_CORRUPT; /* App. B: a stream must end exactly at its start */
    return 0;
}

/* ------------------------------------------------------------------ */
/* frame state                                                         */
/* ------------------------------------------------------------------ */
typedef struct {
    huf_table huf;
    fse_table ll, of, ml;
    int have_ll, have_of, have_ml;
    uint64_t rep[3];
    uint8_t *lit; /* literal scratch, 128 KiB + slack */
    zo_stats *st;
} frame_ctx;

static long decode_literals(frame_ctx *fc, const uint8_t *src, size_t n, size_t *lit_size)
{
    if (n < 1)
        return -ZO_E_TRUNCATED;
    int type = src[0] & 3;
    int sf = (src[0] >> 2) & 3;
    if (type <= 1) { /* Raw / RLE */
        size_t regen, hdr;
        if (sf == 0 || sf == 2) {
            regen = src[0] >> 3;
            hdr = 1;
        } else if (sf == 1) {
            if (n < 2)
                return -ZO_E_TRUNCATED;
            regen = (src[0] >> 4) + ((size_t)src[1] << 4);
            hdr = 2;
        } else {
            if (n < 3)
                return -ZO_E_TRUNCATED;
            regen = (src[0] >> 4) + ((size_t)src[1] << 4) + ((size_t)src[2] << 12);
            hdr = 3;
        }
        if (regen > (128u << 10))
            return -ZO_E_CORRUPT;
        if (type == 0) {
            if (hdr + regen > n)
                return -ZO_E_TRUNCATED;
            memcpy(fc->lit, src + hdr, regen);
            *lit_size = regen;
            if (fc->st)
                fc->st->lit_raw++;
            return (long)(hdr + regen);
        }
        if (hdr + 1 > n)
            return -ZO_E_TRUNCATED;
        memset(fc->lit, src[hdr], regen);
        *lit_size = regen;
        if (fc->st)
            fc->st->lit_rle++;
        return (long)(hdr + 1);
    }
    /* Compressed (2) / Treeless (3) */
    size_t regen, comp, hdr;
    int streams;
    if (sf == 0 || sf == 1) {
        if (n < 3)
            return -ZO_E_TRUNCATED;
        uint32_t v = src[0] | ((uint32_t)src[1] << 8) | ((uint32_t)src[2] << 16);
        regen = (v >> 4) & 0x3FF;
        comp = (v >> 14) & 0x3FF;
        hdr = 3;
        streams = sf == 0 ? 1 : 4;
    } else if (sf == 2) {
        if (n < 4)
            return -ZO_E_TRUNCATED;
        uint32_t v = src[0] | ((uint32_t)src[1] << 8) | ((uint32_t)src[2] << 16) |
                     ((uint32_t)src[3] << 24);
        regen = (v >> 4) & 0x3FFF;
        comp = (v >> 18) & 0x3FFF;
        hdr = 4;
        streams = 4;
    } else {
        if (n < 5)
            return -ZO_E_TRUNCATED;
        uint64_t v = src[0] | ((uint64_t)src[1] << 8) | ((uint64_t)src[2] << 16) |
                     ((uint64_t)src[3] << 24) | ((uint64_t)src[4] << 32);
        regen = (size_t)((v >> 4) & 0x3FFFF);
        comp = (size_t)((v >> 22) & 0x3FFFF);
        hdr = 5;
        streams = 4;
    }
    if (regen > (128u << 10))
        return -ZO_E_CORRUPT;
    if (hdr + comp > n)
        return -ZO_E_TRUNCATED;
    const uint8_t *p = src + hdr;
    size_t rem = comp;
    if (type == 2) {
        long r = huf_read_table(&fc->huf, p, rem);
        if (r < 0)
            return r;
        p += r;
        rem -= (size_t)r;
        if (fc->st)
            fc->st->lit_huf++;
    } else {
        if (fc->huf.max_bits == 0)
            return -ZO_E_CORRUPT; /* treeless without a previous table */
        if (fc->st)
            fc->st->lit_treeless++;
    }
    if (streams == 1) {
        int r = huf_decode_stream(&fc->huf, p, rem, fc->lit, regen);
        if (r < 0)
            return r;
    } else {
        if (rem < 6)
            return -ZO_E_TRUNCATED;
        size_t s1 = p[0] | ((size_t)p[1] << 8);
        size_t s2 = p[2] | ((size_t)p[3] << 8);
        size_t s3 = p[4] | ((size_t)p[5] << 8);
        if (6 + s1 + s2 + s3 > rem)
            return -ZO_E_CORRUPT;
        size_t s4 = rem - 6 - s1 - s2 - s3;
        size_t q = (regen + 3) / 4;
        if (3 * q > regen)
            return -ZO_E_CORRUPT;
        const uint8_t *b = p + 6;
        int r;
        if ((r = huf_decode_stream(&fc->huf, b, s1, fc->lit, q)) < 0)
            return r;
        if ((r = huf_decode_stream(&fc->huf, b + s1, s2, fc->lit + q, q)) < 0)
            return r;
        if ((r = huf_decode_stream(&fc->huf, b + s1 + s2, s3, fc->lit + 2 * q, q)) < 0)
            return r;
        if ((r = huf_decode_stream(&fc->huf, b + s1 + s2 + s3, s4, fc->lit + 3 * q,
                                   regen - 3 * q)) < 0)
            return r;
    }
    *lit_size = regen;
    if (fc->st)
        fc->st->lit_bytes_entropy += regen;
    return (long)(hdr + comp);
}

/* sets up one of the three sequence tables; returns bytes consumed or <0 */
static long setup_seq_table(fse_table *t, int *have, int mode, const uint8_t *src, size_t n,
                            const int16_t *def, int def_n, int def_al, int max_al, int max_sym)
{
    switch (mode) {
    case 0:
        if (fse_build(t, def, def_n, def_al) < 0)
            return -ZO_E_CORRUPT;
        *have = 1;
        return 0;
    case 1:
        if (n < 1)
            return -ZO_E_TRUNCATED;
        if (src[0] > max_sym)
            return -ZO_E_CORRUPT;
        fse_build_rle(t, src[0]);
        *have = 1;
        return 1;
    case 2: {
        int16_t norm[256];
        int nsym, al;
        long r = fse_read_dist(src, n, max_al, max_sym, norm, &nsym, &al);
        if (r < 0)
            return r;
        if (fse_build(t, norm, nsym, al) < 0)
            return -ZO_E_CORRUPT;
        *have = 1;
        return r;
    }
    default:
        if (!*have)
            return -ZO_E_CORRUPT; /* repeat without a previous table */
        return 0;
    }
}

static int decode_block(frame_ctx *fc, const uint8_t *src, size_t n, uint8_t *dst_base,
                        size_t *dpos, size_t dcap, size_t frame_start, size_t window)
{
    size_t lit_size = 0;
    long r = decode_literals(fc, src, n, &lit_size);
    if (r < 0)
        return (int)r;
    const uint8_t *p = src + r;
    size_t rem = n - (size_t)r;
    if (rem < 1)
        return -ZO_E_TRUNCATED;
    size_t nseq;
    if (p[0] == 0) {
        nseq = 0;
        p += 1;
        rem -= 1;
    } else if (p[0] < 128) {
        nseq = p[0];
        p += 1;
        rem -= 1;
    } else if (p[0] < 255) {
        if (rem < 2)
            return -ZO_E_TRUNCATED;
        nseq = ((size_t)(p[0] - 128) << 8) + p[1];
        p += 2;
        rem -= 2;
    } else {
        if (rem < 3)
            return -ZO_E_TRUNCATED;
        nseq = (size_t)p[1] + ((size_t)p[2] << 8) + 0x7F00;
        p += 3;
        rem -= 3;
    }
    if (fc->st) {
        fc->st->sequences += nseq;
        fc->st->lit_bytes += lit_size;
    }
    uint8_t *out = dst_base;
    size_t o = *dpos;
    if (nseq == 0) {
        if (rem != 0)
            return -ZO_E_CORRUPT;
        if (o + lit_size > dcap)
            return -ZO_E_DSTFULL;
        memcpy(out + o, fc->lit, lit_size);
        *dpos = o + lit_size;
        return 0;
    }
    if (rem < 1)
        return -ZO_E_TRUNCATED;
    int modes = p[0];
    if (modes & 3)
        return -ZO_E_CORRUPT; /* reserved bits */
    p += 1;
    rem -= 1;
    if (fc->st) {
        fc->st->seq_mode_count[(modes >> 6) & 3]++;
        fc->st->seq_mode_count[4 + ((modes >> 4) & 3)]++;
        fc->st->seq_mode_count[8 + ((modes >> 2) & 3)]++;
    }
    r = setup_seq_table(&fc->ll, &fc->have_ll, (modes >> 6) & 3, p, rem, LL_DEFAULT, 36, 6, 9, 35);
    if (r < 0)
        return (int)r;
    p += r;
    rem -= (size_t)r;
    r = setup_seq_table(&fc->of, &fc->have_of, (modes >> 4) & 3, p, rem, OF_DEFAULT, 29, 5, 8, 31);
    if (r < 0)
        return (int)r;
    p += r;
    rem -= (size_t)r;
    r = setup_seq_table(&fc->ml, &fc->have_ml, (modes >> 2) & 3, p, rem, ML_DEFAULT, 53, 6, 9, 52);
    if (r < 0)
        return (int)r;
    p += r;
    rem -= (size_t)r;

    back_bits b;
    if (back_init(&b, p, rem) < 0)
        return -ZO_E_CORRUPT;
    uint32_t sll = back_read(&b, fc->ll.al);
    uint32_t sof = back_read(&b, fc->of.al);
    uint32_t sml = back_read(&b, fc->ml.al);
    size_t lpos = 0;
    for (size_t i = 0; i < nseq; i++) {
        uint8_t llc = fc->ll.cell[sll].sym;
        uint8_t ofc = fc->of.cell[sof].sym;
        uint8_t mlc = fc->ml.cell[sml].sym;
        if (llc > 35 || mlc > 52 || ofc > 31)
            return -ZO_E_CORRUPT;
        /* extra bits in order OF, ML, LL */
        uint64_t ov = ((uint64_t)1 << ofc) + back_read(&b, ofc);
        uint32_t ml = ML_BASE[mlc] + back_read(&b, ML_BITS[mlc]);
        uint32_t ll = LL_BASE[llc] + back_read(&b, LL_BITS[llc]);
        /* state updates LL, ML, OF -- not after the last sequence */
        if (i + 1 < nseq) {
            sll = fc->ll.cell[sll].base + back_read(&b, fc->ll.cell[sll].nb);
            sml = fc->ml.cell[sml].base + back_read(&b, fc->ml.cell[sml].nb);
            sof = fc->of.cell[sof].base + back_read(&b, fc->of.cell[sof].nb);
        }
        if (b.pos < 0)
            return -ZO_E_CORRUPT;
        /* repeat offsets */
        uint64_t off;
        if (ov > 3) {
            off = ov - 3;
            fc->rep[2] = fc->rep[1];
            fc->rep[1] = fc->rep[0];
            fc->rep[0] = off;
        } else {
            uint32_t idx = (uint32_t)ov - 1 + (ll == 0);
            if (idx == 0) {
                off = fc->rep[0];
            } else {
                off = idx < 3 ? fc->rep[idx] : fc->rep[0] - 1;
                if (off == 0)
                    return -ZO_E_CORRUPT;
                if (idx > 1)
                    fc->rep[2] = fc->rep[1];
                fc->rep[1] = fc->rep[0];
                fc->rep[0] = off;
            }
        }
        /* execute */
        if (lpos + ll > lit_size)
            return -ZO_E_CORRUPT;
        if (o + ll + ml > dcap)
            return -ZO_E_DSTFULL;
        memcpy(out + o, fc->lit + lpos, ll);
        lpos += ll;
        o += ll;
        if (off > o - frame_start || off > window)
            return -ZO_E_CORRUPT;
        for (uint32_t k = 0; k < ml; k++)
            out[o + k] = out[o + k - off]; /* byte-serial: overlap allowed */
        o += ml;
        if (fc->st)
            fc->st->match_bytes += ml;
    }
    if (b.pos != 0)
        return -ZO_E_CORRUPT;
    size_t tail = lit_size - lpos;
    if (o + tail > dcap)
        return -ZO_E_DSTFULL;
    memcpy(out + o, fc->lit + lpos, tail);
    *dpos = o + tail;
    return 0;
}

/*
 * XXH64, seed 0 (the xxHash specification, as used by RFC 8878 3.1.1 for the
 * Content_Checksum field).
 */
static uint64_t zo_rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static uint64_t zo_rd64(const uint8_t *p)
{
    uint64_t v = 0;
    for (int k = 7; k >= 0; k--)
        v = (v << 8) | p[k];
    return v;
}
uint64_t zo_xxh64(const uint8_t *p, size_t n)
{
    const uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL, P3 = 1609587929392839161ULL,
                   P4 = 9650029242287828579ULL, P5 = 2870177450012600261ULL;
    const uint8_t *end = p + n;
    uint64_t h;
    if (n >= 32) {
        uint64_t v[4] = {P1 + P2, P2, 0, 0 - P1};
        for (; p + 32 <= end; p += 32)
            for (int k = 0; k < 4; k++)
                v[k] = zo_rotl64(v[k] + zo_rd64(p + 8 * k) * P2, 31) * P1;
        h = zo_rotl64(v[0], 1) + zo_rotl64(v[1], 7) + zo_rotl64(v[2], 12) + zo_rotl64(v[3], 18);
        for (int k = 0; k < 4; k++)
            h = (h ^ (zo_rotl64(v[k] * P2, 31) * P1)) * P1 + P4;
    } else {
        h = P5;
    }
    h += (uint64_t)n;
    for (; p + 8 <= end; p += 8) {
        h ^= zo_rotl64(zo_rd64(p) * P2, 31) * P1;
        h = zo_rotl64(h, 27) * P1 + P4;
    }
    if (p + 4 <= end) {
        uint64_t x = (uint64_t)p[0] | ((uint64_t)p[1] << 8) | ((uint64_t)p[2] << 16) | ((uint64_t)p[3] << 24);
        h ^= x * P1;
        h = zo_rotl64(h, 23) * P2 + P3;
        p += 4;
    }
    for (; p < end; p++) {
        h ^= (uint64_t)*p * P5;
        h = zo_rotl64(h, 11) * P1;
    }
    h ^= h >> 33;
    h *= P2;
    h ^= h >> 29;
    h *= P3;
    h ^= h >> 32;
    return h;
}

/*
 * Decode ONE magicless frame starting at src.  *consumed and *dpos are updated.
 * (mod.rs:221-222: the reference feeds the section payload to a zstd stream
 * decoder configured with include_magicbytes(false).)
 */
static int decode_frame(const uint8_t *src, size_t n, size_t *consumed, uint8_t *dst, size_t *dpos,
                        size_t dcap, zo_stats *st)
{
    size_t i = 0;
    if (n < 1)
        return -ZO_E_TRUNCATED;
    uint8_t fhd = src[i++];
    int fcs_flag = fhd >> 6;
    int single = (fhd >> 5) & 1;
    int checksum = (fhd >> 2) & 1;
    int dict_flag = fhd & 3;
    if (fhd & 0x08)
        return -ZO_E_CORRUPT; /* reserved bit */
    size_t window = 0;
    if (!single) {
        if (i >= n)
            return -ZO_E_TRUNCATED;
        uint8_t wd = src[i++];
        int e = wd >> 3, m = wd & 7;
        size_t base = (size_t)1 << (10 + e);
        window = base + (base / 8) * (size_t)m;
    }
    static const int dict_bytes[4] = {0, 1, 2, 4};
    if (i + (size_t)dict_bytes[dict_flag] > n)
        return -ZO_E_TRUNCATED;
    uint32_t dict_id = 0;
    for (int k = 0; k < dict_bytes[dict_flag]; k++)
        dict_id |= (uint32_t)src[i + (size_t)k] << (8 * k);
    i += (size_t)dict_bytes[dict_flag];
    if (dict_id != 0)
        return -ZO_E_UNSUPPORTED;
    int fcs_bytes = fcs_flag == 0 ? (single ? 1 : 0) : (fcs_flag == 1 ? 2 : (fcs_flag == 2 ? 4 : 8));
    if (i + (size_t)fcs_bytes > n)
        return -ZO_E_TRUNCATED;
    uint64_t fcs = 0;
    for (int k = 0; k < fcs_bytes; k++)
        fcs |= (uint64_t)src[i + (size_t)k] << (8 * k);
    if (fcs_bytes == 2)
        fcs += 256;
    i += (size_t)fcs_bytes;
    if (single)
        window = (size_t)fcs;
    if (st) {
        st->frames++;
        st->window = window;
    }

    frame_ctx *fc = (frame_ctx *)calloc(1, sizeof(frame_ctx));
    if (!fc)
        return -ZO_E_NOMEM;
    fc->lit = (uint8_t *)malloc((128u << 10) + 64);
    if (!fc->lit) {
        free(fc);
        return -ZO_E_NOMEM;
    }
    fc->rep[0] = 1;
    fc->rep[1] = 4;
    fc->rep[2] = 8;
    fc->st = st;
    size_t frame_start = *dpos;
    int rc = 0;
    for (;;) {
        if (i + 3 > n) {
            rc = -ZO_E_TRUNCATED;
            break;
        }
        uint32_t bh = src[i] | ((uint32_t)src[i + 1] << 8) | ((uint32_t)src[i + 2] << 16);
        i += 3;
        int last = bh & 1;
        int type = (bh >> 1) & 3;
        size_t bsize = bh >> 3;
        if (st)
            st->blocks++;
        if (type == 0) {
            if (i + bsize > n) {
                rc = -ZO_E_TRUNCATED;
                break;
            }
            if (*dpos + bsize > dcap) {
                rc = -ZO_E_DSTFULL;
                break;
            }
            memcpy(dst + *dpos, src + i, bsize);
            *dpos += bsize;
            i += bsize;
            if (st)
                st->blocks_raw++;
        } else if (type == 1) {
            if (i + 1 > n) {
                rc = -ZO_E_TRUNCATED;
                break;
            }
            if (*dpos + bsize > dcap) {
                rc = -ZO_E_DSTFULL;
                break;
            }
            memset(dst + *dpos, src[i], bsize);
            *dpos += bsize;
            i += 1;
            if (st)
                st->blocks_rle++;
        } else if (type == 2) {
            if (i + bsize > n) {
                rc = -ZO_E_TRUNCATED;
                break;
            }
            if (bsize > (128u << 10)) {
                rc = -ZO_E_CORRUPT;
                break;
            }
            size_t before = *dpos;
            rc = decode_block(fc, src + i, bsize, dst, dpos, dcap, frame_start, window);
            if (rc < 0)
                break;
            if (*dpos - before > (128u << 10)) {
                rc = -ZO_E_CORRUPT;
                break;
            }
            i += bsize;
            if (st)
                st->blocks_compressed++;
        } else {
            rc = -ZO_E_CORRUPT;
            break;
        }
        if (last)
            break;
    }
    if (rc == 0 && checksum) {
        if (i + 4 > n)
            rc = -ZO_E_TRUNCATED;
        else {
            /* Content_Checksum: low 32 bits of XXH64 of the decoded frame; libzstd (which the reference's zstd
             * crate wraps) refuses a frame whose checksum differs */
            uint32_t want = (uint32_t)src[i] | ((uint32_t)src[i + 1] << 8) | ((uint32_t)src[i + 2] << 16) | ((uint32_t)src[i + 3] << 24);
            if ((uint32_t)zo_xxh64(dst + frame_start, *dpos - frame_start) != want)
                rc = -ZO_E_CORRUPT;
            i += 4;
        }
    }
    if (rc == 0 && fcs_bytes && (uint64_t)(*dpos - frame_start) != fcs)
        rc = -ZO_E_CORRUPT;
    free(fc->lit);
    free(fc);
    *consumed = i;
    return rc;
}

/*
 * Decode a NAF section payload: one or more magicless frames back to back
 * (the reference's streaming decoder continues into a
 * following frame).  Returns produced bytes or a negative error.
 */
long zo_decode_section(const uint8_t *src, size_t n, uint8_t *dst, size_t dcap, zo_stats *st)
{
    size_t i = 0, dpos = 0;
    if (st)
        memset(st, 0, sizeof(*st));
    if (n == 0)
        return -ZO_E_TRUNCATED;
    while (i < n) {
        size_t used = 0;
        int rc = decode_frame(src + i, n - i, &used, dst, &dpos, dcap, st);
        if (rc < 0)
            return rc;
        i += used;
    }
    return (long)dpos;
}
