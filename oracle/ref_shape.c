/*
 * oracle/ref_shape.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The reference decode pipeline IN ITS OWN SHAPE, for bench.py's cpu_baseline leg: what
 * althonos/nafcodec v0.3.1 does per record on one thread, stage for stage --
 *   - one streaming Zstandard decoder per section, the system libzstd through dlopen (the reference links the
 *     same C library through the zstd crate: decoder/mod.rs:221-222, magicless frames), drained through a
 *     buffered reader of `buffer_size` = 4096 bytes (mod.rs:223: BufReader::with_capacity(self.buffer_size, ..));
 *   - CStringReader::next  (reader.rs:22-30): read_until(0) into a fresh heap string per record;
 *   - LengthReader::next   (reader.rs:48-67): read_exact(4) while the word is 0xFFFFFFFF;
 *   - SequenceReader::next (reader.rs:88-149): a fresh String::with_capacity(l), then per input byte two
 *     decode() + push(char) steps -- each push with its capacity check -- and the odd-nibble cache;
 *   - read_text            (reader.rs:113-119) + from_utf8 for quality / text sequences;
 *   - mask_sequence        (mod.rs:402-441) with MaskReader::next (reader.rs:198-231).
 * It does not replace oracle/naf_oracle.c as the CHECKER (that one decodes sections whole): it is the thing
 * that is TIMED, and tests/test_oracle_fixtures.py pins it on the same fixtures and against the checker.
 * When libzstd.so.1 cannot be loaded, rs_available() returns 0 and bench.py times the oracle instead.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "naf_oracle.h"

/* ---- libzstd, loaded at run time ------------------------------------------------------------------ */
typedef struct { const void *src; size_t size, pos; } zin;
typedef struct { void *dst; size_t size, pos; } zout;
static void *(*p_createDStream)(void);
static size_t (*p_freeDStream)(void *);
static size_t (*p_initDStream)(void *);
static size_t (*p_decompressStream)(void *, zout *, zin *);
static unsigned (*p_isError)(size_t);
static int g_loaded = -1;

int rs_available(void)
{
    if (g_loaded >= 0)
        return g_loaded;
    void *h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h)
        h = dlopen("libzstd.so", RTLD_NOW | RTLD_GLOBAL);
    if (h) {
        p_createDStream = (void *(*)(void))dlsym(h, "ZSTD_createDStream");
        p_freeDStream = (size_t(*)(void *))dlsym(h, "ZSTD_freeDStream");
        p_initDStream = (size_t(*)(void *))dlsym(h, "ZSTD_initDStream");
        p_decompressStream = (size_t(*)(void *, zout *, zin *))dlsym(h, "ZSTD_decompressStream");
        p_isError = (unsigned (*)(size_t))dlsym(h, "ZSTD_isError");
    }
    g_loaded = h && p_createDStream && p_freeDStream && p_initDStream && p_decompressStream && p_isError;
    return g_loaded;
}

/* ---- BufReader<zstd::Decoder<IoSlice>> (mod.rs:219-223) -------------------------------------------- */
#define RS_BUF 4096 /* DecoderBuilder::buffer_size default, mod.rs:69 */
static const uint8_t MAGIC[4] = {0x28, 0xB5, 0x2F, 0xFD};

typedef struct {
    void *ds;
    const uint8_t *src; /* section payload (magicless) */
    size_t n, fed;      /* fed: payload bytes handed to libzstd */
    int magic_fed;      /* the 4 magic bytes include_magicbytes(false) stands for have been fed */
    uint8_t buf[RS_BUF];
    size_t lo, hi;      /* buf[lo..hi) is unread */
    int eof, err;
    int present;
} bufreader;

static void br_open(bufreader *b, const uint8_t *src, size_t n)
{
    b->ds = p_createDStream();
    p_initDStream(b->ds);
    b->src = src;
    b->n = n;
    b->fed = 0;
    b->magic_fed = 0;
    b->lo = b->hi = 0;
    b->eof = b->err = 0;
    b->present = 1;
}

static void br_close(bufreader *b)
{
    if (b->present && b->ds)
        p_freeDStream(b->ds);
    b->present = 0;
}

/* BufRead::fill_buf: at most one buffer of decoded bytes per call */
static int br_fill(bufreader *b)
{
    if (b->lo < b->hi)
        return 1;
    if (b->eof || b->err)
        return 0;
    zout out = {b->buf, RS_BUF, 0};
    while (out.pos == 0) {
        size_t r;
        if (!b->magic_fed) {
            zin in = {MAGIC, 4, 0};
            r = p_decompressStream(b->ds, &out, &in);
            b->magic_fed = 1;
        } else {
            zin in = {b->src, b->n, b->fed};
            r = p_decompressStream(b->ds, &out, &in);
            b->fed = in.pos;
            if (!p_isError(r) && out.pos == 0 && in.pos == b->n) { /* input exhausted, nothing produced */
                b->eof = 1;
                if (r != 0)
                    b->err = 1; /* frame cut short */
                break;
            }
        }
        if (p_isError(r)) {
            b->err = 1;
            break;
        }
    }
    b->lo = 0;
    b->hi = out.pos;
    return b->hi > 0;
}

/* ---- growable string: String::with_capacity + push (reader.rs:91,131-136) --------------------------- */
typedef struct {
    uint8_t *p;
    size_t len, cap;
} rstring;

static inline void rs_push(rstring *s, uint8_t c)
{
    if (s->len == s->cap) { /* String::push's capacity check (never taken after with_capacity(l), always evaluated) */
        s->cap = s->cap ? 2 * s->cap : 8;
        s->p = (uint8_t *)realloc(s->p, s->cap);
    }
    s->p[s->len++] = c;
}

static const char NUC_T[17] = "-TGKCYSBAWRDMHVN";

static inline uint8_t decode_nuc(uint8_t c, uint8_t t) /* SequenceReader::decode, reader.rs:152-172 */
{
    switch (c) {
    case 0x00: return '-';
    case 0x01: return t;
    case 0x02: return 'G';
    case 0x03: return 'K';
    case 0x04: return 'C';
    case 0x05: return 'Y';
    case 0x06: return 'S';
    case 0x07: return 'B';
    case 0x08: return 'A';
    case 0x09: return 'W';
    case 0x0A: return 'R';
    case 0x0B: return 'D';
    case 0x0C: return 'M';
    case 0x0D: return 'H';
    case 0x0E: return 'V';
    default: return (uint8_t)NUC_T[15];
    }
}

/* ---- the decoder ------------------------------------------------------------------------------------ */
typedef struct {
    no_header h;
    bufreader s[6]; /* ids, comments, lengths, mask, sequence, quality */
    uint64_t n;
    int cache_valid;
    uint8_t cache;
    uint64_t mask_total, mask_current, unit_n;
    int mask_flag, unit_masked;
} rs_decoder;

static uint64_t mix64(uint64_t x)
{
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}

typedef struct { uint64_t h, n, word; } hacc;
static void hfeed(hacc *a, const uint8_t *p, uint64_t n)
{
    for (uint64_t i = 0; i < n; i++) {
        a->word |= (uint64_t)p[i] << (8 * (a->n & 7));
        a->n++;
        if (!(a->n & 7)) {
            a->h += mix64(a->word ^ ((a->n / 8) * 0x9E3779B97F4A7C15ull));
            a->word = 0;
        }
    }
}
static uint64_t hfinish(const hacc *a)
{
    return (a->n & 7) ? a->h + mix64(a->word ^ ((a->n / 8 + 1) * 0x9E3779B97F4A7C15ull)) : a->h;
}

/* CStringReader::next: read_until(0) into a fresh Vec (reader.rs:22-30) */
static int cstring_next(bufreader *b, rstring *out)
{
    out->p = NULL;
    out->len = out->cap = 0;
    int any = 0;
    for (;;) {
        if (!br_fill(b))
            break;
        any = 1;
        const uint8_t *z = (const uint8_t *)memchr(b->buf + b->lo, 0, b->hi - b->lo);
        size_t take = z ? (size_t)(z - (b->buf + b->lo)) + 1 : b->hi - b->lo;
        if (out->len + take > out->cap) {
            out->cap = (out->len + take) * 2;
            out->p = (uint8_t *)realloc(out->p, out->cap);
        }
        memcpy(out->p + out->len, b->buf + b->lo, take);
        out->len += take;
        b->lo += take;
        if (z) {
            out->len--; /* CString drops the NUL */
            return 1;
        }
    }
    return any ? -1 : 0;
}

static int read_exact(bufreader *b, uint8_t *dst, size_t n)
{
    while (n) {
        if (!br_fill(b))
            return 0;
        size_t take = b->hi - b->lo < n ? b->hi - b->lo : n;
        memcpy(dst, b->buf + b->lo, take);
        b->lo += take;
        dst += take;
        n -= take;
    }
    return 1;
}

static int mask_next(rs_decoder *d, uint64_t *n_out, int *masked_out) /* reader.rs:198-231 */
{
    bufreader *b = &d->s[3];
    if (d->mask_current >= d->mask_total)
        return 0;
    uint64_t n = 0;
    int terminated = 0, any = 0;
    for (;;) {
        uint8_t x;
        if (!read_exact(b, &x, 1))
            break;
        any = 1;
        n += x;
        if (x != 0xFF) {
            terminated = 1;
            break;
        }
    }
    if (!terminated && !any)
        return -1;
    d->mask_current += n;
    *n_out = n;
    *masked_out = d->mask_flag;
    d->mask_flag = !d->mask_flag;
    return 1;
}

static int mask_sequence(rs_decoder *d, uint8_t *seq, uint64_t len) /* mod.rs:402-441 */
{
    uint64_t mn = d->unit_n;
    int mm = d->unit_masked;
    for (;;) {
        if (mm) {
            if (mn < len) {
                for (uint64_t i = 0; i < mn; i++)
                    if (seq[i] >= 'A' && seq[i] <= 'Z')
                        seq[i] |= 0x20;
                seq += mn;
                len -= mn;
            } else {
                d->unit_masked = 1;
                d->unit_n = mn - len;
                return 1;
            }
        } else {
            if (mn < len) {
                seq += mn;
                len -= mn;
            } else {
                d->unit_masked = 0;
                d->unit_n = mn - len;
                return 1;
            }
        }
        int rc = mask_next(d, &mn, &mm);
        if (rc <= 0)
            return 0;
    }
}

/* Drains the whole archive as `for record in Decoder::new(..)` does; returns 0, or a negative code on any error. */
int rs_drain_limit(const uint8_t *bytes, size_t n, int want_hash, uint64_t limit, no_drain_result *out);
int rs_drain(const uint8_t *bytes, size_t n, int want_hash, no_drain_result *out)
{
    return rs_drain_limit(bytes, n, want_hash, UINT64_MAX, out);
}

/* ... at most `limit` records of it: what the reference has handed out by then (it streams: a corrupt block in the
 * middle of a section is met after the records in front of it, decoder/mod.rs:356-399 over :221-223).  *out counts and
 * hashes the records yielded before the error or the limit, whichever comes first. */
int rs_drain_limit(const uint8_t *bytes, size_t n, int want_hash, uint64_t limit, no_drain_result *out)
{
    static const uint8_t FLAG[6] = {0x20, 0x10, 0x08, 0x04, 0x02, 0x01};
    memset(out, 0, sizeof *out);
    if (!rs_available())
        return -100;
    rs_decoder *d = (rs_decoder *)calloc(1, sizeof *d);
    size_t i = 0, u;
    int nom = 0;
    if (no_parse_header(bytes, n, &d->h, &i, &nom) != NO_OK) {
        free(d);
        return -1;
    }
    if (d->h.flags & 0x40) {
        uint64_t ts;
        if (no_variable_u64(bytes + i, n - i, &ts, &u, &nom) != NO_OK || ts > n - i - u) {
            free(d);
            return -1;
        }
        i += u + (size_t)ts;
    }
    uint64_t seq_original = 0;
    for (int k = 0; k < 6; k++) { /* setup_block! x6, mod.rs:199-242 */
        if (!(d->h.flags & FLAG[k]))
            continue;
        uint64_t osz, csz;
        if (no_variable_u64(bytes + i, n - i, &osz, &u, &nom) != NO_OK) {
            free(d);
            return -1;
        }
        i += u;
        if (no_variable_u64(bytes + i, n - i, &csz, &u, &nom) != NO_OK || csz > n - i - u) {
            free(d);
            return -1;
        }
        i += u;
        if (k == 4)
            seq_original = osz;
        br_open(&d->s[k], bytes + i, (size_t)csz);
        i += (size_t)csz;
    }
    d->mask_total = seq_original;
    const uint8_t t = d->h.sequence_type == 1 ? 'U' : 'T';
    hacc hs = {0, 0, 0}, hq = {0, 0, 0}, he = {0, 0, 0}, hi = {0, 0, 0}, hc = {0, 0, 0};
    uint64_t end = 0;
    int rc = 0;
    for (d->n = 0; d->n < d->h.number_of_sequences && d->n < limit; d->n++) { /* Iterator::next, mod.rs:444-451 */
        rstring id = {0, 0, 0}, com = {0, 0, 0}, seq = {0, 0, 0}, qual = {0, 0, 0};
        int have_len = 0;
        uint64_t l = 0;
        if (d->s[0].present && cstring_next(&d->s[0], &id) < 0) rc = -2;
        if (d->s[1].present && cstring_next(&d->s[1], &com) < 0) rc = -2;
        if (d->s[2].present) { /* LengthReader::next */
            for (;;) {
                uint8_t w[4];
                if (!read_exact(&d->s[2], w, 4))
                    break;
                uint32_t x = (uint32_t)w[0] | ((uint32_t)w[1] << 8) | ((uint32_t)w[2] << 16) | ((uint32_t)w[3] << 24);
                l += x;
                have_len = 1;
                if (x != 0xFFFFFFFFu)
                    break;
            }
        }
        if (have_len && d->s[4].present) {
            if (d->h.sequence_type <= 1) { /* read_nucleotide, reader.rs:121-149 */
                seq.cap = (size_t)l ? (size_t)l : 1;
                seq.p = (uint8_t *)malloc(seq.cap);
                if (d->cache_valid && l > 0) {
                    rs_push(&seq, d->cache);
                    d->cache_valid = 0;
                }
                bufreader *b = &d->s[4];
                while (seq.len < l) {
                    if (!br_fill(b)) {
                        rc = -3;
                        break;
                    }
                    size_t rem = (size_t)(l - seq.len), avail = b->hi - b->lo;
                    size_t nb = rem / 2 < avail ? rem / 2 : avail;
                    const uint8_t *x = b->buf + b->lo;
                    for (size_t k = 0; k < nb; k++) { /* two checked pushes per byte */
                        rs_push(&seq, decode_nuc(x[k] & 0x0F, t));
                        rs_push(&seq, decode_nuc(x[k] >> 4, t));
                    }
                    if (nb < avail && seq.len == l - 1) {
                        rs_push(&seq, decode_nuc(x[nb] & 0x0F, t));
                        d->cache = decode_nuc(x[nb] >> 4, t);
                        d->cache_valid = 1;
                        nb++;
                    }
                    b->lo += nb;
                }
            } else { /* read_text */
                seq.cap = (size_t)l ? (size_t)l : 1;
                seq.p = (uint8_t *)malloc(seq.cap);
                if (!read_exact(&d->s[4], seq.p, (size_t)l)) rc = -3;
                seq.len = (size_t)l;
            }
            if (rc == 0 && d->s[3].present && !mask_sequence(d, seq.p, seq.len)) rc = -4;
        }
        if (have_len && d->s[5].present) {
            qual.cap = (size_t)l ? (size_t)l : 1;
            qual.p = (uint8_t *)malloc(qual.cap);
            if (!read_exact(&d->s[5], qual.p, (size_t)l)) rc = -3;
            qual.len = (size_t)l;
        }
        if (rc == 0) {
            out->n_records++;
            if (have_len) {
                end += l;
                if (want_hash) hfeed(&he, (const uint8_t *)&end, 8);
            }
            out->n_bases += seq.len;
            out->n_quality += qual.len;
            if (want_hash) {
                static const uint8_t nul = 0;
                hfeed(&hs, seq.p, seq.len);
                hfeed(&hq, qual.p, qual.len);
                if (d->s[0].present) { hfeed(&hi, id.p, id.len); hfeed(&hi, &nul, 1); }
                if (d->s[1].present) { hfeed(&hc, com.p, com.len); hfeed(&hc, &nul, 1); }
            }
        }
        free(id.p); /* the Record (and its four Strings) is dropped by the consumer */
        free(com.p);
        free(seq.p);
        free(qual.p);
        if (rc != 0)
            break;
    }
    out->seq_hash = hfinish(&hs);
    out->qual_hash = hfinish(&hq);
    out->ends_hash = hfinish(&he);
    out->ids_hash = hfinish(&hi);
    out->com_hash = hfinish(&hc);
    for (int k = 0; k < 6; k++)
        br_close(&d->s[k]);
    free(d);
    return rc;
}

/* ---- "all cores" upper bound: T threads, each draining its own copy of the archive ------------------ */
typedef struct {
    const uint8_t *bytes;
    size_t n;
    uint64_t bases;
    int rc;
} rs_job;

static void *rs_worker(void *p)
{
    rs_job *j = (rs_job *)p;
    no_drain_result r;
    j->rc = rs_drain(j->bytes, j->n, 0, &r);
    j->bases = r.n_bases;
    return NULL;
}

/* returns the bases decoded by all threads together (0 on error) */
uint64_t rs_drain_parallel(const uint8_t *bytes, size_t n, int threads)
{
    if (threads < 1 || !rs_available())
        return 0;
    pthread_t *th = (pthread_t *)calloc((size_t)threads, sizeof *th);
    rs_job *jobs = (rs_job *)calloc((size_t)threads, sizeof *jobs);
    uint64_t total = 0;
    for (int k = 0; k < threads; k++) {
        jobs[k].bytes = bytes;
        jobs[k].n = n;
        pthread_create(&th[k], NULL, rs_worker, &jobs[k]);
    }
    for (int k = 0; k < threads; k++) {
        pthread_join(th[k], NULL);
        if (jobs[k].rc == 0)
            total += jobs[k].bases;
    }
    free(th);
    free(jobs);
    return total;
}
