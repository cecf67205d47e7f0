/*
 * oracle/naf_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's NAF decode path, function by function:
 *
 *   no_variable_u64      parser::variable_u64          decoder/parser.rs:27-48
 *   no_parse_header      parser::header + helpers      decoder/parser.rs:50-123
 *   no_open              DecoderBuilder::with_reader   decoder/mod.rs:169-256
 *   cstring_next         CStringReader::next           decoder/reader.rs:22-30
 *   length_next          LengthReader::next            decoder/reader.rs:48-67
 *   sequence_next        SequenceReader::next          decoder/reader.rs:88-111
 *   read_nucleotide      SequenceReader::read_nucleotide  reader.rs:121-149
 *   nuc_decode           SequenceReader::decode        reader.rs:152-172
 *   mask_next            MaskReader::next              decoder/reader.rs:198-231
 *   mask_sequence        Decoder::mask_sequence        decoder/mod.rs:402-441
 *   no_next              Decoder::next / next_record   decoder/mod.rs:356-399,444-451
 *
 * The reference streams each section through a 4 KiB BufReader; its results do
 * not depend on buffer boundaries, so this restatement decodes each section
 * fully (zstd_oracle.c) and then runs the same per-record logic over the
 * decoded bytes.  Places where the reference panics or never returns on
 * malformed input yield NO_E_PANIC / an Io error instead.
 */
#include <stdlib.h>
#include <string.h>

#include "naf_oracle.h"

/* ---------------------------------------------------------------- parser */

int no_variable_u64(const uint8_t *p, size_t n, uint64_t *out, size_t *used, int *nom_code)
{
    /* take_while(byte & 0x80) then one byte, both streaming -> Incomplete on short input */
    size_t k = 0;
    while (k < n && (p[k] & 0x80))
        k++;
    if (k >= n)
        return NO_E_IO_EOF;
    uint64_t num = 0, basis = 1;
    num += (uint64_t)(p[k] & 0x7F) * basis;
    basis *= 128;
    for (size_t j = k; j-- > 0;) {
        uint64_t x = (uint64_t)(p[j] & 0x7F) * basis; /* wraps like release Rust (App. D-6) */
        if (num + x < num) {
            if (nom_code)
                *nom_code = NO_NOM_TOOLARGE;
            return NO_E_NOM;
        }
        num += x;
        basis *= 128;
    }
    *out = num;
    *used = k + 1;
    return NO_OK;
}

int no_parse_header(const uint8_t *p, size_t n, no_header *h, size_t *used, int *nom_code)
{
    size_t i = 0;
    int code = 0;
    if (n < 3)
        return NO_E_IO_EOF;
    if (!(p[0] == 0x01 && p[1] == 0xF9 && p[2] == 0xEC)) { /* parser.rs:50-53 */
        code = NO_NOM_VERIFY;
        goto nom;
    }
    i = 3;
    if (i >= n)
        return NO_E_IO_EOF;
    if (p[i] != 1 && p[i] != 2) { /* parser.rs:55-62 */
        code = NO_NOM_MAPRES;
        goto nom;
    }
    h->format_version = p[i++];
    if (h->format_version == 1) {
        h->sequence_type = 0; /* parser.rs:104-107: v1 => DNA */
    } else {
        if (i >= n)
            return NO_E_IO_EOF;
        if (p[i] > 3) { /* parser.rs:64-73 */
            code = NO_NOM_MAPRES;
            goto nom;
        }
        h->sequence_type = p[i++];
    }
    if (i >= n)
        return NO_E_IO_EOF;
    h->flags = p[i++]; /* parser.rs:75-85 */
    if (i >= n)
        return NO_E_IO_EOF;
    if (p[i] < 0x20 || p[i] > 0x7E) { /* parser.rs:10-12,87-91 */
        code = NO_NOM_VERIFY;
        goto nom;
    }
    h->name_separator = p[i++];
    size_t u;
    int rc = no_variable_u64(p + i, n - i, &h->line_length, &u, nom_code);
    if (rc != NO_OK)
        return rc;
    i += u;
    rc = no_variable_u64(p + i, n - i, &h->number_of_sequences, &u, nom_code);
    if (rc != NO_OK)
        return rc;
    i += u;
    *used = i;
    return NO_OK;
nom:
    if (nom_code)
        *nom_code = code;
    return NO_E_NOM;
}

/* --------------------------------------------------------------- decoder */

typedef struct {
    int flagged, used;
    uint64_t original_size, compressed_size, file_offset;
    uint8_t *data; /* decoded bytes */
    uint64_t n;    /* decoded length */
    int status;    /* NO_OK or error raised when the section is first touched */
    uint64_t pos;  /* read cursor */
} section;

enum { S_IDS, S_COM, S_LEN, S_MASK, S_SEQ, S_QUAL, S_COUNT };

struct no_decoder {
    no_header h;
    no_opts o;
    section s[S_COUNT];
    uint64_t n;
    /* SequenceReader.cache (reader.rs:76) */
    int cache_valid;
    uint8_t cache;
    /* MaskReader state (reader.rs:178-183) */
    uint64_t mask_total, mask_current;
    int mask_flag;
    /* Decoder.unit (mod.rs:295) */
    uint64_t unit_n;
    int unit_masked;
    /* per-record output buffers */
    uint8_t *seqbuf;
    uint64_t seqcap;
};

static const uint8_t SECTION_FLAG[S_COUNT] = {0x20, 0x10, 0x08, 0x04, 0x02, 0x01};

static uint64_t section_capacity(const no_decoder *d, int which, uint64_t original_size)
{
    /* Sequence original_size counts nucleotides for DNA/RNA (mod.rs:241,250) */
    if (which == S_SEQ && d->h.sequence_type <= 1)
        return (original_size + 1) / 2;
    return original_size;
}

static int utf8_valid(const uint8_t *p, uint64_t n);

int no_open(const uint8_t *bytes, size_t n, const no_opts *opts, no_decoder **out, int *nom_code)
{
    no_header h;
    size_t i = 0, u;
    memset(&h, 0, sizeof h);
    int rc = no_parse_header(bytes, n, &h, &i, nom_code); /* mod.rs:173-189 */
    if (rc != NO_OK)
        return rc;
    if (h.flags & 0x40) { /* Title: mod.rs:191-196, parser.rs:125-139 */
        uint64_t tsize;
        rc = no_variable_u64(bytes + i, n - i, &tsize, &u, nom_code);
        if (rc == NO_E_IO_EOF)
            return NO_E_PANIC; /* Incomplete -> todo!() error.rs:50 */
        if (rc != NO_OK)
            return rc;
        i += u;
        if (tsize > n - i)
            return NO_E_PANIC;
        if (!utf8_valid(bytes + i, tsize)) { /* map_res(take(size), from_utf8), parser.rs:133-137 */
            if (nom_code)
                *nom_code = NO_NOM_MAPRES;
            return NO_E_NOM;
        }
        i += (size_t)tsize;
    }
    no_decoder *d = (no_decoder *)calloc(1, sizeof *d);
    if (!d)
        return NO_E_IO_OTHER;
    d->h = h;
    d->o = *opts;
    const int want[S_COUNT] = {opts->id, opts->comment, 1, opts->mask, opts->sequence, opts->quality};
    for (int k = 0; k < S_COUNT; k++) { /* setup_block! x6, mod.rs:199-242 */
        section *s = &d->s[k];
        if (!(h.flags & SECTION_FLAG[k]))
            continue;
        s->flagged = 1;
        rc = no_variable_u64(bytes + (i < n ? i : n), i < n ? n - i : 0, &s->original_size, &u, nom_code);
        if (rc == NO_OK) {
            i += u;
            rc = no_variable_u64(bytes + (i < n ? i : n), i < n ? n - i : 0, &s->compressed_size, &u,
                                 nom_code);
        }
        if (rc != NO_OK) {
            no_close(d);
            return rc == NO_E_IO_EOF ? NO_E_PANIC : rc;
        }
        i += u;
        s->file_offset = i;
        s->used = want[k];
        if (s->used) {
            uint64_t cap = section_capacity(d, k, s->original_size);
            uint64_t avail = i <= n ? n - i : 0;
            uint64_t take = s->compressed_size < avail ? s->compressed_size : avail;
            s->data = (uint8_t *)malloc((size_t)cap + 16);
            if (!s->data) {
                no_close(d);
                return NO_E_IO_OTHER;
            }
            long got = zo_decode_section(bytes + i, (size_t)take, s->data, (size_t)cap, NULL);
            if (got < 0) {
                s->status = take < s->compressed_size ? NO_E_IO_EOF : NO_E_IO_INVALID;
                s->n = 0;
            } else {
                s->n = (uint64_t)got;
            }
        }
        i += (size_t)s->compressed_size; /* seek(Current(+csz)) mod.rs:228; may pass EOF */
    }
    d->mask_total = d->s[S_SEQ].flagged ? d->s[S_SEQ].original_size : 0; /* mod.rs:236,241,250 */
    *out = d;
    return NO_OK;
}

void no_get_header(const no_decoder *d, no_header *h) { *h = d->h; }

uint64_t no_remaining(const no_decoder *d)
{
    return d->h.number_of_sequences - d->n; /* mod.rs:453-456 */
}

void no_close(no_decoder *d)
{
    if (!d)
        return;
    for (int k = 0; k < S_COUNT; k++)
        free(d->s[k].data);
    free(d->seqbuf);
    free(d);
}

int no_section(const no_decoder *d, int which, const uint8_t **p, uint64_t *n, uint64_t *original_size,
               uint64_t *compressed_size, uint64_t *file_offset)
{
    const section *s = &d->s[which];
    if (!s->flagged)
        return 0;
    if (p)
        *p = s->data;
    if (n)
        *n = s->n;
    if (original_size)
        *original_size = s->original_size;
    if (compressed_size)
        *compressed_size = s->compressed_size;
    if (file_offset)
        *file_offset = s->file_offset;
    return s->status == NO_OK ? 1 : s->status;
}

/* CStringReader::next, reader.rs:22-30: 1 = Some(Ok), 0 = None, <0 = error */
static int cstring_next(section *s, no_field *f)
{
    if (s->status != NO_OK)
        return s->status;
    if (s->pos >= s->n)
        return 0; /* read_until -> Ok(0) -> None */
    const uint8_t *start = s->data + s->pos;
    const uint8_t *nul = (const uint8_t *)memchr(start, 0, (size_t)(s->n - s->pos));
    if (!nul)
        return NO_E_PANIC; /* from_vec_with_nul(...).expect(...) */
    f->ptr = start;
    f->len = (uint64_t)(nul - start);
    f->present = 1;
    s->pos += f->len + 1;
    return 1;
}

static int utf8_valid(const uint8_t *p, uint64_t n)
{
    uint64_t i = 0;
    while (i < n) {
        uint8_t c = p[i];
        if (c < 0x80) {
            i++;
            continue;
        }
        int extra;
        uint32_t cp, min;
        if ((c & 0xE0) == 0xC0) {
            extra = 1; cp = c & 0x1F; min = 0x80;
        } else if ((c & 0xF0) == 0xE0) {
            extra = 2; cp = c & 0x0F; min = 0x800;
        } else if ((c & 0xF8) == 0xF0) {
            extra = 3; cp = c & 0x07; min = 0x10000;
        } else {
            return 0;
        }
        if (i + (uint64_t)extra >= n)
            return 0;
        for (int k = 1; k <= extra; k++) {
            if ((p[i + (uint64_t)k] & 0xC0) != 0x80)
                return 0;
            cp = (cp << 6) | (p[i + (uint64_t)k] & 0x3F);
        }
        if (cp < min || cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF))
            return 0;
        i += (uint64_t)extra + 1;
    }
    return 1;
}

/* LengthReader::next, reader.rs:48-67 */
static int length_next(section *s, uint64_t *out)
{
    if (s->status != NO_OK)
        return s->status;
    uint64_t n = 0;
    uint32_t x = 0xFFFFFFFFu;
    while (x == 0xFFFFFFFFu) {
        if (s->n - s->pos < 4) {
            s->pos = s->n; /* read_exact consumed what was there */
            return 0;      /* UnexpectedEof -> None */
        }
        const uint8_t *b = s->data + s->pos;
        x = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
        s->pos += 4;
        n += x;
    }
    *out = n;
    return 1;
}

/* SequenceReader::decode, reader.rs:152-172 */
static uint8_t nuc_decode(uint8_t c, uint8_t t)
{
    static const char LUT[17] = "-TGKCYSBAWRDMHVN";
    return c == 1 ? t : (uint8_t)LUT[c];
}

static int ensure_seqbuf(no_decoder *d, uint64_t l)
{
    if (l + 1 <= d->seqcap)
        return 0;
    uint64_t cap = l + 1;
    uint8_t *nb = (uint8_t *)realloc(d->seqbuf, (size_t)cap);
    if (!nb)
        return -1;
    d->seqbuf = nb;
    d->seqcap = cap;
    return 0;
}

/* SequenceReader::next for nucleotide types, reader.rs:88-103 + 121-149 */
static int sequence_next_nucleotide(no_decoder *d, uint64_t l, no_field *f)
{
    section *s = &d->s[S_SEQ];
    if (s->status != NO_OK)
        return s->status;
    uint8_t t = d->h.sequence_type == 1 ? 'U' : 'T';
    if (ensure_seqbuf(d, l) < 0)
        return NO_E_IO_OTHER;
    uint8_t *seq = d->seqbuf;
    uint64_t len = 0;
    if (d->cache_valid && l > 0) { /* reader.rs:92-94 */
        seq[len++] = d->cache;
        d->cache_valid = 0;
    }
    while (len < l) {
        /* read_nucleotide, with `buffer` = everything still undecoded */
        uint64_t buflen = s->n - s->pos;
        if (buflen == 0)
            return NO_E_IO_EOF; /* reference spins forever here (App. D-4) */
        const uint8_t *buffer = s->data + s->pos;
        uint64_t rem = l - len;
        uint64_t n = buflen < rem / 2 ? buflen : rem / 2;
        for (uint64_t k = 0; k < n; k++) {
            uint8_t x = buffer[k];
            seq[len++] = nuc_decode(x & 0x0F, t);
            seq[len++] = nuc_decode(x >> 4, t);
        }
        if (n < buflen && len == l - 1) {
            seq[len++] = nuc_decode(buffer[n] & 0x0F, t);
            d->cache = nuc_decode(buffer[n] >> 4, t);
            d->cache_valid = 1;
            s->pos += n + 1;
        } else {
            s->pos += n;
        }
    }
    f->ptr = seq;
    f->len = l;
    f->present = 1;
    return 1;
}

/* SequenceReader::next for text types, reader.rs:104-119 */
static int sequence_next_text(section *s, uint64_t l, no_field *f)
{
    if (s->status != NO_OK)
        return s->status;
    if (s->n - s->pos < l)
        return NO_E_IO_EOF; /* reference spins forever (read_text on an empty buffer) */
    f->ptr = s->data + s->pos;
    f->len = l;
    f->present = 1;
    s->pos += l;
    if (!utf8_valid(f->ptr, l))
        return NO_E_IO_INVALID; /* reader.rs:108-109 */
    return 1;
}

/* MaskReader::next, reader.rs:198-231: 1 = Some, 0 = None */
static int mask_next(no_decoder *d, uint64_t *n_out, int *masked_out)
{
    section *s = &d->s[S_MASK];
    if (d->mask_current >= d->mask_total)
        return 0;
    if (s->status != NO_OK)
        return s->status;
    uint64_t n = 0;
    int terminated = 0;
    while (s->pos < s->n) {
        uint8_t b = s->data[s->pos++];
        n += b;
        if (b != 0xFF) {
            terminated = 1;
            break;
        }
    }
    if (!terminated && n == 0)
        return NO_E_IO_EOF; /* reference yields zero-length units forever (App. D-4) */
    d->mask_current += n;
    *n_out = n;
    *masked_out = d->mask_flag;
    d->mask_flag = !d->mask_flag;
    return 1;
}

static void lowercase(uint8_t *p, uint64_t n)
{
    for (uint64_t i = 0; i < n; i++)
        if (p[i] >= 'A' && p[i] <= 'Z')
            p[i] |= 0x20;
}

/* Decoder::mask_sequence, mod.rs:402-441 (incl. the record-end quirk) */
static int mask_sequence(no_decoder *d, uint8_t *seq, uint64_t len)
{
    uint64_t mn = d->unit_n;
    int mmasked = d->unit_masked;
    if (!(d->s[S_MASK].flagged && d->s[S_MASK].used))
        return NO_OK;
    for (;;) {
        if (mmasked) {
            if (mn < len) {
                lowercase(seq, mn);
                seq += mn;
                len -= mn;
            } else {
                if (d->o.spec_mask)
                    lowercase(seq, len); /* NOT what the reference does */
                d->unit_masked = 1;
                d->unit_n = mn - len;
                break;
            }
        } else {
            if (mn < len) {
                seq += mn;
                len -= mn;
            } else {
                d->unit_masked = 0;
                d->unit_n = mn - len;
                break;
            }
        }
        int rc = mask_next(d, &mn, &mmasked);
        if (rc < 0)
            return rc;
        if (rc == 0)
            return NO_E_IO_EOF; /* "failed to get mask unit" mod.rs:430-435 */
    }
    return NO_OK;
}

int no_next(no_decoder *d, no_record *rec)
{
    if (d->n >= d->h.number_of_sequences)
        return NO_END; /* mod.rs:447-449 */
    memset(rec, 0, sizeof *rec);
    int rc;
    /* NOTE: like the reference, state consumed before an error is not rolled back (mod.rs:391) */
    if (d->s[S_IDS].flagged && d->s[S_IDS].used) {
        rc = cstring_next(&d->s[S_IDS], &rec->id);
        if (rc < 0)
            return rc;
        if (rc == 1 && !utf8_valid(rec->id.ptr, rec->id.len))
            return NO_E_PANIC; /* mod.rs:362 expect("TODO") */
    }
    if (d->s[S_COM].flagged && d->s[S_COM].used) {
        rc = cstring_next(&d->s[S_COM], &rec->comment);
        if (rc < 0)
            return rc;
        if (rc == 1 && !utf8_valid(rec->comment.ptr, rec->comment.len))
            return NO_E_PANIC;
    }
    if (d->s[S_LEN].flagged) {
        uint64_t l = 0;
        rc = length_next(&d->s[S_LEN], &l);
        if (rc < 0)
            return rc;
        if (rc == 1) {
            rec->has_length = 1;
            rec->length = l;
        }
    }
    if (rec->has_length) { /* mod.rs:373 */
        uint64_t l = rec->length;
        if (d->s[S_SEQ].flagged && d->s[S_SEQ].used) {
            if (d->h.sequence_type <= 1) {
                rc = sequence_next_nucleotide(d, l, &rec->sequence);
            } else {
                /* text sequences are copied so masking can edit them in place */
                no_field tmp = {0, 0, 0};
                rc = sequence_next_text(&d->s[S_SEQ], l, &tmp);
                if (rc == 1) {
                    if (ensure_seqbuf(d, l) < 0)
                        return NO_E_IO_OTHER;
                    memcpy(d->seqbuf, tmp.ptr, (size_t)l);
                    rec->sequence.ptr = d->seqbuf;
                    rec->sequence.len = l;
                    rec->sequence.present = 1;
                }
            }
            if (rc < 0)
                return rc;
        }
        if (d->s[S_QUAL].flagged && d->s[S_QUAL].used) {
            rc = sequence_next_text(&d->s[S_QUAL], l, &rec->quality);
            if (rc < 0)
                return rc;
        }
        if (rec->sequence.present) { /* mod.rs:386-388 */
            rc = mask_sequence(d, d->seqbuf, rec->sequence.len);
            if (rc < 0)
                return rc;
        }
    }
    d->n += 1;
    return NO_OK;
}

size_t no_mask_units(const no_decoder *d, uint64_t *len, uint8_t *masked, size_t cap)
{
    const section *s = &d->s[S_MASK];
    size_t k = 0;
    uint64_t pos = 0, cur = 0;
    int flag = 0;
    if (!s->flagged || !s->used || s->status != NO_OK)
        return 0;
    while (k < cap && cur < d->mask_total && pos < s->n) {
        uint64_t n = 0;
        while (pos < s->n) {
            uint8_t b = s->data[pos++];
            n += b;
            if (b != 0xFF)
                break;
        }
        len[k] = n;
        masked[k] = (uint8_t)flag;
        flag = !flag;
        cur += n;
        k++;
    }
    return k;
}

/* ---- whole-archive drain with checksums (checker for full-size GPU parity tests) --------
 * Drains Decoder::next and accumulates, per field, the position-keyed checksum the product's
 * hash64.h defines (restated here: the checker computes its own value):
 *   H = sum over 8-byte little-endian words w_j of mix64(w_j ^ (j + 1) * K), last word zero-padded. */
typedef struct {
    uint64_t h, n;      /* running sum, bytes fed */
    uint64_t word;      /* partial word */
} hash_acc;

static uint64_t drain_mix64(uint64_t x)
{
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}

static void hash_feed(hash_acc *a, const uint8_t *p, uint64_t n)
{
    uint64_t i = 0;
    while (i < n && (a->n & 7)) { /* complete the partial word */
        a->word |= (uint64_t)p[i++] << (8 * (a->n & 7));
        a->n++;
        if (!(a->n & 7)) {
            a->h += drain_mix64(a->word ^ ((a->n / 8) * 0x9E3779B97F4A7C15ull));
            a->word = 0;
        }
    }
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        memcpy(&w, p + i, 8);
        a->n += 8;
        a->h += drain_mix64(w ^ ((a->n / 8) * 0x9E3779B97F4A7C15ull));
    }
    for (; i < n; i++) {
        a->word |= (uint64_t)p[i] << (8 * (a->n & 7));
        a->n++;
    }
}

static uint64_t hash_finish(const hash_acc *a)
{
    if (a->n & 7)
        return a->h + drain_mix64(a->word ^ ((a->n / 8 + 1) * 0x9E3779B97F4A7C15ull));
    return a->h;
}

int no_drain(no_decoder *d, int want_hash, no_drain_result *out)
{
    hash_acc seq = {0, 0, 0}, qual = {0, 0, 0}, ends = {0, 0, 0}, ids = {0, 0, 0}, com = {0, 0, 0};
    uint64_t end = 0;
    memset(out, 0, sizeof *out);
    for (;;) {
        no_record r;
        int rc = no_next(d, &r);
        if (rc == NO_END)
            break;
        if (rc != NO_OK)
            return rc;
        out->n_records++;
        if (r.has_length) {
            end += r.length;
            if (want_hash)
                hash_feed(&ends, (const uint8_t *)&end, 8);
        }
        if (r.sequence.present) {
            out->n_bases += r.sequence.len;
            if (want_hash)
                hash_feed(&seq, r.sequence.ptr, r.sequence.len);
        }
        if (r.quality.present) {
            out->n_quality += r.quality.len;
            if (want_hash)
                hash_feed(&qual, r.quality.ptr, r.quality.len);
        }
        if (want_hash && r.id.present) {
            static const uint8_t nul = 0;
            hash_feed(&ids, r.id.ptr, r.id.len);
            hash_feed(&ids, &nul, 1);
        }
        if (want_hash && r.comment.present) {
            static const uint8_t nul = 0;
            hash_feed(&com, r.comment.ptr, r.comment.len);
            hash_feed(&com, &nul, 1);
        }
    }
    out->seq_hash = hash_finish(&seq);
    out->qual_hash = hash_finish(&qual);
    out->ends_hash = hash_finish(&ends);
    out->ids_hash = hash_finish(&ids);
    out->com_hash = hash_finish(&com);
    return NO_OK;
}
