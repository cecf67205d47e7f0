/*
 * oracle/naf_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * C interface of the CPU oracle: a restatement of the reference's NAF decode
 * path (althonos/nafcodec v0.3.1: nafcodec/src/decoder/{parser,reader,mod}.rs)
 * plus a scalar Zstandard decoder (zstd_oracle.c).  Parity is PINNED: see
 * tests/test_oracle_fixtures.py (reference fixtures + every known answer the
 * reference's own tests assert) and tests/test_oracle_zstd.py (libzstd
 * cross-check).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * this.  nafcodec_amd/ never does.
 */
#ifndef NAF_ORACLE_H
#define NAF_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- zstd stage ------------------------------------------------------- */
typedef struct {
    uint64_t frames, blocks, blocks_raw, blocks_rle, blocks_compressed;
    uint64_t lit_raw, lit_rle, lit_huf, lit_treeless;
    uint64_t lit_bytes, lit_bytes_entropy, match_bytes, sequences;
    uint64_t window;
    uint64_t seq_mode_count[12]; /* [0..3] LL modes, [4..7] OF, [8..11] ML */
} zo_stats;

/* decode a NAF section payload (magicless zstd frame(s)); returns produced
 * bytes or a negative error code */
long zo_decode_section(const uint8_t *src, size_t n, uint8_t *dst, size_t dcap, zo_stats *st);

/* ---- container (parser.rs) -------------------------------------------- */
typedef struct {
    uint8_t format_version; /* 1 | 2          parser.rs:55-62  */
    uint8_t sequence_type;  /* 0 dna 1 rna 2 protein 3 text  parser.rs:64-73 */
    uint8_t flags;          /* data.rs:80-97 */
    uint8_t name_separator; /* parser.rs:87-91 */
    uint64_t line_length;
    uint64_t number_of_sequences;
} no_header;

/* error kinds, mirroring error.rs:4-11 and the std::io kinds the path raises */
enum {
    NO_OK = 0,
    NO_END = 1,            /* iterator exhausted */
    NO_E_IO_EOF = -1,      /* Error::Io(UnexpectedEof) */
    NO_E_IO_INVALID = -2,  /* Error::Io(InvalidData): zstd failure / bad UTF-8 in text */
    NO_E_NOM = -3,         /* Error::Nom{code}: see nom_code */
    NO_E_PANIC = -4,       /* the reference panics / never returns here */
    NO_E_IO_OTHER = -5,
};
/* nom::error::ErrorKind values used by parser.rs */
enum { NO_NOM_VERIFY = 1, NO_NOM_MAPRES = 2, NO_NOM_TOOLARGE = 3 };

/* parser::variable_u64 (parser.rs:27-48).  rc: NO_OK, NO_E_IO_EOF (Incomplete) or NO_E_NOM */
int no_variable_u64(const uint8_t *p, size_t n, uint64_t *out, size_t *used, int *nom_code);
/* parser::header (parser.rs:101-123) */
int no_parse_header(const uint8_t *p, size_t n, no_header *h, size_t *used, int *nom_code);

/* ---- decoder (mod.rs + reader.rs) -------------------------------------- */
typedef struct {
    uint8_t id, comment, sequence, quality, mask; /* DecoderBuilder, mod.rs:53-76 */
    uint8_t spec_mask;                            /* 1 = fix App. D-1 (not reference behaviour) */
} no_opts;

typedef struct {
    const uint8_t *ptr;
    uint64_t len;
    uint8_t present;
} no_field;

typedef struct {
    no_field id, comment, sequence, quality; /* data.rs:29-40 */
    uint64_t length;
    uint8_t has_length;
} no_record;

typedef struct no_decoder no_decoder;

int no_open(const uint8_t *bytes, size_t n, const no_opts *opts, no_decoder **out, int *nom_code);
void no_get_header(const no_decoder *d, no_header *h);
uint64_t no_remaining(const no_decoder *d);
/* Decoder::next (mod.rs:444-451).  Field pointers stay valid until the next call. */
int no_next(no_decoder *d, no_record *rec);
void no_close(no_decoder *d);

/* raw decoded section access for tests: which = 0 ids,1 comments,2 lengths,3 mask,4 sequence,5 quality */
int no_section(const no_decoder *d, int which, const uint8_t **p, uint64_t *n, uint64_t *original_size,
               uint64_t *compressed_size, uint64_t *file_offset);
/* the first `cap` mask units (reader.rs:198-231); returns count, sets masked[i] */
size_t no_mask_units(const no_decoder *d, uint64_t *len, uint8_t *masked, size_t cap);

/* Drains the iterator.  want_hash: also accumulate, per field, the position-keyed 64-bit checksum of
 * the concatenated bytes (sequence after masking; quality; ids / comments each followed by NUL; the u64
 * table of inclusive record ends) -- the checker's value for full-size GPU parity tests. */
typedef struct {
    uint64_t n_records, n_bases, n_quality;
    uint64_t seq_hash, qual_hash, ends_hash, ids_hash, com_hash;
} no_drain_result;
int no_drain(no_decoder *d, int want_hash, no_drain_result *out);

#ifdef __cplusplus
}
#endif
#endif
