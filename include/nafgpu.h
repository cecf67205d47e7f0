/*
 * nafgpu.h -- C-ABI of libnafgpu, the MI355X-native NAF decode path.
 *
 * This is the drop-in boundary for the hot path of althonos/nafcodec v0.3.1
 * (/root/reference): everything below the crate's public `Decoder` iterator --
 * container parse, per-section Zstandard decompression, 4-bit -> IUPAC unpack,
 * mask application, record slicing -- runs behind these entry points, on the GPU.
 * The reference has no FFI seam inside the path; the seam is its
 * public L3 API, so each entry point names the reference item it replaces.
 * A Rust / C++ / Python shim re-creates `Decoder`/`DecoderBuilder`/`Record` on
 * top (INTEGRATION.md shows the Rust one; include/nafcodec.hpp and
 * nafcodec_amd/ are the C++ and Python ones).
 *
 * Conventions: plain pointers and sizes only; no HIP or torch types.  Every
 * function returns NAFGPU_OK (0) or a negative nafgpu_status unless stated.
 * A decoder may move between threads but calls on one decoder must be
 * serialised by the caller (mirrors `Send + !Sync`, nafcodec/src/lib.rs:24-28).
 * Nothing here aborts or throws across the boundary.
 */
#ifndef NAFGPU_H
#define NAFGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NAFGPU_ABI_VERSION 2

/* ---- status / error kinds ------------------------------------------------
 * Mirrors nafcodec::Error (nafcodec/src/error.rs:4-11) plus the std::io kinds
 * the decode path raises (decoder/mod.rs:180-185, 430-435; reader.rs:108-109). */
typedef enum {
    NAFGPU_OK = 0,
    NAFGPU_END = 1,              /* Iterator::next -> None (decoder/mod.rs:447-449) */
    NAFGPU_E_IO = -1,            /* Error::Io; see nafgpu_error.io_kind */
    NAFGPU_E_NOM = -2,           /* Error::Nom; see nafgpu_error.nom_code */
    NAFGPU_E_UTF8 = -3,          /* Error::Utf8 */
    NAFGPU_E_PANIC = -4,         /* the reference panics / never returns on this input */
    NAFGPU_E_DEVICE = -5,        /* HIP runtime failure, no usable GPU, out of device memory */
    NAFGPU_E_INVALID_ARG = -6,
    NAFGPU_E_MISSING_FIELD = -7,     /* Error::MissingField (encoder/mod.rs:254,265,298,316) */
    NAFGPU_E_INVALID_LENGTH = -8,    /* Error::InvalidLength (encoder/mod.rs:273-276,303-306) */
    NAFGPU_E_INVALID_SEQUENCE = -9   /* Error::InvalidSequence (encoder/mod.rs:284-290, writer.rs:31-56) */
} nafgpu_status;

typedef enum {                   /* std::io::ErrorKind values used on the path */
    NAFGPU_IO_NONE = 0,
    NAFGPU_IO_UNEXPECTED_EOF = 1,
    NAFGPU_IO_INVALID_DATA = 2,  /* zstd stream corrupt, bad UTF-8 in text sequence/quality */
    NAFGPU_IO_NOT_FOUND = 3,
    NAFGPU_IO_IS_A_DIRECTORY = 4,
    NAFGPU_IO_PERMISSION_DENIED = 5,
    NAFGPU_IO_OTHER = 6
} nafgpu_io_kind;

typedef enum {                   /* nom::error::ErrorKind values parser.rs produces */
    NAFGPU_NOM_NONE = 0,
    NAFGPU_NOM_VERIFY = 1,       /* bad magic / separator (parser.rs:50-53, 87-91) */
    NAFGPU_NOM_MAPRES = 2,       /* bad version / sequence type (parser.rs:55-73) */
    NAFGPU_NOM_TOOLARGE = 3      /* varint overflow (parser.rs:38-45) */
} nafgpu_nom_code;

typedef struct {
    int32_t status;              /* nafgpu_status */
    int32_t io_kind;             /* nafgpu_io_kind */
    int32_t os_errno;            /* errno for NAFGPU_E_IO raised by the OS, else 0 */
    int32_t nom_code;            /* nafgpu_nom_code */
    char message[192];           /* NUL-terminated, human readable */
} nafgpu_error;

/* ---- DecoderBuilder (decoder/mod.rs:53-148) -------------------------------- */
typedef struct {
    uint8_t id;                  /* mod.rs:117  default 1 */
    uint8_t comment;             /* mod.rs:124  default 1 */
    uint8_t sequence;            /* mod.rs:131  default 1 */
    uint8_t quality;             /* mod.rs:138  default 1 */
    uint8_t mask;                /* mod.rs:145  default 1 */
    uint8_t spec_mask;           /* 0 = reference behaviour incl. the record-end mask quirk
                                    (mod.rs:410-415); 1 = lower-case every masked base */
    uint8_t shard_protocol;      /* 1: the decoder is driven through nafgpu_shard_* (below): sections WITH LZ sequences are
                                    sharded too, and the Quality section as well as the Sequence section */
    uint8_t reserved[1];
    uint64_t buffer_size;        /* mod.rs:110: accepted for API parity; sizes the host read-back window */
    int32_t device;              /* HIP device ordinal; -1 = current device */
    int32_t shard_rank;          /* multi-GPU: this process decodes shard `shard_rank` of `shard_count` */
    int32_t shard_count;         /* contiguous zstd-block ranges of the sequence section; 1 = everything */
    int32_t tile_mib;            /* decode the sequence / quality sections in tiles of about this many MiB of output, so that
                                    neither the compressed bytes nor the scratch memory -- nor, for nafgpu_next, the output --
                                    are resident whole (the reference streams any size through 4 KiB buffers, mod.rs:223);
                                    0 = when the archive would not fit in the device's free memory, and -- for nafgpu_next /
                                    nafgpu_next_batch only -- for a section of 4 GiB or more, which the iterator takes in 2 GiB
                                    tiles (the next tile's compressed bytes travel while this one is read back) */
} nafgpu_opts;

/* DecoderBuilder::new() (mod.rs:67-76) */
void nafgpu_opts_default(nafgpu_opts *opts);
/* DecoderBuilder::from_flags (mod.rs:93-101): quality/sequence/mask/comment follow `flags`
 * (a data.rs:80-97 flag byte); `id` stays on. */
void nafgpu_opts_from_flags(nafgpu_opts *opts, uint8_t flags);

/* ---- Header (data.rs:198-237) ------------------------------------------------ */
typedef struct {
    uint8_t format_version;      /* 1 | 2                          data.rs:46-50 */
    uint8_t sequence_type;       /* 0 dna, 1 rna, 2 protein, 3 text  data.rs:56-73 */
    uint8_t flags;               /* bit0 quality, 1 sequence, 2 mask, 3 length, 4 comment, 5 id,
                                    6 title, 7 extended             data.rs:80-97 */
    uint8_t name_separator;
    uint32_t reserved;
    uint64_t line_length;
    uint64_t number_of_sequences;
} nafgpu_header;

/* ---- Record (data.rs:29-40) --------------------------------------------------- */
typedef struct {
    const uint8_t *ptr;          /* host memory owned by the decoder; NOT NUL-terminated */
    uint64_t len;
    uint8_t present;             /* Option::is_some() */
    uint8_t reserved[7];
} nafgpu_field;

typedef struct {
    nafgpu_field id, comment, sequence, quality;
    uint64_t length;
    uint8_t has_length;
    uint8_t reserved[7];
} nafgpu_record;

typedef struct nafgpu_decoder nafgpu_decoder;

/* reader callbacks for nafgpu_open_io: R: Read + Seek (decoder/mod.rs:169-172, ioslice.rs) */
typedef int64_t (*nafgpu_read_fn)(void *ctx, uint8_t *buf, uint64_t cap); /* bytes read, 0 = EOF, <0 = -errno */
typedef int64_t (*nafgpu_seek_fn)(void *ctx, int64_t offset, int whence); /* new position or <0 = -errno */

/* DecoderBuilder::with_path (mod.rs:159-166) / Decoder::from_path (mod.rs:304-306) */
int nafgpu_open_path(const char *path, const nafgpu_opts *opts, nafgpu_decoder **out, nafgpu_error *err);
/* DecoderBuilder::with_bytes (mod.rs:151-156); `bytes` is borrowed until nafgpu_close */
int nafgpu_open_bytes(const uint8_t *bytes, size_t n, const nafgpu_opts *opts, nafgpu_decoder **out,
                      nafgpu_error *err);
/* DecoderBuilder::with_reader (mod.rs:169-256) */
int nafgpu_open_io(nafgpu_read_fn read, nafgpu_seek_fn seek, void *ctx, const nafgpu_opts *opts,
                   nafgpu_decoder **out, nafgpu_error *err);

/* Decoder::header (mod.rs:322-325) */
void nafgpu_get_header(const nafgpu_decoder *dec, nafgpu_header *out);
/* ExactSizeIterator::len (mod.rs:453-457): number_of_sequences - records yielded */
uint64_t nafgpu_remaining(const nafgpu_decoder *dec);
/* Iterator::next (mod.rs:444-451) -> NAFGPU_OK with *rec filled, NAFGPU_END, or an error.
 * Field pointers stay valid until the next call on this decoder; the shim copies them into
 * owned strings to honour Record<'static>.  An error does not fuse the iterator (mod.rs:391). */
int nafgpu_next(nafgpu_decoder *dec, nafgpu_record *rec);
/* Up to `cap` calls of Iterator::next (mod.rs:444-451) in one crossing of the boundary: a binding that reads millions of
 * 151-base records pays one call per few thousand of them.  Fills recs[0 .. *n) -- exactly the records, in order, that *n
 * calls of nafgpu_next would have handed out -- and returns NAFGPU_OK (with *n >= 1; *n < cap when the next record's bytes
 * are not on the host yet: call again), NAFGPU_END (*n == 0: nothing left), or the error the (*n + 1)-th call of nafgpu_next
 * would have returned: recs[0 .. *n) are valid then too, the iterator is not fused, and the next call goes on behind the
 * record that failed (mod.rs:391).  All views of one batch stay valid until the next call on this decoder. */
int nafgpu_next_batch(nafgpu_decoder *dec, nafgpu_record *recs, uint64_t cap, uint64_t *n);
/* Drop for Decoder / into_inner (mod.rs:343-350) */
void nafgpu_close(nafgpu_decoder *dec);
/* last error raised on this decoder (or by the failed open when dec == NULL is not possible:
 * open failures report through their `err` argument) */
void nafgpu_last_error(const nafgpu_decoder *dec, nafgpu_error *err);

/* ---- bulk device path (bench / GPU consumers) -------------------------------------
 * Decodes every selected section on the GPU and leaves the results in HBM.  Replaces a
 * full drain of the iterator (`for r in decoder`) without the per-record host copies. */
typedef struct {
    /* device pointers (HIP device memory owned by the decoder, valid until close / next decode_all) */
    const uint8_t *d_sequence;   /* masked ASCII bases (DNA/RNA) or raw text; length n_bases */
    const uint8_t *d_quality;    /* quality bytes; length n_quality */
    const uint64_t *d_record_end;/* inclusive prefix sum of record lengths; n_records entries */
    const uint8_t *d_ids;        /* NUL-terminated ids, concatenated */
    const uint8_t *d_comments;
    uint64_t n_bases, n_quality, n_records, n_ids_bytes, n_comments_bytes;
    uint64_t packed_bytes;       /* decoded 4-bit bytes of the sequence section (DNA/RNA), else 0 */
    uint64_t compressed_bytes;   /* compressed bytes read by the GPU over all decoded sections */
    uint64_t seq_compressed_bytes;
    uint64_t n_zstd_blocks;      /* sequence section */
    uint64_t n_huf_streams;      /* sequence section */
    uint64_t first_record;       /* shard: index of the first record that STARTS in this shard */
    uint8_t carry;               /* shard: 1 if the shard begins inside a record (its head belongs to record first_record-1) */
    uint8_t sharded;             /* 1 if shard_count > 1 was honoured (sequence section without LZ sequences) */
    uint8_t reserved[6];
    /* timing of the last decode_all, milliseconds (HIP events on the decoder's stream) */
    float ms_total;              /* first kernel launch -> last kernel done */
    float ms_huf;                /* sum over launches of the Huffman literal kernel */
    float ms_unpack;             /* 4-bit -> ASCII kernel */
    float ms_seq_lz;             /* FSE sequence decode + LZ execute kernels */
    float ms_other;
    float ms_host_plan;          /* host: frame walk + table build (outside ms_total) */
    float ms_h2d;                /* host->device upload of archive + task lists (outside ms_total) */
    uint32_t n_huf_launches;
    uint32_t reserved3;
    uint64_t lz_residue_matches; /* LZ matches the parallel passes left to the pointer-jumping stage (all sections) */
    uint64_t base_offset;        /* shard: global index of d_sequence[0] (0 unless sharded) */
    /* ids / comments split on NUL on the device (CStringReader, reader.rs:22-30): string k of d_ids is
     * d_ids[d_id_end[k-1] .. d_id_end[k] - 1) (d_id_end[k] = offset just past its NUL); at most
     * number_of_sequences strings are listed */
    const uint64_t *d_id_end;
    const uint64_t *d_comment_end;
    uint64_t n_ids, n_comments;
    /* bit s set: section s (0 ids, 1 comments, 4 sequence of a protein/text archive, 5 quality) is not valid
     * UTF-8 -- where the reference returns Error::Utf8 (reader.rs:108-109) or panics (mod.rs:362,368) */
    uint32_t utf8_invalid;
    uint32_t reserved4;
    uint64_t quality_offset;     /* shard protocol: index in the Quality section of d_quality[0] (0 otherwise) */
} nafgpu_device_result;

int nafgpu_decode_all_device(nafgpu_decoder *dec, nafgpu_device_result *out);
/* Host front end only: frame walk, table build, upload of the archive and task lists to HBM.
 * Idempotent; decode_all_device / next call it implicitly.  Lets a caller separate "compressed
 * bytes resident in HBM" from the decode itself. */
int nafgpu_upload(nafgpu_decoder *dec);

/* ---- shard protocol: ONE archive over several GPUs, sections with LZ sequences included --------
 * A section is one Zstandard frame: a block may copy from up to a window in front of it and inherits three repeat
 * offsets.  Every rank (opts.shard_rank of opts.shard_count, opts.shard_protocol = 1) walks the whole block directory
 * and decodes one contiguous block range; what a range needs from the ranges in front of it arrives in two steps:
 *
 *   nafgpu_shard_begin    entropy decode of the range (Huffman literals, FSE sequences), block sizes, and the range's
 *                         repeat-offset map -- nothing here needs the other ranks.  Fills `mine`.
 *   (all-gather of the 64-byte nafgpu_shard_summary: RCCL over xGMI; the only collective)
 *   nafgpu_shard_place    every rank now knows where its range begins in the decoded section, which repeat offsets it
 *                         inherits and how much of the window in front of it exists: literals to their places and all
 *                         matches that do not reach -- directly or through other matches -- into that window.
 *   nafgpu_shard_halo     how many bytes this rank receives from rank - 1 (recv_bytes) and sends to rank + 1
 *                         (send_bytes: the last window of its output), and whether what it sends is final already
 *                         (tail_ready; always 1 for the first rank and wherever no pending match touches the tail).
 *   nafgpu_shard_export_tail / nafgpu_shard_import_halo    the point-to-point step (ncclSend / ncclRecv): a rank whose
 *                         tail is ready sends first and receives afterwards, the others receive, finish, send.
 *                         `dst` / `src` are device (or host) buffers of send_bytes / recv_bytes; import also runs what
 *                         was waiting for the window.
 *   nafgpu_shard_finish   mask, record table placement; the result describes this rank's share of the Sequence and the
 *                         Quality sections (d_sequence / base_offset / n_bases, d_quality / quality_offset / n_quality).
 * Sections without LZ sequences take the same calls (recv_bytes = send_bytes = 0).  `section`: 0 Sequence, 1 Quality. */
typedef struct {
    uint64_t decoded[2];         /* elements (decoded zstd bytes) of this rank's block range; [0] Sequence, [1] Quality */
    uint64_t frame_tail[2];      /* of which: behind the start of the last frame that begins inside the range (all, if none does) */
    uint32_t rep_map[2][3];      /* the range's repeat-offset map: the three offsets it leaves behind, as values or as
                                    "inherited offset k minus d" tokens (plan.h: kRepToken) */
    uint8_t failed[2];           /* the rank could not decode its range (every rank then reports the error) */
    uint8_t reserved[6];
} nafgpu_shard_summary;          /* 64 bytes, no pointers: gathered as it is */
int nafgpu_shard_begin(nafgpu_decoder *dec, nafgpu_shard_summary *mine);
int nafgpu_shard_place(nafgpu_decoder *dec, const nafgpu_shard_summary *all, int n_ranks);
int nafgpu_shard_halo(nafgpu_decoder *dec, int section, uint64_t *recv_bytes, uint64_t *send_bytes, int *tail_ready);
int nafgpu_shard_export_tail(nafgpu_decoder *dec, int section, void *dst, uint64_t n);
int nafgpu_shard_import_halo(nafgpu_decoder *dec, int section, const void *src, uint64_t n);
int nafgpu_shard_finish(nafgpu_decoder *dec, nafgpu_device_result *out);

/* ---- text output on the device (what a consumer of the iterator does next, cf. unnaf) ----
 * FASTA, or FASTQ when the archive has a Quality section and opts.quality is set, of the records the
 * iterator would yield, built in HBM from the decoded buffers:
 *   FASTA  '>' id [name_separator comment] '\n', then the sequence in lines of header.line_length characters
 *   FASTQ  '@' id [name_separator comment] '\n' sequence '\n' '+' '\n' quality '\n'
 * (separator and comment only when the comment is not empty; line_length 0 = one line).  Runs
 * decode_all_device first if nothing is decoded yet.  d_text stays valid until close / the next call. */
typedef struct {
    const uint8_t *d_text;       /* device pointer */
    uint64_t n_text;             /* bytes */
    uint64_t n_records;
    float ms;                    /* sizes + scan + write kernels, HIP events */
    uint8_t fastq;
    uint8_t reserved[3];
} nafgpu_text_result;
int nafgpu_format_device(nafgpu_decoder *dec, nafgpu_text_result *out);
/* device -> host copy of `n` bytes of a buffer a result struct points to (tests, small consumers) */
int nafgpu_copy_to_host(nafgpu_decoder *dec, const void *d_ptr, uint64_t n, void *dst);
/* hipDeviceSynchronize on `device` (-1 = current) */
int nafgpu_device_synchronize(int device);
/* The library keeps the device memory of closed decoders for the next one (mapped ranges of 32 MiB and more, up to 16 GiB
 * of them idle per device; small buffers up to 64 MiB in all): this call gives all of it back to the driver -- before
 * another allocator or another process needs the device (no counterpart in the reference; -1 = current device). */
int nafgpu_trim_device_memory(int device);

/* ---- L0 replacement on its own: one NAF section payload ------------------------------
 * Replaces zstd::stream::read::Decoder + include_magicbytes(false) (decoder/mod.rs:221-223)
 * for a whole section: `src` is the magicless Zstandard frame(s), `dst` receives up to `cap`
 * decoded bytes.  Host buffers in, host buffer out; the decode runs on the GPU. */
int nafgpu_zstd_decompress(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *produced,
                           int device, nafgpu_error *err);

/* ---- synthetic archives (format per encoder/mod.rs:334-384) ------
 * Writes a NAF v1 DNA archive with Length+Sequence (+Mask) sections whose sequence section is
 * ONE magicless zstd frame of 128 KiB Huffman-literal blocks (what `ennaf` / zstd level 1
 * produce on DNA).  Deterministic in (seed, n_bases, flags).  Host-only, multi-threaded. */
typedef struct {
    uint64_t n_bases;            /* total nucleotides */
    uint64_t seed;
    uint8_t with_mask;           /* add a Mask section (runs stay inside records) */
    uint8_t iupac_permille;      /* 0..255: per-mille of non-ACGT codes (N, R, Y, ...) */
    uint8_t part_count;          /* > 1: write only part `part_rank` of the archive (several processes write one archive
                                    together, each its share of the sequence section's zstd blocks: nafgpu_synth_archive.bytes
                                    then holds that share alone, and nafgpu_synth_head writes what goes in front) */
    uint8_t part_rank;
    uint8_t reserved[4];
    uint32_t threads;            /* 0 = hardware concurrency */
    uint32_t reserved2;
} nafgpu_synth_spec;

typedef struct {
    uint8_t *bytes;              /* malloc'ed archive; free with nafgpu_synth_free */
    uint64_t n;
    uint64_t n_records, n_bases;
    uint64_t seq_hash;           /* nafgpu_hash64 of the expected ASCII bases (after masking) */
    uint64_t offsets_hash;       /* nafgpu_hash64 of the u64 record_end table */
} nafgpu_synth_archive;

int nafgpu_synth_write(const nafgpu_synth_spec *spec, nafgpu_synth_archive *out);
/* What precedes the parts of an archive written in parts: container header, Length (and Mask) sections, the
 * sequence section's two sizes and its frame header.  seq_part_bytes = sum of the parts' sizes.  out->bytes / n:
 * that head; the archive is head followed by part 0, part 1, ...; seq_hash is 0 (the parts' values add up). */
int nafgpu_synth_head(const nafgpu_synth_spec *spec, uint64_t seq_part_bytes, nafgpu_synth_archive *out);
void nafgpu_synth_free(nafgpu_synth_archive *a);

/* ---- Encoder (EncoderBuilder / Encoder, encoder/mod.rs:46-384, writer.rs) ----------------
 * Host code: the reference's encoder is CPU code too, and the decode path above is what runs on the GPU.  Same
 * surface and checks as the reference; every section is written as one magicless Zstandard frame of 128 KiB
 * blocks -- Huffman / RLE / raw literals, and at `compression_level` 0 (the default level) or >= 3 greedy hash
 * matches coded as sequences with the predefined FSE tables (levels 1-2: literals only) -- which the reference's
 * decoder, libzstd and this library all read.  Sections are kept in memory until the archive is written (the
 * reference's `Memory` storage, storage.rs).  Like the reference's encoder it never writes a Mask section and
 * accepts upper-case IUPAC letters only. */
typedef struct {
    uint8_t sequence_type;       /* 0 dna, 1 rna, 2 protein, 3 text (EncoderBuilder::new, mod.rs:81-90) */
    uint8_t id, comment, sequence, quality;   /* opt-in fields (mod.rs:112-145); all 0 by default */
    uint8_t reserved[3];
    int32_t compression_level;   /* mod.rs:147-157; 0 or >= 3: with LZ matches, 1-2: literals only (see above) */
    uint32_t threads;            /* blocks are encoded in parallel when the archive is written; 0 = hardware concurrency */
} nafgpu_encoder_opts;
typedef struct nafgpu_encoder nafgpu_encoder;

void nafgpu_encoder_opts_default(uint8_t sequence_type, nafgpu_encoder_opts *opts);
/* EncoderBuilder::from_flags (mod.rs:92-110): NAF flag bits 0x20 id, 0x10 comment, 0x02 sequence, 0x01 quality */
void nafgpu_encoder_opts_from_flags(uint8_t sequence_type, uint8_t flags, nafgpu_encoder_opts *opts);
/* EncoderBuilder::with_memory (mod.rs:161-163) */
int nafgpu_encoder_new(const nafgpu_encoder_opts *opts, nafgpu_encoder **out, nafgpu_error *err);
/* Encoder::push (mod.rs:236-323).  Only the enabled fields are read.  A record that is refused (missing field,
 * inconsistent length, invalid letter) leaves the encoder as it was -- the reference has by then written the
 * fields in front of the offending one. */
int nafgpu_encoder_push(nafgpu_encoder *enc, const nafgpu_record *rec, nafgpu_error *err);
/* Encoder::write (mod.rs:325-384) into memory: *bytes stays valid until nafgpu_encoder_free; no push afterwards */
int nafgpu_encoder_finish(nafgpu_encoder *enc, const uint8_t **bytes, uint64_t *n, nafgpu_error *err);
void nafgpu_encoder_free(nafgpu_encoder *enc);

/* order-sensitive 64-bit checksum used for full-size parity checks: sum over the 8-byte words w_j of
 * mix64(w_j ^ (j + 1) * K) -- every word is mixed non-linearly with its position before it is added, so
 * byte errors cannot cancel -- see hash64.h */
uint64_t nafgpu_hash64_host(const uint8_t *p, uint64_t n);
int nafgpu_hash64_device(const nafgpu_decoder *dec, const void *d_ptr, uint64_t n, uint64_t *out);
/* same, for a buffer that starts at 4 KiB chunk `first_chunk` of a larger object: the values of
 * consecutive shards add up (mod 2^64) to the checksum of the whole object */
int nafgpu_hash64_device_at(const nafgpu_decoder *dec, const void *d_ptr, uint64_t n, uint64_t first_chunk, uint64_t *out);
uint64_t nafgpu_hash64_host_at(const uint8_t *p, uint64_t n, uint64_t first_chunk);

/* library / device identification, for logs */
int nafgpu_abi_version(void);
/* tests and measurements only: after nafgpu_test_hooks(1) the library reads its NAFGPU_* experiment variables
 *  from the environment; by default it reads none of them. */
void nafgpu_test_hooks(int enable);
int nafgpu_device_info(int device, char *name, size_t cap, uint64_t *hbm_bytes, int *compute_units);

#ifdef __cplusplus
}
#endif
#endif /* NAFGPU_H */
