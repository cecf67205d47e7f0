// nafcodec.hpp -- header-only C++17 mirror of the reference's public decode API on top of the
// C-ABI (include/nafgpu.h).  Same names, defaults and error behaviour as
// nafcodec/src/decoder/mod.rs (DecoderBuilder :53-257, Decoder :285-461), data.rs (Record :29-40,
// Header :198-237, Flag(s) :80-189, SequenceType :56-73, FormatVersion :46-50) and error.rs.
#pragma once
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "nafgpu.h"

namespace nafcodec {

enum class FormatVersion : uint8_t { V1 = 1, V2 = 2 };
enum class SequenceType : uint8_t { Dna = 0, Rna = 1, Protein = 2, Text = 3 };
inline bool is_nucleotide(SequenceType t) { return t == SequenceType::Dna || t == SequenceType::Rna; }

enum class Flag : uint8_t { Quality = 0x1, Sequence = 0x2, Mask = 0x4, Length = 0x8, Comment = 0x10, Id = 0x20, Title = 0x40, Extended = 0x80 };
struct Flags {
    uint8_t bits = 0;
    bool test(Flag f) const { return (bits & static_cast<uint8_t>(f)) != 0; }
    void set(Flag f) { bits |= static_cast<uint8_t>(f); }
    void unset(Flag f) { bits &= static_cast<uint8_t>(~static_cast<uint8_t>(f)); }
    uint8_t as_byte() const { return bits; }
};
inline Flags operator|(Flag a, Flag b) { return Flags{static_cast<uint8_t>(static_cast<uint8_t>(a) | static_cast<uint8_t>(b))}; }
inline Flags operator|(Flags a, Flag b) { return Flags{static_cast<uint8_t>(a.bits | static_cast<uint8_t>(b))}; }

// error.rs:4-11 -- Io / Nom / Utf8 (+ the C-ABI's Panic and Device kinds)
struct Error : std::runtime_error {
    nafgpu_error raw;
    explicit Error(const nafgpu_error &e) : std::runtime_error(e.message), raw(e) {}
    bool is_io() const { return raw.status == NAFGPU_E_IO; }
    bool is_unexpected_eof() const { return is_io() && raw.io_kind == NAFGPU_IO_UNEXPECTED_EOF; }
    bool is_nom() const { return raw.status == NAFGPU_E_NOM; }
};

struct Record {   // data.rs:29-40: five public Option fields, owned strings (Record<'static>)
    std::optional<std::string> id, comment, sequence, quality;
    std::optional<uint64_t> length;
};

class Header {    // data.rs:198-237
public:
    explicit Header(const nafgpu_header &h) : h_(h) {}
    Flags flags() const { return Flags{h_.flags}; }
    uint64_t line_length() const { return h_.line_length; }
    char name_separator() const { return static_cast<char>(h_.name_separator); }
    uint64_t number_of_sequences() const { return h_.number_of_sequences; }
    SequenceType sequence_type() const { return static_cast<SequenceType>(h_.sequence_type); }
    FormatVersion format_version() const { return static_cast<FormatVersion>(h_.format_version); }

private:
    nafgpu_header h_;
};

class Decoder {   // mod.rs:285-461
public:
    Decoder(Decoder &&o) noexcept
        : d_(std::exchange(o.d_, nullptr)), batch_(std::move(o.batch_)), at_(std::exchange(o.at_, 0)), n_(std::exchange(o.n_, 0)),
          pending_(std::exchange(o.pending_, NAFGPU_OK)), pending_err_(o.pending_err_) {}
    Decoder &operator=(Decoder &&o) noexcept {
        if (this != &o) {
            nafgpu_close(d_);
            d_ = std::exchange(o.d_, nullptr);
            batch_ = std::move(o.batch_);
            at_ = std::exchange(o.at_, 0);
            n_ = std::exchange(o.n_, 0);
            pending_ = std::exchange(o.pending_, NAFGPU_OK);
            pending_err_ = o.pending_err_;
        }
        return *this;
    }
    Decoder(const Decoder &) = delete;
    ~Decoder() { nafgpu_close(d_); }

    static Decoder from_path(const std::string &path);   // mod.rs:304-306
    Header header() const {
        nafgpu_header h;
        nafgpu_get_header(d_, &h);
        return Header(h);
    }
    SequenceType sequence_type() const { return header().sequence_type(); }
    size_t len() const { return static_cast<size_t>(nafgpu_remaining(d_)) + (n_ - at_); }   // ExactSizeIterator (records fetched ahead count)

    // Iterator::next: nullopt at the end; throws Error (the iterator stays usable, mod.rs:391).
    // Records cross the C boundary a batch at a time (nafgpu_next_batch); what the caller sees is what one nafgpu_next
    // per record gives: an error met by record k of a batch is thrown when record k is asked for, not before.
    std::optional<Record> next() {
        if (at_ == n_) {
            if (pending_ != NAFGPU_OK) {
                const int rc = std::exchange(pending_, NAFGPU_OK);
                if (rc == NAFGPU_END) return std::nullopt;
                throw Error(pending_err_);
            }
            if (batch_.empty()) batch_.resize(kBatch);
            uint64_t got = 0;
            const int rc = nafgpu_next_batch(d_, batch_.data(), kBatch, &got);
            at_ = 0;
            n_ = static_cast<size_t>(got);
            if (rc != NAFGPU_OK) {
                if (rc != NAFGPU_END) nafgpu_last_error(d_, &pending_err_);
                if (n_ == 0) {
                    if (rc == NAFGPU_END) return std::nullopt;
                    throw Error(pending_err_);
                }
                pending_ = rc;
            }
        }
        const nafgpu_record &r = batch_[at_++];
        auto own = [](const nafgpu_field &f) -> std::optional<std::string> {
            if (!f.present) return std::nullopt;
            return std::string(reinterpret_cast<const char *>(f.ptr), static_cast<size_t>(f.len));
        };
        Record out;
        out.id = own(r.id);
        out.comment = own(r.comment);
        out.sequence = own(r.sequence);
        out.quality = own(r.quality);
        if (r.has_length) out.length = r.length;
        return out;
    }
    // The whole archive as FASTA (FASTQ when it has qualities and `quality` is selected), formatted on
    // the GPU from the decoded buffers (nafgpu_format_device) and copied to the host: what `unnaf` prints.
    std::string to_text() {
        nafgpu_text_result t;
        if (nafgpu_format_device(d_, &t) != NAFGPU_OK) {
            nafgpu_error e;
            nafgpu_last_error(d_, &e);
            throw Error(e);
        }
        std::string out(static_cast<size_t>(t.n_text), '\0');
        if (t.n_text && nafgpu_copy_to_host(d_, t.d_text, t.n_text, out.data()) != NAFGPU_OK) {
            nafgpu_error e;
            nafgpu_last_error(d_, &e);
            throw Error(e);
        }
        return out;
    }
    nafgpu_decoder *raw() const { return d_; }

private:
    friend class DecoderBuilder;
    explicit Decoder(nafgpu_decoder *d) : d_(d) {}
    nafgpu_decoder *d_ = nullptr;
    static constexpr size_t kBatch = 1024;
    std::vector<nafgpu_record> batch_;
    size_t at_ = 0, n_ = 0;
    int pending_ = NAFGPU_OK;          // what the batch call returned behind its records (an error, or the end)
    nafgpu_error pending_err_{};
};

class DecoderBuilder {   // mod.rs:53-257
public:
    DecoderBuilder() { nafgpu_opts_default(&o_); }                                  // mod.rs:67-76
    static DecoderBuilder from_flags(Flags f) {                                     // mod.rs:93-101
        DecoderBuilder b;
        nafgpu_opts_from_flags(&b.o_, f.as_byte());
        return b;
    }
    DecoderBuilder &buffer_size(size_t n) { o_.buffer_size = n; return *this; }      // mod.rs:110
    DecoderBuilder &id(bool v) { o_.id = v; return *this; }
    DecoderBuilder &comment(bool v) { o_.comment = v; return *this; }
    DecoderBuilder &sequence(bool v) { o_.sequence = v; return *this; }
    DecoderBuilder &quality(bool v) { o_.quality = v; return *this; }
    DecoderBuilder &mask(bool v) { o_.mask = v; return *this; }
    DecoderBuilder &device(int ordinal) { o_.device = ordinal; return *this; }       // MI355X-specific knob

    Decoder with_bytes(const uint8_t *p, size_t n) const {                           // mod.rs:151-156
        nafgpu_decoder *d = nullptr;
        nafgpu_error e;
        if (nafgpu_open_bytes(p, n, &o_, &d, &e) != NAFGPU_OK) throw Error(e);
        return Decoder(d);
    }
    Decoder with_path(const std::string &path) const {                               // mod.rs:159-166
        nafgpu_decoder *d = nullptr;
        nafgpu_error e;
        if (nafgpu_open_path(path.c_str(), &o_, &d, &e) != NAFGPU_OK) throw Error(e);
        return Decoder(d);
    }
    Decoder with_reader(nafgpu_read_fn read, nafgpu_seek_fn seek, void *ctx) const {  // mod.rs:169-256
        nafgpu_decoder *d = nullptr;
        nafgpu_error e;
        if (nafgpu_open_io(read, seek, ctx, &o_, &d, &e) != NAFGPU_OK) throw Error(e);
        return Decoder(d);
    }

private:
    nafgpu_opts o_;
};

inline Decoder Decoder::from_path(const std::string &path) { return DecoderBuilder().with_path(path); }

// ---- Encoder (encoder/mod.rs:46-384).  Host code, as in the reference; see include/nafgpu.h for what is written.
class Encoder {      // mod.rs:215-384 (Memory storage)
public:
    Encoder(Encoder &&o) noexcept : e_(o.e_) { o.e_ = nullptr; }
    Encoder(const Encoder &) = delete;
    ~Encoder() { if (e_) nafgpu_encoder_free(e_); }
    void push(const Record &r) {                                                     // mod.rs:236-323
        nafgpu_record c{};
        auto field = [](const std::optional<std::string> &s, nafgpu_field *f) {
            if (!s) return;
            f->ptr = reinterpret_cast<const uint8_t *>(s->data());
            f->len = s->size();
            f->present = 1;
        };
        field(r.id, &c.id);
        field(r.comment, &c.comment);
        field(r.sequence, &c.sequence);
        field(r.quality, &c.quality);
        if (r.length) {
            c.length = *r.length;
            c.has_length = 1;
        }
        nafgpu_error err{};
        if (nafgpu_encoder_push(e_, &c, &err) != NAFGPU_OK) throw Error(err);
    }
    std::string write() {                                                            // mod.rs:325-384 into a string
        const uint8_t *p = nullptr;
        uint64_t n = 0;
        nafgpu_error err{};
        if (nafgpu_encoder_finish(e_, &p, &n, &err) != NAFGPU_OK) throw Error(err);
        return std::string(reinterpret_cast<const char *>(p), n);
    }

private:
    friend class EncoderBuilder;
    explicit Encoder(nafgpu_encoder *e) : e_(e) {}
    nafgpu_encoder *e_ = nullptr;
};

class EncoderBuilder {   // mod.rs:46-213
public:
    explicit EncoderBuilder(SequenceType t) { nafgpu_encoder_opts_default(static_cast<uint8_t>(t), &o_); }   // mod.rs:81-90
    static EncoderBuilder from_flags(SequenceType t, Flags f) {                      // mod.rs:92-110
        EncoderBuilder b(t);
        nafgpu_encoder_opts_from_flags(static_cast<uint8_t>(t), f.as_byte(), &b.o_);
        return b;
    }
    EncoderBuilder &id(bool v) { o_.id = v; return *this; }
    EncoderBuilder &comment(bool v) { o_.comment = v; return *this; }
    EncoderBuilder &sequence(bool v) { o_.sequence = v; return *this; }
    EncoderBuilder &quality(bool v) { o_.quality = v; return *this; }
    EncoderBuilder &compression_level(int v) { o_.compression_level = v; return *this; }
    Encoder with_memory() const {                                                    // mod.rs:161-163
        nafgpu_encoder *e = nullptr;
        nafgpu_error err{};
        if (nafgpu_encoder_new(&o_, &e, &err) != NAFGPU_OK) throw Error(err);
        return Encoder(e);
    }

private:
    nafgpu_encoder_opts o_{};
};

// (no counterpart in the reference: the library keeps device memory of closed decoders for the next one -- nafgpu.h)
inline void trim_device_memory(int device = -1) { (void)nafgpu_trim_device_memory(device); }

}  // namespace nafcodec
