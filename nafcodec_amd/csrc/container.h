// container.h -- NAF container parse on the host (bytes, not gigabytes).
// Replaces nafcodec/src/decoder/parser.rs and the section walk of
// DecoderBuilder::with_reader (nafcodec/src/decoder/mod.rs:169-256).
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <string>

#include "../../include/nafgpu.h"

namespace nafgpu {

enum Section { kIds = 0, kComments, kLengths, kMask, kSequence, kQuality, kNumSections };
extern const uint8_t kSectionFlag[kNumSections];

struct SectionInfo {
    bool present = false;
    uint64_t original_size = 0;    // Sequence: nucleotides for DNA/RNA (mod.rs:241,250)
    uint64_t compressed_size = 0;
    uint64_t offset = 0;           // payload offset in the archive
};

struct Failure {                   // maps 1:1 onto nafgpu_error
    int status = NAFGPU_OK;
    int io_kind = 0, os_errno = 0, nom_code = 0;
    std::string message;
    bool ok() const { return status == NAFGPU_OK; }
    static Failure io(int kind, const std::string &m, int err = 0) {
        Failure f; f.status = NAFGPU_E_IO; f.io_kind = kind; f.os_errno = err; f.message = m; return f;
    }
    static Failure nom(int code, const std::string &m) {
        Failure f; f.status = NAFGPU_E_NOM; f.nom_code = code; f.message = m; return f;
    }
    static Failure make(int status, const std::string &m) {
        Failure f; f.status = status; f.message = m; return f;
    }
    void to_c(nafgpu_error *e) const;
};

// what std::str::from_utf8 accepts
bool utf8_valid(const uint8_t *p, uint64_t n);
// parser::variable_u64 (parser.rs:27-48).  incomplete=true <=> nom::Err::Incomplete.
Failure parse_varint(const uint8_t *p, size_t n, uint64_t *value, size_t *used, bool *incomplete);
// parser::header (parser.rs:101-123)
Failure parse_header(const uint8_t *p, size_t n, nafgpu_header *h, size_t *used, bool *incomplete);
// header + optional title + the six (original_size, compressed_size) pairs (mod.rs:173-242).
// `need(offset, length)` is called before bytes are parsed: a source that loads lazily (a reader with
// seek, nafgpu_open_io) fetches just those, as the reference reads the header and seeks over payloads.
using NeedFn = std::function<void(size_t, size_t)>;
Failure parse_archive(const uint8_t *p, size_t n, nafgpu_header *h, SectionInfo sec[kNumSections], const NeedFn *need = nullptr);

}  // namespace nafgpu
