#include "container.h"

#include <cstring>

namespace nafgpu {

const uint8_t kSectionFlag[kNumSections] = {0x20, 0x10, 0x08, 0x04, 0x02, 0x01};

void Failure::to_c(nafgpu_error *e) const {
    if (!e) return;
    e->status = status;
    e->io_kind = io_kind;
    e->os_errno = os_errno;
    e->nom_code = nom_code;
    std::strncpy(e->message, message.c_str(), sizeof(e->message) - 1);
    e->message[sizeof(e->message) - 1] = 0;
}

// std::str::from_utf8 (reader.rs:108-109, parser.rs:133-137, mod.rs:362,368)
bool utf8_valid(const uint8_t *p, uint64_t n) {
    uint64_t i = 0;
    while (i < n) {
        // (ASCII, the only thing names, sequences and qualities hold in practice: eight bytes a step)
        while (i + 8 <= n) {
            uint64_t w;
            std::memcpy(&w, p + i, 8);
            if (w & 0x8080808080808080ull) break;
            i += 8;
        }
        if (i >= n) break;
        const uint8_t c = p[i];
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
            return false;
        }
        if (i + uint64_t(extra) >= n) return false;
        for (int k = 1; k <= extra; k++) {
            if ((p[i + uint64_t(k)] & 0xC0) != 0x80) return false;
            cp = (cp << 6) | (p[i + uint64_t(k)] & 0x3F);
        }
        if (cp < min || cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF)) return false;
        i += uint64_t(extra) + 1;
    }
    return true;
}

Failure parse_varint(const uint8_t *p, size_t n, uint64_t *value, size_t *used, bool *incomplete) {
    *incomplete = false;
    size_t k = 0;
    while (k < n && (p[k] & 0x80)) k++;      // limbs carry the continuation bit
    if (k >= n) {                            // streaming take_while / take(1) -> Incomplete
        *incomplete = true;
        return Failure::io(NAFGPU_IO_UNEXPECTED_EOF, "incomplete variable-length integer");
    }
    uint64_t num = p[k] & 0x7F, basis = 128;
    for (size_t j = k; j-- > 0;) {
        // parser.rs:38 guards only the addition; the multiply wraps in a release build (App. D-6)
        uint64_t term = static_cast<uint64_t>(p[j] & 0x7F) * basis;
        if (num + term < num) return Failure::nom(NAFGPU_NOM_TOOLARGE, "variable-length integer too large");
        num += term;
        basis *= 128;
    }
    *value = num;
    *used = k + 1;
    return Failure();
}

Failure parse_header(const uint8_t *p, size_t n, nafgpu_header *h, size_t *used, bool *incomplete) {
    *incomplete = false;
    auto eof = [&]() {
        *incomplete = true;
        return Failure::io(NAFGPU_IO_UNEXPECTED_EOF, "failed to read header");   // mod.rs:180-185
    };
    std::memset(h, 0, sizeof *h);
    if (n < 3) return eof();
    if (p[0] != 0x01 || p[1] != 0xF9 || p[2] != 0xEC)
        return Failure::nom(NAFGPU_NOM_VERIFY, "invalid format descriptor");
    size_t i = 3;
    if (i >= n) return eof();
    if (p[i] != 1 && p[i] != 2) return Failure::nom(NAFGPU_NOM_MAPRES, "invalid format version");
    h->format_version = p[i++];
    if (h->format_version == 2) {
        if (i >= n) return eof();
        if (p[i] > 3) return Failure::nom(NAFGPU_NOM_MAPRES, "invalid sequence type");
        h->sequence_type = p[i++];
    }
    if (i >= n) return eof();
    h->flags = p[i++];
    if (i >= n) return eof();
    if (p[i] < 0x20 || p[i] > 0x7E) return Failure::nom(NAFGPU_NOM_VERIFY, "name separator is not printable");
    h->name_separator = p[i++];
    size_t u = 0;
    bool inc = false;
    Failure f = parse_varint(p + i, n - i, &h->line_length, &u, &inc);
    if (!f.ok()) return inc ? eof() : f;
    i += u;
    f = parse_varint(p + i, n - i, &h->number_of_sequences, &u, &inc);
    if (!f.ok()) return inc ? eof() : f;
    i += u;
    *used = i;
    return Failure();
}

Failure parse_archive(const uint8_t *p, size_t n, nafgpu_header *h, SectionInfo sec[kNumSections], const NeedFn *need) {
    size_t i = 0, u = 0;
    bool inc = false;
    auto want = [&](size_t off, size_t len) {          // bytes about to be parsed (a lazily loaded source fetches them)
        if (need && off < n) (*need)(off, len < n - off ? len : n - off);
    };
    want(0, 64);
    Failure f = parse_header(p, n, h, &i, &inc);
    if (!f.ok()) return f;
    // The reference hits `todo!()` (error.rs:50) when a title or size pair is cut short; we
    // report that as NAFGPU_E_PANIC instead of aborting.
    auto cut = []() { return Failure::make(NAFGPU_E_PANIC, "archive ends inside a section header"); };
    if (h->flags & 0x40) {                                         // Title, mod.rs:191-196
        uint64_t tsize = 0;
        want(i, 16);
        f = parse_varint(p + i, n - i, &tsize, &u, &inc);
        if (!f.ok()) return inc ? cut() : f;
        i += u;
        if (tsize > n - i) return cut();
        want(i, static_cast<size_t>(tsize));
        if (!utf8_valid(p + i, tsize))                             // map_res(take(size), from_utf8), parser.rs:133-137
            return Failure::nom(NAFGPU_NOM_MAPRES, "title is not valid UTF-8");
        i += static_cast<size_t>(tsize);
    }
    for (int k = 0; k < kNumSections; k++) {                       // setup_block! x6, mod.rs:235-242
        if (!(h->flags & kSectionFlag[k])) continue;
        size_t at = i < n ? i : n;
        want(at, 32);
        f = parse_varint(p + at, n - at, &sec[k].original_size, &u, &inc);
        if (!f.ok()) return inc ? cut() : f;
        i += u;
        at = i < n ? i : n;
        f = parse_varint(p + at, n - at, &sec[k].compressed_size, &u, &inc);
        if (!f.ok()) return inc ? cut() : f;
        i += u;
        sec[k].present = true;
        sec[k].offset = i;
        // seek(Current(+compressed_size)) may run past EOF without error (mod.rs:228); a short
        // payload only fails when that section is actually decoded.
        if (sec[k].compressed_size > UINT64_MAX - i) i = SIZE_MAX; else i += sec[k].compressed_size;
    }
    return Failure();
}

}  // namespace nafgpu
