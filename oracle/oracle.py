"""ctypes binding of the CPU oracle (oracle/*.c) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (nafcodec_amd) never does.
"""
import ctypes
import os
import subprocess
from ctypes import POINTER, byref, c_char_p, c_int, c_long, c_size_t, c_uint8, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libnaforacle.so")


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("zstd_oracle.c", "naf_oracle.c", "ref_shape.c", "naf_oracle.h", "Makefile")]
    stale = force or not os.path.exists(_SO) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-s", "-C", _HERE], stdout=subprocess.DEVNULL)
    return _SO


class ZoStats(ctypes.Structure):
    _fields_ = [(n, c_uint64) for n in (
        "frames", "blocks", "blocks_raw", "blocks_rle", "blocks_compressed",
        "lit_raw", "lit_rle", "lit_huf", "lit_treeless",
        "lit_bytes", "lit_bytes_entropy", "match_bytes", "sequences", "window")] + [
        ("seq_mode_count", c_uint64 * 12)]


class NoHeader(ctypes.Structure):
    _fields_ = [("format_version", c_uint8), ("sequence_type", c_uint8), ("flags", c_uint8),
                ("name_separator", c_uint8), ("line_length", c_uint64),
                ("number_of_sequences", c_uint64)]


class NoOpts(ctypes.Structure):
    _fields_ = [(n, c_uint8) for n in ("id", "comment", "sequence", "quality", "mask", "spec_mask")]


class NoField(ctypes.Structure):
    _fields_ = [("ptr", c_void_p), ("len", c_uint64), ("present", c_uint8)]


class NoRecord(ctypes.Structure):
    _fields_ = [("id", NoField), ("comment", NoField), ("sequence", NoField), ("quality", NoField),
                ("length", c_uint64), ("has_length", c_uint8)]


class DrainResult(ctypes.Structure):
    _fields_ = [(n, c_uint64) for n in ("n_records", "n_bases", "n_quality", "seq_hash", "qual_hash", "ends_hash",
                                        "ids_hash", "com_hash")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        L.zo_decode_section.restype = c_long
        L.zo_decode_section.argtypes = [c_char_p, c_size_t, c_void_p, c_size_t, POINTER(ZoStats)]
        L.no_variable_u64.argtypes = [c_char_p, c_size_t, POINTER(c_uint64), POINTER(c_size_t), POINTER(c_int)]
        L.no_parse_header.argtypes = [c_char_p, c_size_t, POINTER(NoHeader), POINTER(c_size_t), POINTER(c_int)]
        L.no_open.argtypes = [c_char_p, c_size_t, POINTER(NoOpts), POINTER(c_void_p), POINTER(c_int)]
        L.no_get_header.argtypes = [c_void_p, POINTER(NoHeader)]
        L.no_remaining.restype = c_uint64
        L.no_remaining.argtypes = [c_void_p]
        L.no_next.argtypes = [c_void_p, POINTER(NoRecord)]
        L.no_close.argtypes = [c_void_p]
        L.no_section.argtypes = [c_void_p, c_int, POINTER(c_void_p), POINTER(c_uint64), POINTER(c_uint64),
                                 POINTER(c_uint64), POINTER(c_uint64)]
        L.no_mask_units.restype = c_size_t
        L.no_mask_units.argtypes = [c_void_p, POINTER(c_uint64), POINTER(c_uint8), c_size_t]
        L.no_drain.argtypes = [c_void_p, c_int, POINTER(DrainResult)]
        L.rs_available.restype = c_int
        L.rs_drain.argtypes = [c_char_p, c_size_t, c_int, POINTER(DrainResult)]
        L.rs_drain_limit.argtypes = [c_char_p, c_size_t, c_int, c_uint64, POINTER(DrainResult)]
        L.rs_drain_parallel.restype = c_uint64
        L.rs_drain_parallel.argtypes = [c_char_p, c_size_t, c_int]
        _lib = L
    return _lib


class OracleError(Exception):
    def __init__(self, kind, nom_code=0):
        super().__init__("oracle error kind=%d nom=%d" % (kind, nom_code))
        self.kind = kind
        self.nom_code = nom_code


E_IO_EOF, E_IO_INVALID, E_NOM, E_PANIC = -1, -2, -3, -4
NOM_VERIFY, NOM_MAPRES, NOM_TOOLARGE = 1, 2, 3
SEQUENCE_TYPES = ("dna", "rna", "protein", "text")


def zstd_decode(payload: bytes, capacity: int, stats=False):
    """Decode a magicless zstd section payload; returns bytes (and ZoStats)."""
    buf = ctypes.create_string_buffer(max(capacity, 1))
    st = ZoStats()
    n = lib().zo_decode_section(payload, len(payload), buf, capacity, byref(st))
    if n < 0:
        raise OracleError(int(n))
    out = buf.raw[:n]
    return (out, st) if stats else out


def variable_u64(data: bytes):
    out, used, code = c_uint64(), c_size_t(), c_int()
    rc = lib().no_variable_u64(data, len(data), byref(out), byref(used), byref(code))
    if rc != 0:
        raise OracleError(rc, code.value)
    return out.value, used.value


def parse_header(data: bytes):
    h, used, code = NoHeader(), c_size_t(), c_int()
    rc = lib().no_parse_header(data, len(data), byref(h), byref(used), byref(code))
    if rc != 0:
        raise OracleError(rc, code.value)
    return h, used.value


def ref_shape_available():
    """True when the reference-shaped pipeline (oracle/ref_shape.c) can run: it needs the system libzstd."""
    return bool(lib().rs_available())


def ref_shape_drain(data: bytes, want_hash=True):
    """The reference pipeline in its own shape (streaming libzstd through 4 KiB buffers, per-nibble push, one heap
    string per field and record; oracle/ref_shape.c) drained over a whole archive -> DrainResult."""
    out = DrainResult()
    rc = lib().rs_drain(data, len(data), int(want_hash), byref(out))
    if rc != 0:
        raise OracleError(rc)
    return out


def ref_shape_stream(data: bytes, limit=(1 << 64) - 1, want_hash=True):
    """The same, as far as it gets: -> (rc, DrainResult) where rc is 0 or the error the STREAMING pipeline meets, and the
    result counts and hashes the records it had handed out before that (or before `limit` records).  This is the
    reference's error timing: records in front of a corrupt block are yielded, then Err (mod.rs:356-399)."""
    out = DrainResult()
    rc = lib().rs_drain_limit(data, len(data), int(want_hash), limit, byref(out))
    return rc, out


def ref_shape_drain_parallel(data: bytes, threads: int) -> int:
    """`threads` independent drains of the same archive at once (an upper bound for "all cores"); bases decoded in all."""
    return int(lib().rs_drain_parallel(data, len(data), threads))


class Record:
    __slots__ = ("id", "comment", "sequence", "quality", "length")

    def __init__(self, id=None, comment=None, sequence=None, quality=None, length=None):
        self.id, self.comment, self.sequence, self.quality, self.length = id, comment, sequence, quality, length

    def __repr__(self):
        return "Record(id=%r, length=%r)" % (self.id, self.length)


def _field(f):
    if not f.present:
        return None
    return ctypes.string_at(f.ptr, f.len) if f.len else b""


class Decoder:
    """Mirror of nafcodec.Decoder (nafcodec-py/nafcodec/lib.pyi:35-67) over the CPU oracle."""

    def __init__(self, data: bytes, *, id=True, comment=True, sequence=True, quality=True, mask=True,
                 spec_mask=False, raw=False):
        self._h = None
        self._data = bytes(data)
        self._raw = raw
        opts = NoOpts(int(id), int(comment), int(sequence), int(quality), int(mask), int(spec_mask))
        h, code = c_void_p(), c_int()
        rc = lib().no_open(self._data, len(self._data), byref(opts), byref(h), byref(code))
        if rc != 0:
            raise OracleError(rc, code.value)
        self._h = h
        self.header = NoHeader()
        lib().no_get_header(self._h, byref(self.header))

    def close(self):
        if self._h:
            lib().no_close(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def __iter__(self):
        return self

    def __len__(self):
        return int(lib().no_remaining(self._h))

    def __next__(self):
        rec = NoRecord()
        rc = lib().no_next(self._h, byref(rec))
        if rc == 1:
            raise StopIteration
        if rc != 0:
            raise OracleError(rc)
        conv = (lambda b: b) if self._raw else (lambda b: None if b is None else b.decode("utf-8"))
        return Record(conv(_field(rec.id)), conv(_field(rec.comment)), conv(_field(rec.sequence)),
                      conv(_field(rec.quality)), rec.length if rec.has_length else None)

    @property
    def sequence_type(self):
        return SEQUENCE_TYPES[self.header.sequence_type]

    @property
    def format_version(self):
        return "v%d" % self.header.format_version

    @property
    def line_length(self):
        return self.header.line_length

    @property
    def name_separator(self):
        return chr(self.header.name_separator)

    @property
    def number_of_sequences(self):
        return self.header.number_of_sequences

    def section(self, which):
        """(decoded bytes, original_size, compressed_size, file_offset) of section `which`
        (0 ids, 1 comments, 2 lengths, 3 mask, 4 sequence, 5 quality) or None."""
        p, n, o, c, off = c_void_p(), c_uint64(), c_uint64(), c_uint64(), c_uint64()
        rc = lib().no_section(self._h, which, byref(p), byref(n), byref(o), byref(c), byref(off))
        if rc == 0:
            return None
        if rc < 0:
            raise OracleError(rc)
        data = ctypes.string_at(p, n.value) if p.value and n.value else b""
        return data, o.value, c.value, off.value

    def drain(self, want_hash=True):
        """Drains the iterator in C (no Python objects per record) -> DrainResult with counts and, if asked,
        the position-keyed checksums of the concatenated fields (the checker for full-size GPU parity)."""
        out = DrainResult()
        rc = lib().no_drain(self._h, int(want_hash), byref(out))
        if rc != 0:
            raise OracleError(rc)
        return out

    def mask_units(self, cap=1 << 16):
        ln = (c_uint64 * cap)()
        mk = (c_uint8 * cap)()
        k = lib().no_mask_units(self._h, ln, mk, cap)
        return [(bool(mk[i]), ln[i]) for i in range(k)]
