"""Encoder -- same names, arguments and error behaviour as the reference's Python API
(nafcodec-py/nafcodec/lib.pyi:69-87, lib.rs:463-600; the Rust side: encoder/mod.rs:46-384).  Host code, like the
reference's; every section is written as Huffman-literal Zstandard blocks (include/nafgpu.h: Encoder)."""
import ctypes
import os
from ctypes import byref, c_uint64, c_void_p

from . import _ffi
from .decoder import SEQUENCE_TYPES, Record


class Encoder:
    """lib.pyi:69-87.  `file` is a path or a binary file-like object; the archive is written by close()."""

    def __init__(self, file, sequence_type="dna", *, id=False, comment=False, sequence=False, quality=False,
                 compression_level=0, _lib=None):
        if sequence_type not in SEQUENCE_TYPES:
            raise ValueError("expected 'dna', 'rna', 'protein' or 'text', got %r" % (sequence_type,))   # lib.rs:487-495
        self._lib = _lib or _ffi.default()
        self._file = file
        self._h = None
        if not isinstance(file, (str, bytes, os.PathLike)) and not hasattr(file, "write"):
            raise TypeError("expected a path or a binary file-like object")
        opts = _ffi.EncoderOpts()
        self._lib.c.nafgpu_encoder_opts_default(SEQUENCE_TYPES.index(sequence_type), byref(opts))
        opts.id, opts.comment, opts.sequence, opts.quality = map(int, (id, comment, sequence, quality))
        opts.compression_level = int(compression_level)
        h, err = c_void_p(), _ffi.Error()
        if self._lib.c.nafgpu_encoder_new(byref(opts), byref(h), byref(err)) != _ffi.OK:
            raise _ffi.NafError.from_c(err)
        self._h = h
        if isinstance(file, (str, bytes, os.PathLike)):      # fail now, as the reference does when it creates the file
            self._out = open_binary(file)
        else:
            self._out = None

    def write(self, record):
        """lib.pyi:85 -- push one record; ValueError for a missing field, an inconsistent length or an invalid letter
        (lib.rs:39-52)."""
        if self._h is None:
            raise RuntimeError("operation on closed encoder.")                         # lib.rs:584
        rec, keep = _ffi.Record(), []
        for name in ("id", "comment", "sequence", "quality"):
            value = getattr(record, name)
            if value is None:
                continue
            data = value.encode("utf-8") if isinstance(value, str) else bytes(value)
            buf = ctypes.create_string_buffer(data, len(data)) if data else ctypes.create_string_buffer(1)
            keep.append(buf)
            f = getattr(rec, name)
            f.ptr, f.len, f.present = ctypes.cast(buf, c_void_p), len(data), 1
        if record.length is not None:
            rec.length, rec.has_length = int(record.length), 1
        err = _ffi.Error()
        rc = self._lib.c.nafgpu_encoder_push(self._h, byref(rec), byref(err))
        if rc in (_ffi.E_MISSING_FIELD, _ffi.E_INVALID_LENGTH, _ffi.E_INVALID_SEQUENCE):
            raise ValueError("invalid characters found in sequence" if rc == _ffi.E_INVALID_SEQUENCE
                             else err.message.decode("utf-8", "replace"))
        if rc != _ffi.OK:
            raise _ffi.NafError.from_c(err)

    def close(self):
        """lib.pyi:86 -- build the archive (Encoder::write, mod.rs:325-384) and write it to the file."""
        if self._h is None:
            return
        h, self._h = self._h, None
        try:
            p, n, err = c_void_p(), c_uint64(), _ffi.Error()
            if self._lib.c.nafgpu_encoder_finish(h, byref(p), byref(n), byref(err)) != _ffi.OK:
                raise _ffi.NafError.from_c(err)
            data = ctypes.string_at(p, n.value)
            if self._out is not None:
                with self._out as f:
                    f.write(data)
            else:
                self._file.write(data)
        finally:
            self._lib.c.nafgpu_encoder_free(h)

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        self.close()
        return False

    def __del__(self):
        if getattr(self, "_h", None) is not None:
            try:
                self._lib.c.nafgpu_encoder_free(self._h)
            except Exception:
                pass
            self._h = None


def open_binary(path):
    import builtins
    return builtins.open(os.fspath(path), "wb")


__all__ = ["Encoder", "Record"]
