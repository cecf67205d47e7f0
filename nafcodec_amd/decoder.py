"""Decoder / Record / open -- same names, arguments and error behaviour as the reference's Python
API (nafcodec-py/nafcodec/lib.pyi:18-67, lib.rs:324-461, error mapping lib.rs:39-77)."""
import ctypes
import errno as _errno
import io
import os
from ctypes import byref, c_void_p

from . import _ffi

SEQUENCE_TYPES = ("dna", "rna", "protein", "text")


class Record:
    """lib.pyi:18-33 -- five optional fields."""
    __slots__ = ("id", "comment", "sequence", "quality", "length")

    def __init__(self, *, id=None, comment=None, sequence=None, quality=None, length=None):
        self.id, self.comment, self.sequence, self.quality, self.length = id, comment, sequence, quality, length

    def __repr__(self):
        args = ", ".join("%s=%r" % (k, getattr(self, k)) for k in self.__slots__ if getattr(self, k) is not None)
        return "Record(%s)" % args


def _raise(err):
    """Error -> Python exception, as nafcodec-py/nafcodec/lib.rs:39-77 does."""
    e = _ffi.NafError.from_c(err)
    if e.status == _ffi.E_IO:
        if e.io_kind == _ffi.IO_NOT_FOUND:
            raise FileNotFoundError(e.os_errno or _errno.ENOENT, e.message)
        if e.io_kind == _ffi.IO_IS_A_DIRECTORY:
            raise IsADirectoryError(e.os_errno or _errno.EISDIR, e.message)
        if e.io_kind == _ffi.IO_PERMISSION_DENIED:
            raise PermissionError(e.os_errno or _errno.EACCES, e.message)
        if e.io_kind == _ffi.IO_UNEXPECTED_EOF:
            raise EOFError(e.message)
        raise OSError(e.os_errno, e.message)
    if e.status in (_ffi.E_NOM, _ffi.E_UTF8):
        raise ValueError(e.message)
    raise e


class Decoder:
    """lib.pyi:35-67.  `file` is a path or a binary file-like object."""

    def __init__(self, file, *, id=True, comment=True, sequence=True, quality=True, mask=True, buffer_size=None,
                 device=-1, spec_mask=False, shard_rank=0, shard_count=1, shard_protocol=False, _lib=None):
        self._lib = _lib or _ffi.default()
        self._h = None
        self._pending_error = None
        opts = _ffi.Opts()
        self._lib.c.nafgpu_opts_default(byref(opts))
        opts.id, opts.comment, opts.sequence, opts.quality, opts.mask = map(int, (id, comment, sequence, quality, mask))
        opts.spec_mask = int(spec_mask)
        opts.buffer_size = io.DEFAULT_BUFFER_SIZE if buffer_size is None else int(buffer_size)  # lib.rs:350-354
        opts.device = device
        opts.shard_rank, opts.shard_count = shard_rank, shard_count   # bulk device path only (decode_all_device)
        opts.shard_protocol = int(bool(shard_protocol))               # ... or the shard protocol (nafcodec_amd.sharding)
        h, err = c_void_p(), _ffi.Error()
        if isinstance(file, (str, bytes, os.PathLike)):
            path = os.fsencode(file)
            rc = self._lib.c.nafgpu_open_path(path, byref(opts), byref(h), byref(err))
        else:
            rc = self._open_filelike(file, opts, h, err)
        if rc != _ffi.OK:
            _raise(err)
        self._h = h
        self._header = _ffi.Header()
        self._lib.c.nafgpu_get_header(self._h, byref(self._header))

    def _open_filelike(self, file, opts, h, err):
        """with_reader (mod.rs:169-256) over a Python file-like, as PyFileRead does (pyfile.rs:88-187): the
        library calls back into `readinto` (or `read`) and `seek`; with a working seek it reads only the
        header and the selected sections.  An exception raised by the file object is re-raised as it is."""
        if not (hasattr(file, "readinto") or hasattr(file, "read")):
            raise TypeError("expected a path or a binary file-like object")
        pending = []                                       # exception raised inside a callback

        def on_read(_ctx, buf, cap):
            try:
                cap = int(cap)
                if hasattr(file, "readinto"):              # pyfile.rs:88-132
                    view = (ctypes.c_uint8 * cap).from_address(ctypes.addressof(buf.contents))
                    n = file.readinto(memoryview(view).cast("B"))
                    return int(n or 0)
                data = file.read(cap)                      # pyfile.rs:134-187
                if not isinstance(data, (bytes, bytearray, memoryview)):
                    raise TypeError("expected bytes from read(), found %s" % type(data).__name__)
                if len(data) > cap:
                    raise OSError(_errno.EIO, "read() returned more bytes than asked for")
                ctypes.memmove(buf, bytes(data), len(data))
                return len(data)
            except BaseException as e:                     # noqa: BLE001 -- carried across the C boundary
                pending.append(e)
                return -(getattr(e, "errno", None) or _errno.EIO)

        def on_seek(_ctx, offset, whence):
            try:
                return int(file.seek(int(offset), int(whence)))
            except BaseException as e:                     # noqa: BLE001
                pending.append(e)
                return -(getattr(e, "errno", None) or _errno.ESPIPE)

        seekable = hasattr(file, "seek")
        try:
            seekable = seekable and (file.seekable() if hasattr(file, "seekable") else True)
        except Exception:
            seekable = False
        read_cb = _ffi.READ_FN(on_read)
        seek_cb = _ffi.SEEK_FN(on_seek) if seekable else _ffi.SEEK_FN()
        rc = self._lib.c.nafgpu_open_io(read_cb, seek_cb, None, byref(opts), byref(h), byref(err))
        if pending and rc != _ffi.OK:
            raise pending[0]
        return rc

    # ---- iteration -----------------------------------------------------------------------
    def __iter__(self):
        return self

    def __next__(self):
        rec = self.read()
        if rec is None:
            raise StopIteration
        return rec

    def read(self):
        """lib.pyi:67 -- the next record, or None at the end of the archive."""
        rec = _ffi.Record()
        rc = self._lib.c.nafgpu_next(self._h, byref(rec))
        if rc == _ffi.END:
            return None
        if rc != _ffi.OK:
            err = _ffi.Error()
            self._lib.c.nafgpu_last_error(self._h, byref(err))
            _raise(err)

        def text(f):
            if not f.present:
                return None
            return ctypes.string_at(f.ptr, f.len).decode("utf-8") if f.len else ""

        return Record(id=text(rec.id), comment=text(rec.comment), sequence=text(rec.sequence),
                      quality=text(rec.quality), length=rec.length if rec.has_length else None)

    def read_batch(self, n=4096):
        """Up to `n` records in ONE call into the library (nafgpu_next_batch): the records `n` calls of read() would return, in
        order; fewer when the archive ends or the next record's bytes are not on the host yet, [] at the end.  An error
        raised by record k of the batch is raised by this call AFTER records 0 .. k-1 were collected: they are returned by
        the call, the exception is kept and raised by the next one (so no record is lost and the order of events is the
        one read() gives)."""
        if self._pending_error is not None:
            e, self._pending_error = self._pending_error, None
            raise e
        recs = (_ffi.Record * n)()
        got = ctypes.c_uint64()
        rc = self._lib.c.nafgpu_next_batch(self._h, recs, n, byref(got))

        def text(f):
            if not f.present:
                return None
            return ctypes.string_at(f.ptr, f.len).decode("utf-8") if f.len else ""

        out = [Record(id=text(r.id), comment=text(r.comment), sequence=text(r.sequence), quality=text(r.quality),
                      length=r.length if r.has_length else None) for r in recs[:got.value]]
        if rc not in (_ffi.OK, _ffi.END):
            err = _ffi.Error()
            self._lib.c.nafgpu_last_error(self._h, byref(err))
            try:
                _raise(err)
            except Exception as e:                          # noqa: BLE001
                if not out:
                    raise
                self._pending_error = e
        return out

    def __len__(self):
        return int(self._lib.c.nafgpu_remaining(self._h))

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        self.close()
        return False

    def close(self):
        if self._h:
            self._lib.c.nafgpu_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- header properties (lib.pyi:56-66) ---------------------------------------------------
    @property
    def sequence_type(self):
        return SEQUENCE_TYPES[self._header.sequence_type]

    @property
    def format_version(self):
        return "v%d" % self._header.format_version

    @property
    def line_length(self):
        return int(self._header.line_length)

    @property
    def name_separator(self):
        return chr(self._header.name_separator)

    @property
    def number_of_sequences(self):
        return int(self._header.number_of_sequences)

    @property
    def flags(self):
        return int(self._header.flags)

    # ---- bulk device path (no reference counterpart: what `for r in decoder` becomes on a GPU) ----
    def decode_all_device(self):
        """Decode every selected section on the GPU; returns the nafgpu_device_result struct."""
        res = _ffi.DeviceResult()
        rc = self._lib.c.nafgpu_decode_all_device(self._h, byref(res))
        if rc != _ffi.OK:
            err = _ffi.Error()
            self._lib.c.nafgpu_last_error(self._h, byref(err))
            _raise(err)
        return res

    # ---- the shard protocol (include/nafgpu.h: nafgpu_shard_*; nafcodec_amd.sharding drives it) ----
    def _check(self, rc):
        if rc != _ffi.OK:
            err = _ffi.Error()
            self._lib.c.nafgpu_last_error(self._h, byref(err))
            _raise(err)

    def shard_begin(self):
        """-> this rank's 64-byte summary (bytes)"""
        mine = _ffi.ShardSummary()
        self._check(self._lib.c.nafgpu_shard_begin(self._h, byref(mine)))
        return bytes(mine)

    def shard_place(self, summaries):
        """summaries: the ranks' summaries back to back (bytes-like, 64 bytes per rank, rank order)"""
        buf = ctypes.create_string_buffer(bytes(summaries), len(summaries))
        self._check(self._lib.c.nafgpu_shard_place(self._h, ctypes.cast(buf, c_void_p), len(summaries) // ctypes.sizeof(_ffi.ShardSummary)))

    def shard_halo(self, section):
        """-> (recv_bytes, send_bytes, tail_ready) for section 0 (Sequence) / 1 (Quality)"""
        recv, send, ready = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_int()
        self._check(self._lib.c.nafgpu_shard_halo(self._h, section, byref(recv), byref(send), byref(ready)))
        return recv.value, send.value, bool(ready.value)

    def shard_export_tail(self, section, ptr, n):
        self._check(self._lib.c.nafgpu_shard_export_tail(self._h, section, ptr, n))

    def shard_import_halo(self, section, ptr, n):
        self._check(self._lib.c.nafgpu_shard_import_halo(self._h, section, ptr, n))

    def shard_finish(self):
        res = _ffi.DeviceResult()
        self._check(self._lib.c.nafgpu_shard_finish(self._h, byref(res)))
        return res

    def format_device(self):
        """FASTA (FASTQ when the archive has qualities and `quality` is selected) text of every record,
        built on the GPU from the decoded buffers; returns the nafgpu_text_result struct (device pointer)."""
        res = _ffi.TextResult()
        rc = self._lib.c.nafgpu_format_device(self._h, byref(res))
        if rc != _ffi.OK:
            err = _ffi.Error()
            self._lib.c.nafgpu_last_error(self._h, byref(err))
            _raise(err)
        return res

    def to_text(self):
        """format_device() copied to the host: the archive as FASTA / FASTQ bytes (what `unnaf` prints)."""
        res = self.format_device()
        return self.copy_to_host(res.d_text, res.n_text)

    def copy_to_host(self, d_ptr, n):
        buf = ctypes.create_string_buffer(int(n)) if n else ctypes.create_string_buffer(1)
        rc = self._lib.c.nafgpu_copy_to_host(self._h, d_ptr, int(n), ctypes.cast(buf, ctypes.c_void_p))
        if rc != _ffi.OK:
            raise RuntimeError("nafgpu_copy_to_host failed: %d" % rc)
        return buf.raw[:int(n)]

    def hash_device(self, d_ptr, n, first_chunk=0):
        out = ctypes.c_uint64()
        rc = self._lib.c.nafgpu_hash64_device_at(self._h, d_ptr, n, first_chunk, byref(out))
        if rc != _ffi.OK:
            raise RuntimeError("nafgpu_hash64_device failed: %d" % rc)
        return out.value


def open(file, mode="r", **options):
    """nafcodec.open (lib.pyi:89-108, nafcodec/__init__.py): "r" -> Decoder, "w" -> Encoder."""
    if mode == "r":
        return Decoder(file, **options)
    if mode == "w":
        from .encoder import Encoder
        return Encoder(file, **options)
    raise ValueError("invalid mode: %r" % (mode,))
