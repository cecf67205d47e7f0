"""System libzstd through ctypes: used by tests to (a) cross-check the oracle's zstd
stage and (b) write test archives at real compression levels.  Test helper only."""
import ctypes
import ctypes.util

_lib = None


def lib():
    global _lib
    if _lib is None:
        for name in ("libzstd.so.1", ctypes.util.find_library("zstd")):
            if not name:
                continue
            try:
                L = ctypes.CDLL(name)
            except OSError:
                continue
            L.ZSTD_compressBound.restype = ctypes.c_size_t
            L.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
            L.ZSTD_isError.argtypes = [ctypes.c_size_t]
            L.ZSTD_createCCtx.restype = ctypes.c_void_p
            L.ZSTD_freeCCtx.argtypes = [ctypes.c_void_p]
            L.ZSTD_CCtx_setParameter.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
            L.ZSTD_CCtx_setParameter.restype = ctypes.c_size_t
            L.ZSTD_compress2.restype = ctypes.c_size_t
            L.ZSTD_compress2.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                         ctypes.c_char_p, ctypes.c_size_t]
            L.ZSTD_compressStream2.restype = ctypes.c_size_t
            L.ZSTD_decompress.restype = ctypes.c_size_t
            L.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
            L.ZSTD_versionNumber.restype = ctypes.c_uint
            _lib = L
            break
        else:
            _lib = False
    return _lib or None


def available():
    return lib() is not None


ZSTD_c_compressionLevel = 100
ZSTD_c_windowLog = 101
ZSTD_c_contentSizeFlag = 200
ZSTD_c_checksumFlag = 201
ZSTD_MAGIC = b"\x28\xb5\x2f\xfd"


class _Buf(ctypes.Structure):
    _fields_ = [("p", ctypes.c_void_p), ("size", ctypes.c_size_t), ("pos", ctypes.c_size_t)]


def compress_magicless(data: bytes, level=1, streaming=True, checksum=False, window_log=None, flush_every=None):
    """One zstd frame WITHOUT the 4-byte magic (what a NAF section holds; mod.rs:221-222).

    streaming=True uses ZSTD_compressStream2 like ennaf / the reference encoder (FHD 0x00, no
    content size); flush_every=N forces a block boundary every N input bytes (the reference
    encoder flushes per record, encoder/mod.rs:271,298,319)."""
    L = lib()
    cctx = L.ZSTD_createCCtx()
    try:
        L.ZSTD_CCtx_setParameter(cctx, ZSTD_c_compressionLevel, level)
        if checksum:
            L.ZSTD_CCtx_setParameter(cctx, ZSTD_c_checksumFlag, 1)
        if window_log:
            L.ZSTD_CCtx_setParameter(cctx, ZSTD_c_windowLog, window_log)
        bound = L.ZSTD_compressBound(len(data)) + 1024 + (0 if not flush_every else 16 * (len(data) // flush_every + 2))
        out = ctypes.create_string_buffer(bound)
        if not streaming:
            n = L.ZSTD_compress2(cctx, out, bound, data, len(data))
            assert not L.ZSTD_isError(n)
            frame = out.raw[:n]
        else:
            src = ctypes.create_string_buffer(data, len(data)) if data else ctypes.create_string_buffer(1)
            ob = _Buf(ctypes.cast(out, ctypes.c_void_p), bound, 0)
            step = flush_every or max(len(data), 1)
            pos = 0
            while pos < len(data):
                end = min(len(data), pos + step)
                ib = _Buf(ctypes.cast(src, ctypes.c_void_p), end, pos)
                mode = 1 if flush_every else 0  # ZSTD_e_flush / ZSTD_e_continue
                while True:
                    r = L.ZSTD_compressStream2(ctypes.c_void_p(cctx), ctypes.byref(ob), ctypes.byref(ib), mode)
                    assert not L.ZSTD_isError(r)
                    if ib.pos == end and (mode == 0 or r == 0):
                        break
                pos = end
            ib = _Buf(ctypes.cast(src, ctypes.c_void_p), len(data), len(data))
            while True:
                r = L.ZSTD_compressStream2(ctypes.c_void_p(cctx), ctypes.byref(ob), ctypes.byref(ib), 2)  # ZSTD_e_end
                assert not L.ZSTD_isError(r)
                if r == 0:
                    break
            frame = out.raw[:ob.pos]
    finally:
        L.ZSTD_freeCCtx(cctx)
    assert frame[:4] == ZSTD_MAGIC
    return frame[4:]


def decompress_magicless(payload: bytes, capacity: int) -> bytes:
    L = lib()
    out = ctypes.create_string_buffer(max(capacity, 1))
    framed = ZSTD_MAGIC + payload
    n = L.ZSTD_decompress(out, capacity, framed, len(framed))
    if L.ZSTD_isError(n):
        raise ValueError("libzstd error %d" % n)
    return out.raw[:n]
