"""ctypes binding of include/nafgpu.h (libnafgpu.so, built by __graft_entry__.build()).

There is no CPU fallback: if the HIP library is missing or no GPU is usable every decode
raises.  (`Library(path)` exists so tests can point the same binding at another build of
the same sources.)"""
import ctypes
import os
from ctypes import (POINTER, Structure, byref, c_char, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t,
                    c_uint8, c_uint32, c_uint64, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_PATH = os.path.join(_HERE, "libnafgpu.so")

OK, END = 0, 1
E_IO, E_NOM, E_UTF8, E_PANIC, E_DEVICE, E_INVALID_ARG = -1, -2, -3, -4, -5, -6
E_MISSING_FIELD, E_INVALID_LENGTH, E_INVALID_SEQUENCE = -7, -8, -9
IO_UNEXPECTED_EOF, IO_INVALID_DATA, IO_NOT_FOUND, IO_IS_A_DIRECTORY, IO_PERMISSION_DENIED, IO_OTHER = 1, 2, 3, 4, 5, 6
NOM_VERIFY, NOM_MAPRES, NOM_TOOLARGE = 1, 2, 3


class Error(Structure):
    _fields_ = [("status", c_int32), ("io_kind", c_int32), ("os_errno", c_int32), ("nom_code", c_int32),
                ("message", c_char * 192)]


class Opts(Structure):
    _fields_ = [("id", c_uint8), ("comment", c_uint8), ("sequence", c_uint8), ("quality", c_uint8),
                ("mask", c_uint8), ("spec_mask", c_uint8), ("shard_protocol", c_uint8), ("reserved", c_uint8 * 1),
                ("buffer_size", c_uint64), ("device", c_int32), ("shard_rank", c_int32),
                ("shard_count", c_int32), ("tile_mib", c_int32)]


class Header(Structure):
    _fields_ = [("format_version", c_uint8), ("sequence_type", c_uint8), ("flags", c_uint8),
                ("name_separator", c_uint8), ("reserved", c_uint32), ("line_length", c_uint64),
                ("number_of_sequences", c_uint64)]


class Field(Structure):
    _fields_ = [("ptr", c_void_p), ("len", c_uint64), ("present", c_uint8), ("reserved", c_uint8 * 7)]


class Record(Structure):
    _fields_ = [("id", Field), ("comment", Field), ("sequence", Field), ("quality", Field),
                ("length", c_uint64), ("has_length", c_uint8), ("reserved", c_uint8 * 7)]


class DeviceResult(Structure):
    _fields_ = [("d_sequence", c_void_p), ("d_quality", c_void_p), ("d_record_end", c_void_p),
                ("d_ids", c_void_p), ("d_comments", c_void_p),
                ("n_bases", c_uint64), ("n_quality", c_uint64), ("n_records", c_uint64),
                ("n_ids_bytes", c_uint64), ("n_comments_bytes", c_uint64),
                ("packed_bytes", c_uint64), ("compressed_bytes", c_uint64), ("seq_compressed_bytes", c_uint64),
                ("n_zstd_blocks", c_uint64), ("n_huf_streams", c_uint64), ("first_record", c_uint64),
                ("carry", c_uint8), ("sharded", c_uint8), ("reserved", c_uint8 * 6),
                ("ms_total", c_float), ("ms_huf", c_float), ("ms_unpack", c_float), ("ms_seq_lz", c_float),
                ("ms_other", c_float), ("ms_host_plan", c_float), ("ms_h2d", c_float),
                ("n_huf_launches", c_uint32), ("reserved3", c_uint32), ("lz_residue_matches", c_uint64),
                ("base_offset", c_uint64),
                ("d_id_end", c_void_p), ("d_comment_end", c_void_p), ("n_ids", c_uint64), ("n_comments", c_uint64),
                ("utf8_invalid", c_uint32), ("reserved4", c_uint32), ("quality_offset", c_uint64)]


class ShardSummary(Structure):
    """nafgpu_shard_summary: 64 bytes, gathered over the ranks as it is."""
    _fields_ = [("decoded", c_uint64 * 2), ("frame_tail", c_uint64 * 2), ("rep_map", (c_uint32 * 3) * 2),
                ("failed", c_uint8 * 2), ("reserved", c_uint8 * 6)]


class TextResult(Structure):
    _fields_ = [("d_text", c_void_p), ("n_text", c_uint64), ("n_records", c_uint64), ("ms", c_float),
                ("fastq", c_uint8), ("reserved", c_uint8 * 3)]


class SynthSpec(Structure):
    _fields_ = [("n_bases", c_uint64), ("seed", c_uint64), ("with_mask", c_uint8), ("iupac_permille", c_uint8),
                ("part_count", c_uint8), ("part_rank", c_uint8), ("reserved", c_uint8 * 4), ("threads", c_uint32), ("reserved2", c_uint32)]


class SynthArchive(Structure):
    _fields_ = [("bytes", c_void_p), ("n", c_uint64), ("n_records", c_uint64), ("n_bases", c_uint64),
                ("seq_hash", c_uint64), ("offsets_hash", c_uint64)]


class EncoderOpts(Structure):
    _fields_ = [("sequence_type", c_uint8), ("id", c_uint8), ("comment", c_uint8), ("sequence", c_uint8), ("quality", c_uint8),
                ("reserved", c_uint8 * 3), ("compression_level", ctypes.c_int32), ("threads", c_uint32)]


READ_FN = ctypes.CFUNCTYPE(c_int64, c_void_p, POINTER(c_uint8), c_uint64)
SEEK_FN = ctypes.CFUNCTYPE(c_int64, c_void_p, c_int64, c_int)

# every symbol include/nafgpu.h declares
EXPORTS = [
    "nafgpu_opts_default", "nafgpu_opts_from_flags", "nafgpu_open_path", "nafgpu_open_bytes", "nafgpu_open_io",
    "nafgpu_get_header", "nafgpu_remaining", "nafgpu_next", "nafgpu_close", "nafgpu_last_error",
    "nafgpu_decode_all_device", "nafgpu_zstd_decompress", "nafgpu_synth_write", "nafgpu_synth_free",
    "nafgpu_hash64_host", "nafgpu_hash64_device", "nafgpu_abi_version", "nafgpu_device_info",
    "nafgpu_upload", "nafgpu_device_synchronize", "nafgpu_hash64_device_at",
    "nafgpu_format_device", "nafgpu_copy_to_host", "nafgpu_synth_head",
    "nafgpu_encoder_opts_default", "nafgpu_encoder_opts_from_flags", "nafgpu_encoder_new", "nafgpu_encoder_push",
    "nafgpu_encoder_finish", "nafgpu_encoder_free", "nafgpu_test_hooks",
    "nafgpu_hash64_host_at", "nafgpu_shard_begin", "nafgpu_shard_place", "nafgpu_shard_halo", "nafgpu_shard_export_tail",
    "nafgpu_shard_import_halo", "nafgpu_shard_finish", "nafgpu_next_batch", "nafgpu_trim_device_memory",
]


class Library:
    def __init__(self, path=DEFAULT_PATH):
        if not os.path.exists(path):
            raise ImportError(
                "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950); nafcodec_amd has no CPU fallback" % path)
        L = self.c = ctypes.CDLL(path)
        self.path = path
        L.nafgpu_opts_default.argtypes = [POINTER(Opts)]
        L.nafgpu_opts_default.restype = None
        L.nafgpu_opts_from_flags.argtypes = [POINTER(Opts), c_uint8]
        L.nafgpu_opts_from_flags.restype = None
        L.nafgpu_open_path.argtypes = [c_char_p, POINTER(Opts), POINTER(c_void_p), POINTER(Error)]
        L.nafgpu_open_bytes.argtypes = [c_char_p, c_size_t, POINTER(Opts), POINTER(c_void_p), POINTER(Error)]
        L.nafgpu_open_io.argtypes = [READ_FN, SEEK_FN, c_void_p, POINTER(Opts), POINTER(c_void_p), POINTER(Error)]
        L.nafgpu_get_header.argtypes = [c_void_p, POINTER(Header)]
        L.nafgpu_get_header.restype = None
        L.nafgpu_remaining.argtypes = [c_void_p]
        L.nafgpu_remaining.restype = c_uint64
        L.nafgpu_next.argtypes = [c_void_p, POINTER(Record)]
        if hasattr(L, "nafgpu_next_batch"):                  # (absent from older builds loaded for A/B runs: tools/*_probe.py)
            L.nafgpu_next_batch.argtypes = [c_void_p, POINTER(Record), ctypes.c_uint64, POINTER(ctypes.c_uint64)]
        L.nafgpu_close.argtypes = [c_void_p]
        L.nafgpu_close.restype = None
        L.nafgpu_last_error.argtypes = [c_void_p, POINTER(Error)]
        L.nafgpu_last_error.restype = None
        L.nafgpu_decode_all_device.argtypes = [c_void_p, POINTER(DeviceResult)]
        L.nafgpu_upload.argtypes = [c_void_p]
        L.nafgpu_device_synchronize.argtypes = [c_int]
        if hasattr(L, "nafgpu_trim_device_memory"):          # (absent from older builds loaded for A/B runs)
            L.nafgpu_trim_device_memory.argtypes = [c_int]
        L.nafgpu_zstd_decompress.argtypes = [c_char_p, c_size_t, c_void_p, c_size_t, POINTER(c_size_t), c_int,
                                             POINTER(Error)]
        L.nafgpu_encoder_opts_default.argtypes = [c_uint8, POINTER(EncoderOpts)]
        L.nafgpu_encoder_opts_default.restype = None
        L.nafgpu_encoder_opts_from_flags.argtypes = [c_uint8, c_uint8, POINTER(EncoderOpts)]
        L.nafgpu_encoder_opts_from_flags.restype = None
        L.nafgpu_encoder_new.argtypes = [POINTER(EncoderOpts), POINTER(c_void_p), POINTER(Error)]
        L.nafgpu_encoder_push.argtypes = [c_void_p, POINTER(Record), POINTER(Error)]
        L.nafgpu_encoder_finish.argtypes = [c_void_p, POINTER(c_void_p), POINTER(c_uint64), POINTER(Error)]
        L.nafgpu_encoder_free.argtypes = [c_void_p]
        L.nafgpu_encoder_free.restype = None
        L.nafgpu_synth_write.argtypes = [POINTER(SynthSpec), POINTER(SynthArchive)]
        L.nafgpu_synth_head.argtypes = [POINTER(SynthSpec), c_uint64, POINTER(SynthArchive)]
        L.nafgpu_synth_free.argtypes = [POINTER(SynthArchive)]
        L.nafgpu_synth_free.restype = None
        L.nafgpu_hash64_host.argtypes = [c_char_p, c_uint64]
        L.nafgpu_hash64_host.restype = c_uint64
        L.nafgpu_hash64_host_at.argtypes = [c_char_p, c_uint64, c_uint64]
        L.nafgpu_hash64_host_at.restype = c_uint64
        L.nafgpu_hash64_device.argtypes = [c_void_p, c_void_p, c_uint64, POINTER(c_uint64)]
        L.nafgpu_hash64_device_at.argtypes = [c_void_p, c_void_p, c_uint64, c_uint64, POINTER(c_uint64)]
        L.nafgpu_device_info.argtypes = [c_int, c_char_p, c_size_t, POINTER(c_uint64), POINTER(c_int)]
        L.nafgpu_format_device.argtypes = [c_void_p, POINTER(TextResult)]
        L.nafgpu_copy_to_host.argtypes = [c_void_p, c_void_p, c_uint64, c_void_p]
        L.nafgpu_shard_begin.argtypes = [c_void_p, POINTER(ShardSummary)]
        L.nafgpu_shard_place.argtypes = [c_void_p, c_void_p, c_int]
        L.nafgpu_shard_halo.argtypes = [c_void_p, c_int, POINTER(c_uint64), POINTER(c_uint64), POINTER(c_int)]
        L.nafgpu_shard_export_tail.argtypes = [c_void_p, c_int, c_void_p, c_uint64]
        L.nafgpu_shard_import_halo.argtypes = [c_void_p, c_int, c_void_p, c_uint64]
        L.nafgpu_shard_finish.argtypes = [c_void_p, POINTER(DeviceResult)]
        L.nafgpu_test_hooks.argtypes = [c_int]
        L.nafgpu_test_hooks.restype = None

    # ---- helpers ---------------------------------------------------------------------------
    def zstd_decompress(self, payload: bytes, size: int, device: int = -1) -> bytes:
        """Decode one NAF section payload (magicless zstd frame(s)) on the GPU."""
        buf = ctypes.create_string_buffer(max(size, 1))
        produced, err = c_size_t(0), Error()
        rc = self.c.nafgpu_zstd_decompress(payload, len(payload), buf, size, byref(produced), device, byref(err))
        if rc != OK:
            raise NafError.from_c(err)
        return buf.raw[:produced.value]

    def synth(self, n_bases, seed=0x4E4146, with_mask=False, iupac_permille=0, threads=0, part_rank=0, part_count=1):
        """nafgpu_synth_write: the whole archive, or (part_count > 1) this process's share of its sequence blocks."""
        spec = SynthSpec(n_bases=n_bases, seed=seed, with_mask=int(with_mask), iupac_permille=iupac_permille,
                         threads=threads, part_rank=part_rank, part_count=part_count)
        arc = SynthArchive()
        rc = self.c.nafgpu_synth_write(byref(spec), byref(arc))
        if rc != OK:
            raise RuntimeError("nafgpu_synth_write failed: %d" % rc)
        return arc

    def synth_head(self, n_bases, seq_part_bytes, seed=0x4E4146, with_mask=False, iupac_permille=0):
        """nafgpu_synth_head: what goes in front of the parts of an archive written in parts."""
        spec = SynthSpec(n_bases=n_bases, seed=seed, with_mask=int(with_mask), iupac_permille=iupac_permille)
        arc = SynthArchive()
        rc = self.c.nafgpu_synth_head(byref(spec), seq_part_bytes, byref(arc))
        if rc != OK:
            raise RuntimeError("nafgpu_synth_head failed: %d" % rc)
        return arc

    def device_info(self, device=0):
        name = ctypes.create_string_buffer(256)
        hbm, cus = c_uint64(), c_int()
        rc = self.c.nafgpu_device_info(device, name, 256, byref(hbm), byref(cus))
        if rc != OK:
            raise RuntimeError("no usable HIP device (status %d)" % rc)
        return name.value.decode(), hbm.value, cus.value


class NafError(Exception):
    """Mirror of nafcodec::Error (error.rs:4-11)."""

    def __init__(self, status, io_kind=0, os_errno=0, nom_code=0, message=""):
        super().__init__(message or "nafgpu error %d" % status)
        self.status, self.io_kind, self.os_errno, self.nom_code, self.message = status, io_kind, os_errno, nom_code, message

    @classmethod
    def from_c(cls, e):
        return cls(e.status, e.io_kind, e.os_errno, e.nom_code, e.message.decode("utf-8", "replace"))


_default = None


def default():
    global _default
    if _default is None:
        _default = Library(DEFAULT_PATH)
    return _default
