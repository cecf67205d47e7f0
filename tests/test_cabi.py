"""The C-ABI library builds for gfx950 here (hipcc cross-compiles without a GPU), loads, and exports
every symbol include/nafgpu.h declares.  No compute calls: this container has no GPU -- and the
product must say so loudly instead of falling back to a CPU path."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT, golden_bytes


@pytest.fixture(scope="module")
def lib():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "nafcodec_amd", "csrc")])
    from nafcodec_amd import _ffi
    return _ffi.Library(_ffi.DEFAULT_PATH)


def test_every_declared_symbol_is_exported(lib):
    from nafcodec_amd import _ffi
    header = open(os.path.join(ROOT, "include", "nafgpu.h")).read()
    declared = set(re.findall(r"\b(nafgpu_[a-z0-9_]+)\s*\(", header)) - {"nafgpu_read_fn", "nafgpu_seek_fn"}
    assert declared == set(_ffi.EXPORTS)
    for name in declared:
        assert hasattr(lib.c, name), name
    assert lib.c.nafgpu_abi_version() == 2


def test_struct_layouts_match_header(lib):
    from nafcodec_amd import _ffi
    assert ctypes.sizeof(_ffi.Opts) == 32 and ctypes.sizeof(_ffi.Header) == 24
    assert ctypes.sizeof(_ffi.Field) == 24 and ctypes.sizeof(_ffi.Record) == 4 * 24 + 16
    assert ctypes.sizeof(_ffi.Error) == 16 + 192
    assert ctypes.sizeof(_ffi.ShardSummary) == 64          # gathered over the ranks as it is (nafcodec_amd/sharding.py)
    o = _ffi.Opts()
    lib.c.nafgpu_opts_default(ctypes.byref(o))   # DecoderBuilder::new(), mod.rs:67-76
    assert (o.id, o.comment, o.sequence, o.quality, o.mask, o.buffer_size, o.device, o.shard_count) == (1, 1, 1, 1, 1, 4096, -1, 1)
    lib.c.nafgpu_opts_from_flags(ctypes.byref(o), 0x01 | 0x20)   # from_flags(Id | Quality), mod.rs:93-101
    assert (o.id, o.comment, o.sequence, o.quality, o.mask) == (1, 0, 0, 1, 0)
    lib.c.nafgpu_opts_from_flags(ctypes.byref(o), 0x02)          # `id` is never switched off (App. D-2)
    assert (o.id, o.sequence) == (1, 1)


def test_open_parses_header_without_a_gpu(lib):
    """open = header + section table only (as the reference: no bulk work at open, mod.rs:169-256)"""
    from nafcodec_amd.decoder import Decoder
    import io
    d = Decoder(io.BytesIO(golden_bytes("LuxC.naf")), _lib=lib)
    assert (d.sequence_type, d.format_version, d.line_length, d.name_separator, d.number_of_sequences, len(d)) == \
           ("protein", "v2", 60, " ", 12, 12)
    with pytest.raises(EOFError):          # decoder/mod.rs:470-476 error_empty
        Decoder(io.BytesIO(b""), _lib=lib)
    with pytest.raises(ValueError):
        Decoder(io.BytesIO(b"\x01\xF9\xED\x01\x3E\x20\x3C\x20"), _lib=lib)
    with pytest.raises(FileNotFoundError):
        Decoder("", _lib=lib)
    with pytest.raises(IsADirectoryError):
        Decoder(ROOT, _lib=lib)


def test_no_gpu_means_a_loud_error_not_a_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from nafcodec_amd import NafError, _ffi
    from nafcodec_amd.decoder import Decoder
    import io
    d = Decoder(io.BytesIO(golden_bytes("phix.naf")), _lib=lib)
    with pytest.raises(NafError) as e:
        next(d)
    assert e.value.status == _ffi.E_DEVICE
    with pytest.raises(NafError) as e:
        lib.zstd_decompress(b"\x00\x48\x01\x00\x00", 0)
    assert e.value.status == _ffi.E_DEVICE


def test_product_does_not_reference_oracle_or_emu():
    pkg = os.path.join(ROOT, "nafcodec_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip", "Makefile")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("oracle/", "oracle/").lower() or f == "Makefile" or \
                    all("import" not in line and "include" not in line and "dlopen" not in line and "CDLL" not in line
                        for line in text.splitlines() if "oracle" in line.lower()), f
    out = subprocess.run(["ldd", os.path.join(pkg, "libnafgpu.so")], capture_output=True, text=True).stdout
    assert "naforacle" not in out and "zstd" not in out
