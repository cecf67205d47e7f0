"""nafgpu_open_io = DecoderBuilder::with_reader (mod.rs:169-256) through callbacks: the Python mirror hands a
file-like to the library the way PyFileRead does (pyfile.rs:88-187: readinto / read + seek).  With a working
seek only the header and the SELECTED sections are read -- the reference seeks over the others (mod.rs:228)."""
import io
import os
import subprocess

import numpy as np
import pytest

import cases
import naf_writer as nw
import zstd_ref
from conftest import ROOT, golden_bytes

EMU_DIR = os.path.join(ROOT, "tests", "emu", "_build")
CSRC = os.path.join(ROOT, "nafcodec_amd", "csrc")


class Spy(io.BytesIO):
    """a seekable reader that records what was asked of it"""

    def __init__(self, data):
        super().__init__(data)
        self.n_read, self.n_seeks = 0, 0

    def readinto(self, b):
        n = super().readinto(b)
        self.n_read += n
        return n

    def seek(self, off, whence=0):
        self.n_seeks += 1
        return super().seek(off, whence)


class ReadOnly:
    """`read` only: no readinto, no seek (pyfile.rs falls back to read(); the library drains it)"""

    def __init__(self, data):
        self._b = io.BytesIO(data)

    def read(self, n=-1):
        return self._b.read(n)


def big_archive():
    rng = np.random.default_rng(99)
    recs = cases.make_records(rng, [600000, 151, 0, 400001, 7], iupac=0.01, quality=True)
    return nw.write_naf(recs, quality=True, level=1, mask_runs=[1000, 50, 999109])


def check(lib):
    from nafcodec_amd.decoder import Decoder
    kw = {} if lib is None else {"_lib": lib}
    blob = big_archive()
    assert len(blob) > 5 * 65536                       # several lazy-load chunks
    want_all = cases.run_oracle(blob, {})
    # seekable reader: same records; everything selected -> everything read once
    f = Spy(blob)
    got = [tuple(getattr(r, k) for k in cases.FIELDS) for r in Decoder(f, **kw)]
    assert (got, None) == want_all and f.n_seeks >= 3 and f.n_read <= len(blob) + 65536
    # sequence and quality switched off: their payloads (almost the whole file) are never read
    f = Spy(blob)
    got = [tuple(getattr(r, k) for k in cases.FIELDS) for r in Decoder(f, sequence=False, quality=False, **kw)]
    assert (got, None) == cases.run_oracle(blob, {"sequence": False, "quality": False})
    assert f.n_read < len(blob) // 2, (f.n_read, len(blob))
    # the archive starts where the reader stands (fill_buf from the current position)
    f = Spy(b"not a naf header" + blob)
    f.read(16)
    assert len(list(Decoder(f, **kw))) == 5
    # read() only: drained front to back
    got = [tuple(getattr(r, k) for k in cases.FIELDS) for r in Decoder(ReadOnly(blob), **kw)]
    assert (got, None) == want_all
    # an exception raised by the file object comes back as it is

    class Broken(io.BytesIO):
        def readinto(self, b):
            raise PermissionError(13, "no reading today")

    with pytest.raises(PermissionError):
        Decoder(Broken(blob), **kw)
    # truncated / empty readers: the errors of the reference (mod.rs:180-185)
    with pytest.raises(EOFError):
        Decoder(io.BytesIO(b""), **kw)
    with pytest.raises(EOFError):
        Decoder(io.BytesIO(blob[:5]), **kw)
    for name in ("phix", "LuxC"):
        data = golden_bytes(name + ".naf")
        got = [tuple(getattr(r, k) for k in cases.FIELDS) for r in Decoder(Spy(data), **kw)]
        assert (got, None) == cases.run_oracle(data, {})


@pytest.mark.skipif(not zstd_ref.available(), reason="libzstd not loadable (cases are written with it)")
def test_open_io_on_the_cpu_harness():
    subprocess.check_call(["make", "-s", "-C", CSRC, "emu"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    from nafcodec_amd import _ffi
    check(_ffi.Library(os.path.join(EMU_DIR, "libnafgpu_emu.so")))


@pytest.mark.gpu
def test_open_io_on_the_gpu():
    check(None)
