"""Cross-checks the oracle's scalar zstd decoder (oracle/zstd_oracle.c) against the system
libzstd on the reference's fixture sections and on synthetic streams written at several levels.
(The reference's own zstd dependency is un-vendored -- nafcodec/Cargo.toml:16-18 -- so libzstd is
the closest runnable stand-in for it)"""
import numpy as np
import pytest

import zstd_ref
from oracle import oracle
from conftest import golden_bytes

pytestmark = pytest.mark.skipif(not zstd_ref.available(), reason="libzstd.so.1 not loadable")

FIXTURES = ["LuxC", "masked", "phix", "CP040672", "NZ_AAEN01000029"]


@pytest.mark.parametrize("name", FIXTURES)
def test_fixture_sections_match_libzstd(name):
    raw = golden_bytes(name + ".naf")
    d = oracle.Decoder(raw)
    n = 0
    for which in range(6):
        sec = d.section(which)
        if sec is None:
            continue
        data, orig, comp, off = sec
        assert zstd_ref.decompress_magicless(raw[off:off + comp], len(data) + 8) == data
        n += 1
    assert n >= 4


def packed_dna(rng, n, alphabet=(1, 2, 4, 8)):
    codes = np.array(alphabet, dtype=np.uint8)
    lo = codes[rng.integers(0, len(codes), n)]
    hi = codes[rng.integers(0, len(codes), n)]
    return (lo | (hi << 4)).astype(np.uint8).tobytes()


@pytest.mark.parametrize("level", [1, 3, 9, 19])
@pytest.mark.parametrize("n", [0, 1, 7, 255, 1000, 70000, 1 << 20])
def test_synthetic_dna_roundtrip(level, n):
    rng = np.random.default_rng(n * 31 + level)
    data = packed_dna(rng, n)
    for streaming in (True, False):
        payload = zstd_ref.compress_magicless(data, level, streaming)
        assert oracle.zstd_decode(payload, len(data)) == data


def test_repeats_and_rle_and_raw():
    rng = np.random.default_rng(7)
    unit = packed_dna(rng, 4000)
    data = (unit * 40) + bytes(5000) + b"\x11" * 300000 + rng.integers(0, 256, 200000, dtype=np.uint8).tobytes()
    for level in (1, 3, 19):
        for kw in ({}, {"checksum": True}, {"flush_every": 9999}, {"window_log": 17}):
            payload = zstd_ref.compress_magicless(data, level, True, **kw)
            out, st = oracle.zstd_decode(payload, len(data), stats=True)
            assert out == data
    assert st.sequences > 0


def test_text_per_record_flush():
    # the reference encoder flushes per record -> many tiny blocks, repeat-mode tables (App. D-10)
    recs = [b"@SRR%d some read\n" % i for i in range(3000)]
    data = b"".join(recs)
    repeats = 0
    for level in (1, 3, 9):
        payload = zstd_ref.compress_magicless(data, level, True, flush_every=100)
        out, st = oracle.zstd_decode(payload, len(data), stats=True)
        assert out == data and st.blocks > 400
        repeats += st.seq_mode_count[3] + st.seq_mode_count[7] + st.seq_mode_count[11]
    assert repeats > 0  # repeat-mode FSE tables exercised


def test_quality_like():
    rng = np.random.default_rng(3)
    q = rng.choice(np.frombuffer(b"#8CGGGGGGGG<AFFJJ", dtype=np.uint8), 500000).tobytes()
    for level in (1, 3, 12):
        payload = zstd_ref.compress_magicless(q, level, True)
        assert oracle.zstd_decode(payload, len(q)) == q


def test_corrupt_streams_error_not_crash():
    rng = np.random.default_rng(11)
    data = packed_dna(rng, 50000)
    payload = bytearray(zstd_ref.compress_magicless(data, 3, True))
    for cut in (1, 2, 5, len(payload) // 2, len(payload) - 1):
        with pytest.raises(oracle.OracleError):
            oracle.zstd_decode(bytes(payload[:cut]), len(data))
    bad = 0
    for k in range(40):
        p = bytearray(payload)
        p[int(rng.integers(0, len(p)))] ^= 1 << int(rng.integers(0, 8))
        try:
            out = oracle.zstd_decode(bytes(p), len(data))
            bad += out != data
        except oracle.OracleError:
            bad += 1
    assert bad > 0
