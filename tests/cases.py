"""Archive cases shared by the CPU-harness tests (tests/test_emu_parity.py) and the GPU parity
tests (tests/test_gpu_parity.py).  Every case is (name, archive bytes, decoder options); the
expected result is whatever the CPU oracle produces for the same bytes and options.

Covers the reference-test gaps listed in SURVEY.md section 4: masked runs reaching / crossing record
ends, runs of exactly 255*k, lengths >= 2^32-1, many-block frames, truncated streams,
zero-length records, RNA, title section."""
import numpy as np

import naf_writer as nw


def rand_dna(rng, n, alphabet="ACGT", iupac=0.0):
    s = rng.choice(list(alphabet), n)
    if iupac and n:
        k = rng.random(n) < iupac
        s[k] = rng.choice(list("NRYKMSWBDHV-"), int(k.sum()))
    return "".join(s)


def make_records(rng, lengths, alphabet="ACGT", iupac=0.0, quality=False):
    recs = []
    for i, n in enumerate(lengths):
        r = {"id": "rec%d" % i, "comment": "" if i % 3 == 0 else "comment number %d é" % i,
             "sequence": rand_dna(rng, n, alphabet, iupac)}
        if quality:
            r["quality"] = "".join(rng.choice(list("#8CGGGGGG<AFFJJ"), n))
        recs.append(r)
    return recs


def build_cases(scale=1):
    """scale=1: sizes the CPU harness decodes in seconds; larger scales for the GPU."""
    rng = np.random.default_rng(0x4E4146)
    cases = []

    def add(name, blob, **opts):
        cases.append((name, blob, opts))

    lens = [0, 1, 2, 151, 151, 0, 17, 21, 1000, 4097, 70001 * scale, 3, 0, 999]
    for level in (1, 3, 19):
        recs = make_records(rng, lens, iupac=0.01)
        add("dna_l%d" % level, nw.write_naf(recs, level=level))
    add("dna_l3_big", nw.write_naf(make_records(rng, [300000 * scale, 151, 500001 * scale]), level=3))
    add("dna_repeats_l1", nw.write_naf([{"id": "r", "sequence": rand_dna(rng, 5000) * (40 * scale)}], level=1))
    add("dna_homopolymer", nw.write_naf([{"id": "a", "sequence": "A" * (400001 * scale)},
                                         {"id": "n", "sequence": "N" * 70000}], level=3))
    add("rna_l3", nw.write_naf(make_records(rng, [33, 0, 100001, 7], alphabet="ACGU"), sequence_type="rna", level=3))
    add("protein_l3", nw.write_naf(make_records(rng, [488, 477, 0, 30001], alphabet="ACDEFGHIKLMNPQRSTVWY"),
                                   sequence_type="protein", level=3))
    add("text_quality", nw.write_naf(make_records(rng, [301] * 40 + [95, 0, 7], iupac=0.02, quality=True),
                                     quality=True, level=3, line_length=301))
    add("fastq_flush_per_record", nw.write_naf(make_records(rng, [151] * 300, quality=True), quality=True, level=3,
                                               zstd_kwargs={"flush_every": 151}))
    # dense short-offset match chains (what real quality strings look like to zstd): deeper than any
    # fixed number of match passes, resolved by the pointer-jumping stage
    dense = "".join(rng.choice(list("GGGGGGGJJJJF#"), 60000 * scale))
    add("text_dense_chains", nw.write_naf([{"id": "q", "sequence": dense}], sequence_type="text", level=3))
    add("dna_dense_chains", nw.write_naf([{"id": "d", "sequence": "".join(rng.choice(list("AAAAAAAT"), 150001 * scale))}], level=3))
    # repeat offsets, densely: a 211-character motif copied over and over with a substitution every ~17 characters and a
    # deletion now and then -- almost every sequence of the blocks is "same offset as before" (or the one before that, or
    # that minus one after a literal-free match), thousands in a row, so k_seq_values' maps stay open across its lanes
    # and tiles and the history has to be carried through both
    motif = rng.choice(list("ACGTRYKMSWBDHVN"), 211)
    rep = np.tile(motif, 1400 * scale)
    hits = np.flatnonzero(rng.random(rep.size) < 1 / 17)
    rep[hits] = rng.choice(list("ACGT"), hits.size)
    rep = np.delete(rep, np.flatnonzero(rng.random(rep.size) < 1 / 900))
    for lvl in (1, 3, 9):
        add("text_repeat_offsets_l%d" % lvl, nw.write_naf([{"id": "r", "sequence": "".join(rep)}], sequence_type="text", level=lvl))
    add("dna_repeat_offsets", nw.write_naf([{"id": "r", "sequence": "".join(rep)}], level=3))
    add("title_and_extended", nw.write_naf(make_records(rng, [10, 20, 30]), title="a title ✓", extended=True))
    add("v2_dna", nw.write_naf(make_records(rng, [100, 101]), version=2))
    add("no_ids_no_comments", nw.write_naf(make_records(rng, [5, 6, 7]), ids=False, comments=False))
    add("lengths_only_big", nw.write_naf([{"id": "x"}], sequence=False, comments=False,
                                         raw_sections={"lengths": nw.length_words([0xFFFFFFFF + 5, 7, 0xFFFFFFFF * 2])},
                                         number_of_sequences=3))
    add("more_records_than_lengths", nw.write_naf(make_records(rng, [4, 5]), number_of_sequences=4))
    add("fewer_records_than_lengths", nw.write_naf(make_records(rng, [4, 5, 6]), number_of_sequences=2))
    add("checksum_frames", nw.write_naf(make_records(rng, [50000, 3]), level=3, zstd_kwargs={"checksum": True}))
    # Content_Checksum over several blocks: nucleotides (verified on the characters the section was expanded to), text
    # with matches; and one whose stored checksum is wrong although everything decodes (libzstd refuses the frame)
    ck_dna = nw.write_naf(make_records(rng, [700000 * scale + 1, 777], iupac=0.01), level=1, zstd_kwargs={"checksum": True})
    add("checksum_dna_blocks", ck_dna)
    add("checksum_text_l3", nw.write_naf([{"id": "q", "sequence": "".join(rng.choice(list("GGGGGGGJJJJF#"), 300000 * scale + 13))}],
                                         sequence_type="text", level=3, zstd_kwargs={"checksum": True}))
    bad = bytearray(ck_dna)
    bad[-2] ^= 0x04                                                        # (the sequence section is the last one: its checksum ends the file)
    add("checksum_wrong", bytes(bad))
    # number_of_sequences is an untrusted varint: it must not size anything (ids/comment tables, scan scratch);
    # the iterator keeps yielding (empty) records until it is reached, so only the first few are compared
    huge = make_records(rng, [40, 50, 60])
    add("huge_nseq_2p40", nw.write_naf(huge, number_of_sequences=1 << 40), _limit=6)
    add("huge_nseq_u64max", nw.write_naf(huge, number_of_sequences=(1 << 64) - 1), _limit=6)
    add("huge_nseq_no_names", nw.write_naf(huge, number_of_sequences=(1 << 64) - 1, ids=False, comments=False), _limit=6)
    add("title_bad_utf8", nw.write_naf(huge, title=b"caf\xe9 \xff"))              # from_utf8 fails: Nom(MapRes), parser.rs:133-137

    # ---- Huffman table formats of k_huf_decode (plan.h: HufTblKind) x segment-aware variant -------------------------
    # several trees with a small joint alphabet -> dictionary tables; rare IUPAC codes -> escape sub-tables; a few far
    # matches -> blocks with a handful of sequences, whose literals go to their final positions segment by segment
    def skewed(n, probs, extra="", pe=0.0):
        t = rng.choice(list("ACGT"), n, p=probs)
        if pe:
            k = rng.random(n) < pe
            t[k] = rng.choice(list(extra), int(k.sum()))
        return "".join(t)
    parts = []
    for i in range(4 + 2 * scale):
        p = np.array([0.1 + 0.1 * (i % 6), 0.4 - 0.05 * (i % 6), 0.3 - 0.03 * (i % 6), 0])
        p[3] = 1 - p[:3].sum()
        parts.append(skewed(262144 + i * 7, p, "NRY", 0.0015))
    sk = "".join(parts)
    sk = sk[:900000] + sk[1000:1500] + sk[900000:] + sk[5000:5300]
    add("dna_skewed_blocks_dict_seg", nw.write_naf([{"id": "a", "sequence": sk}, {"id": "b", "sequence": sk[77:77 + 151]}], level=1))
    # two dozen rare symbols with codes longer than the table index: an escape in almost every round of a lane (a round
    # that resolves one must still produce no more than the row holds)
    dense = "".join(skewed(262144 + i, np.array([0.4 - 0.1 * i, 0.2, 0.1 + 0.1 * i, 0.3]), "NRY", 0.012) for i in range(3))
    add("dna_dense_escapes_dict", nw.write_naf([{"id": "e", "sequence": dense}], level=1))
    dense_dna = dense
    # every IUPAC code in use: more than 64 byte values between the trees -> compact tables
    add("dna_multi_tree_compact", nw.write_naf(make_records(rng, [700001 * scale], iupac=0.05), level=1))
    # text with ~40 symbols, several blocks: dictionary tables without the ASCII expansion
    qa = list("!\"#$%&'()*+,-./0123456789:;<=>?@ABCDEFGHIJ")
    qtext = "".join(rng.choice(qa, 500000 * scale, p=np.r_[np.full(10, 0.002), np.full(32, (1 - 0.02) / 32)]))
    add("text_multi_tree_dict", nw.write_naf([{"id": "q", "sequence": qtext}], sequence_type="text", level=1))

    # ---- mask ------------------------------------------------------------------------------
    recs = make_records(rng, [1550, 1800, 0, 700, 255, 510, 1000])
    total = sum(len(r["sequence"]) for r in recs)
    inside = [657, 19, 635, 39, 725, 96, 99, 13]                     # masked.naf-like, runs inside records
    inside.append(total - sum(inside))
    add("mask_inside_records", nw.write_naf(recs, mask_runs=inside))
    crossing = [1500, 100, 1700, 50, 0, 0, 700, 255, 255, 510, 100]  # runs reaching / crossing record ends
    crossing.append(total - sum(crossing))
    add("mask_crossing_record_ends", nw.write_naf(recs, mask_runs=crossing))                 # App. D-1 quirk
    add("mask_crossing_spec", nw.write_naf(recs, mask_runs=crossing), spec_mask=True)
    add("mask_exact_255k", nw.write_naf(recs, mask_runs=[255, 255, 510, 765, 0, 1020, total - 2805]))
    add("mask_all_masked_from_0", nw.write_naf(recs, mask_runs=[0, total]))
    add("mask_short_stream", nw.write_naf(recs, mask_runs=[1000, 500]))                      # runs end early: error
    add("mask_overshoot", nw.write_naf(recs, mask_runs=[total - 10, 500]))
    add("mask_off", nw.write_naf(recs, mask_runs=inside), mask=False)
    add("mask_no_sequence", nw.write_naf(recs, mask_runs=inside), sequence=False)
    add("mask_on_protein", nw.write_naf(make_records(rng, [100, 200], alphabet="ACDEFGHIKLMNPQRSTVWY"),
                                        sequence_type="protein", mask_runs=[50, 30, 40, 100, 80]))
    # many short runs: several masked runs share one 16-byte chunk (k_mask_apply's atomic edge path), runs of
    # length 0, 1 and 15-17 around chunk boundaries, > 256 masked runs (more than one workgroup batch)
    dense_recs = make_records(rng, [5000 * scale + 7, 333, 12000])
    dt = sum(len(r["sequence"]) for r in dense_recs)
    dense, left = [], dt
    pat = [3, 2, 1, 3, 0, 5, 9, 1, 1, 1, 2, 16, 15, 17, 1, 31, 33, 4, 0, 0, 7, 64, 5, 100]
    while left > 400:
        for v in pat:
            dense.append(v)
            left -= v
    dense.append(left)
    add("mask_dense_short_runs", nw.write_naf(dense_recs, mask_runs=dense))
    add("mask_dense_short_runs_spec", nw.write_naf(dense_recs, mask_runs=dense), spec_mask=True)
    long_run = make_records(rng, [70000 * scale + 3, 40000])
    lt = sum(len(r["sequence"]) for r in long_run)
    add("mask_run_gt_65535", nw.write_naf(long_run, mask_runs=[10, 66000, lt - 66010]))

    # the mask applied by the sequence's own writers (sections without LZ sequences: ArchiveJob::decode builds a bit map
    # and k_huf_decode / k_copy_fill OR it in): each table format of K1, raw and RLE blocks, several blocks and tasks,
    # runs of every length against unit and block boundaries; and a sequence WITH matches, which takes the separate pass
    def random_runs(total, mean):
        runs, left = [], total
        while left > 0:
            v = int(min(left, rng.integers(0, mean) if rng.random() < 0.7 else rng.integers(0, 40 * mean)))
            runs.append(v)
            left -= v
        return runs
    for nm, seq in (("mask_dict_tables", dense_dna),
                    ("mask_compact_tables", rand_dna(rng, 700001 * scale, "ACGT", 0.05)),
                    ("mask_raw_rle_blocks", "".join(rng.choice(list(nw.NUC), 262144)) + "C" * (524288 * scale) +     # raw and RLE blocks only
                                            "".join(rng.choice(list(nw.NUC), 262144))),
                    ("mask_with_matches", rand_dna(rng, 5000) * (40 * scale))):
        cut = len(seq) // 3
        recs_m = [{"id": "m0", "sequence": seq[:cut]}, {"id": "m1", "sequence": seq[cut:cut + 151]}, {"id": "m2", "sequence": seq[cut + 151:]}]
        add(nm, nw.write_naf(recs_m, level=1, mask_runs=random_runs(len(seq), 60)))
    add("mask_dict_tables_spec", nw.write_naf([{"id": "e", "sequence": dense_dna}], level=1, mask_runs=random_runs(len(dense_dna), 700)), spec_mask=True)

    # ---- field selection ---------------------------------------------------------------------
    fq = nw.write_naf(make_records(rng, [151] * 9, quality=True), quality=True, mask_runs=[100, 20, 151 * 9 - 120])
    for off in ("id", "comment", "sequence", "quality", "mask"):
        add("fastq_no_" + off, fq, **{off: False})

    # ---- malformed ---------------------------------------------------------------------------
    good = nw.write_naf(make_records(rng, [3000, 2000]), level=3)
    add("truncated_tail", good[:-7])
    add("truncated_mid", good[:len(good) // 2])
    add("truncated_header", good[:6])
    add("bad_magic", b"\x01\xF9\xED" + good[3:])
    add("bad_version", good[:3] + b"\x07" + good[4:])
    add("bad_separator", good[:5] + b"\x07" + good[6:])
    flip = bytearray(good)
    flip[len(flip) - 40] ^= 0x10
    add("bitflip_sequence", bytes(flip))
    add("empty", b"")
    add("lengths_exceed_sequence", nw.write_naf(make_records(rng, [10, 20]),
                                                raw_sections={"lengths": nw.length_words([10, 25])}))
    return cases


FIELDS = ("id", "comment", "sequence", "quality", "length")


def run_oracle(blob, opts):
    """-> (records, error) where error is None or a normalised kind string"""
    from oracle import oracle
    kinds = {oracle.E_IO_EOF: "io:eof", oracle.E_IO_INVALID: "io:invalid", oracle.E_NOM: "nom", oracle.E_PANIC: "panic"}
    recs = []
    opts = dict(opts)
    limit = opts.pop("_limit", None)
    try:
        d = oracle.Decoder(blob, **opts)
        for r in d:
            recs.append(tuple(getattr(r, f) for f in FIELDS))
            if limit is not None and len(recs) >= limit:
                break
    except oracle.OracleError as e:
        return recs, kinds.get(e.kind, "other")
    except UnicodeDecodeError:
        return recs, "panic"
    return recs, None


def run_product(blob, opts, lib=None):
    import io
    from nafcodec_amd import _ffi
    from nafcodec_amd.decoder import Decoder
    recs = []
    kw = dict(opts)
    limit = kw.pop("_limit", None)
    if lib is not None:
        kw["_lib"] = lib
    try:
        d = Decoder(io.BytesIO(blob), **kw)
        for r in d:
            recs.append(tuple(getattr(r, f) for f in FIELDS))
            if limit is not None and len(recs) >= limit:
                break
    except EOFError:
        return recs, "io:eof"
    except ValueError:
        return recs, "nom"
    except _ffi.NafError as e:
        return recs, {_ffi.E_PANIC: "panic", _ffi.E_IO: "io:invalid"}.get(e.status, "other:%d" % e.status)
    except OSError:
        return recs, "io:invalid"
    return recs, None


def fuzz_cases(seed=1, n=120):
    """Corrupted variants (bit flips, byte overwrites, truncations) of small valid archives.
    Every decoder must terminate on them; where both the oracle and the product succeed they must
    agree (error *detection* may differ in strictness, the data may not)."""
    from conftest import golden_bytes
    rng = np.random.default_rng(seed)
    seeds = [golden_bytes(name + ".naf") for name in ("phix", "masked", "CP040672", "LuxC")]
    recs = make_records(rng, [3000, 2000, 151, 0, 77], iupac=0.02, quality=True)
    seeds.append(nw.write_naf(recs, level=3, quality=True, mask_runs=[100, 50, 5000, 20, 58]))
    seeds.append(nw.write_naf(make_records(rng, [40000]), level=1))
    dense = "".join(rng.choice(list("GGGGGGGJJJJF#"), 30000))
    seeds.append(nw.write_naf([{"id": "q", "sequence": dense}], sequence_type="text", level=3))
    # three 128 KiB blocks with their own deep Huffman trees: compact decode tables with escape sub-tables
    seeds.append(nw.write_naf(make_records(rng, [700000, 151], iupac=0.05), level=1))
    # an archive of this repository's own Encoder: sequences coded with the PREDEFINED FSE tables, fresh offsets only
    import io as _io
    from nafcodec_amd import Encoder, Record
    buf = _io.BytesIO()
    motif = rand_dna(rng, 700)
    with Encoder(buf, "dna", id=True, comment=True, sequence=True, quality=True) as enc:
        for i in range(9):
            seq = motif * (2 + i % 3) + rand_dna(rng, 300 * i)
            enc.write(Record(id="e%d" % i, comment="c", sequence=seq, quality=("IIHHG#" * len(seq))[:len(seq)]))
    seeds.append(buf.getvalue())
    out = []
    for it in range(n):
        blob = bytearray(seeds[it % len(seeds)])
        for _ in range(int(rng.integers(1, 4))):
            mode = int(rng.integers(0, 3))
            pos = int(rng.integers(0, len(blob)))
            if mode == 0:
                blob[pos] ^= 1 << int(rng.integers(0, 8))
            elif mode == 1:
                blob[pos] = int(rng.integers(0, 256))
            else:
                blob = blob[:pos]
            if not blob:
                break
        out.append(("fuzz%d" % it, bytes(blob), {}))
    return out


def fuzz_disagreements(cases_list, lib=None):
    """Names of the cases on which product and checker disagree: both succeed with different records, both fail with
    errors of different kinds or after different records, or ONE of them fails -- an archive the checker refuses and the
    product accepts (or the other way round) is a red test, not a difference in strictness.  The single named exception:
    an archive that is both cut short and corrupt may be Io(UnexpectedEof) to one side and Io(InvalidData) to the other."""
    bad = []
    for name, blob, opts in cases_list:
        got, want = run_product(blob, opts, lib), run_oracle(blob, opts)
        if got[1] is None and want[1] is None:
            if got != want:
                bad.append(name)
        elif got[1] is not None and want[1] is not None:
            # Both fail: the same records in front of the error, and the same KIND of error -- except that an archive which
            # is both cut short and corrupt may be Io(UnexpectedEof) to one and Io(InvalidData) to the other (the product's
            # host walk sees every block header before the device decodes a byte; the checker decodes front to back).
            kinds = {got[1], want[1]}
            if got[0] != want[0] or (len(kinds) > 1 and kinds != {"io:eof", "io:invalid"}):
                bad.append("%s: %s after %d records, checker %s after %d" % (name, got[1], len(got[0]), want[1], len(want[0])))
        else:
            bad.append("%s: one-sided -- product %s after %d records, checker %s after %d"
                       % (name, got[1] or "ok", len(got[0]), want[1] or "ok", len(want[0])))
    return bad


def error_timing(blob, opts, lib=None, eager=True, slack=0):
    """An archive that is malformed somewhere inside a section, three ways:
      * the reference's timing (oracle/ref_shape.c, streaming like mod.rs:356-399 over :221-223): k records, then the error;
      * the product through the record iterator: it decodes a section whole (or tile by tile) when the first record needs
        it, so the error comes at that record -- DESIGN.md section 8 -- i.e. after j <= k records, and those j are the
        streaming pipeline's first j;
      * the eager checker (oracle/naf_oracle.c), which the product equals exactly.
    Returns (k, streaming rc, j, product error kind)."""
    from nafcodec_amd import _ffi
    from oracle import oracle
    L = lib or _ffi.default()
    rc, st = oracle.ref_shape_stream(blob)
    got, err = run_product(blob, opts, lib)
    if eager:                                              # (decoded in tiles, the product gets further than the eager checker)
        assert (got, err) == run_oracle(blob, opts)
    assert (rc != 0) == (err is not None), (rc, err)
    # (slack: libzstd's streaming decoder meets the bad block while it fills the 4 KiB request that ENDS the block in front of
    #  it, and the reference loses that request's bytes with the error -- a record that ends in those 4 KiB is one this
    #  library, which fails tile by tile, still hands out)
    assert len(got) <= st.n_records + slack, (len(got), st.n_records)
    # what was handed out is the streaming pipeline's first records (its checksums over exactly that many)
    m = min(len(got), st.n_records)
    rc2, first = oracle.ref_shape_stream(blob, limit=m)
    seq = "".join(r[2] or "" for r in got[:m]).encode()
    qual = "".join(r[3] or "" for r in got[:m]).encode()
    assert first.n_records == m and first.n_bases == len(seq)
    assert first.seq_hash == L.c.nafgpu_hash64_host(seq, len(seq)) and first.qual_hash == L.c.nafgpu_hash64_host(qual, len(qual))
    return st.n_records, rc, len(got), err


def check_next_batch(blob, opts=None, lib=None, caps=(1, 3, 64, 4096)):
    """nafgpu_next_batch against nafgpu_next on the same archive: the same records in the same order, the same error at
    the same place (returned by the call that reaches it, with the records in front of it delivered), the iterator not
    fused by it, NAFGPU_END only once nothing is left -- for several batch sizes."""
    import ctypes
    import io
    from nafcodec_amd import _ffi
    from nafcodec_amd.decoder import Decoder
    opts = opts or {}
    kw = {} if lib is None else {"_lib": lib}
    L = lib or _ffi.default()

    def fields(r):
        def t(f):
            return ctypes.string_at(f.ptr, f.len) if f.present else None
        return (t(r.id), t(r.comment), t(r.sequence), t(r.quality), r.length if r.has_length else None)

    def events_single():
        try:
            dec = Decoder(io.BytesIO(blob), **opts, **kw)
        except Exception as e:                                  # noqa: BLE001 -- open errors are the same object either way
            return [("open", type(e).__name__)]
        ev, rec = [], _ffi.Record()
        for _ in range(len(dec) + 3):
            rc = L.c.nafgpu_next(dec._h, ctypes.byref(rec))
            if rc == _ffi.END:
                break
            ev.append(fields(rec) if rc == _ffi.OK else ("error", rc))
        dec.close()
        return ev

    want = events_single()
    for cap in caps:
        try:
            dec = Decoder(io.BytesIO(blob), **opts, **kw)
        except Exception as e:                                  # noqa: BLE001
            assert want == [("open", type(e).__name__)]
            continue
        ev, recs, got = [], (_ffi.Record * cap)(), ctypes.c_uint64()
        ended = False
        while len(ev) < len(want) + 1 and not ended:           # (an archive may fail on every call for ever: as many events as the single calls saw)
            rc = L.c.nafgpu_next_batch(dec._h, recs, cap, ctypes.byref(got))
            ev += [fields(r) for r in recs[:got.value]]
            if rc == _ffi.END:
                assert got.value == 0
                ended = True
            elif rc != _ffi.OK:
                ev.append(("error", rc))
            else:
                assert 1 <= got.value <= cap
        dec.close()
        assert ev[:len(want)] == want and (ended or len(ev) > len(want) or not want), (cap, len(ev), len(want))
        if ended:
            assert len(ev) == len(want), (cap, len(ev), len(want))
    return len(want)


def check_archive_ends(lib, n_bases, seed, n_blk):
    """A synthetic DNA-only archive of any size: its first and its last `n_blk` zstd blocks (the tail cut at one of the
    writer's 64-block units, where a block brings its own Huffman tree) are re-framed as small archives of ONE record each,
    the CPU oracle decodes those, and the bytes the product wrote for the WHOLE archive at those positions must be the
    same -- so that a full-size decode (positions beyond 2^35 for the tail of the 40-Gbase archive) is not only checked
    against the writer's own checksums.  The archive is opened by path (mapped, as bench.py's iterator leg does)."""
    import ctypes
    import os
    import shutil
    import tempfile
    from nafcodec_amd import _ffi
    from nafcodec_amd.decoder import Decoder
    from oracle import oracle
    L = lib or _ffi.default()
    kw = {} if lib is None else {"_lib": lib}
    arc = L.synth(n_bases, seed=seed)
    path = os.path.join("/dev/shm" if shutil.disk_usage("/dev/shm").free > 2 * arc.n else tempfile.gettempdir(), "nafgpu_ends_%d.naf" % os.getpid())
    try:
        mv = memoryview((ctypes.c_uint8 * arc.n).from_address(arc.bytes)).cast("B")
        # container walk (parser.rs:101-123, mod.rs:199-233): header, Length section, then the Sequence section's frame
        at = 6

        def varint():
            nonlocal at
            v = 0
            while True:
                b = mv[at]
                at += 1
                v = (v << 7) | (b & 0x7F)
                if not b & 0x80:
                    return v
        varint()
        n_rec = varint()
        _len_orig, len_comp = varint(), varint()
        at += len_comp
        assert varint() == n_bases
        seq_comp = varint()
        frame0, frame1 = at, at + seq_comp
        assert frame1 == arc.n and mv[frame0] == 0x00           # FHD 0: a window descriptor follows, no content size
        # block directory: 3-byte headers {last, type, size} (RFC 8878 3.1.1.2)
        offs, pos = [], frame0 + 2
        while pos < frame1:
            h = mv[pos] | (mv[pos + 1] << 8) | (mv[pos + 2] << 16)
            assert (h >> 1) & 3 == 2                            # compressed blocks only
            offs.append(pos)
            pos += 3 + (h >> 3)
            if h & 1:
                break
        blk = 131072
        assert pos == frame1 and len(offs) == ((n_bases + 1) // 2 + blk - 1) // blk and len(offs) > n_blk

        def small_archive(b0, b1, bases):
            body = bytearray(mv[offs[b0]:(offs[b1] if b1 < len(offs) else frame1)])
            body[offs[b1 - 1] - offs[b0]] |= 1                  # the last block of the cut closes the frame
            frame = bytes(mv[frame0:frame0 + 2]) + bytes(body)
            lens = nw.length_words([bases])
            len_frame = bytes([0x20, len(lens), (len(lens) << 3) | 1, 0, 0]) + lens      # single-segment frame, one raw block
            return (bytes([1, 0xF9, 0xEC, 1, 0x0A, 0x20]) + nw.varint(60) + nw.varint(1) + nw.varint(len(lens)) + nw.varint(len(len_frame)) +
                    len_frame + nw.varint(bases) + nw.varint(len(frame)) + frame)

        with open(path, "wb") as f:
            f.write(mv)
        first_tail_blk = (len(offs) - n_blk) // 64 * 64         # a 64-block unit of the writer begins here
        tail_base = first_tail_blk * 2 * blk
        dec = Decoder(path, **kw)
        res = dec.decode_all_device()
        assert (res.n_bases, res.n_records) == (n_bases, arc.n_records) and n_rec == arc.n_records
        for b0, b1, base0, n in ((0, n_blk, 0, n_blk * 2 * blk), (first_tail_blk, len(offs), tail_base, n_bases - tail_base)):
            want = oracle.Decoder(small_archive(b0, b1, n)).drain()
            assert (want.n_bases, want.n_records) == (n, 1)
            got = dec.copy_to_host(res.d_sequence + base0, n)
            assert L.c.nafgpu_hash64_host(got, n) == want.seq_hash, (b0, b1)
            del got
        dec.close()
    finally:
        L.c.nafgpu_synth_free(ctypes.byref(arc))
        if os.path.exists(path):
            os.unlink(path)


def check_sharding(lib, n_bases, mask, worlds=(2, 3, 5), seed=5):
    """Block-range sharding of ONE archive (nafgpu_opts.shard_rank/shard_count): the shards tile the
    base stream, their checksums add up to the whole archive's, and first_record / carry place each
    shard in the global record table."""
    import ctypes
    import io
    from nafcodec_amd.decoder import Decoder
    kw = {} if lib is None else {"_lib": lib}
    from nafcodec_amd import _ffi
    L = lib or _ffi.default()
    arc = L.synth(n_bases, seed=seed, with_mask=mask)
    try:
        blob = ctypes.string_at(arc.bytes, arc.n)
        ends, pos = [], 0
        for r in run_oracle(blob, {"sequence": False, "mask": False})[0]:
            pos += r[4]
            ends.append(pos)
        starts = [0] + ends[:-1]
        from nafcodec_amd.sharding import decode_sharded_local
        for world in worlds:
            for protocol in (False, True):           # decode_all_device per shard; the shard protocol (the same ranges, nothing to exchange)
                nxt, hsum = 0, 0
                decs = [Decoder(io.BytesIO(blob), shard_rank=rank, shard_count=world, shard_protocol=protocol, **kw) for rank in range(world)]
                results = decode_sharded_local(decs) if protocol else [d.decode_all_device() for d in decs]
                for d, res in zip(decs, results):
                    assert res.sharded == 1 and res.base_offset == nxt
                    assert res.n_bases == 0 or res.base_offset % 4096 == 0
                    hsum = (hsum + d.hash_device(res.d_sequence, res.n_bases, res.base_offset // 4096)) % (1 << 64)
                    want_first = sum(1 for st in starts if st < res.base_offset)
                    assert res.first_record == want_first, (world, res.first_record, want_first)
                    assert res.carry == (0 if res.base_offset in starts or res.base_offset in (0, pos) else 1)
                    assert res.n_records == len(ends)
                    nxt = res.base_offset + res.n_bases
                    d.close()
                assert nxt == arc.n_bases and hsum == arc.seq_hash, (world, protocol, nxt, hsum)
    finally:
        L.c.nafgpu_synth_free(ctypes.byref(arc))


def raw_naf(packed=None, n_bases=0, text=None, lens=(), quality=None, level=1, frames=1, line_length=60):
    """A NAF v1 DNA (packed: numpy uint8 array of 4-bit pairs) or v2 text archive put together from ready-made section
    contents, the large sections written by libzstd in `frames` frames back to back (a frame ends with a short block and
    the next one starts from fresh repeat offsets and an empty window: what the shard protocol has to get right)."""
    import numpy as np
    import zstd_ref

    def comp(data, lvl=level, n_frames=1):
        data = bytes(data)
        if n_frames <= 1:
            return zstd_ref.compress_magicless(data, lvl, True)
        cut = [len(data) * k // n_frames for k in range(n_frames + 1)]
        return b"".join(zstd_ref.compress_magicless(data[a:b], lvl, True) for a, b in zip(cut, cut[1:]))

    words = nw.length_words(lens)
    if packed is not None:
        head = bytes([1, 0xF9, 0xEC, 1])
        flags = 0x0A | (0x01 if quality is not None else 0)
        secs = [(len(words), comp(words, 1)), (n_bases, comp(packed, level, frames))]
    else:
        head = bytes([1, 0xF9, 0xEC, 2, 3])
        flags = 0x0A | (0x01 if quality is not None else 0)
        secs = [(len(words), comp(words, 1)), (len(text), comp(text, level, frames))]
    if quality is not None:
        secs.append((len(quality), comp(quality, level, frames)))
    blob = bytearray(head) + bytes([flags, 0x20]) + nw.varint(line_length) + nw.varint(len(lens))
    for orig, payload in secs:
        blob += nw.varint(orig) + nw.varint(len(payload)) + payload
    return bytes(blob)


def lz_shard_archives(scale=1):
    """(name, archive bytes, expected sequence bytes, expected quality bytes or None, record lengths) -- archives whose
    sections hold LZ sequences, for the shard protocol: the statistics of a real genome (the reference's fixture tiled,
    libzstd level 1: a few far matches per block), level-3 DNA (a quarter of the bytes are matches at random distances:
    swept), FASTQ-like reads with qualities (both sections swept, the Length section one long chain), the same in
    several frames, and text with dense short-offset chains."""
    import numpy as np
    from conftest import golden_bytes
    from oracle import oracle
    rng = np.random.default_rng(99)
    lut = np.frombuffer(nw.NUC.encode(), dtype=np.uint8)
    code = np.zeros(256, dtype=np.uint8)
    for i, c in enumerate(nw.NUC.encode()):
        code[c] = i
    out = []

    def dna(name, packed, level, frames=1, lens=None, quality=None):
        n_bases = 2 * len(packed)
        want = np.empty(n_bases, dtype=np.uint8)
        want[0::2] = lut[packed & 15]
        want[1::2] = lut[packed >> 4]
        lens = [n_bases] if lens is None else lens
        out.append((name, raw_naf(packed=packed.tobytes(), n_bases=n_bases, lens=lens, quality=quality, level=level, frames=frames),
                    want.tobytes(), quality, lens))

    fixture = "".join(r.sequence.upper() for r in oracle.Decoder(golden_bytes("NZ_AAEN01000029.naf"))).encode()
    nib = code[np.frombuffer(fixture, dtype=np.uint8)]
    one = (nib[0:len(nib) & ~1:2] | (nib[1::2] << 4)).astype(np.uint8)
    dna("real_genome_l1", np.tile(one[:len(one) // (4 if scale == 1 else 1)], 2 * scale), 1)
    codes = np.array([1, 2, 4, 8], dtype=np.uint8)
    n = 600_000 * scale
    dna("random_dna_l3", (codes[rng.integers(0, 4, n)] | (codes[rng.integers(0, 4, n)] << 4)).astype(np.uint8), 3)
    dna("random_dna_l3_frames", (codes[rng.integers(0, 4, n)] | (codes[rng.integers(0, 4, n)] << 4)).astype(np.uint8), 3, frames=3)
    n_reads = 1500 * scale
    nb = n_reads * 151 + (n_reads * 151 & 1)
    nibs = codes[rng.integers(0, 4, nb)]
    qalpha = np.frombuffer(b"#8CGGGGGGGGGG<AFFFJJJJJJJJJJJJJJ", dtype=np.uint8)
    for lvl, frames in ((1, 1), (3, 2)):
        qual = qalpha[rng.integers(0, len(qalpha), n_reads * 151)].tobytes()
        packed = (nibs[0::2] | (nibs[1::2] << 4)).astype(np.uint8)
        n_bases = n_reads * 151
        want = np.empty(2 * len(packed), dtype=np.uint8)
        want[0::2] = lut[packed & 15]
        want[1::2] = lut[packed >> 4]
        out.append(("fastq_like_l%d" % lvl, raw_naf(packed=packed.tobytes(), n_bases=n_bases, lens=[151] * n_reads, quality=qual, level=lvl, frames=frames),
                    want.tobytes()[:n_bases], qual, [151] * n_reads))
    dense = bytes(rng.choice(np.frombuffer(b"GGGGGGGJJJJF#", dtype=np.uint8), 400_000 * scale))
    out.append(("text_dense_chains_frames", raw_naf(text=dense, lens=[len(dense)], level=3, frames=2), dense, None, [len(dense)]))
    return out


def check_lz_sharding(lib, scale=1, worlds=(2, 3, 8), names=None, force_modes=(None,)):
    """The shard protocol (nafgpu_shard_*, nafcodec_amd.sharding.decode_sharded_local: every rank in this process) on
    archives whose sections hold LZ sequences: every rank reports sharded = 1, the shards tile the Sequence and the Quality
    sections, put together they are what the archive decodes to (and what the oracle says it decodes to), and first_record /
    carry place every shard in the record table."""
    import io
    import os
    from nafcodec_amd import _ffi
    from nafcodec_amd.decoder import Decoder
    from nafcodec_amd.sharding import decode_sharded_local
    from oracle import oracle
    L = lib or _ffi.default()
    kw = {} if lib is None else {"_lib": lib}
    for name, blob, want_seq, want_qual, lens in lz_shard_archives(scale):
        if names is not None and name not in names:
            continue
        chk = oracle.Decoder(blob).drain(want_hash=True)                      # the checker's word on the whole archive
        assert chk.n_bases == len(want_seq) and chk.seq_hash == L.c.nafgpu_hash64_host(want_seq, len(want_seq)), name
        if want_qual is not None:
            assert chk.qual_hash == L.c.nafgpu_hash64_host(want_qual, len(want_qual)), name
        ends, pos = [], 0
        for n in lens:
            pos += n
            ends.append(pos)
        starts = [0] + ends[:-1]
        for mode in force_modes:
            if mode is not None:
                L.c.nafgpu_test_hooks(1)
                os.environ["NAFGPU_LZ_MODE"] = mode
            try:
                for world in worlds:
                    decs = [Decoder(io.BytesIO(blob), shard_rank=r, shard_count=world, shard_protocol=True, **kw) for r in range(world)]
                    res = decode_sharded_local(decs)
                    nxt = qnxt = 0
                    hsum, aligned = 0, True
                    for d, x in zip(decs, res):
                        assert x.sharded == 1 and x.base_offset == nxt and x.quality_offset == qnxt, (name, world, mode)
                        got = d.copy_to_host(x.d_sequence, x.n_bases)
                        assert got == want_seq[nxt:nxt + x.n_bases], (name, world, mode, "sequence of rank", decs.index(d))
                        if want_qual is not None:
                            assert d.copy_to_host(x.d_quality, x.n_quality) == want_qual[qnxt:qnxt + x.n_quality], (name, world, mode, "quality")
                        aligned = aligned and x.base_offset % 4096 == 0
                        if aligned:
                            hsum = (hsum + d.hash_device(x.d_sequence, x.n_bases, x.base_offset // 4096)) % (1 << 64)
                        assert x.first_record == sum(1 for st in starts if st < x.base_offset), (name, world)
                        assert x.carry == (0 if x.base_offset in starts or x.base_offset in (0, pos) else 1), (name, world)
                        assert x.n_records == len(lens)
                        nxt += x.n_bases
                        qnxt += x.n_quality
                    assert nxt == len(want_seq) and qnxt == (len(want_qual) if want_qual is not None else 0), (name, world)
                    if aligned:
                        assert hsum == chk.seq_hash, (name, world, "summed checksums against the oracle's")
                    for d in decs:
                        d.close()
            finally:
                os.environ.pop("NAFGPU_LZ_MODE", None)


def zstd_payload_cases(scale=1):
    """(name, payload, decoded) triples at the zstd-section level (nafgpu_zstd_decompress): several
    magicless frames back to back with > 64 blocks each, so that the repeat-offset
    composition (k_rep_partial / k_rep_scan / k_rep_apply) crosses chunk AND frame boundaries, and
    blocks without sequences sit between blocks with sequences."""
    import zstd_ref
    rng = np.random.default_rng(77)
    out = []
    motif = bytes(rng.integers(65, 91, 61, dtype=np.uint8))
    parts = []
    for f in range(3):
        body = b"".join(motif[(i * 7 + f) % 50:][:11] + bytes([65 + (i * i + f) % 26]) * (i % 5) for i in range(2600 * scale))
        noise = bytes(rng.integers(0, 256, 4000, dtype=np.uint8))            # literal-only / raw blocks in the middle
        parts.append(body[:len(body) // 2] + noise + body[len(body) // 2:])
    for level, step in ((3, 97), (1, 211), (19, 151)):
        payload = b"".join(zstd_ref.compress_magicless(p, level, True, flush_every=step) for p in parts)
        out.append(("multi_frame_l%d_flush%d" % (level, step), payload, b"".join(parts)))
    # ... and every frame with its Content_Checksum (one workgroup of k_xxh64_frames per frame)
    payload = b"".join(zstd_ref.compress_magicless(p, 3, True, flush_every=97, checksum=(k != 1)) for k, p in enumerate(parts))
    out.append(("multi_frame_checksums", payload, b"".join(parts)))
    # the Length section of equal-length reads: one whole-block run per block, each copying from the block
    # before it -- a chain of a few LONG matches, finished pass by pass (launch_lz_more_passes), not pointer-jumped
    words = (151).to_bytes(4, "little") * (400000 * scale)
    for level in (1, 3):
        out.append(("equal_length_words_l%d" % level, zstd_ref.compress_magicless(words, level, True), words))
    return out


def oracle_text(blob, opts=None):
    """FASTA / FASTQ text of an archive from the ORACLE's records (the checker for nafgpu_format_device):
    '>' id [sep comment] newline + the sequence in lines of header.line_length characters, or
    '@' id [sep comment] newline sequence newline '+' newline quality newline when the archive has qualities."""
    from oracle import oracle
    d = oracle.Decoder(blob, **(opts or {}))
    sep = chr(d.header.name_separator)
    width = int(d.header.line_length)
    out = []
    for r in d:
        name = (r.id or "") + (sep + r.comment if r.comment else "")
        seq = r.sequence or ""
        if r.quality is not None:
            out.append("@%s\n%s\n+\n%s\n" % (name, seq, r.quality))
        else:
            lines = [seq[i:i + width] for i in range(0, len(seq), width)] if width else ([seq] if seq else [])
            out.append(">%s\n" % name + "".join(l + "\n" for l in lines))
    return "".join(out).encode()


def text_cases(scale=1):
    """(name, blob) pairs for the text-output parity test: line lengths that do and do not divide the
    record lengths, empty records, records longer than a 16 KiB text chunk, hundreds of short records per
    chunk, no comments / no ids, line_length 0, FASTQ, protein, a mask crossing line ends."""
    rng = np.random.default_rng(4242)
    out = []
    lens = [0, 1, 59, 60, 61, 120, 0, 7, 33000 * scale, 5, 16384, 16383, 1]
    out.append(("fasta_l60", nw.write_naf(make_records(rng, lens), line_length=60)))
    out.append(("fasta_l7_masked", nw.write_naf(make_records(rng, [100, 0, 50000, 3]), line_length=7,
                                                 mask_runs=[5, 20, 30, 45, 2000, 12, 50103 - 2112])))
    out.append(("fasta_l0", nw.write_naf(make_records(rng, [10, 0, 3000]), line_length=0)))
    out.append(("fasta_many_short", nw.write_naf(make_records(rng, [int(x) for x in rng.integers(0, 90, 700 * scale)]), line_length=50)))
    out.append(("fasta_no_comments", nw.write_naf(make_records(rng, [100, 200, 300]), comments=False)))
    out.append(("fasta_no_ids", nw.write_naf(make_records(rng, [100, 200, 300]), ids=False, comments=False)))
    out.append(("fasta_protein", nw.write_naf(make_records(rng, [488, 0, 30001], alphabet="ACDEFGHIKLMNPQRSTVWY"),
                                              sequence_type="protein", line_length=80, level=3)))
    out.append(("fastq_reads", nw.write_naf(make_records(rng, [151] * (300 * scale) + [0, 1, 301], quality=True), quality=True,
                                            level=3, line_length=301)))
    return out
