"""include/nafcodec.hpp (the C++ mirror of the reference's Decoder API) compiled and run for real:
against the CPU harness build of the library here, against libnafgpu.so on a GPU box."""
import os
import subprocess

import pytest

from conftest import ROOT, golden_bytes

CSRC = os.path.join(ROOT, "nafcodec_amd", "csrc")
GOLDEN = os.path.join(ROOT, "tests", "golden")

PROGRAM = r"""
#include <cstdio>
#include <fstream>
#include <iterator>
#include "nafcodec.hpp"
int main(int argc, char **argv) {
    using namespace nafcodec;
    try {
        Decoder dec = DecoderBuilder().with_path(argv[1]);          // mod.rs:159-166
        const Header h = dec.header();
        std::printf("records %llu line %llu type %d remaining %zu\n", (unsigned long long)h.number_of_sequences(),
                    (unsigned long long)h.line_length(), (int)h.sequence_type(), dec.len());
        size_t n = 0, bases = 0;
        std::string first;
        while (auto rec = dec.next()) {                             // Iterator::next
            if (n == 0) first = *rec->id;
            bases += rec->sequence->size();
            n++;
        }
        std::printf("iterated %zu bases %zu first %s\n", n, bases, first.c_str());
        Decoder again = Decoder::from_path(argv[1]);
        const std::string text = again.to_text();                   // FASTA / FASTQ built on the device
        std::ifstream f(argv[2], std::ios::binary);
        const std::string want((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        std::printf("text %s\n", text == want ? "equal" : "DIFFERENT");
        // with_reader (mod.rs:169-256): R: Read + Seek as two callbacks over a FILE*, sequence switched off
        {
            std::FILE *fp = std::fopen(argv[1], "rb");
            struct Ctx { std::FILE *fp; unsigned long long read = 0; int seeks = 0; } ctx{fp};
            auto rd = [](void *c, uint8_t *buf, uint64_t cap) -> int64_t {
                Ctx *x = static_cast<Ctx *>(c);
                const size_t got = std::fread(buf, 1, cap, x->fp);
                x->read += got;
                return std::ferror(x->fp) ? -5 : static_cast<int64_t>(got);
            };
            auto sk = [](void *c, int64_t off, int whence) -> int64_t {
                Ctx *x = static_cast<Ctx *>(c);
                x->seeks++;
                if (std::fseek(x->fp, static_cast<long>(off), whence) != 0) return -29;
                return std::ftell(x->fp);
            };
            Decoder r = DecoderBuilder().sequence(false).with_reader(rd, sk, &ctx);
            size_t k = 0, with_seq = 0, lens = 0;
            while (auto rec = r.next()) {
                k++;
                with_seq += rec->sequence.has_value();
                lens += *rec->length;
            }
            std::printf("reader %zu records, %zu with sequence, lengths %zu, seeks>0 %d\n", k, with_seq, lens, ctx.seeks > 0);
            std::fclose(fp);
        }
        try {
            DecoderBuilder().with_path("/nonexistent.naf");
            std::printf("no error?\n");
        } catch (const Error &e) {
            std::printf("open error io=%d\n", (int)e.is_io());
        }
        // EncoderBuilder / Encoder (encoder/mod.rs:46-384; nafcodec/tests/encoder.rs): written, then read back by the Decoder
        {
            Encoder enc = EncoderBuilder::from_flags(SequenceType::Dna, Flag::Id | Flag::Sequence).quality(true).with_memory();
            Record a, b;
            a.id = "r1"; a.comment = "record 1"; a.sequence = "NGCTCTTAAACCTGCTA"; a.quality = "#8CCCGGGGGGGGGGGG"; a.length = 17;
            b.id = "r2"; b.sequence = "NTAATAAGCAATGACGGCAGC"; b.quality = "#8AACCFF<FFGGFGE@@@@@";
            enc.push(a);
            enc.push(b);
            int refused = 0;
            try { Record c; c.id = "r3"; c.sequence = "ACGX"; c.quality = "IIII"; enc.push(c); } catch (const Error &e) { refused += e.raw.status == NAFGPU_E_INVALID_SEQUENCE; }
            try { Record c; c.id = "r3"; c.sequence = "ACG"; c.quality = "IIII"; enc.push(c); } catch (const Error &e) { refused += e.raw.status == NAFGPU_E_INVALID_LENGTH; }
            try { Record c; c.sequence = "ACG"; c.quality = "III"; enc.push(c); } catch (const Error &e) { refused += e.raw.status == NAFGPU_E_MISSING_FIELD; }
            const std::string archive = enc.write();
            Decoder back = DecoderBuilder().with_bytes(reinterpret_cast<const uint8_t *>(archive.data()), archive.size());
            size_t k = 0;
            bool same = true;
            while (auto rec = back.next()) {
                const Record &w = k == 0 ? a : b;
                same = same && rec->id == w.id && !rec->comment && rec->sequence == w.sequence && rec->quality == w.quality &&
                       rec->length == std::optional<uint64_t>(w.sequence->size());
                k++;
            }
            std::printf("encoder %zu records back, same %d, refused %d\n", k, (int)same, refused);
        }
    } catch (const Error &e) {
        std::printf("Error: %s\n", e.what());
        return 1;
    }
    return 0;
}
"""


def build_and_run(tmp_path, libdir, libname, extra_env=None):
    src = tmp_path / "mirror.cpp"
    src.write_text(PROGRAM)
    exe = tmp_path / "mirror"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-l:" + libname, "-Wl,-rpath," + libdir])
    env = dict(os.environ, **(extra_env or {}))
    out = subprocess.run([str(exe), os.path.join(GOLDEN, "phix.naf"), os.path.join(GOLDEN, "phix.fastq")],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.strip().splitlines()
    assert lines[0] == "records 42 line 301 type 0 remaining 42"
    assert lines[1] == "iterated 42 bases 12436 first SRR1377138.1"
    assert lines[2] == "text equal"
    assert lines[3] == "reader 42 records, 0 with sequence, lengths 12436, seeks>0 1"
    assert lines[4] == "open error io=1"
    assert lines[5] == "encoder 2 records back, same 1, refused 3"


def test_cpp_mirror_on_the_cpu_harness(tmp_path):
    subprocess.check_call(["make", "-s", "-C", CSRC, "emu"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    build_and_run(tmp_path, os.path.join(ROOT, "tests", "emu", "_build"), "libnafgpu_emu.so")


@pytest.mark.gpu
def test_cpp_mirror_on_the_gpu(tmp_path):
    build_and_run(tmp_path, os.path.join(ROOT, "nafcodec_amd"), "libnafgpu.so")
