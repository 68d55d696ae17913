"""Differential fuzz: seeded random inputs and flag combinations through filtlong_amd/bin/filtlong AND the reference binary
(oracle/_ref/filtlong, built from /root/reference/src by oracle/Makefile; it travels to the GPU box), run on the same files
with the same argv.  Exit code, stdout bytes and stderr as a terminal shows it must be the same for every case — valid runs
in all three scoring modes (src/read.cpp:35-58), every hard cut-off and weight (src/arguments.cpp:126-221), --trim / --split
down to 1 (src/read.cpp:86-141), exact score ties from records with equal content (src/main.cpp:247-257), odd record
grammar (src/kseq.h:176-224) and the error paths of src/main.cpp:80-116."""
import gzip
import os
import random
import subprocess
import zlib

import pytest

import _oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "filtlong_amd", "bin", "filtlong")

LENGTHS = [0, 1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 99, 100, 101, 249, 250, 251, 500, 1000, 1023, 1024, 1025, 2000, 4000]
WINDOWS = [1, 2, 15, 16, 17, 50, 100, 249, 250, 251, 600, 1500]
SPLITS = [1, 2, 5, 31, 32, 33, 100, 250, 1000]


def rand_bases(rng, n):
    return bytes(rng.choice(b"ACGT") for _ in range(n))


def mutate(rng, s, rate):
    if rate == 0 or not s:
        return s
    b = bytearray(s)
    for i in range(len(b)):
        if rng.random() < rate:
            b[i] = rng.choice(b"ACGT")
    return bytes(b)


def make_case(seed):
    """-> dict(files={name: bytes}, argv=[...] with file names relative to the case directory)"""
    rng = random.Random(seed)
    mode = rng.choices(["phred", "asm", "sr"], [0.5, 0.3, 0.2])[0]
    files = {}
    argv = []
    contigs = []
    if mode != "phred":
        contigs = [rand_bases(rng, rng.choice([40, 300, 1500, 5000])) for _ in range(rng.choice([1, 1, 2, 3]))]
        if rng.random() < 0.15:
            contigs.append(rand_bases(rng, rng.choice([0, 5, 15, 16])))  # references shorter than a 16-mer
    if mode == "asm":
        fa = b"".join(b">c%d some text\n" % i + (c.lower() if rng.random() < 0.2 else c) + b"\n" for i, c in enumerate(contigs))
        files["ref.fasta"] = fa
        argv += ["-a", "ref.fasta"]
    elif mode == "sr":
        # 100-mers tiled at a depth that puts most 16-mers in the set (4 copies, or 3 and a Bloom false positive: src/kmers.cpp:142-166)
        pairs = [[], []]
        for c in contigs:
            if len(c) < 100:
                continue
            for _ in range(max(1, len(c) * rng.choice([3, 8, 12]) // 100)):
                p = rng.randrange(0, len(c) - 99)
                pairs[rng.randrange(2)].append(c[p:p + 100])
        for k in (0, 1):
            fq = b"".join(b"@s%d\n%s\n+\n%s\n" % (i, s, b"I" * len(s)) for i, s in enumerate(pairs[k]))
            files["sr_%d.fastq" % (k + 1)] = fq
        argv += ["-1", "sr_1.fastq", "-2", "sr_2.fastq"]

    if files and rng.random() < 0.3:  # gzip-compressed references (read through gzopen by the reference: src/kmers.cpp:86-90)
        for name in list(files):
            files[name + ".gz"] = gzip.compress(files.pop(name), rng.choice([1, 6, 9]), mtime=0)
        argv = [a + ".gz" if a in ("ref.fasta", "sr_1.fastq", "sr_2.fastq") else a for a in argv]

    # ---- reads ----
    n = rng.choice([0, 1, 2, 3, 5, 10, 30, 60])
    reads = []
    total = 0
    for i in range(n):
        L = rng.choice(LENGTHS) if rng.random() < 0.6 else rng.randrange(1, 3000)
        if contigs and rng.random() < 0.85:
            c = rng.choice(contigs)
            if len(c) > 0:
                p = rng.randrange(0, len(c))
                s = (c * (1 + L // max(1, len(c)) + 1))[p:p + L]
                s = mutate(rng, s, rng.choice([0, 0, 0.02, 0.1, 0.3]))
                if rng.random() < 0.5 and L > 60:  # junk inside and at the ends: split and trim have something to find
                    b = bytearray(s)
                    for _ in range(rng.choice([1, 1, 2, 3])):
                        a = rng.randrange(0, L)
                        w = rng.choice([1, 5, 20, 40, 120, 400])
                        b[a:a + w] = rand_bases(rng, len(b[a:a + w]))
                    if rng.random() < 0.5:
                        w = rng.randrange(1, 60)
                        b[:w] = rand_bases(rng, w)
                    if rng.random() < 0.5:
                        w = rng.randrange(1, 60)
                        b[-w:] = rand_bases(rng, w)
                    s = bytes(b)
            else:
                s = rand_bases(rng, L)
            if rng.random() < 0.3:
                s = _oracle_revcomp(s)
        else:
            s = rand_bases(rng, L)
        if rng.random() < 0.1:
            s = s.lower()
        if rng.random() < 0.1 and L > 0:
            b = bytearray(s)
            for _ in range(rng.randrange(1, 6)):
                b[rng.randrange(0, L)] = rng.choice(b"NnRYKM-*")
            s = bytes(b)
        centre = rng.randrange(33, 100)
        spread = rng.choice([0, 2, 8, 30])
        if rng.random() < 0.05:
            q = bytes([33]) * L
        elif rng.random() < 0.05:
            q = bytes(rng.randrange(33, 127) for _ in range(L))
        else:
            q = bytes(min(126, max(33, centre + rng.randrange(-spread, spread + 1))) for _ in range(L))
        name = b"r%d" % i
        comment = rng.choice([b"", b"", b" len=%d" % L, b"\tx y z", b" "])
        reads.append([name, comment, s, q])
        total += L
    if n >= 2 and rng.random() < 0.25:  # records of equal content under different names: exact score ties around the cut
        for _ in range(rng.randrange(1, 1 + n // 2)):
            a, b = rng.randrange(n), rng.randrange(n)
            if a != b:
                total += len(reads[a][2]) - len(reads[b][2])
                reads[b][2], reads[b][3] = reads[a][2], reads[a][3]
    error_kind = None
    if n >= 2 and rng.random() < 0.06:
        reads[rng.randrange(1, n)][0] = reads[0][0]  # duplicate name (src/main.cpp:113-117)
        error_kind = "dup"

    fasta = mode != "phred" and rng.random() < 0.2
    if mode == "phred" and rng.random() < 0.04:
        fasta = True  # FASTA without an external reference (src/main.cpp:103-106)
    style = rng.choices(["plain", "wrapped", "crlf", "no_final_newline", "blank_lines", "truncated", "mixed"],
                        [0.55, 0.1, 0.1, 0.1, 0.05, 0.05, 0.05])[0]
    nl = b"\r\n" if style == "crlf" else b"\n"
    out = bytearray()
    for k, (name, comment, s, q) in enumerate(reads):
        as_fasta = fasta != (style == "mixed" and k == n - 1 and n > 1)
        head = (b">" if as_fasta else b"@") + name + comment + nl
        if style == "wrapped" and len(s) > 0:
            w = rng.choice([1, 7, 60, 61])
            body = nl.join(s[i:i + w] for i in range(0, len(s), w)) + nl
            qual = nl.join(q[i:i + w] for i in range(0, len(q), w)) + nl
        else:
            body, qual = s + nl, q + nl
        out += head + body
        if not as_fasta:
            out += b"+" + (name if rng.random() < 0.1 else b"") + nl + qual
        if style == "blank_lines" and rng.random() < 0.5:
            out += nl
    if style == "no_final_newline" and out.endswith(nl):
        del out[-len(nl):]
    if style == "truncated" and len(out) > 10:
        del out[-rng.randrange(1, min(len(out), 200)):]
    data = bytes(out)
    inname = "reads.fasta" if fasta else "reads.fastq"
    if rng.random() < 0.12:
        data = gzip.compress(data, mtime=0)
        inname += ".gz"
    files[inname] = data

    # ---- flags ----
    if rng.random() < 0.3:
        argv += ["--min_length", str(rng.choice([1, 16, 100, 250, 1000, 5000] + [0] * (rng.random() < 0.1)))]
    if rng.random() < 0.12:
        argv += ["--max_length", str(rng.choice([1, 100, 1000, 2500, 100000]))]
    if rng.random() < 0.2:
        argv += ["--min_mean_q", "%.3f" % rng.uniform(0, 100)]
    if rng.random() < 0.2:
        argv += ["--min_window_q", "%.3f" % rng.uniform(0, 100)]
    r = rng.random()
    if rng.random() < 0.03:
        r = 0.45 if rng.random() < 0.5 else 2.0  # 2.0: no threshold at all (an argument error unless a hard cut-off is given)
    if r < 0.4 or 0.9 < r <= 1.0:
        argv += ["--keep_percent", rng.choice(["%d" % rng.randrange(1, 100), "%.2f" % rng.uniform(0.5, 99.9), "50", "99.999", "0.001"] + ["100"] * (rng.random() < 0.1))]
    if 0.3 < r < 0.95 or 0.97 < r <= 1.0:
        argv += ["--target_bases", str(rng.choice([1] * (rng.random() < 0.2) + [max(1, total // 3), max(1, total // 2), max(1, total - 1), total + 1, 10 ** 9,
                                                     rng.randrange(1, total + 2)]))]
    if rng.random() < 0.25:
        for flag in ("--length_weight", "--mean_q_weight", "--window_q_weight"):
            if rng.random() < 0.6:
                argv += [flag, rng.choice(["0", "0.5", "1", "2", "10", "0.001"])]
    if rng.random() < 0.35:
        argv += ["--window_size", str(rng.choice(WINDOWS))]
    if mode != "phred" or rng.random() < 0.05:
        if rng.random() < 0.45:
            argv += ["--trim"]
        if rng.random() < 0.45:
            argv += ["--split", str(rng.choice(SPLITS))]
    if rng.random() < 0.15:
        argv += ["--verbose"]
    if rng.random() < 0.2:  # the short forms, glued to their value or not (src/arguments.cpp:126-221)
        short = {"--target_bases": "-t", "--keep_percent": "-p", "--min_length": "-l", "--max_length": "-L", "--min_mean_q": "-q"}
        if rng.random() < 0.1:
            short["--min_window_q"] = "-w"  # (no such flag: "Error: flag could not be matched: 'w'")
        new = []
        for tok in argv:
            if new and new[-1] in short.values() and rng.random() < 0.5:
                new[-1] += tok  # -t100
            else:
                new.append(short.get(tok, tok))
        argv = new
    argv += [inname]
    return {"files": files, "argv": argv, "mode": mode, "style": style, "error": error_kind}


_COMP = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")


def _oracle_revcomp(s):
    return s.translate(_COMP)[::-1]


def shown(err):
    """stderr RAW, line by line (round 5: every "\\r  N reads (M bp)" / "\\r  file (M bp)" progress update is part of the comparison —
    up to round 4 only the last carriage-return segment of a line was looked at, i.e. what a terminal ends up showing)"""
    return err.split("\n")


def run_both(case, td, extra_env=None, new_argv_prefix=()):
    for name, data in case["files"].items():
        with open(os.path.join(td, name), "wb") as f:
            f.write(data)
    env = dict(os.environ, LANG="C", LC_ALL="C")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    ref = subprocess.run([_oracle.REF_FILTLONG] + case["argv"], cwd=td, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    env.update(extra_env or {})
    # gzip inputs and references (about one case in eight): through the block-parallel inflater, in chunks of a few hundred bytes
    env.update(FLX_CLI_PINFLATE_MIN="1", FLX_CLI_PINFLATE_CHUNK="300")
    new = subprocess.run([BIN] + list(new_argv_prefix) + case["argv"], cwd=td, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    return ref, new


def need_reference_binary():
    """oracle/_ref/filtlong is built by __graft_entry__.build() / `make -C oracle` where /root/reference exists and travels with the
    snapshot; without it these tests have nothing to compare with (the golden-vector tests do not need it)."""
    if not os.path.exists(_oracle.REF_FILTLONG):
        pytest.skip("oracle/_ref/filtlong missing: run `make -C oracle` where /root/reference exists")


N_CASES = int(os.environ.get("FLX_FUZZ_CASES", "240"))      # a longer campaign: FLX_FUZZ_CASES=2000 FLX_FUZZ_BASE=campaign-2
SEED_BASE = os.environ.get("FLX_FUZZ_BASE", "filtlong-fuzz").encode()
INGEST = {
    "default": {},
    "chunked": {"FLX_CLI_CHUNK_BYTES": "9000"},
    "blocks": {"FLX_CLI_FORCE_STREAM": "1", "FLX_CLI_BLOCK_BYTES": "5000"},
}


@pytest.mark.parametrize("part", range(4))
def test_random_invocations_match_the_reference_binary(tmp_path, part):
    need_reference_binary()
    seen = {"ok": 0, "error": 0, "children": 0, "empty_out": 0}
    for i in range(part, N_CASES, 4):
        seed = zlib.crc32(SEED_BASE + b"-%d" % i)
        case = make_case(seed)
        td = tmp_path / ("case%d" % i)
        td.mkdir()
        ingest = sorted(INGEST)[(i // 4) % 3] if i % 8 >= 4 else "default"
        ref, new = run_both(case, str(td), INGEST[ingest])
        what = (i, seed, ingest, case["argv"], case["mode"], case["style"])
        assert ref.returncode in (0, 1), (what, ref.stderr[-400:])
        assert new.returncode == ref.returncode, (what, new.stderr.decode(errors="replace")[-600:], ref.stderr.decode(errors="replace")[-600:])
        assert new.stdout == ref.stdout, (what, len(new.stdout), len(ref.stdout))
        if b"usage:" not in ref.stderr and b"--help" not in ref.stderr:
            assert shown(new.stderr.decode(errors="replace")) == shown(ref.stderr.decode(errors="replace")), what
        seen["ok" if ref.returncode == 0 else "error"] += 1
        seen["children"] += ref.stdout.count(b"-") > 0 and ("--trim" in case["argv"] or "--split" in case["argv"])
        seen["empty_out"] += len(ref.stdout) == 0
    assert seen["ok"] >= 20 and seen["error"] >= 1, seen


def _damaged_case(seed):
    """A gzip-compressed input (about 100-300 kB of text: many 16 KiB gzread calls) with one kind of damage, or a damaged
    gzip reference.  What the reference makes of it is decided by zlib's gzread and kseq's handling of its error state
    (src/kseq.h:71-76,98-108,176-224; src/main.cpp:77-88; src/kmers.cpp:91-94): a truncated stream is read as far as it decodes,
    a data error costs the 16 KiB call it is noticed in and ends in -2 / -3 / a record of another format, depending on where."""
    rng = random.Random(seed)
    mode = rng.choice(["phred", "phred", "asm"])
    files, argv = {}, []
    contig = rand_bases(rng, 6000)
    if mode == "asm":
        files["ref.fasta"] = b">c0\n" + contig + b"\n" + b">c1 second\n" + rand_bases(rng, 3000) + b"\n"
        argv += ["-a", "ref.fasta"]
    fasta = mode == "asm" and rng.random() < 0.3
    n = rng.choice([60, 120, 250])
    recs, total = [], 0
    nl = b"\r\n" if rng.random() < 0.15 else b"\n"
    for i in range(n):
        L = rng.choice([0, 1, 40, 250]) if rng.random() < 0.05 else rng.randrange(200, 1600)
        if mode == "asm" and rng.random() < 0.8:
            a = rng.randrange(0, len(contig))
            s = mutate(rng, (contig * 2)[a:a + L], rng.choice([0, 0.03, 0.15]))
        else:
            s = rand_bases(rng, L)
        centre = rng.randrange(40, 100)
        q = bytes(min(126, max(33, centre + rng.randrange(-6, 7))) for _ in range(L))
        head = (b">" if fasta else b"@") + b"r%d" % i + rng.choice([b"", b" c=%d" % i, b"\tt"]) + nl
        recs.append(head + s + nl + (b"" if fasta else b"+" + nl + q + nl))
        total += L
    kind = rng.choice(["cut", "cut", "cut_boundary", "flip", "flip", "flip", "crc", "isize", "two_members_flip", "ref_cut", "ref_flip"])
    if kind.startswith("ref") and mode != "asm":
        kind = "flip"
    if mode == "asm" and rng.random() < 0.3:
        kind = rng.choice(["ref_cut", "ref_flip"])
    text = b"".join(recs)
    if kind == "cut_boundary":  # the compressed stream ends exactly behind a whole record (sync flush there): the reads so far, exit 0
        k = rng.randrange(1, n)
        c = zlib.compressobj(6, zlib.DEFLATED, 31)
        blob = c.compress(b"".join(recs[:k])) + c.flush(zlib.Z_SYNC_FLUSH)
    elif kind == "two_members_flip":
        k = rng.randrange(1, n)
        first = gzip.compress(b"".join(recs[:k]), 6, mtime=0)
        second = bytearray(gzip.compress(b"".join(recs[k:]), 6, mtime=0))
        second[rng.randrange(12, len(second))] ^= 1 << rng.randrange(8)
        blob = first + bytes(second)
    else:
        blob = bytearray(gzip.compress(text, rng.choice([1, 6, 9]), mtime=0))
        if kind == "cut":
            del blob[rng.randrange(1, len(blob)):]
        elif kind == "flip":
            blob[rng.randrange(0, len(blob))] ^= 1 << rng.randrange(8)
        elif kind == "crc":
            blob[-8 + rng.randrange(4)] ^= 1 << rng.randrange(8)
        elif kind == "isize":
            blob[-4 + rng.randrange(4)] ^= 1 << rng.randrange(8)
        blob = bytes(blob)
    inname = ("reads.fasta" if fasta else "reads.fastq") + ".gz"
    files[inname] = blob
    if kind.startswith("ref"):
        ref = bytearray(gzip.compress(files.pop("ref.fasta"), 6, mtime=0))
        if kind == "ref_cut":
            del ref[rng.randrange(20, len(ref)):]
        else:
            ref[rng.randrange(10, len(ref))] ^= 1 << rng.randrange(8)
        files["ref.fasta.gz"] = bytes(ref)
        argv = ["-a", "ref.fasta.gz"]
    argv += rng.choice([["-t", str(max(1, total // 2))], ["-p", "80"], ["--min_length", "300", "-t", str(max(1, total // 3))]])
    if mode == "asm" and rng.random() < 0.5:
        argv += rng.choice([["--trim"], ["--split", "100"], ["--trim", "--split", "250"]])
    if rng.random() < 0.25:
        argv += ["--verbose"]
    argv += [inname]
    return {"files": files, "argv": argv, "mode": mode, "style": kind}


DAMAGED_INGEST = {
    "streamed": {},                                                               # gzip on one GPU: blocks of 256 MiB, block-parallel inflate
    "streamed_small_blocks": {"FLX_CLI_BLOCK_BYTES": "5000", "FLX_CLI_SPAN_BYTES": "20000"},
    "streamed_zlib": {"FLX_CLI_PINFLATE": "0", "FLX_CLI_BLOCK_BYTES": "40000"},
    "in_memory": {"FLX_CLI_NO_STREAM": "1"},                                      # what pipes and several ranks use
    "in_memory_chunked": {"FLX_CLI_NO_STREAM": "1", "FLX_CLI_CHUNK_BYTES": "9000"},
}


@pytest.mark.parametrize("part", range(3))
def test_damaged_gzip_inputs_match_the_reference_binary(tmp_path, part):
    """VERDICT r3, missing 2: the reference on a damaged / truncated gzip — exit code, stdout and stderr — through every ingest
    path.  45 seeded cases per run (FLX_FUZZ_DAMAGED raises it)."""
    need_reference_binary()
    n_cases = int(os.environ.get("FLX_FUZZ_DAMAGED", "45"))
    seen = {}
    for i in range(part, n_cases, 3):
        seed = zlib.crc32(SEED_BASE + b"-damaged-%d" % i)
        case = _damaged_case(seed)
        for ingest in sorted(DAMAGED_INGEST):
            td = tmp_path / ("case%d_%s" % (i, ingest))
            td.mkdir()
            ref, new = run_both(case, str(td), DAMAGED_INGEST[ingest])
            what = (i, seed, ingest, case["argv"], case["style"])
            assert ref.returncode in (0, 1), (what, ref.stderr[-400:])
            assert new.returncode == ref.returncode, (what, new.stderr.decode(errors="replace")[-600:], ref.stderr.decode(errors="replace")[-600:])
            assert new.stdout == ref.stdout, (what, len(new.stdout), len(ref.stdout))
            assert shown(new.stderr.decode(errors="replace")) == shown(ref.stderr.decode(errors="replace")), what
        last = [l for l in (x.split("\r")[-1] for x in shown(ref.stderr.decode(errors="replace"))) if l.startswith("Error") or l.startswith("  problem")]
        key = (case["style"], ref.returncode, last[0].split(" for read")[0].split(" reads.")[0] if last else "")
        seen[key] = seen.get(key, 0) + 1
    assert len(seen) >= 4, seen  # clean ends, cut-off records (-2), the stream's error state (-3), ...


def test_damaged_gzip_inputs_with_forked_ranks(tmp_path):
    """The same damage through `--gpus 2` (every rank takes the file into memory)."""
    need_reference_binary()
    shim_dir = os.path.join(ROOT, "tests", "shim")
    subprocess.check_call(["make", "-s", "-C", shim_dir])
    env = {"FLX_RCCL_LIB": os.path.join(shim_dir, "libloopback_rccl.so"), "FLX_DEVICE": "0"}
    n = 0
    for i in range(12):
        case = _damaged_case(zlib.crc32(SEED_BASE + b"-damaged-ranks-%d" % i))
        td = tmp_path / ("case%d" % i)
        td.mkdir()
        ref, new = run_both(case, str(td), env, ["--gpus", "2"])
        what = (i, case["argv"], case["style"])
        assert new.returncode == ref.returncode, (what, new.stderr.decode(errors="replace")[-600:], ref.stderr.decode(errors="replace")[-600:])
        assert new.stdout == ref.stdout, (what, len(new.stdout), len(ref.stdout))
        assert shown(new.stderr.decode(errors="replace")) == shown(ref.stderr.decode(errors="replace")), what
        n += 1
    assert n == 12


def test_header_only_records_print_what_the_reference_prints(tmp_path):
    """A record that is only a header (kseq: length 0, no '+' line) comes out of the reference's FASTQ writer with the quality string
    of the last record in front of it that had a '+' line (src/main.cpp:279 prints kseq's buffer, of which only the length was
    reset); with no such record std::cout goes bad and the output ends behind that record's "+" line.  Same bytes here, for every
    way of reading the input."""
    need_reference_binary()
    rng = random.Random(99)

    def rec(i, L, wrap=0):
        s, q = rand_bases(rng, L), bytes(rng.randrange(40, 90) for _ in range(L))
        if wrap:
            s = b"\n".join(s[a:a + wrap] for a in range(0, L, wrap))
            q = b"\n".join(q[a:a + wrap] for a in range(0, L, wrap))
        return b"@r%d\n" % i + s + b"\n+\n" + q + b"\n"

    inputs = {
        "middle": rec(0, 300) + rec(1, 500) + b"@lonely\n" + rec(2, 400) + rec(3, 200),
        "at_eof_no_newline": rec(0, 300) + rec(1, 500) + b"@lonely",
        "at_eof_comment": rec(0, 300) + rec(1, 500) + b"@lonely with a comment\n",
        "two_in_a_row": rec(0, 300) + b"@l1\n@l2 c\n" + rec(1, 500) + b"@l3\n",
        "behind_wrapped_record": rec(0, 300, wrap=70) + b"@lonely\n" + rec(1, 500),
        "first_record": b"@lonely\n" + rec(0, 300) + rec(1, 500),
        "first_and_later": b"@lonely\n" + rec(0, 300) + b"@later\n" + rec(1, 500),
        "behind_empty_plus_record": rec(0, 300) + b"@e\n\n+\n\n" + b"@lonely\n" + rec(1, 200),
        "fasta_header_in_fastq": rec(0, 300) + b">lonely\n" + rec(1, 200),
    }
    flag_sets = [["-t", "1000000"], ["-p", "90"], ["--min_mean_q", "1"], ["--min_length", "1", "-t", "100000"], ["-t", "600"]]
    n = 0
    for name, data in sorted(inputs.items()):
        for flags in flag_sets:
            for ingest in sorted(INGEST):
                td = tmp_path / ("%s_%d" % (name, n))
                td.mkdir()
                case = {"files": {"reads.fastq": data}, "argv": flags + ["reads.fastq"]}
                ref, new = run_both(case, str(td), INGEST[ingest])
                what = (name, flags, ingest)
                assert new.returncode == ref.returncode, (what, new.stderr[-300:], ref.stderr[-300:])
                assert new.stdout == ref.stdout, (what, new.stdout[-80:], ref.stdout[-80:])
                assert shown(new.stderr.decode(errors="replace")) == shown(ref.stderr.decode(errors="replace")), what
                n += 1
    # gzip input: the streamed reader, the header-only record and its predecessor in different blocks
    data = b"".join(rec(i, 900) for i in range(12)) + b"@lonely\n" + b"".join(rec(i, 700) for i in range(12, 20)) + b"@last"
    for block in ("1900", "7000", "100000"):
        td = tmp_path / ("gz_%s" % block)
        td.mkdir()
        case = {"files": {"reads.fastq.gz": gzip.compress(data, 6, mtime=0)}, "argv": ["-t", "1000000", "reads.fastq.gz"]}
        ref, new = run_both(case, str(td), {"FLX_CLI_BLOCK_BYTES": block, "FLX_CLI_SPAN_BYTES": "2500"})
        assert new.returncode == ref.returncode == 0 and new.stdout == ref.stdout, block
        assert b"@lonely\n\n+\n" in ref.stdout and ref.stdout.count(b"\n") == 4 * 22


def test_random_invocations_with_forked_ranks(tmp_path):
    """The same kind of random command lines through `--gpus 2` / `--gpus 3` (ranks forked by the command line, the library's
    communicator over tests/shim's loopback RCCL on one GPU): reads sharded by count, the global stage across ranks, part files
    stitched by rank 0 — still the reference binary's exit code, stdout and stderr (--verbose included: blocks and table rows of
    every rank's reads in file order)."""
    need_reference_binary()
    shim_dir = os.path.join(ROOT, "tests", "shim")
    subprocess.check_call(["make", "-s", "-C", shim_dir])
    env = {"FLX_RCCL_LIB": os.path.join(shim_dir, "libloopback_rccl.so"), "FLX_DEVICE": "0"}
    n = 0
    for i in range(0, N_CASES, 5):
        case = make_case(zlib.crc32(SEED_BASE + b"-ranks-%d" % i))
        td = tmp_path / ("case%d" % i)
        td.mkdir()
        gpus = "2" if i % 10 == 0 else "3"
        ref, new = run_both(case, str(td), env, ["--gpus", gpus])
        what = (i, gpus, case["argv"], case["mode"], case["style"])
        assert new.returncode == ref.returncode, (what, new.stderr.decode(errors="replace")[-600:], ref.stderr.decode(errors="replace")[-600:])
        assert new.stdout == ref.stdout, (what, len(new.stdout), len(ref.stdout))
        if b"usage:" not in ref.stderr:
            assert shown(new.stderr.decode(errors="replace")) == shown(ref.stderr.decode(errors="replace")), what
        n += 1
    assert n >= 30
