"""A damaged gzip input must end for the drop-in's parser exactly where — and how — it ends for the reference's: kseq over
zlib's gzread, 16 KiB per call (src/kseq.h:71-76,98-108,234; src/main.cpp:70-88).  gzread delivers everything decodable of a
TRUNCATED stream and then the end of the file; a DATA ERROR (flipped bit, wrong CRC-32 / length) loses the whole call in which
inflate notices it and puts kseq into its error state at the last call boundary, where kseq_read's own handling of -3 decides
what the cut-off record becomes.  Every ingest path (file taken into memory, streamed blocks with the 32 KiB hold-back, zlib alone
or the block-parallel decoder in front of it) is compared with `oracle/_ref/kseq_probe` — our harness around the reference's own
kseq.h + gzread — on record count, end status, the name at a -2 and a digest of every field.  Runs without a GPU."""
import gzip
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from test_cli_pinflate import BIN, ROOT, bgzf, read_like_fastq, run

PROBE = os.path.join(ROOT, "oracle", "_ref", "kseq_probe")
pytestmark = pytest.mark.skipif(not os.path.exists(PROBE), reason="oracle/_ref/kseq_probe not built (needs /root/reference)")


def probe(path):
    p = subprocess.run([PROBE, path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr
    return p.stdout.decode().strip()


PATHS = [
    ("seq", dict(FLX_CLI_PINFLATE=0)),
    ("seq", dict(FLX_CLI_PINFLATE_MIN=1, FLX_CLI_PINFLATE_CHUNK=3000)),
    ("blk", dict(FLX_CLI_PINFLATE=0, FLX_CLI_BLOCK_BYTES=5000)),
    ("blk", dict(FLX_CLI_PINFLATE_MIN=1, FLX_CLI_PINFLATE_CHUNK=3000, FLX_CLI_BLOCK_BYTES=70000)),
    ("blk", dict(FLX_CLI_PINFLATE_MIN=1, FLX_CLI_PINFLATE_CHUNK=40000)),
]


def check(path, tag):
    want = probe(path)
    for mode, env in PATHS:
        rc, got, err = run(path, mode, 5, **env)
        assert rc == 0, (tag, mode, env, err[-300:])
        assert got == want, (tag, mode, env)
    return want


def fasta_like(rng, n, mean_len, width=0):
    out = bytearray()
    for i in range(n):
        L = max(1, int(rng.gamma(4, mean_len / 4)))
        seq = bytes(rng.choice(list(b"ACGT"), size=L).astype(np.uint8))
        out += b">contig_%d len=%d\n" % (i, L)
        if width:
            for a in range(0, L, width):
                out += seq[a:a + width] + b"\n"
        else:
            out += seq + b"\n"
    return bytes(out)


def damage_cases(rng, blob, n_cut, n_flip, tag):
    cases = {}
    for k in range(n_cut):
        cases["%s_cut%d" % (tag, k)] = bytes(blob[:int(rng.randint(1, len(blob)))])
    for k in range(n_flip):
        b = bytearray(blob)
        b[int(rng.randint(0, len(b)))] ^= 1 << int(rng.randint(8))
        cases["%s_flip%d" % (tag, k)] = bytes(b)
    b = bytearray(blob); b[-8] ^= 0x55; cases[tag + "_crc"] = bytes(b)
    b = bytearray(blob); b[-1] ^= 0x01; cases[tag + "_isize"] = bytes(b)
    b = bytearray(blob); b[-4] ^= 0x10; cases[tag + "_isize_low"] = bytes(b)
    cases[tag + "_no_trailer"] = bytes(blob[:-8])
    cases[tag + "_half_trailer"] = bytes(blob[:-3])
    cases[tag + "_garbage_behind"] = bytes(blob) + b"\x1f\x00garbage"
    cases[tag + "_lonely_magic_behind"] = bytes(blob) + b"\x1f\x8b"
    cases[tag + "_bad_member_behind"] = bytes(blob) + b"\x1f\x8b\x08\x00\0\0\0\0\0\xff\xff\xff\xff"
    return cases


@pytest.mark.parametrize("shape", ["fastq", "fastq_crlf", "fasta_wrapped", "fastq_short_reads"])
def test_damaged_gzip_ends_like_the_reference_reader(tmp_path, shape):
    rng = np.random.RandomState(zlib.crc32(shape.encode()))
    if shape == "fastq":
        data = read_like_fastq(rng, 150, 2500)
    elif shape == "fastq_crlf":
        data = read_like_fastq(rng, 150, 1200).replace(b"\n", b"\r\n")
    elif shape == "fasta_wrapped":
        data = fasta_like(rng, 60, 6000, width=60)
    else:
        data = read_like_fastq(rng, 3000, 40)  # many record boundaries per 16 KiB call
    blob = gzip.compress(data, 6, mtime=0)
    cases = damage_cases(rng, blob, 7, 14, shape)
    statuses = {}
    for name, bb in sorted(cases.items()):
        path = str(tmp_path / (name + ".gz"))
        open(path, "wb").write(bb)
        want = check(path, name)
        st = want.split(" status ")[1].split()[0]
        statuses[st] = statuses.get(st, 0) + 1
    # the campaign must have seen clean ends (truncation on a record boundary / harmless damage), cut-off records and the
    # stream's error state
    assert len(statuses) >= 2 and sum(statuses.values()) == len(cases), statuses


def test_error_at_every_stage_of_a_record(tmp_path):
    """The error state begins on a 16 KiB boundary of the output; records laid out so that the boundary falls into the header
    search, the name, the comment, the sequence, the '+' line, the quality (one byte short, exactly complete, complete with '\\r'),
    into a header-only record, and between records — each followed by a bit flip far enough behind to be the first damage."""
    rng = np.random.RandomState(77)
    call = 16384
    tails = {
        "between": b"",
        "garbage": b"\n\n  \n",
        "name": b"@the_name_that_is_cut",
        "name_done": b"@name ",
        "comment": b"@name some comment that goes o",
        "comment_cr": b"@name comment\r",
        "header_done": b"@name comment\n",
        "seq": b"@name c\nACGTACGTAC",
        "seq_cr": b"@name c\nACGTACGTAC\r",
        "seq_done": b"@name c\nACGTACGTAC\n",
        "seq_two_lines": b"@name c\nACGTA\nCGTAC",
        "plus": b"@name c\nACGTACGTAC\n+",
        "plus_text": b"@name c\nACGTACGTAC\n+name agai",
        "plus_done": b"@name c\nACGTACGTAC\n+\n",
        "qual_short": b"@name c\nACGTACGTAC\n+\nIIIIIIIII",
        "qual_full": b"@name c\nACGTACGTAC\n+\nIIIIIIIIII",
        "qual_full_cr": b"@name c\nACGTACGTAC\n+\nIIIIIIIIII\r",
        "qual_long": b"@name c\nACGTACGTAC\n+\nIIIIIIIIIII",
        "qual_two_lines": b"@name c\nACGTACGTAC\n+\nIIIII\nIIII",
        "empty_plus": b"@name\n+",
        "empty_plus_done": b"@name\n+\n",
        "header_only": b"@name",
        "fasta_seq": b">name c\nACGTACGTAC",
        "fasta_header": b">name",
        "dup_name": b"@read_0 again\nACGT",
    }
    for tag, tail in sorted(tails.items()):
        body = read_like_fastq(rng, 40, 900)
        keep = (len(body) + len(tail) + call - 1) // call * call - len(tail)
        # pad with whole records so that `tail` ends exactly on a call boundary
        pad = keep - len(body)
        while pad < 12:
            pad += call
        filler = b"@f c\n" + b"A" * (pad - 11) + b"\n+\n"  # 5 + (pad-11) + 3 ... quality added below
        L = pad - 11
        # a record costs 5 + L + 3 + L + 1 bytes: solve for L with the parity it needs, the rest as blank lines
        L = (pad - 9) // 2
        filler = b"@f c\n" + b"A" * L + b"\n+\n" + b"I" * L + b"\n" + b"\n" * (pad - 9 - 2 * L)
        assert len(filler) == pad
        front = body + filler + tail
        assert len(front) % call == 0
        behind = read_like_fastq(rng, 30, 900)
        data = front + behind
        # deflate so that the damage lies in the blocks behind the boundary: two streams' worth, stored block + compressed tail
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        raw = c.compress(front) + c.flush(zlib.Z_FULL_FLUSH)
        at = len(raw)
        raw += c.compress(behind) + c.flush()
        blob = bytearray(b"\x1f\x8b\x08\x00\0\0\0\0\0\xff" + raw + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data) & 0xffffffff))
        assert gzip.decompress(bytes(blob)) == data
        blob[10 + at + 1] ^= 0xff  # the first bytes of the block behind the flush point: a header zlib rejects at once
        blob[10 + at + 2] ^= 0xff
        path = str(tmp_path / (tag + ".gz"))
        open(path, "wb").write(bytes(blob))
        want = check(path, tag)
        assert " status -1 " not in want, (tag, want)  # (the damage is real: every one of these ends in -2 or -3)
    # sanity of the probe itself: an intact file ends with -1
    path = str(tmp_path / "intact.gz")
    open(path, "wb").write(gzip.compress(read_like_fastq(rng, 10, 500)))
    assert " status -1 " in check(path, "intact")


def test_members_and_bgzf_damaged(tmp_path):
    """Damage in a second member and in BGZF blocks: the failing member's own 16 KiB grid starts at its first byte, the calls
    of kseq keep the grid of the whole stream (inflate_stream.h: gzread_delivered_before_error)."""
    rng = np.random.RandomState(99)
    data = read_like_fastq(rng, 200, 1500)
    third = len(data) // 3
    blobs = {
        "two": gzip.compress(data[:third], 6, mtime=0) + gzip.compress(data[third:], 6, mtime=0),
        "three": gzip.compress(data[:third], 1, mtime=0) + gzip.compress(data[third:2 * third], 9, mtime=0) + gzip.compress(data[2 * third:], 6, mtime=0),
        "bgzf": bgzf(data),
        "bgzf_small": bgzf(data, 3000, 6),
        "plain_then_bgzf": gzip.compress(data[:third], 6, mtime=0) + bgzf(data[third:]),
    }
    n = 0
    for tag, blob in sorted(blobs.items()):
        for name, bb in sorted(damage_cases(rng, blob, 3, 7, tag).items()):
            path = str(tmp_path / (name + ".gz"))
            open(path, "wb").write(bb)
            check(path, name)
            n += 1
    assert n >= 80


def test_forged_bgzf_length_goes_to_zlib(tmp_path):
    """A BGZF block whose ISIZE claims gigabytes (advisor, round 3): no allocation of the claimed size, zlib has the word."""
    rng = np.random.RandomState(5)
    data = read_like_fastq(rng, 50, 1500)
    blob = bytearray(bgzf(data))
    first_len = struct.unpack("<H", blob[16:18])[0] + 1
    blob[first_len - 4:first_len] = struct.pack("<I", 0xfffffff0)
    path = str(tmp_path / "forged.gz")
    open(path, "wb").write(bytes(blob))
    env = dict(os.environ, FLX_CLI_PARSE_ONLY="blk", FLX_CLI_THREADS="4", FLX_CLI_PINFLATE_MIN="1", LANG="C")
    p = subprocess.run("ulimit -v 8000000; exec '%s' --target_bases 1 '%s'" % (BIN, path), shell=True, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, env=env, timeout=120)
    assert p.returncode == 0, p.stderr[-300:]
    assert p.stdout.decode().strip().replace(" parallel 1 ", " parallel 0 ") == probe(path)


def test_forced_hand_over_still_checks_the_trailer(tmp_path):
    """A text that compresses better than 24 : 1 makes the chunk decoder hand the member over to zlib in mid-stream (raw inflate
    from a block boundary); the member's CRC-32 and length are still checked there (advisor, round 3: they were skipped)."""
    unit = b"@r%d\n" + b"ACGT" * 2500 + b"\n+\n" + b"I" * 10000 + b"\n"
    data = b"".join(unit % i for i in range(2000))
    blob = bytearray(gzip.compress(data, 6, mtime=0))
    assert len(data) > 100 * len(blob)
    for tag, at, mask in (("crc", -8, 0x01), ("crc_high", -5, 0x80), ("isize", -4, 0x01), ("isize_high", -1, 0x40)):
        b = bytearray(blob)
        b[at] ^= mask
        path = str(tmp_path / (tag + ".gz"))
        open(path, "wb").write(bytes(b))
        want = probe(path)
        assert " status -1 " not in want
        for mode in ("seq", "blk"):
            rc, got, err = run(path, mode, 4, FLX_CLI_PINFLATE_MIN=1, FLX_CLI_PINFLATE_CHUNK=4096)
            assert rc == 0 and got == want, (tag, mode, got, want, err[-300:])
    path = str(tmp_path / "fine.gz")
    open(path, "wb").write(bytes(blob))
    assert " status -1 " in check(path, "fine")
