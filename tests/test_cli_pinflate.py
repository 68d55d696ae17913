"""Block-parallel inflate (filtlong_amd/cli/pinflate.h): pass 1 of a gzip input decoded on several threads from block starts found
by search, with markers for the unknown window, zlib behind the marker decoder once 32 KiB are marker free.  Whatever the file, the
records parsed from it — and the pieces re-inflated from the access points it leaves (FLX_CLI_PARSE_ONLY=unit) — must be exactly
those of the sequential parse of the uncompressed bytes (the reference reads through gzread: src/kseq.h:87-110), and a damaged
file must end the way it ends through zlib alone.  Runs without a GPU."""
import gzip
import io
import os
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "filtlong_amd", "bin", "filtlong")


def run(path, mode, threads=4, **env_over):
    env = dict(os.environ, FLX_CLI_PARSE_ONLY=mode, FLX_CLI_PARALLEL_PARSE_MIN="1", FLX_CLI_THREADS=str(threads), LANG="C", LC_ALL="C")
    env.update({k: str(v) for k, v in env_over.items()})
    p = subprocess.run([BIN, "--target_bases", "1", path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    return p.returncode, p.stdout.decode().strip().replace(" parallel 1 ", " parallel 0 "), p.stderr.decode()


def stats(err):
    """(bytes from the parallel path, of those by zlib tails, rounds, dropped chunks) from the hook's diagnostic line"""
    for l in err.split("\n"):
        if l.startswith("inflate:"):
            t = l.replace("(", " ").split()
            nums = [int(x) for x in t if x.isdigit()]
            return nums[0], nums[2], nums[3], nums[4]
    raise AssertionError(err)


def read_like_fastq(rng, n_reads, mean_len, genome=None, periodic=False):
    """FASTQ whose sequences are error-laden substrings of a small genome (matches at all distances, like real reads) or, with
    periodic=True, a repeated unit (every sequence a copy of the one before: markers never die out)"""
    out = bytearray()
    if genome is None:
        genome = bytes(rng.choice(list(b"ACGT"), size=50000).astype(np.uint8))
    for i in range(n_reads):
        L = max(1, int(rng.gamma(4, mean_len / 4)))
        if periodic:
            seq = (b"ACGT" * (L // 4 + 1))[:L]
        else:
            s = int(rng.randint(0, len(genome)))
            seq = bytearray((genome * (L // len(genome) + 2))[s:s + L])
            for p in np.nonzero(rng.random_sample(L) < 0.05)[0]:
                seq[p] = b"ACGT"[int(rng.randint(4))]
            seq = bytes(seq)
        centre = int(rng.randint(40, 70))
        qual = bytes(np.clip(centre + rng.randint(-8, 9, size=L), 33, 126).astype(np.uint8))
        out += b"@read_%d ch=%d start_time=%d\n" % (i, i % 512, i * 17) + seq + b"\n+\n" + qual + b"\n"
    return bytes(out)


def gz_with_name(data, level):
    buf = io.BytesIO()
    with gzip.GzipFile(filename="reads with a name.fastq", mode="wb", compresslevel=level, fileobj=buf, mtime=0) as f:
        f.write(data)
    return buf.getvalue()


@pytest.fixture(scope="module")
def reads(tmp_path_factory):
    rng = np.random.RandomState(20250925)
    d = tmp_path_factory.mktemp("pinflate")
    data = read_like_fastq(rng, 600, 5000)
    plain = str(d / "reads.fastq")
    open(plain, "wb").write(data)
    rc, want, _ = run(plain, "seq")
    assert rc == 0
    return d, data, want


@pytest.mark.parametrize("level", [1, 2, 4, 6, 9])
@pytest.mark.parametrize("chunk,threads", [(700, 16), (4096, 5), (65536, 3), (300000, 2)])
def test_parallel_inflate_gives_the_sequential_records(reads, level, chunk, threads):
    d, data, want = reads
    gz = str(d / ("l%d.fastq.gz" % level))
    if not os.path.exists(gz):
        open(gz, "wb").write(gz_with_name(data, level) if level % 2 == 0 else gzip.compress(data, level, mtime=0))
    for variant in ("hybrid", "nozlib"):
        env = dict(FLX_CLI_PINFLATE_MIN=1, FLX_CLI_PINFLATE_CHUNK=chunk, FLX_CLI_SPAN_BYTES=200000, FLX_CLI_BLOCK_BYTES=1 << 20)
        if variant == "nozlib":
            env["FLX_CLI_PINFLATE"] = "nozlib"
        rc, got, err = run(gz, "unit", threads, **env)
        assert rc == 0, err
        assert got == want, (level, chunk, threads, variant)
        par, tail, rounds, dropped = stats(err)
        assert par == len(data), (par, len(data), err)  # nothing fell back to the serial reader
        assert rounds >= 1
        if variant == "nozlib":
            assert tail == 0
    # the same through zlib alone
    rc, got, err = run(gz, "unit", threads, FLX_CLI_PINFLATE=0, FLX_CLI_SPAN_BYTES=200000, FLX_CLI_BLOCK_BYTES=1 << 20)
    assert rc == 0 and got == want and stats(err)[0] == 0
    # the whole file taken into memory (references, FLX_CLI_NO_STREAM): the same decoder behind Input::open
    rc, got, err = run(gz, "seq", threads, FLX_CLI_PINFLATE_MIN=1, FLX_CLI_PINFLATE_CHUNK=chunk)
    assert rc == 0 and got == want


def test_zlib_takes_the_tail_of_a_chunk_once_the_markers_are_gone(reads):
    """Where a chunk has 32 KiB without a marker behind it, zlib finishes it (four-letter sequences are coded as chains of short
    matches, so markers of the unknown window live long: a tenth of the bytes here); a file of periodic sequences keeps its markers
    for ever and the marker decoder does all of it.  Same records both ways."""
    d, _, _ = reads
    rng = np.random.RandomState(3)
    genome = bytes(rng.choice(list(b"ACGT"), size=3000000).astype(np.uint8))  # (reads rarely overlap, as in a real run)
    data = read_like_fastq(rng, 500, 6000, genome=genome)
    plain = str(d / "sparse.fastq")
    open(plain, "wb").write(data)
    rc, want, _ = run(plain, "seq")
    gz = plain + ".gz"
    open(gz, "wb").write(gzip.compress(data, 6, mtime=0))
    rc2, got, err = run(gz, "unit", 4, FLX_CLI_PINFLATE_MIN=1, FLX_CLI_PINFLATE_CHUNK=300000, FLX_CLI_SPAN_BYTES=300000)
    assert rc == 0 and rc2 == 0 and got == want
    par, tail, _, _ = stats(err)
    assert par == len(data) and tail > len(data) // 50, err
    rng = np.random.RandomState(7)
    pdata = read_like_fastq(rng, 300, 5000, periodic=True)
    plain = str(d / "periodic.fastq")
    open(plain, "wb").write(pdata)
    open(plain + ".gz", "wb").write(gzip.compress(pdata, 6, mtime=0))
    rc, pw, _ = run(plain, "seq")
    rc2, got, err = run(plain + ".gz", "unit", 4, FLX_CLI_PINFLATE_MIN=1, FLX_CLI_PINFLATE_CHUNK=100000, FLX_CLI_SPAN_BYTES=100000)
    assert rc == 0 and rc2 == 0 and got == pw
    assert stats(err)[0] == len(pdata)


def test_odd_streams(tmp_path):
    """Stored blocks, fixed-Huffman blocks, several members, a member of nothing, text the search does not take for text, input that
    compresses 1000 : 1 — all must come out as through zlib; what the parallel decoder declines, zlib reads."""
    rng = np.random.RandomState(11)
    data = read_like_fastq(rng, 150, 3000)
    plain = str(tmp_path / "in.fastq")
    open(plain, "wb").write(data)
    rc, want, _ = run(plain, "seq")
    assert rc == 0

    def deflate(d, **kw):
        c = zlib.compressobj(kw.pop("level", 6), zlib.DEFLATED, 31, 9, kw.pop("strategy", zlib.Z_DEFAULT_STRATEGY))
        return c.compress(d) + c.flush()

    def flushed(d, every, how):
        c = zlib.compressobj(6, zlib.DEFLATED, 31)
        out = b""
        for a in range(0, len(d), every):
            out += c.compress(d[a:a + every]) + c.flush(how)  # an empty stored block after every piece (pigz does this)
        return out + c.flush()

    variants = {
        "stored": gzip.compress(data, 0, mtime=0),
        "fixed": deflate(data, strategy=zlib.Z_FIXED),
        "huffman_only": deflate(data, strategy=zlib.Z_HUFFMAN_ONLY),
        "rle": deflate(data, strategy=zlib.Z_RLE),
        "sync_flushed": flushed(data, 30000, zlib.Z_SYNC_FLUSH),
        "full_flushed": flushed(data, 50000, zlib.Z_FULL_FLUSH),
        "members": b"".join(gzip.compress(data[a:b], 5, mtime=0) for a, b in zip([0, 100000, 100001, len(data) // 2], [100000, 100001, len(data) // 2, len(data)])),
        "empty_member_first": gzip.compress(b"", 6, mtime=0) + gzip.compress(data, 6, mtime=0),
        "trailing_garbage": gzip.compress(data, 6, mtime=0) + b"\0" * 100,
    }
    for name, blob in sorted(variants.items()):
        gz = str(tmp_path / (name + ".gz"))
        open(gz, "wb").write(blob)
        for chunk in (900, 20000):
            rc, got, err = run(gz, "unit", 6, FLX_CLI_PINFLATE_MIN=1, FLX_CLI_PINFLATE_CHUNK=chunk, FLX_CLI_SPAN_BYTES=50000)
            assert rc == 0, (name, err)
            assert got == want, (name, chunk)
    # control characters in a name: the search for block starts only accepts text, so the chunks are not found and chunk 0 reads on
    weird = data.replace(b"@read_7 ", b"@read_7\x01\x02 ", 1)
    open(plain, "wb").write(weird)
    rc, ww, _ = run(plain, "seq")
    open(plain + ".gz", "wb").write(gzip.compress(weird, 6, mtime=0))
    rc2, got, err = run(plain + ".gz", "unit", 6, FLX_CLI_PINFLATE_MIN=1, FLX_CLI_PINFLATE_CHUNK=5000, FLX_CLI_SPAN_BYTES=50000)
    assert rc == 0 and rc2 == 0 and got == ww
    # 1000 : 1: the decoder gives up beyond 24 times the chunk and zlib reads on
    mono = b"@r\n" + b"A" * 3000000 + b"\n+\n" + b"I" * 3000000 + b"\n"
    open(plain, "wb").write(mono)
    rc, mw, _ = run(plain, "seq")
    open(plain + ".gz", "wb").write(gzip.compress(mono, 6, mtime=0))
    rc2, got, err = run(plain + ".gz", "blk", 6, FLX_CLI_PINFLATE_MIN=1, FLX_CLI_PINFLATE_CHUNK=1000)
    assert rc == 0 and rc2 == 0 and got == mw
    assert stats(err)[0] < len(mono)


def bgzf(data, piece=60000, level=6):
    """the blocked gzip of bgzip / htslib (SAM specification 4.1): members of at most 64 KiB that carry their own size"""
    import struct
    out = b""
    for a in list(range(0, len(data), piece)) + [None]:
        d = b"" if a is None else data[a:a + piece]
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        cd = c.compress(d) + c.flush()
        out += (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(cd) + 25) + cd
                + struct.pack("<II", zlib.crc32(d) & 0xffffffff, len(d)))
    return out


def test_bgzf_and_members_side_by_side(tmp_path):
    """A bgzip file is inflated block-parallel through zlib; plain members in front of it or behind it go their own way; a damaged
    block ends the run like zlib alone."""
    rng = np.random.RandomState(13)
    data = read_like_fastq(rng, 200, 4000)
    plain = str(tmp_path / "in.fastq")
    open(plain, "wb").write(data)
    rc, want, _ = run(plain, "seq")
    assert rc == 0
    assert gzip.decompress(bgzf(data)) == data
    half = len(data) // 2
    variants = {
        "bgzf": bgzf(data),
        "bgzf_small_blocks": bgzf(data, 700, 1),
        "bgzf_then_plain": bgzf(data[:half]) + gzip.compress(data[half:], 6, mtime=0),
        "plain_then_bgzf": gzip.compress(data[:half], 6, mtime=0) + bgzf(data[half:]),
        "bgzf_garbage_behind": bgzf(data) + b"\0\0\0",
    }
    for name, blob in sorted(variants.items()):
        gz = str(tmp_path / (name + ".fastq.gz"))
        open(gz, "wb").write(blob)
        for threads in (2, 7):
            rc, got, err = run(gz, "unit", threads, FLX_CLI_PINFLATE_MIN=1, FLX_CLI_PINFLATE_CHUNK=4000, FLX_CLI_SPAN_BYTES=70000)
            assert rc == 0, (name, err)
            assert got == want, (name, threads)
            assert stats(err)[0] == len(data), (name, err)
        rc, got, err = run(gz, "seq", 4, FLX_CLI_PINFLATE_MIN=1)
        assert rc == 0 and got == want, name
    blob = bytearray(bgzf(data))
    for k in range(6):
        b = bytearray(blob)
        b[int(rng.randint(30, len(b) - 30))] ^= 1 << int(rng.randint(8))
        gz = str(tmp_path / ("bad%d.fastq.gz" % k))
        open(gz, "wb").write(bytes(b))
        for mode in ("blk", "seq"):
            a = run(gz, mode, 5, FLX_CLI_PINFLATE=0)
            p = run(gz, mode, 5, FLX_CLI_PINFLATE_MIN=1)
            assert a[:2] == p[:2], (k, mode, a[2][-200:], p[2][-200:])


def test_damaged_files_end_like_through_zlib(tmp_path):
    """Truncated anywhere, a flipped bit anywhere, a wrong CRC-32, a wrong length: records and end state as with FLX_CLI_PINFLATE=0."""
    rng = np.random.RandomState(5)
    data = read_like_fastq(rng, 120, 3000)
    blob = bytearray(gzip.compress(data, 6, mtime=0))
    cases = {}
    for k in range(8):
        cases["cut%d" % k] = bytes(blob[:int(rng.randint(20, len(blob)))])
    for k in range(12):
        b = bytearray(blob)
        pos = int(rng.randint(10, len(b) - 8))
        b[pos] ^= 1 << int(rng.randint(8))
        cases["flip%d" % k] = bytes(b)
    b = bytearray(blob); b[-8] ^= 0x55; cases["crc"] = bytes(b)
    b = bytearray(blob); b[-1] ^= 0x01; cases["isize"] = bytes(b)
    differing = 0
    for name, bb in sorted(cases.items()):
        gz = str(tmp_path / (name + ".fastq.gz"))
        open(gz, "wb").write(bb)
        a = run(gz, "blk", 5, FLX_CLI_PINFLATE=0)
        p = run(gz, "blk", 5, FLX_CLI_PINFLATE_MIN=1, FLX_CLI_PINFLATE_CHUNK=3000)
        assert a[0] == 0 and a[:2] == p[:2], (name, a[2][-300:], p[2][-300:])
        differing += " status -1 " not in a[1]  # (what the end states must BE: tests/test_cli_damaged_gzip.py, against the reference's reader)
        # taken into memory: whatever gzread makes of the damaged file, with and without the parallel decoder in front of it
        a = run(gz, "seq", 5, FLX_CLI_PINFLATE=0)
        p = run(gz, "seq", 5, FLX_CLI_PINFLATE_MIN=1, FLX_CLI_PINFLATE_CHUNK=3000)
        assert a[:2] == p[:2], name
    assert differing >= 10  # (most of these are errors; a flipped bit in a literal is caught by the CRC-32 either way)
