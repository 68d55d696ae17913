"""End-to-end through the C++ command line on the GPU: stdout must be BYTE-IDENTICAL to the reference binary's
(sha256 recorded in tests/golden/e2e.json by tests/golden/make_golden.py) and the stderr summary lines equal."""
import hashlib
import json
import os
import subprocess
import tempfile

import pytest

import _cases
import _e2e_checks

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "filtlong_amd", "bin", "filtlong")
FIX = _cases.FIXTURES


def run(args, cwd, extra_env=None):
    env = dict(os.environ, LANG="C", LC_ALL="C")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env.update(extra_env or {})
    p = subprocess.run([BIN] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=cwd, env=env)
    err = p.stderr.decode(errors="replace")
    keep = [l.strip() for l in err.replace("\r", "\n").split("\n")
            if any(t in l for t in ("target:", "keeping", "not enough", "already fall", "after ", "Error", "16-mers"))]
    return p.returncode, p.stdout, keep, err


MODES = {
    "default": {},
    # streaming ingest forced into many small chunks (two pinned slots, worker thread): same bytes out
    "chunked": {"FLX_CLI_CHUNK_BYTES": "20000"},
    # the gzip path: input read block by block (tail carried over, blocks and pipeline slots grown for long records),
    # nothing of the input kept but names and lengths, second pass over the file for the output
    # (gzip files, the reference's short-read fixtures among them, through the block-parallel inflater in chunks of 2000 bytes)
    "blocks": {"FLX_CLI_FORCE_STREAM": "1", "FLX_CLI_BLOCK_BYTES": "6000", "FLX_CLI_PINFLATE_MIN": "1", "FLX_CLI_PINFLATE_CHUNK": "2000"},
    # the one-process-per-GPU path with a single rank: RCCL communicator, flx_rank_and_cut_comm, part files
    "rank-env": {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "FLX_COMM_ID_FILE": "/tmp/flx_test_comm.id"},
}


@pytest.mark.parametrize("mode", sorted(MODES))
def test_cli_matches_reference_binary_byte_for_byte(mode):
    gold = _e2e_checks.load_e2e()
    inp = _e2e_checks.Inputs()
    with tempfile.TemporaryDirectory() as td:
        def w(name, data):
            path = os.path.join(td, name)
            open(path, "wb").write(data)
            return path
        fa = w("ref.fasta", _cases.fasta_bytes(inp.contigs))
        p1 = w("sr_1.fastq", _cases.fastq_bytes(inp.sr[0]))
        p2 = w("sr_2.fastq", _cases.fastq_bytes(inp.sr[1]))
        c1 = w("c1.fastq", _cases.c1_fastq_bytes())
        pin = w("synth_phred.fastq", _cases.long_fastq_bytes(inp.preads))
        oin = w("odd.fastq", _cases.odd_fastq_bytes(inp.preads))
        kin = w("synth_kmer.fastq", _cases.long_fastq_bytes(inp.kreads))
        cin = w("cr_at_eof.fasta", _cases.fasta_cr_at_eof_bytes(inp.kreads))
        n = 0
        todo = sorted(gold.items())
        if mode != "default":  # the other ingest / rank paths: every third golden plus all trim/split ones
            todo = [kv for i, kv in enumerate(todo) if i % 3 == 0 or "trimsplit" in kv[0] or kv[0].startswith(("trim", "split"))]
        for key, g in todo:
            parts = key.split("|")
            args = list(g["args"])
            # the golden argv holds the generator's temp paths; map them back by flag
            for i, a in enumerate(args):
                if a == "-a":
                    args[i + 1] = os.path.join(FIX, "test_reference.fasta") if parts[0] in ("sort", "trim", "split") else fa
                elif a == "-1":
                    args[i + 1] = os.path.join(FIX, "test_reference_1.fastq.gz") if parts[0] in ("sort", "trim", "split") else p1
                elif a == "-2":
                    args[i + 1] = os.path.join(FIX, "test_reference_2.fastq.gz") if parts[0] in ("sort", "trim", "split") else p2
            if key == "bad_fastq":
                inpath = os.path.join(FIX, "test_bad_fastq.fastq")
            elif parts[0] == "odd_format":
                # multi-line FASTQ, CRLF, blank lines: the kseq record grammar (src/kseq.h:176-224); a last line that is a bare '\r'
                inpath = cin if parts[1] == "cr_at_eof" else oin
            elif parts[0] == "c1":
                inpath = c1  # BASELINE.json configs[0]: 10k reads x 5 kbp --min_length 1000 --keep_percent 90
            elif parts[0] in ("sort", "trim", "split"):
                inpath = os.path.join(FIX, "test_%s.fastq" % parts[0])
            elif parts[0] == "synth_phred":
                inpath = pin
            else:
                inpath = kin
            rc, out, keep, err = run(args + [inpath], td, MODES[mode])
            assert rc == g["rc"], (key, err)
            assert len(out) == g["stdout_len"] and hashlib.sha256(out).hexdigest() == g["stdout_sha256"], key
            # reference stderr lines hold the generator's temp file names in the hashing section; compare the rest
            want = [l for l in g["stderr"] if "Hashing" not in l]
            got = [l for l in keep if "Hashing" not in l]
            assert got == want, (key, got, want)
            n += 1
        assert n == len(todo)


def test_cli_fasta_input_and_gz(tmp_path):
    """FASTA in -> FASTA out with a reference; gzipped input; FASTA without reference is an error."""
    fx = os.path.join(FIX, "test_sort.fasta")
    rc, out, keep, err = run(["-a", os.path.join(FIX, "test_reference.fasta"), "--target_bases", "10000", fx], str(tmp_path))
    assert rc == 0 and out.startswith(b">test_sort_1") and out.count(b">") == 2 and b"+\n" not in out
    rc, out, keep, err = run(["--target_bases", "10000", fx], str(tmp_path))
    assert rc == 1 and "Error: FASTA input not supported without an external reference" in err
    import gzip
    gz = tmp_path / "in.fastq.gz"
    gz.write_bytes(gzip.compress(open(os.path.join(FIX, "test_sort.fastq"), "rb").read()))
    rc, out, keep, err = run(["--target_bases", "5000", str(gz)], str(tmp_path))
    assert rc == 0 and out.startswith(b"@test_sort_2")
    # gzip input is streamed block by block by default; the in-memory path must give the same bytes
    rc2, out2, keep2, err2 = run(["--target_bases", "5000", str(gz)], str(tmp_path), {"FLX_CLI_NO_STREAM": "1"})
    rc3, out3, keep3, err3 = run(["--target_bases", "5000", str(gz)], str(tmp_path), {"FLX_CLI_BLOCK_BYTES": "700"})
    assert (rc2, out2, keep2) == (rc, out, keep) and (rc3, out3, keep3) == (rc, out, keep)


def test_cli_window_size_narrows_like_the_reference(tmp_path):
    """The reference stores --window_size in an int (src/arguments.h:90): 2^32 + 100 scores with a window of 100."""
    fx = os.path.join(FIX, "test_sort.fastq")
    a = run(["--window_size", "100", "--target_bases", "10001", "--verbose", fx], str(tmp_path))
    b = run(["--window_size", "4294967396", "--target_bases", "10001", "--verbose", fx], str(tmp_path))
    c = run(["--window_size", "250", "--target_bases", "10001", "--verbose", fx], str(tmp_path))
    assert a[0] == 0 and a == b and a[3] != c[3]


def test_cli_verbose_scores(tmp_path):
    """--verbose final scores (2 decimals): Phred 0.00 / 70.70 / 61.54."""
    rc, out, keep, err = run(["--target_bases", "100000", "--verbose", os.path.join(FIX, "test_sort.fastq")], str(tmp_path))
    assert rc == 0
    rows = {l.split("\t")[0].strip(): l.split("\t") for l in err.split("\n") if l.startswith("test_sort_") and "\t" in l}
    assert [rows["test_sort_%d" % i][4].strip() for i in (1, 2, 3)] == ["0.00", "70.70", "61.54"]


@pytest.mark.parametrize("mode", ["default", "blocks", "gpus2", "gpus3"])
def test_cli_verbose_matches_reference_stderr(tmp_path, mode):
    """--verbose stderr, character for character after the hashing section (tests/golden/verbose.json, written by
    make_verbose_golden.py from the reference binary): per-read blocks with `bad ranges` / `child ranges`
    (src/read.cpp:169-194), the blank line after them (main.cpp:129), the score table with host-libm final scores.
    gpus2 / gpus3: `--gpus N` (forked ranks over the loopback communicator on one GPU) — every rank leaves the blocks and the
    table rows of its reads in the job's directory, rank 0 prints them in file order; on the error paths rank 0 alone scores
    the reads in front of the error."""
    gold = json.load(open(os.path.join(_cases.GOLDEN, "verbose.json")))
    prefix = []
    if mode.startswith("gpus"):
        shim_dir = os.path.join(ROOT, "tests", "shim")
        subprocess.check_call(["make", "-s", "-C", shim_dir])
        MODES[mode] = {"FLX_RCCL_LIB": os.path.join(shim_dir, "libloopback_rccl.so"), "FLX_DEVICE": "0"}
        prefix = ["--gpus", mode[4:]]
    for key, g in sorted(gold.items()):
        args = [os.path.join(FIX, "test_reference.fasta") if a == "REF" else a for a in g["args"]]
        inp = g["input"]
        if inp.startswith("CAT:"):  # fixtures concatenated (the error-path cases: duplicate names, mixed formats)
            path = str(tmp_path / ("cat_" + inp[4:].replace("+", "_")))
            with open(path, "wb") as f:
                for part in inp[4:].split("+"):
                    f.write(open(os.path.join(FIX, part), "rb").read())
        else:
            path = os.path.join(FIX, inp)
        rc, out, keep, err = run(prefix + args + [path], str(tmp_path), MODES[mode])
        assert rc == g["rc"], (key, err)
        cut = err.find("16-mers\n\n")
        got = err[cut + len("16-mers\n\n"):] if cut >= 0 else err
        assert got == g["stderr"], (key, got, g["stderr"])


def test_cli_gpus_flag_single(tmp_path):
    """--gpus 1 is the plain run; --gpus 0 / non-numeric are argument errors."""
    fq = os.path.join(FIX, "test_sort.fastq")
    a = run(["--target_bases", "10000", fq], str(tmp_path))
    b = run(["--target_bases", "10000", "--gpus", "1", fq], str(tmp_path))
    assert a[0] == b[0] == 0 and a[1] == b[1]
    assert run(["--target_bases", "10000", "--gpus", "0", fq], str(tmp_path))[0] == 1


def test_cli_unit_suffixes(tmp_path):
    """The reference's test/test_unit_suffixes.py:39-205, success cases: k/kb/m/mb/g/gb, case-insensitive, decimals, on
    every flag that takes them; the stderr target line shows the expanded value (C locale here: no grouping)."""
    fq = os.path.join(FIX, "test_sort.fastq")
    asm = os.path.join(FIX, "test_reference.fasta")
    for val, shown in (("10k", "10000"), ("10kb", "10000"), ("1g", "1000000000"), ("1gb", "1000000000"), ("0.01m", "10000"),
                       ("0.01mb", "10000"), ("10K", "10000"), ("10KB", "10000"), ("0.01M", "10000"), ("0.01MB", "10000"),
                       ("3.5mb", "3500000"), ("1kb", "1000"), ("1k", "1000"), ("3.5k", "3500"), ("10000", "10000")):
        rc, out, keep, err = run(["--target_bases", val, fq], str(tmp_path))
        assert rc == 0 and ("target: %s bp" % shown) in err, (val, err)
    same = None
    for flags, n_out in ((["--min_length", "1k"], 3), (["--min_length", "1g"], 0), (["--max_length", "10k"], 3),
                         (["--max_length", "1gb"], 3), (["-l", "5k"], 3), (["-L", "10k"], 3),
                         (["-a", asm, "--split", "1k"], None), (["-a", asm, "--split", "1g"], None)):
        rc, out, keep, err = run(flags + [fq], str(tmp_path))
        assert rc == 0, (flags, err)
        if n_out is not None:
            assert out.count(b"\n@test_sort_") + out.startswith(b"@test_sort_") == n_out, flags
    # a suffix is the same run as the plain number
    a = run(["--target_bases", "10k", fq], str(tmp_path))[1]
    b = run(["--target_bases", "10000", fq], str(tmp_path))[1]
    assert a == b and len(a) > 0


def test_cli_output_sinks(tmp_path):
    """The output pass writes a regular-file stdout with concurrent pwrite()s at precomputed offsets, a pipe or an
    O_APPEND file in order: the same bytes every way (the reference writes through std::cout, src/main.cpp:263-313)."""
    fq = tmp_path / "c1.fastq"
    fq.write_bytes(_cases.c1_fastq_bytes())
    args = [BIN, "--target_bases", "20000000", str(fq)]
    env = dict(os.environ, LANG="C", LC_ALL="C")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    piped = subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert piped.returncode == 0 and len(piped.stdout) > 30_000_000
    f1 = tmp_path / "direct.out"
    with open(f1, "wb") as fh:
        assert subprocess.run(args, stdout=fh, stderr=subprocess.PIPE, env=env).returncode == 0
    assert f1.read_bytes() == piped.stdout
    f2 = tmp_path / "append.out"
    f2.write_bytes(b"HEAD\n")
    with open(f2, "ab") as fh:
        assert subprocess.run(args, stdout=fh, stderr=subprocess.PIPE, env=env).returncode == 0
    assert f2.read_bytes() == b"HEAD\n" + piped.stdout
    f3 = tmp_path / "offset.out"   # a file descriptor that is not at offset 0 and whose file is longer than the output
    f3.write_bytes(b"x" * 7 + b"y" * (len(piped.stdout) + 100))
    with open(f3, "r+b") as fh:
        fh.seek(7)
        assert subprocess.run(args, stdout=fh, stderr=subprocess.PIPE, env=env).returncode == 0
    got = f3.read_bytes()
    assert got[:7] == b"x" * 7 and got[7:7 + len(piped.stdout)] == piped.stdout and got[7 + len(piped.stdout):] == b"y" * 100
    with open(f1, "wb") as fh:
        assert subprocess.run(args, stdout=fh, stderr=subprocess.PIPE, env=dict(env, FLX_CLI_ORDERED_OUTPUT="1")).returncode == 0
    assert f1.read_bytes() == piped.stdout


@pytest.mark.parametrize("gpus", ["2", "3"])
def test_cli_output_sinks_with_forked_ranks(tmp_path, gpus):
    """`--gpus N` (ranks forked over the loopback communicator on one GPU): when the job's stdout is a regular file every rank writes
    its passed records straight into it at its own offset (round-3 review, item 8: one output file, no part files, nothing written
    twice) — a fresh file, a descriptor that is not at offset 0 of a longer file, k-mer mode with children; a pipe, an O_APPEND file and
    FLX_CLI_ORDERED_OUTPUT still go through part files that rank 0 streams out.  The same bytes as one rank every way, and the
    descriptor's position ends behind them (a second command appended to the same descriptor lands behind the first)."""
    shim_dir = os.path.join(ROOT, "tests", "shim")
    subprocess.check_call(["make", "-s", "-C", shim_dir])
    env = dict(os.environ, LANG="C", LC_ALL="C", FLX_RCCL_LIB=os.path.join(shim_dir, "libloopback_rccl.so"), FLX_DEVICE="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    fq = tmp_path / "c1.fastq"
    fq.write_bytes(_cases.c1_fastq_bytes())
    inp = _e2e_checks.Inputs()
    kfq = tmp_path / "kmer.fastq"
    kfq.write_bytes(_cases.long_fastq_bytes(inp.kreads))
    fa = tmp_path / "ref.fasta"
    fa.write_bytes(_cases.fasta_bytes(inp.contigs))
    jobs = {"phred": ["--target_bases", "20000000", str(fq)],
            "kmer_children": ["-a", str(fa), "--trim", "--split", "100", "--keep_percent", "80", str(kfq)]}
    for tag, args in jobs.items():
        one = subprocess.run([BIN] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert one.returncode == 0 and len(one.stdout) > 100_000, (tag, one.stderr[-300:])
        cmd = [BIN, "--gpus", gpus] + args
        piped = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert piped.returncode == 0 and piped.stdout == one.stdout, (tag, piped.stderr[-300:])
        f1 = tmp_path / (tag + "_direct.out")
        with open(f1, "wb") as fh:
            assert subprocess.run(cmd, stdout=fh, stderr=subprocess.PIPE, env=env).returncode == 0
            fh.write(b"TAIL\n")  # (the parent's descriptor: its position must be behind the ranks' output)
        assert f1.read_bytes() == one.stdout + b"TAIL\n", tag
        f2 = tmp_path / (tag + "_append.out")
        f2.write_bytes(b"HEAD\n")
        with open(f2, "ab") as fh:
            assert subprocess.run(cmd, stdout=fh, stderr=subprocess.PIPE, env=env).returncode == 0
        assert f2.read_bytes() == b"HEAD\n" + one.stdout, tag
        f3 = tmp_path / (tag + "_offset.out")
        f3.write_bytes(b"x" * 7 + b"y" * (len(one.stdout) + 100))
        with open(f3, "r+b") as fh:
            fh.seek(7)
            assert subprocess.run(cmd, stdout=fh, stderr=subprocess.PIPE, env=env).returncode == 0
            assert os.lseek(fh.fileno(), 0, os.SEEK_CUR) == 7 + len(one.stdout)
        got = f3.read_bytes()
        assert got[:7] == b"x" * 7 and got[7:7 + len(one.stdout)] == one.stdout and got[7 + len(one.stdout):] == b"y" * 100, tag
        with open(f1, "wb") as fh:
            assert subprocess.run(cmd, stdout=fh, stderr=subprocess.PIPE, env=dict(env, FLX_CLI_ORDERED_OUTPUT="1")).returncode == 0
        assert f1.read_bytes() == one.stdout, tag
        # two commands in a row on one descriptor (a shell's `{ a; b; } > file`)
        f4 = tmp_path / (tag + "_twice.out")
        with open(f4, "wb") as fh:
            for _ in range(2):
                assert subprocess.run(cmd, stdout=fh, stderr=subprocess.PIPE, env=env).returncode == 0
        assert f4.read_bytes() == one.stdout * 2, tag


@pytest.mark.parametrize("gpus", [None, "2"])
def test_cli_sink_that_cannot_be_written(tmp_path, gpus):
    """A sink that does not take the output (/dev/full: every write fails with ENOSPC) ends the job with status 1 and one message —
    one rank, and ranks whose parts rank 0 streams out; nothing stays behind in the temporary directory."""
    shim_dir = os.path.join(ROOT, "tests", "shim")
    subprocess.check_call(["make", "-s", "-C", shim_dir])
    tmp = tmp_path / "tmp"
    tmp.mkdir()
    env = dict(os.environ, LANG="C", LC_ALL="C", FLX_RCCL_LIB=os.path.join(shim_dir, "libloopback_rccl.so"), FLX_DEVICE="0", TMPDIR=str(tmp))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    fq = tmp_path / "c1.fastq"
    fq.write_bytes(_cases.c1_fastq_bytes())
    cmd = [BIN] + (["--gpus", gpus] if gpus else []) + ["--target_bases", "20000000", str(fq)]
    with open("/dev/full", "wb") as fh:
        res = subprocess.run(cmd, stdout=fh, stderr=subprocess.PIPE, env=env, timeout=300)
    assert res.returncode == 1, res.stderr[-300:]
    assert res.stderr.count(b"Error: could not write the output") == 1, res.stderr[-300:]
    assert list(tmp.iterdir()) == []


# ---- raw stderr: every progress update of the hashing and the scoring loop ----------------------------------------------------
_RAW = {}


def _raw_inputs(td):
    """A reference of 1.7 Mbp in three contigs (wrapped lines), 24 000 error-free short-read pairs of 100 bp and 260 long reads
    of ~9 kbp: every loop that prints "\\r  ... (N bp)" (src/kmers.cpp:123-126, src/main.cpp:119-123) crosses its 483 611-base step
    several times.  Written once per session."""
    import gzip
    import numpy as np
    from filtlong_amd import synth
    if _RAW:
        return _RAW
    import atexit
    import shutil
    d = tempfile.mkdtemp(prefix="flx_raw_")
    atexit.register(shutil.rmtree, d, True)
    ref = synth.bases_read(synth.STREAM_REF, 0, 0, 1_700_000)
    cuts = [0, 600_011, 1_250_007, 1_700_000]
    with open(os.path.join(d, "ref.fasta"), "wb") as f:
        for k in range(3):
            f.write(b">contig_%d some words\n" % (k + 1))
            s = ref[cuts[k]:cuts[k + 1]].tobytes()
            f.write(b"\n".join(s[i:i + 70] for i in range(0, len(s), 70)) + b"\n")
    comp = np.zeros(256, dtype=np.uint8)
    comp[list(b"ACGT")] = list(b"TGCA")
    n_pairs = 12_000
    starts = (synth.mix(synth.SEED, synth.STREAM_START, np.arange(n_pairs, dtype=np.uint64) + np.uint64(1 << 41), 0)
              % np.uint64(len(ref) - 500)).astype(np.int64)
    # ~3.4x of the first 350 kbp only (every 16-mer there is seen often enough for the multi-copy rule), the rest of the pairs anywhere
    starts[: n_pairs // 2] %= 350_000
    idx = starts[:, None] + np.arange(100)[None, :]
    for name, rows in (("sr_1.fastq", ref[idx]), ("sr_2.fastq", comp[ref[idx + 350]][:, ::-1])):
        with open(os.path.join(d, name), "wb") as f:
            for k, r in enumerate(rows):
                f.write(b"@p%d/%s\n%s\n+\n%s\n" % (k, name[3:4].encode(), r.tobytes(), b"I" * 100))
    with open(os.path.join(d, "sr_1.fastq"), "rb") as f, gzip.open(os.path.join(d, "sr_1.fastq.gz"), "wb", compresslevel=3) as g:
        g.write(f.read())
    with open(os.path.join(d, "ref.fasta"), "rb") as f, gzip.open(os.path.join(d, "ref.fasta.gz"), "wb", compresslevel=6) as g:
        g.write(f.read())
    lens = np.maximum(synth.lengths(260, first=900) * 9 // 10, 300)
    with open(os.path.join(d, "reads.fastq"), "wb") as f:
        for i, L in enumerate(lens):
            f.write(b"@read_%d\n%s\n+\n%s\n" % (i, synth.seq_read(900 + i, int(L), ref).tobytes(), synth.qual_read(900 + i, int(L)).tobytes()))
    _RAW.update(dir=d, n_bases=int(lens.sum()))
    return _RAW


_RAW_FLAGS = {
    "phred": ["--target_bases", "1500000"],
    "assembly": ["-a", "ref.fasta", "--keep_percent", "70"],
    "assembly_gz": ["-a", "ref.fasta.gz", "--min_mean_q", "60", "--target_bases", "1400000"],
    "short_reads": ["-1", "sr_1.fastq", "-2", "sr_2.fastq", "--keep_percent", "80"],
    "short_reads_gz": ["-1", "sr_1.fastq.gz", "-2", "sr_2.fastq", "--trim", "--split", "300", "--target_bases", "1000000"],
    "assembly_and_short_reads": ["-a", "ref.fasta", "-1", "sr_1.fastq", "-2", "sr_2.fastq", "--trim", "--split", "500", "--keep_percent", "90"],
}
_RAW_ENVS = {
    "default": {},
    "many_batches": {"FLX_CLI_REF_BATCH_BYTES": "150000", "FLX_CLI_CHUNK_BYTES": "300000"},
    "in_memory": {"FLX_CLI_NO_STREAM": "1"},
    "small_blocks": {"FLX_CLI_FORCE_STREAM": "1", "FLX_CLI_BLOCK_BYTES": "70000", "FLX_CLI_PINFLATE_MIN": "1", "FLX_CLI_PINFLATE_CHUNK": "30000"},
    "gpus2": None,
    "gpus3": None,
}


@pytest.mark.parametrize("flags", sorted(_RAW_FLAGS))
def test_cli_stderr_is_the_references_byte_for_byte(tmp_path, flags):
    """Round-4 review: stderr was compared as a terminal shows it (the last "\\r" segment of a line).  Here RAW, against the
    reference binary, on inputs whose hashing and scoring loops print several progress updates — every scoring mode, the streamed
    reference reader with whole and with tiny batches / blocks, the in-memory reader, gzip references, and two and three forked ranks
    (which index only their byte range of the reads: the progress lines are replayed by rank 0 from the gathered lengths)."""
    import _oracle
    if not os.path.exists(_oracle.REF_FILTLONG):
        pytest.skip("reference binary not built")
    raw = _raw_inputs(str(tmp_path))
    argv = _RAW_FLAGS[flags] + ["reads.fastq"]
    env0 = dict(os.environ, LANG="C", LC_ALL="C")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env0.pop(k, None)
    ref = subprocess.run([_oracle.REF_FILTLONG] + argv, cwd=raw["dir"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env0)
    assert ref.returncode == 0 and ref.stderr.count(b"\r") >= (4 if flags == "phred" else 8), ref.stderr[-300:]
    shim_dir = os.path.join(ROOT, "tests", "shim")
    for name, extra in sorted(_RAW_ENVS.items()):
        prefix = []
        if extra is None:
            subprocess.check_call(["make", "-s", "-C", shim_dir])
            extra = {"FLX_RCCL_LIB": os.path.join(shim_dir, "libloopback_rccl.so"), "FLX_DEVICE": "0"}
            prefix = ["--gpus", name[4:]]
        new = subprocess.run([BIN] + prefix + argv, cwd=raw["dir"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(env0, **extra))
        assert new.returncode == 0, (name, new.stderr[-500:])
        assert new.stderr == ref.stderr, (name, new.stderr[-600:], ref.stderr[-600:])
        assert hashlib.sha256(new.stdout).hexdigest() == hashlib.sha256(ref.stdout).hexdigest(), name


def test_cli_rank_that_cannot_write_its_share(tmp_path):
    """Advisor, round 4: a rank > 0 whose writes into the job's common output file fail (a full disk under its share; here forced
    with FLX_CLI_FAIL_WRITE_RANK) used to leave before the exchange in which every rank learns of it — the others then waited in
    that exchange for ever.  Now every rank reaches it, the job ends with status 1 and rank 0's one message; the same for rank 0
    itself and for part files (a pipe as the job's stdout)."""
    shim_dir = os.path.join(ROOT, "tests", "shim")
    subprocess.check_call(["make", "-s", "-C", shim_dir])
    tmp = tmp_path / "tmp"
    tmp.mkdir()
    env = dict(os.environ, LANG="C", LC_ALL="C", FLX_RCCL_LIB=os.path.join(shim_dir, "libloopback_rccl.so"), FLX_DEVICE="0", TMPDIR=str(tmp))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    fq = tmp_path / "c1.fastq"
    fq.write_bytes(_cases.c1_fastq_bytes()[:6_000_000].rsplit(b"\n@", 1)[0] + b"\n")
    cmd = [BIN, "--gpus", "3", "--target_bases", "2000000", str(fq)]
    for failing in ("0", "1", "2"):
        for sink in ("file", "pipe"):
            e = dict(env, FLX_CLI_FAIL_WRITE_RANK=failing)
            if sink == "file":
                with open(tmp_path / "cut.fastq", "wb") as fh:
                    res = subprocess.run(cmd, stdout=fh, stderr=subprocess.PIPE, env=e, timeout=120)
            else:
                res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=120)
            assert res.returncode == 1, (failing, sink, res.stderr[-300:])
            assert res.stderr.count(b"Error: could not write the output") == 1, (failing, sink, res.stderr[-300:])
            assert b"ended early" not in res.stderr
            assert list(tmp.iterdir()) == []


@pytest.mark.parametrize("gpus", ["2", "3", "8"])
def test_cli_gzip_input_streamed_by_every_rank(tmp_path, gpus):
    """Round 5: with several ranks a gzip input is no longer inflated whole into every rank's memory — rank 0 counts the records, every
    rank streams the file block by block, checks every record and scores only its share (cli/main.cpp).  Same stdout and stderr as one
    rank, with whole blocks and with blocks of a few records (shares that begin and end inside blocks and access-point units), Phred
    and k-mer mode with children, more ranks than records, and an input whose last record is cut off."""
    import gzip
    shim_dir = os.path.join(ROOT, "tests", "shim")
    subprocess.check_call(["make", "-s", "-C", shim_dir])
    base = dict(os.environ, LANG="C", LC_ALL="C", FLX_RCCL_LIB=os.path.join(shim_dir, "libloopback_rccl.so"), FLX_DEVICE="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        base.pop(k, None)
    c1 = _cases.c1_fastq_bytes()[:3_000_000].rsplit(b"\n@", 1)[0] + b"\n"
    few = b"".join(c1.split(b"\n")[i] + b"\n" for i in range(8))  # two records
    cut = c1[:200_000]  # ends inside a record
    files = {}
    for name, data in (("c1", c1), ("few", few), ("cut", cut)):
        p = tmp_path / (name + ".fastq.gz")
        with gzip.open(p, "wb", compresslevel=4) as g:
            g.write(data)
        files[name] = str(p)
    inp = _e2e_checks.Inputs()
    fa = tmp_path / "ref.fasta"
    fa.write_bytes(_cases.fasta_bytes(inp.contigs))
    kin = tmp_path / "kmer.fastq.gz"
    with gzip.open(kin, "wb") as g:
        g.write(_cases.long_fastq_bytes(inp.kreads))
    runs = [(["--target_bases", "700000", files["c1"]], 0), (["--min_length", "100", "--keep_percent", "50", files["few"]], 0),
            (["--target_bases", "50000", files["cut"]], 1), (["-a", str(fa), "--trim", "--split", "100", "--keep_percent", "80", str(kin)], 0)]
    for argv, want_rc in runs:
        one = subprocess.run([BIN] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=base)
        assert one.returncode == want_rc, one.stderr[-300:]
        for blocks in ({}, {"FLX_CLI_BLOCK_BYTES": "40000", "FLX_CLI_SPAN_BYTES": "15000", "FLX_CLI_PINFLATE_MIN": "1", "FLX_CLI_PINFLATE_CHUNK": "9000"}):
            env = dict(base, FLX_CLI_TIMING="1", FLX_CLI_RANK_STREAM="1", **blocks)  # (=1: also below the size from which it is the default)
            res = subprocess.run([BIN, "--gpus", gpus] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
            assert res.returncode == want_rc, (argv, blocks, res.stderr[-400:])
            assert b"count pass (rank 0)" in res.stderr, (argv, blocks)  # the streamed path was taken
            shown = b"\n".join(l for l in res.stderr.replace(b"\r", b"\n").split(b"\n") if not l.startswith(b"[timing]") and b"[timing]" not in l)
            ref = b"\n".join(l for l in one.stderr.replace(b"\r", b"\n").split(b"\n"))
            assert res.stdout == one.stdout, (argv, blocks)
            assert [l for l in shown.split(b"\n") if l.strip()] == [l for l in ref.split(b"\n") if l.strip()], (argv, blocks, res.stderr[-400:])
            old = subprocess.run([BIN, "--gpus", gpus] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(base, FLX_CLI_RANK_STREAM="0", **blocks), timeout=300)
            assert old.returncode == want_rc and old.stdout == one.stdout and old.stderr == one.stderr, (argv, blocks, "in memory")
