#!/usr/bin/env python3
"""Golden --verbose stderr of the REAL reference binary (oracle/_ref/filtlong) on its own fixtures, including the
`bad ranges = ...` / `child ranges = ...` lines of Read::print_verbose_read_info (src/read.cpp:169-194) under --trim/--split.
Run in the build container after `make -C oracle`:   python tests/golden/make_verbose_golden.py
Stored: everything the reference prints to stderr AFTER the reference-hashing section (whose lines hold file paths)."""
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
FIX = os.path.join(HERE, "ref_fixtures")
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "filtlong")

CASES = {
    "phred|sort|t10000": (["--verbose", "--target_bases", "10000"], "test_sort.fastq"),
    "asm|trim": (["--verbose", "-a", "REF", "--trim"], "test_trim.fastq"),
    "asm|split250": (["--verbose", "-a", "REF", "--split", "250"], "test_split.fastq"),
    "asm|trim|split100|t8000": (["--verbose", "-a", "REF", "--trim", "--split", "100", "--target_bases", "8000"], "test_split.fastq"),
    "asm|split1|keep50": (["--verbose", "-a", "REF", "--split", "1", "--keep_percent", "50"], "test_trim.fastq"),
    "asm|split100000": (["--verbose", "-a", "REF", "--split", "100000"], "test_split.fastq"),
    # error paths: the reference prints every read's block inside its pass-1 loop, so the blocks of the reads in front of the
    # failing record (for a duplicate name: that record's too) precede the error line (src/main.cpp:108-117)
    "error|bad_fastq": (["--verbose", "--target_bases", "1000"], "test_bad_fastq.fastq"),
    "error|duplicate_name": (["--verbose", "--target_bases", "1000"], "CAT:test_sort.fastq+test_sort.fastq"),
    "error|duplicate_name|trim": (["--verbose", "-a", "REF", "--trim", "--split", "100"], "CAT:test_split.fastq+test_trim.fastq+test_split.fastq"),
    "error|mixed_formats": (["--verbose", "-a", "REF", "--target_bases", "1000"], "CAT:test_sort.fastq+test_sort.fasta"),
    "error|fasta_without_reference": (["--verbose", "--target_bases", "1000"], "test_sort.fasta"),
}


def input_path(spec, tmpdir):
    """A fixture name, or CAT:a+b+c = those fixtures concatenated into a temporary file."""
    if not spec.startswith("CAT:"):
        return os.path.join(FIX, spec)
    path = os.path.join(tmpdir, "cat_" + spec[4:].replace("+", "_"))
    with open(path, "wb") as out:
        for part in spec[4:].split("+"):
            out.write(open(os.path.join(FIX, part), "rb").read())
    return path


def after_hashing(err):
    cut = err.find("16-mers\n\n")
    return err[cut + len("16-mers\n\n"):] if cut >= 0 else err


def main():
    import tempfile
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for key, (args, fixture) in CASES.items():
            argv = [os.path.join(FIX, "test_reference.fasta") if a == "REF" else a for a in args]
            p = subprocess.run([REF_BIN] + argv + [input_path(fixture, td)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                               env=dict(os.environ, LANG="C", LC_ALL="C"))
            out[key] = {"args": args, "input": fixture, "rc": p.returncode, "stderr": after_hashing(p.stderr.decode())}
    with open(os.path.join(HERE, "verbose.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote", len(out), "cases")


if __name__ == "__main__":
    main()
