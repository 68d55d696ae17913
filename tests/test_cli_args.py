"""The C++ command line (filtlong_amd/bin/filtlong) mirrors the reference's argument handling: same validation
order, messages and exit codes (reference src/arguments.cpp:298-393, pinned by its test/test_error_messages.py
and test/test_unit_suffixes.py).  Argument errors are raised before any GPU work, so these run on CPU."""
import os
import subprocess

import pytest

import _cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "filtlong_amd", "bin", "filtlong")
INPUT = os.path.join(_cases.FIXTURES, "test_sort.fastq")
ASM = os.path.join(_cases.FIXTURES, "test_reference.fasta")


@pytest.fixture(scope="module", autouse=True)
def built():
    if not os.path.exists(BIN):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "filtlong_amd", "csrc"), "-s", "-j8"])
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "filtlong_amd", "cli"), "-s"])


def run(*args):
    p = subprocess.run([BIN] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, LANG="C", LC_ALL="C"))
    return p.returncode, p.stdout, p.stderr.decode()


CASES = [
    (["INPUT"], "Error: no thresholds set"),
    (["--target_bases", "1000", "BAD_FILENAME"], "Error: cannot find file"),
    (["-a", "BAD_FILENAME", "--target_bases", "1000", "INPUT"], "Error: cannot find file"),
    (["--target_bases", "0", "INPUT"], "Error: the value for --target_bases must be a positive integer"),
    (["--target_bases", "-10", "INPUT"], "Error: the value for --target_bases must be a positive integer"),
    (["--keep_percent", "0", "INPUT"], "Error: the value for --keep_percent must be greater than 0 and less than 100"),
    (["--keep_percent", "100", "INPUT"], "Error: the value for --keep_percent must be greater than 0 and less than 100"),
    (["--keep_percent", "111.1", "INPUT"], "Error: the value for --keep_percent must be greater than 0 and less than 100"),
    (["--min_length", "0", "INPUT"], "Error: the value for --min_length must be a positive integer"),
    (["--min_length", "-10", "INPUT"], "Error: the value for --min_length must be a positive integer"),
    (["--min_mean_q", "0", "INPUT"], "Error: the value for --min_mean_q must be greater than 0"),
    (["--min_window_q", "0", "INPUT"], "Error: the value for --min_window_q must be greater than 0"),
    (["--trim", "INPUT"], "Error: assembly or read reference is required to use --trim"),
    (["--split", "250", "INPUT"], "Error: assembly or read reference is required to use --split"),
    (["-a", "ASSEMBLY", "--split", "0", "INPUT"], "Error: the value for --split must be a positive integer"),
    (["-a", "ASSEMBLY", "--split", "-10", "INPUT"], "Error: the value for --split must be a positive integer"),
    (["--min_length", "1000", "--window_size", "0", "INPUT"], "Error: the value for --window_size must be a positive integer"),
    (["--min_length", "1000", "--window_size", "-10", "INPUT"], "Error: the value for --window_size must be a positive integer"),
    # the reference narrows --window_size to an int before it validates it (src/arguments.cpp:295,389)
    (["--min_length", "1000", "--window_size", "2147483648", "INPUT"], "Error: the value for --window_size must be a positive integer"),
    (["-l", "-10", "INPUT"], "Error: the value for --min_length must be a positive integer"),
    (["-L", "-10", "INPUT"], "Error: the value for --max_length must be a positive integer"),
    (["-q", "0", "INPUT"], "Error: the value for --min_mean_q must be greater than 0"),
    (["--length_weight", "-1", "--target_bases", "5", "INPUT"], "received invalid value type"),
    (["--target_bases", "12x", "INPUT"], "received invalid value '12x'"),
    (["--target_bases=100", "INPUT"], "Error: flag could not be matched: target_bases=100"),
    # unit suffixes (reference test/test_unit_suffixes.py:157-209)
    (["--target_bases", "10xyz", "INPUT"], "invalid value"),
    (["--target_bases", "k", "INPUT"], "invalid value"),
    (["--target_bases", "-10k", "INPUT"], "Error: the value for --target_bases must be a positive integer"),
    (["--min_length", "-5kb", "INPUT"], "Error: the value for --min_length must be a positive integer"),
    (["-l", "-10k", "INPUT"], "Error: the value for --min_length must be a positive integer"),
    (["-L", "-5kb", "INPUT"], "Error: the value for --max_length must be a positive integer"),
]


@pytest.mark.parametrize("argv,msg", CASES)
def test_error_messages(argv, msg):
    argv = [INPUT if a == "INPUT" else ASM if a == "ASSEMBLY" else a for a in argv]
    rc, out, err = run(*argv)
    assert rc == 1 and msg in err and out == b""


def test_help_and_version():
    rc, out, err = run()
    assert rc == 0 and "usage:" in err and "Filtlong:" in err
    rc, out, err = run("--help")
    assert rc == 0 and "usage:" in err
    # the menu of the reference, byte for byte, as recorded from its binary with stdout not a terminal (tests/golden/ref_suite.json)
    import json
    gold = json.load(open(os.path.join(_cases.GOLDEN, "ref_suite.json")))
    menus = [inv["stderr"] for inv in gold["invocations"] if "usage:" in inv["stderr"]]
    assert menus and err.replace(BIN, "PROG", 1) == menus[0].replace("/root/repo/oracle/_ref/filtlong", "PROG", 1)
    assert out == b""
    rc, out, err = run("--version")
    assert rc == 0 and out == b"Filtlong v0.3.1\n"


def test_rejected_command_lines_match_the_reference_binary():
    """tests/golden/arg_errors.json (make_arg_error_golden.py, from the reference binary): unknown flags, missing and malformed values,
    the `--` terminator, the default reader's quirks (src/args.h:1609-1627: an empty value keeps the default, a number beyond long
    long saturates), range checks and their order — same exit code, same stderr, nothing on stdout."""
    import json
    gold = json.load(open(os.path.join(_cases.GOLDEN, "arg_errors.json")))
    assert len(gold) >= 80
    for g in gold:
        argv = [INPUT if a == "INPUT" else ASM if a == "ASSEMBLY" else a for a in g["argv"]]
        p = subprocess.run([BIN] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd="/tmp", env=dict(os.environ, LANG="C", LC_ALL="C"))
        err = p.stderr.decode(errors="replace").replace(INPUT, "INPUT").replace(ASM, "ASSEMBLY")
        assert (p.returncode, len(p.stdout), err) == (g["rc"], g["stdout_len"], g["stderr"]), g["argv"]


def test_random_rejected_command_lines_match_the_reference_binary():
    """Random soups of flags, values and positionals (600 seeded cases) through the reference binary (oracle/_ref/filtlong, built where
    /root/reference exists; skipped elsewhere) and this one: whenever the reference rejects the command line before it reads anything —
    parser errors IN PARSE ORDER (a second positional is reported where it stands), validation errors in the order of
    src/arguments.cpp:298-393, missing files — exit code and stderr are the same.  Command lines the reference accepts need a GPU here
    and are left to tests/test_gpu_fuzz.py."""
    import random
    import zlib
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "filtlong")
    if not os.path.exists(ref_bin):
        pytest.skip("oracle/_ref/filtlong not built (needs /root/reference)")
    flags = ["-t", "--target_bases", "-p", "--keep_percent", "-l", "--min_length", "-L", "--max_length", "-q", "--min_mean_q", "-w",
             "--min_window_q", "-a", "--assembly", "-1", "--short_1", "-2", "--short_2", "--length_weight", "--mean_q_weight",
             "--window_q_weight", "--trim", "--split", "--window_size", "--verbose", "--version", "-h", "--help", "--", "-x", "--bogus", "-",
             "--trim=1", "-t5", "-p50", "-l1k", "--split=5", "-tabc"]
    vals = ["5", "0", "-1", "1k", "1.5k", "3g", "abc", "", "1e3", "100", "99.9", "100.1", "0.0", "+5", " 5", "5 ", "2147483648",
            "99999999999999999999", "1.5", "-0", "0x10", "1kb", "1MB", "kb", ".", ASM, "nonexist"]
    env = dict(os.environ, LANG="C", LC_ALL="C")

    def run_bin(b, argv):
        p = subprocess.run([b] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd="/tmp")
        e = p.stderr.decode(errors="replace")
        return p.returncode, len(p.stdout), "USAGE" if ("usage:" in e or "Filtlong:" in e) else e

    compared = 0
    for i in range(600):
        rng = random.Random(zlib.crc32(b"argfuzz-%d" % i))
        argv = []
        for _ in range(rng.randrange(0, 6)):
            argv.append(rng.choice(flags))
            if rng.random() < 0.75:
                argv.append(rng.choice(vals))
        inp = rng.choice(["x", INPUT, INPUT, None])
        if inp:
            argv.insert(rng.randrange(0, len(argv) + 1) if rng.random() < 0.3 else len(argv), inp)
        r = run_bin(ref_bin, argv)
        if (r[0] == 0 and r[2] != "USAGE" and r[1] == 0) or "Scoring long reads" in r[2] or "Hashing" in r[2]:
            continue  # the reference started to work: not an argument matter
        assert run_bin(BIN, argv) == r, argv
        compared += 1
    assert compared >= 500


def _menu_on_a_terminal(binary, width, args):
    """stderr of `binary args` with a pseudo-terminal of `width` columns as stdout (the reference's formatter asks STDOUT for its
    width, src/arguments.cpp:131-133); run under the same argv[0] so that the usage line is the same"""
    import fcntl
    import pty
    import struct
    import termios
    m, s = pty.openpty()
    try:
        fcntl.ioctl(s, termios.TIOCSWINSZ, struct.pack("HHHH", 40, width, 0, 0))
        p = subprocess.run(["./filtlong"] + list(args), stdout=s, stderr=subprocess.PIPE, cwd=os.path.dirname(binary),
                           env=dict(os.environ, LANG="C", LC_ALL="C", LD_LIBRARY_PATH=os.path.join(ROOT, "filtlong_amd", "lib")))
    finally:
        os.close(m)
        os.close(s)
    return p.returncode, p.stderr


def test_help_menu_is_the_references_at_every_terminal_width():
    """src/arguments.cpp:126-221 + the formatter of src/args.h: the layout depends on the terminal's width (indents 1-4, wrapped
    descriptions, flags whose help moves to the next line).  Against the reference binary, widths 1..250 and beyond, --help and no
    argument at all."""
    import _oracle
    if not _oracle.have_ref():
        pytest.skip("reference binary not built")
    widths = list(range(1, 251)) + [400, 1000, 65535]
    for w in widths:
        for args in ((), ("--help",), ("-h",)):
            if args != ("--help",) and w % 10:
                continue
            assert _menu_on_a_terminal(_oracle.REF_FILTLONG, w, args) == _menu_on_a_terminal(BIN, w, args), (w, args)
