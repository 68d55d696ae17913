#!/usr/bin/env python3
"""Argument errors of the REAL reference binary (oracle/_ref/filtlong): exit code and stderr for command lines its parser or its
validation rejects (src/args.h ParseCLI, src/arguments.cpp:222-394) — unknown flags, missing and malformed values, the `--`
terminator, value-range checks and their order.  Run in the build container after `make -C oracle`:
    python tests/golden/make_arg_error_golden.py
Stored in tests/golden/arg_errors.json with INPUT / ASSEMBLY standing for the fixture paths; only rejected command lines are kept."""
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
FIX = os.path.join(HERE, "ref_fixtures")
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "filtlong")
I, A = "INPUT", "ASSEMBLY"

CASES = [
    ["-w", "5", "x"], ["--bogus", "5", "x"], ["-t"], ["--target_bases"], ["-t", "5", "a", "b"], ["--window_size", "abc", "-t", "5", "x"],
    ["-tx", "x"], ["-t", "5", "-x", "f"], ["-t5", "-p"], ["--keep_percent"], ["-p", "abc", "x"], ["-l", "1.5", "-t", "5", "x"],
    ["-l", "abc", "-t", "5", "x"], ["--split"], ["--split", "x", "-t", "1", "f"], ["-a"], ["-1"], ["-1", "a", "-t", "5", "x"],
    ["-2", "b", "-t", "5", "x"], ["x", "-t"], ["-t", "5", "--trim", "x", "y"], ["--min_mean_q", "1e", "-t", "5", "x"], ["-q", "", "-t", "5", "x"],
    ["--length_weight", "x", "-t", "5", "f"], ["-t", "5k5", "x"], ["-t", "kb", "x"], ["-t", "-5", "x"], ["--window_size", "0", "-t", "5", "x"],
    ["--window_size", "-3", "-t", "5", "x"], ["-p", "0", "x"], ["-p", "100", "x"], ["-p", "-1", "x"], ["-t", "0", "x"], ["-l", "0", "-t", "5", "x"],
    ["-L", "0", "-t", "5", "x"], ["x"], ["-t", "5"], ["--trim", "-t", "5", "x"], ["--split", "5", "-t", "5", "x"], ["-t", "5", "-"],
    ["-t", "5", "--", "x"], ["--", "-t"], ["-t", "5", "--verbose=1", "x"], ["--target_bases=5", "x"], ["-t=5", "x"], ["-tt", "x"],
    ["-t", "5", "-1", "a", "x"], ["--min_window_q", "abc", "-t", "1", "x"], ["--mean_q_weight", "-1", "-t", "1", "x"],
    ["--length_weight", "-0.1", "-t", "1", "x"], ["--window_q_weight", "-2", "-t", "1", "x"], ["-t", "1", "--split", "0", "-a", "r", "x"],
    ["-t", "1", "--split", "-1", "-a", "r", "x"], ["-t", "99999999999999999999", "x"], ["-l", "99999999999", "-t", "1", "x"],
    ["--window_size", "99999999999999999999", "-t", "1", "x"],
]
CASES += [["--window_size", v, "-t", "1", I] for v in ["99999999999999999999", "9223372036854775807", "-5", "0", "2147483648", "4294967296", "abc", "5 "]]
CASES += [["-p", v, I] for v in ["0", "100", "100.5", "-1", "abc", ""]]
CASES += [["-l", v, "-t", "1", I] for v in ["0", "-1", "abc", "2147483648", "3g"]] + [["-L", v, "-t", "1", I] for v in ["0", "-1", "3g"]]
CASES += [["--split", v, "-t", "1", "-a", A, I] for v in ["0", "-1", "abc", "3g"]]
CASES += [["--split", "5", "-t", "1", I], ["--trim", "-t", "1", I], [I], ["-q", "-1", I], ["--length_weight", "-1", "-t", "1", I],
          ["--length_weight", "abc", "-t", "1", I], ["-t", "0", I], ["-t", "-1", I], ["-a", "nonexist", "-t", "1", I],
          ["-1", "nonexist", "-2", A, "-t", "1", I], ["--min_mean_q", "abc", "-t", "1", I], ["-t", "1", I, I], ["-t", "1", "--", I, "--trim"]]


def main():
    out = []
    sub = {I: os.path.join(FIX, "test_sort.fastq"), A: os.path.join(FIX, "test_reference.fasta")}
    for argv in CASES:
        p = subprocess.run([REF_BIN] + [sub.get(a, a) for a in argv], stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd="/tmp",
                           env=dict(os.environ, LANG="C", LC_ALL="C"))
        err = p.stderr.decode(errors="replace")
        if p.returncode == 0 or not err.startswith("Error"):
            print("not an argument error, left out:", argv)
            continue
        for k, v in sub.items():
            err = err.replace(v, k)
        out.append({"argv": argv, "rc": p.returncode, "stdout_len": len(p.stdout), "stderr": err})
    with open(os.path.join(HERE, "arg_errors.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", len(out), "cases")


if __name__ == "__main__":
    main()
