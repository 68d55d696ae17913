"""The integer-grid window fold (filtlong_amd/csrc/score_kmer.hip: GridTab, k_kmer_fold<.., GRID>) restated on the host —
tools/sim_fold_grid.cpp, the same regime logic the kernel runs — against the plain floating-point recurrence of the reference
(src/read.cpp:216-236 with qualities 0.0 / 1.0), bit for bit.  No GPU: this is the exhaustion argument behind the kernel (DESIGN.md
§4.3); the kernel itself is held against the oracle and against the floating-point kernel in tests/test_gpu_kmer.py and
tests/test_gpu_fullsize.py."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_regime_fold_equals_the_plain_recurrence(tmp_path):
    exe = str(tmp_path / "sim_fold_grid")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tools", "sim_fold_grid.cpp")])
    out = subprocess.run([exe, "400"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900).stdout.decode()
    m = re.search(r"exactness: (\d+) cases, (\d+) mismatches", out)
    assert m and int(m.group(1)) == 400 * 30 and int(m.group(2)) == 0, out[-2000:]
    # the default window: one regime from 2^-4 up (d* = d + 4 ulp on every binade), a tie at 2^-5
    assert "ws 250: groups [2^-10, 2^-5) [2^-4, 2^2)   ties at: 2^-5" in out
    # ... and what the rule for a regime's top is worth (the share of a wave's words in which some lane replays)
    rates = [float(x) for x in re.findall(r"ws 250: lane-words \d+, slow \d+ \([0-9.]+ %\); wave-words \d+, with a slow lane \d+ \(([0-9.]+) %\)", out)]
    assert len(rates) == 2 and rates[1] < 0.7 * rates[0] and rates[1] < 25.0, out[-1500:]
