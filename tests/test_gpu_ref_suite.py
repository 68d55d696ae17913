"""The reference's own black-box suite (/root/reference/test/test_*.py: 93 tests, 100 command lines), replayed BY INVOCATION
against filtlong_amd/bin/filtlong: tests/golden/ref_suite.json holds every command line the suite runs and what the
reference binary answered (recorded by tests/golden/make_ref_suite_golden.py with the suite's modules imported where they
lie and subprocess.Popen replaced by a recorder; no suite source is copied).  The new binary must give the same exit code,
the same stderr and byte-identical output files for every one of them, so each assertion of the suite sees the same
answer and the suite's pass/fail vector (48 pass, 45 fail under LANG=C: the failing ones expect comma-grouped numbers,
SURVEY §8c) is the same for both binaries.  The --help / no-argument menu is compared byte for byte as well (round 5:
cli/args.h restates the reference's formatter; tests/test_cli_args.py holds it against the reference binary at every terminal width)."""
import hashlib
import json
import os
import subprocess

import pytest

import _cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "filtlong_amd", "bin", "filtlong")
FIX = _cases.FIXTURES


def shown(err, fixdir):
    """stderr RAW (every carriage-return progress update included: round 5), fixture directory normalised"""
    return [l.replace(fixdir + "/", "FIXDIR/") for l in err.split("\n")]


@pytest.mark.parametrize("ingest", ["default", "blocks"])
def test_every_invocation_of_the_reference_suite(tmp_path, ingest):
    gold = json.load(open(os.path.join(_cases.GOLDEN, "ref_suite.json")))
    assert gold["n_tests"] == 93 and len(gold["invocations"]) == 100
    env = dict(os.environ, LANG="C", LC_ALL="C")
    if ingest == "blocks":  # every input through the block-wise reader of the gzip path, in blocks of a few records
        env.update(FLX_CLI_FORCE_STREAM="1", FLX_CLI_BLOCK_BYTES="30000")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    # the recorded stderr holds the generator's fixture directory; find it from a recorded hashing line
    rec_fix = None
    for inv in gold["invocations"]:
        for l in inv["stderr"].split("\n"):
            if "/test_reference" in l and rec_fix is None:
                rec_fix = l.split("\r")[-1].strip().split(" ")[0].rsplit("/", 1)[0]
    assert rec_fix
    checked = 0
    for inv in gold["invocations"]:
        for f in os.listdir(tmp_path):
            os.remove(os.path.join(tmp_path, f))
        cmd = inv["command"].replace("BIN", BIN).replace("FIXDIR/", FIX + "/").replace("TEMPOUT", os.path.join(str(tmp_path), "out"))
        p = subprocess.run(cmd, shell=True, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        what = (inv["test"], inv["command"])
        assert p.returncode == inv["rc"], (what, p.stderr.decode()[-500:])
        err = p.stderr.decode(errors="replace")
        if "usage:" in inv["stderr"]:  # the help menu, byte for byte (only the program's own path differs: argv[0] on the usage line)
            assert err.replace(BIN, "PROG", 1) == inv["stderr"].replace("/root/repo/oracle/_ref/filtlong", "PROG", 1), what
        else:
            assert shown(err, FIX) == shown(inv["stderr"], rec_fix), (what, err, inv["stderr"])
        assert len(p.stdout) == inv["stdout_len"] and hashlib.sha256(p.stdout).hexdigest() == inv["stdout_sha256"], what
        files = {}
        for f in sorted(os.listdir(tmp_path)):
            data = open(os.path.join(tmp_path, f), "rb").read()
            files[f] = {"len": len(data), "sha256": hashlib.sha256(data).hexdigest()}
        assert files == inv["files"], what
        checked += 1
    assert checked == 100
