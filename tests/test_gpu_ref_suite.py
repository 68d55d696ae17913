"""The reference's own black-box suite (/root/reference/test/test_*.py: 93 tests, 100 command lines), replayed BY INVOCATION
against filtlong_amd/bin/filtlong: tests/golden/ref_suite.json holds every command line the suite runs and what the
reference binary answered (recorded by tests/golden/make_ref_suite_golden.py with the suite's modules imported where they
lie and subprocess.Popen replaced by a recorder; no suite source is copied).  The new binary must give the same exit code,
the same stderr and byte-identical output files for every one of them, so each assertion of the suite sees the same
answer and the suite's pass/fail vector (48 pass, 45 fail under LANG=C: the failing ones expect comma-grouped numbers,
SURVEY §8c) is the same for both binaries.  The only tolerated difference is the wording of the --help text (out of
scope), where the suite itself only looks for 'usage:' and 'Filtlong:'."""
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
    """stderr as a terminal would show it: the last carriage-return segment of every line, fixture directory normalised"""
    lines = [l.split("\r")[-1] for l in err.split("\n")]
    return [l.replace(fixdir + "/", "FIXDIR/") for l in lines]


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
        if "usage:" in inv["stderr"]:
            assert "usage:" in err and "Filtlong:" in err, what
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
