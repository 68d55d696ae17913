#!/usr/bin/env python3
"""Record every command line the REFERENCE'S OWN black-box test suite (/root/reference/test/test_*.py, 93 tests) runs,
together with what the reference binary (oracle/_ref/filtlong) answers: exit code, stderr, and the bytes it wrote to the
redirected output file.  Nothing of the suite's source is copied: its modules are imported where they lie, with
subprocess.Popen replaced by a recorder, and each test's own verdict under the reference binary is stored as well.

Run in the build container (where /root/reference exists) after `make -C oracle`:
    python tests/golden/make_ref_suite_golden.py        -> tests/golden/ref_suite.json
tests/test_gpu_ref_suite.py replays every recorded invocation against filtlong_amd/bin/filtlong on the GPU box and requires
the same exit code, the same stderr (outside the path-bearing hashing lines) and byte-identical output files — so every
assertion the suite makes about the reference's answers holds for the new binary, and its pass/fail vector is the same."""
import hashlib
import importlib.util
import io
import json
import os
import subprocess
import sys
import tempfile
import unittest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_TEST = "/root/reference/test"
REF_BIN_TEMPLATE = "/root/reference/bin/filtlong"
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "filtlong")
FIX = os.path.join(HERE, "ref_fixtures")

records = []
current = {"test": None}
real_popen = subprocess.Popen
workdir = tempfile.mkdtemp(prefix="flx_ref_suite_")


class Recorder:
    def __init__(self, command, **kw):
        pid_tag = "TEMP_" + str(os.getpid())
        template = command.replace(REF_BIN_TEMPLATE, "BIN").replace(REF_TEST + "/", "FIXDIR/").replace(pid_tag, "TEMPOUT")
        real = template.replace("BIN", REF_BIN).replace("FIXDIR/", FIX + "/").replace("TEMPOUT", os.path.join(workdir, "out"))
        for f in os.listdir(workdir):
            os.remove(os.path.join(workdir, f))
        p = real_popen(real, stdout=subprocess.PIPE, stderr=subprocess.PIPE, shell=True, cwd=workdir,
                       env=dict(os.environ, LANG="C", LC_ALL="C"))
        self._out, self._err = p.communicate()
        self.returncode = p.returncode
        files = {}
        for f in sorted(os.listdir(workdir)):
            data = open(os.path.join(workdir, f), "rb").read()
            files[f] = {"len": len(data), "sha256": hashlib.sha256(data).hexdigest()}
            # the suite reads the output file at the name it chose: put a copy there
            target = [x for x in command.split() if pid_tag in x]
            if target:
                open(target[0].replace(">", ""), "wb").write(data)
        records.append({"test": current["test"], "command": template, "rc": p.returncode, "stdout_len": len(self._out),
                        "stdout_sha256": hashlib.sha256(self._out).hexdigest(), "stderr": self._err.decode(errors="replace"),
                        "files": files})

    def communicate(self):
        return self._out, self._err


def main():
    verdicts = {}
    os.chdir(workdir)
    for name in ("test_sort", "test_trim", "test_split", "test_error_messages", "test_unit_suffixes"):
        spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF_TEST, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.subprocess.Popen = Recorder  # the module's own `subprocess` name: every run_command goes through the recorder
        suite = unittest.defaultTestLoader.loadTestsFromModule(mod)
        for case in suite:
            for t in case:
                current["test"] = name + "." + t.id().split(".")[-1]
                res = unittest.TestResult()
                t.run(res)
                verdicts[current["test"]] = "pass" if res.wasSuccessful() else "fail"
    subprocess.Popen = real_popen
    out = {"n_tests": len(verdicts), "verdicts_with_reference_binary": verdicts, "invocations": records}
    with open(os.path.join(HERE, "ref_suite.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    npass = sum(v == "pass" for v in verdicts.values())
    print("tests %d (pass %d, fail %d under LANG=C), invocations %d" % (len(verdicts), npass, len(verdicts) - npass, len(records)))


if __name__ == "__main__":
    main()
