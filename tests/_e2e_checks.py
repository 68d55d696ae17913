"""Shared end-to-end checks against tests/golden/e2e.json (outputs of the real reference binary)."""
import json
import os

import _cases
import _oracle
import _pipeline

FIX = _cases.FIXTURES


def load_e2e():
    return json.load(open(os.path.join(_cases.GOLDEN, "e2e.json")))


def _stderr_num(lines, key):
    for l in lines:
        if key in l:
            digits = "".join(ch for ch in l.split(key, 1)[1].split("bp")[0] if ch.isdigit())
            return int(digits)
    return None


class Inputs:
    """Regenerates the inputs make_golden.py used (fixtures + seeded synthetic)."""

    def __init__(self):
        self.fix = {f: _oracle.read_fastx(os.path.join(FIX, f))
                    for f in ("test_sort.fastq", "test_trim.fastq", "test_split.fastq")}
        self.fix_asm = [s for _, s, _ in _oracle.read_fastx(os.path.join(FIX, "test_reference.fasta"))]
        self.fix_sr = [[s for _, s, _ in _oracle.read_fastx(os.path.join(FIX, f))]
                       for f in ("test_reference_1.fastq.gz", "test_reference_2.fastq.gz")]
        self.contigs = _cases.synth_reference()
        self.sr = list(_cases.short_read_pairs(self.contigs))
        self.preads = [r for r in _cases.phred_reads(weird=False) if len(r[1]) > 0]
        self.kreads = [r for r in _cases.kmer_reads(self.contigs) if len(r[1]) > 0]


def check_all(backend, inputs=None, only=None):
    """Every e2e golden: ordered output names (incl. child coordinates in the names), summary numbers."""
    gold = load_e2e()
    inp = inputs or Inputs()
    sets = {}

    def get_set(kind):
        if kind not in sets:
            if kind == "fix_asm": sets[kind] = backend.kmers(assembly=inp.fix_asm)
            elif kind == "fix_short": sets[kind] = backend.kmers(short_files=inp.fix_sr)
            elif kind == "syn_asm": sets[kind] = backend.kmers(assembly=inp.contigs)
            elif kind == "syn_short": sets[kind] = backend.kmers(short_files=inp.sr)
            else: sets[kind] = None
        return sets[kind]

    n = 0
    for key, g in sorted(gold.items()):
        if key == "bad_fastq" or key.startswith("c1|") or key.startswith("odd_format"):
            continue  # parser error path / the C1 configuration: covered by the CLI tests
        if only and not only(key):
            continue
        parts = key.split("|")
        pkw, kw, ref = _pipeline.golden_args_to_kwargs(g["args"])
        if parts[0] in ("sort", "trim", "split"):
            reads = inp.fix["test_%s.fastq" % parts[0]]
            ks = get_set({"phred": None, "asm": "fix_asm", "short": "fix_short"}[parts[1]])
        elif parts[0] == "synth_phred":
            reads, ks = inp.preads, None
        else:
            reads = inp.kreads
            ks = get_set("syn_asm" if parts[1] == "asm" else "syn_short")
        names, after_n, after_b, target, kept, outcome = _pipeline.run_filter(backend, reads, ks, pkw, **kw)
        assert names == g["names"], "%s: output names differ" % key
        t = _stderr_num(g["stderr"], "target:")
        if t is not None:
            assert target == t, key
        k = _stderr_num(g["stderr"], "keeping")
        if k is not None:
            assert kept == k and outcome == 3, key
        if any("not enough" in l for l in g["stderr"]):
            assert outcome == 1, key
        if any("already fall" in l for l in g["stderr"]):
            assert outcome == 2, key
        for l in g["stderr"]:
            if l.startswith("after "):
                cnt = int(l.split(":")[1].split("reads")[0].strip())
                assert after_n == cnt and after_b == _stderr_num([l], "("), key
        n += 1
    return n
