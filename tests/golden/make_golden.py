#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Run in the build container (where /root/reference exists) after `make -C oracle`:

    python tests/golden/make_golden.py

Sources of truth:
  * oracle/_ref/ref_probe  — our probe harness linked against the reference's own objects
                             (read.o kmers.o arguments.o misc.o); dumps Read fields as hex floats.
  * oracle/_ref/filtlong   — the reference binary, for end-to-end stdout/stderr.
Inputs are either the reference's own fixtures (copied to tests/golden/ref_fixtures/ because
/root/reference does not exist on the GPU box) or seeded synthetic reads from tests/_cases.py.
Outputs: probe_*.json (per-read fields), e2e_*.json (ordered output names, child coordinates,
stderr summary lines, sha256 of stdout).
"""
import hashlib
import json
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import _cases  # noqa: E402
import _oracle  # noqa: E402

REF_TEST = "/root/reference/test"
FIX = os.path.join(HERE, "ref_fixtures")
FIXTURE_FILES = ["test_sort.fastq", "test_sort.fasta", "test_trim.fastq", "test_split.fastq", "test_reference.fasta",
                 "test_reference_1.fastq.gz", "test_reference_2.fastq.gz", "test_bad_fastq.fastq"]


def hexify(reads):
    out = []
    for r in reads:
        d = dict(r)
        for k in ("length_score", "mean_q", "window_q"):
            d[k] = float(d[k]).hex()
        d["children"] = hexify(d.get("children", []))
        out.append(d)
    return out


def probe_case(reads, pkw, ref_args):
    p = _oracle.make_params(**pkw)
    argv = _oracle.params_to_argv(p) + ref_args
    if not (p.trim or p.split_set or p.min_length_set or p.max_length_set or p.min_mean_q_set or p.min_window_q_set):
        argv += ["--target_bases", "1000000000"]  # the reference insists on one threshold (arguments.cpp:337-343)
    empty, _, res = _oracle.ref_probe(reads, argv)
    return {"params": pkw, "kmers_empty": empty, "reads": hexify(res)}


def e2e_case(input_path, args, cwd):
    rc, out, err = _oracle.run_ref_filtlong(args + [input_path], cwd=cwd)
    names = [l[1:].split()[0].decode() for l in out.split(b"\n")[0::4] if l[:1] == b"@"] if out[:1] == b"@" else \
            [l[1:].split()[0].decode() for l in out.split(b"\n") if l[:1] == b">"]
    keep = [l.strip() for l in err.replace("\r", "\n").split("\n")
            if any(t in l for t in ("target:", "keeping", "not enough", "already fall", "after ", "Error", "16-mers"))]
    return {"args": args, "rc": rc, "names": names, "stderr": keep, "stdout_sha256": hashlib.sha256(out).hexdigest(),
            "stdout_len": len(out)}


def main():
    assert _oracle.have_ref(), "build oracle/_ref first (make -C oracle)"
    os.makedirs(FIX, exist_ok=True)
    for f in FIXTURE_FILES:
        shutil.copyfile(os.path.join(REF_TEST, f), os.path.join(FIX, f))
    with open(os.path.join(FIX, "README.md"), "w") as f:
        f.write("Test *data* files copied verbatim from the reference repository (rrwick/Filtlong, test/, GPLv3)\n"
                "so that the golden-vector tests can run on the GPU box, where /root/reference does not exist.\n"
                "No reference source code is copied.  Regenerate with tests/golden/make_golden.py.\n")

    asm = os.path.join(FIX, "test_reference.fasta")
    sr1 = os.path.join(FIX, "test_reference_1.fastq.gz")
    sr2 = os.path.join(FIX, "test_reference_2.fastq.gz")
    refmodes = {"phred": [], "asm": ["-a", asm], "short": ["-1", sr1, "-2", sr2]}

    # ---- 1. reference fixtures through the probe ---------------------------------------------
    gold = {}
    for fx in ("test_sort.fastq", "test_trim.fastq", "test_split.fastq"):
        reads = _oracle.read_fastx(os.path.join(FIX, fx))
        for mode, rargs in refmodes.items():
            psets = [dict()]
            if mode != "phred":
                psets += [dict(trim=True), dict(trim=True, split=250)]
                psets += [dict(split=s) for s in (250, 201, 200, 175, 75, 51, 50, 25)]
            for pkw in psets:
                key = "%s|%s|%s" % (fx, mode, json.dumps(pkw, sort_keys=True))
                gold[key] = probe_case(reads, pkw, rargs)
    with open(os.path.join(HERE, "probe_fixtures.json"), "w") as f:
        json.dump(gold, f, indent=0, sort_keys=True)

    # ---- 2. seeded synthetic Phred reads ------------------------------------------------------
    gold = {}
    reads = _cases.phred_reads()
    for pkw in _cases.PHRED_PARAM_SETS:
        gold[json.dumps(pkw, sort_keys=True)] = probe_case(reads, pkw, [])
    with open(os.path.join(HERE, "probe_synth_phred.json"), "w") as f:
        json.dump(gold, f, indent=0, sort_keys=True)

    # ---- 3. seeded synthetic k-mer reads (assembly and short-read references) -------------------
    gold = {}
    with tempfile.TemporaryDirectory() as td:
        contigs = _cases.synth_reference()
        fa = os.path.join(td, "ref.fasta")
        with open(fa, "wb") as f:
            f.write(_cases.fasta_bytes(contigs))
        r1, r2 = _cases.short_read_pairs(contigs)
        p1, p2 = os.path.join(td, "sr_1.fastq"), os.path.join(td, "sr_2.fastq")
        with open(p1, "wb") as f:
            f.write(_cases.fastq_bytes(r1))
        with open(p2, "wb") as f:
            f.write(_cases.fastq_bytes(r2))
        reads = _cases.kmer_reads(contigs)
        for mode, rargs in (("asm", ["-a", fa]), ("short", ["-1", p1, "-2", p2]), ("both", ["-a", fa, "-1", p1])):
            for pkw in _cases.KMER_PARAM_SETS:
                gold["%s|%s" % (mode, json.dumps(pkw, sort_keys=True))] = probe_case(reads, pkw, rargs)

        # ---- 4. k-mer set membership -----------------------------------------------------------
        import numpy as np
        rng = np.random.RandomState(5)
        sets = {}
        for mode, rargs in (("asm", ["-a", fa]), ("short", ["-1", p1, "-2", p2]),
                            ("fix_asm", ["-a", asm]), ("fix_short", ["-1", sr1, "-2", sr2])):
            # candidate queries: every 16-mer the oracle says is present or was seen at all, plus random ones
            ks = _oracle.KmerSet()
            if mode == "asm":
                ks.add_assembly(contigs); seen_src = contigs
            elif mode == "short":
                ks.add_short_reads(r1); ks.add_short_reads(r2); seen_src = r1 + r2
            elif mode == "fix_asm":
                seqs = [s for _, s, _ in _oracle.read_fastx(asm)]
                ks.add_assembly(seqs); seen_src = seqs
            else:
                s1 = [s for _, s, _ in _oracle.read_fastx(sr1)]
                s2 = [s for _, s, _ in _oracle.read_fastx(sr2)]
                ks.add_short_reads(s1); ks.add_short_reads(s2); seen_src = s1 + s2
            seen = _oracle.KmerSet()
            seen.add_assembly(seen_src)  # all 16-mers occurring at least once, both strands
            q = np.unique(np.concatenate([seen.dump(), rng.randint(0, 2 ** 32, size=20000, dtype=np.uint64)
                                          .astype(np.uint32)]))
            _, present, _ = _oracle.ref_probe([], ["--target_bases", "1"] + rargs, kmer_queries=q)
            pres = np.array(sorted(k for k, v in present.items() if v), dtype=np.uint32)
            sets[mode] = {"n_queries": int(len(q)), "n_present": int(len(pres)),
                          "present_sha256": hashlib.sha256(pres.tobytes()).hexdigest(),
                          "queries_sha256": hashlib.sha256(q.tobytes()).hexdigest(),
                          "oracle_size": len(ks)}
        gold["__sets__"] = sets
        import gzip
        with gzip.open(os.path.join(HERE, "probe_synth_kmer.json.gz"), "wt") as f:
            json.dump(gold, f, sort_keys=True)

        # ---- 5. end-to-end through the reference binary -------------------------------------------
        e2e = {}
        srt = os.path.join(FIX, "test_sort.fastq")
        for mode, rargs in refmodes.items():
            for t in (100000, 10001, 10000, 5001, 5000, 1):
                e2e["sort|%s|%d" % (mode, t)] = e2e_case(srt, rargs + ["--target_bases", str(t)], td)
        for mode in ("asm", "short"):
            e2e["trim|%s" % mode] = e2e_case(os.path.join(FIX, "test_trim.fastq"), refmodes[mode] + ["--trim"], td)
            for s in (250, 201, 200, 175, 75, 51, 50, 25):
                e2e["split|%s|%d" % (mode, s)] = e2e_case(os.path.join(FIX, "test_split.fastq"),
                                                          refmodes[mode] + ["--split", str(s)], td)
        e2e["bad_fastq"] = e2e_case(os.path.join(FIX, "test_bad_fastq.fastq"), ["--target_bases", "1000"], td)

        # synthetic long reads, many cut positions
        preads = [r for r in _cases.phred_reads(weird=False) if len(r[1]) > 0]
        pin = os.path.join(td, "synth_phred.fastq")
        with open(pin, "wb") as f:
            f.write(_cases.long_fastq_bytes(preads))
        tot = sum(len(r[1]) for r in preads)
        for frac in (0.05, 0.2, 0.5, 0.8, 0.95):
            e2e["synth_phred|t%.2f" % frac] = e2e_case(pin, ["--target_bases", str(int(tot * frac))], td)
        for kp in (10, 50, 90):
            e2e["synth_phred|p%d" % kp] = e2e_case(pin, ["--keep_percent", str(kp), "--min_length", "300"], td)
        oin = os.path.join(td, "odd.fastq")  # multi-line records, CRLF, blank lines (kseq grammar)
        with open(oin, "wb") as f:
            f.write(_cases.odd_fastq_bytes(preads))
        e2e["odd_format|t0.50"] = e2e_case(oin, ["--target_bases", str(tot // 2)], td)
        e2e["synth_phred|weights"] = e2e_case(pin, ["--keep_percent", "60", "--length_weight", "2.5",
                                                    "--mean_q_weight", "0.5", "--window_q_weight", "3"], td)
        e2e["synth_phred|cutoffs"] = e2e_case(pin, ["--min_mean_q", "90", "--min_window_q", "80", "--max_length",
                                                    "15000", "--window_size", "100"], td)
        kin = os.path.join(td, "synth_kmer.fastq")
        kreads = [r for r in reads if len(r[1]) > 0]
        with open(kin, "wb") as f:
            f.write(_cases.long_fastq_bytes(kreads))
        ktot = sum(len(r[1]) for r in kreads)
        for mode, rargs in (("asm", ["-a", fa]), ("short", ["-1", p1, "-2", p2])):
            e2e["synth_kmer|%s|plain" % mode] = e2e_case(kin, rargs + ["--target_bases", str(ktot // 2)], td)
            e2e["synth_kmer|%s|trimsplit" % mode] = e2e_case(kin, rargs + ["--trim", "--split", "100", "--keep_percent",
                                                                          "70"], td)
            e2e["synth_kmer|%s|split40" % mode] = e2e_case(kin, rargs + ["--trim", "--split", "40", "--min_length",
                                                                        "200", "--target_bases", str(ktot // 3)], td)
        cin = os.path.join(td, "cr_at_eof.fasta")  # last line a bare '\r' without a newline: kseq keeps it (src/kseq.h:141-146)
        with open(cin, "wb") as f:
            f.write(_cases.fasta_cr_at_eof_bytes(kreads))
        e2e["odd_format|cr_at_eof"] = e2e_case(cin, ["-a", fa, "--target_bases", str(ktot // 2)], td)
        # engineered Bloom false positive (3 sightings are enough when all 13 bits were pre-set; kmers.cpp:148-155)
        import numpy as np
        f1, f2, target, control = _cases.bloom_fp_case()
        b1, b2 = os.path.join(td, "bfp_1.fastq"), os.path.join(td, "bfp_2.fastq")
        with open(b1, "wb") as f:
            f.write(_cases.fastq_bytes(f1))
        with open(b2, "wb") as f:
            f.write(_cases.fastq_bytes(f2))
        L = _oracle.lib()
        q = np.array([target, control, L.flo_start_kmer_rev(_cases.kmer_to_seq(target)),
                      L.flo_start_kmer_rev(_cases.kmer_to_seq(control))], dtype=np.uint32)
        _, present, _ = _oracle.ref_probe([], ["--target_bases", "1", "-1", b1, "-2", b2], kmer_queries=q)
        with open(os.path.join(HERE, "bloom_fp.json"), "w") as f:
            json.dump({"target": target, "control": control, "present": {str(int(k)): bool(v) for k, v in present.items()}},
                      f, indent=0, sort_keys=True)

        # BASELINE.json configs[0] (C1): 10k reads x 5 kbp, Phred-only --min_length 1000 --keep_percent 90
        c1 = os.path.join(td, "c1.fastq")
        with open(c1, "wb") as f:
            f.write(_cases.c1_fastq_bytes())
        r = e2e_case(c1, ["--min_length", "1000", "--keep_percent", "90"], td)
        r["names"] = r["names"][:20] + ["..."]  # ~9000 names: the stdout digest pins them
        e2e["c1|min_length1000|keep90"] = r
        with open(os.path.join(HERE, "e2e.json"), "w") as f:
            json.dump(e2e, f, indent=0, sort_keys=True)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
