#!/bin/bash
# End-to-end (file -> stdout) wall clock of the drop-in CLI vs the reference binary on the same FASTQ, same box.
# usage: tools/bench_e2e.sh [n_reads] ; synthetic Phred-only reads (gamma lengths), --target_bases 50 %
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
N=${1:-100000}
cd $R
python - <<PY
import sys, numpy as np
sys.path.insert(0, "$R")
from filtlong_amd import synth
n = $N
L = synth.lengths(n)
with open("/tmp/e2e.fastq", "wb") as f:
    seq_unit = b"ACGT" * 50001
    for i in range(n):
        f.write(b"@r%d\n" % i); f.write(seq_unit[:int(L[i])]); f.write(b"\n+\n"); f.write(synth.qual_read(i, int(L[i])).tobytes()); f.write(b"\n")
print("bases", int(L.astype(np.int64).sum()))
PY
B=$(python -c "import sys; sys.path.insert(0,'$R'); from filtlong_amd import synth; import numpy as np; print(int(synth.lengths($N).astype(np.int64).sum())//2)")
ls -la /tmp/e2e.fastq
export LANG=C LC_ALL=C
python - <<PY
import subprocess, time
for rep in range(2):
    for name, exe, out in (("reference", "$R/oracle/_ref/filtlong", "/tmp/ref"), ("filtlong-amd", "$R/filtlong_amd/bin/filtlong", "/tmp/amd")):
        t = time.time()
        subprocess.run([exe, "--target_bases", "$B", "/tmp/e2e.fastq"], stdout=open(out + ".out", "wb"), stderr=open(out + ".err", "wb"), check=True)
        dt = time.time() - t
        print("%-13s %7.2f s wall  (%.1f Mbases/s end to end)" % (name, dt, 2 * $B / dt / 1e6))
PY
cmp /tmp/ref.out /tmp/amd.out && echo "stdout identical ($(wc -c < /tmp/amd.out) bytes)"
grep -E "target|keeping" /tmp/ref.err /tmp/amd.err
