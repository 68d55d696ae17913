#!/bin/bash
# round 3: stage clocks of the command line on the 2 GB FASTQ bench.py's end-to-end extra uses (100k reads) and on 10 GB
cd $GRAFT_REPO_ROOT
export LANG=C LC_ALL=C
for N in 100000 500000; do
B=$(tools/gen_fastq $N /tmp/e2e.fastq); T=$((B/2))
echo "== $N reads, $B bases, $(stat -c %s /tmp/e2e.fastq) bytes"
for rep in 1 2; do
S=$(date +%s%N); FLX_CLI_TIMING=1 filtlong_amd/bin/filtlong --target_bases $T /tmp/e2e.fastq > /tmp/e2e.out 2> /tmp/e2e.err; E=$(date +%s%N)
echo "run $rep: $(python -c "print(($E-$S)/1e9)") s"
done
tr '\r' '\n' < /tmp/e2e.err | grep timing
done
rm -f /tmp/e2e.fastq /tmp/e2e.out
