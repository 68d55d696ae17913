#!/bin/bash
# host threads of the command line (page-in, parse, pack, output): 10 GB FASTQ, wall seconds by FLX_CLI_THREADS
cd $GRAFT_REPO_ROOT
export LANG=C LC_ALL=C
nproc
N=500000
S=$(date +%s%N); B=$(tools/gen_fastq $N /tmp/e2e.fastq); T=$((B/2)); E=$(date +%s%N)
echo "== $N reads, $B bases, generated in $(python -c "print(($E-$S)/1e9)") s"
filtlong_amd/bin/filtlong --target_bases $T /tmp/e2e.fastq > /tmp/e2e.out 2> /dev/null
sha256sum /tmp/e2e.out | cut -c1-16
for TH in 8 16 32 64 128; do
python - <<PY
import subprocess, time, os
best=1e9
for rep in range(3):
    try: os.remove("/tmp/e2e.out")
    except OSError: pass
    t0=time.time()
    p=subprocess.run(["filtlong_amd/bin/filtlong","--target_bases","$T","/tmp/e2e.fastq"],stdout=open("/tmp/e2e.out","wb"),stderr=subprocess.DEVNULL,env=dict(os.environ,FLX_CLI_THREADS="$TH"))
    best=min(best,time.time()-t0)
print("threads $TH: %.3f s  (%.2f Gbases/s) rc %d" % (best, $B/best/1e9, p.returncode))
PY
sha256sum /tmp/e2e.out | cut -c1-16
done
for TH in 16 64; do
rm -f /tmp/e2e.out
FLX_CLI_THREADS=$TH FLX_CLI_TIMING=1 filtlong_amd/bin/filtlong --target_bases $T /tmp/e2e.fastq > /tmp/e2e.out 2> /tmp/e2e.err
echo "-- stage clocks, $TH threads"; tr '\r' '\n' < /tmp/e2e.err | grep timing
done
rm -f /tmp/e2e.fastq /tmp/e2e.out
