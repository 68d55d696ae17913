#!/bin/bash
cd $GRAFT_REPO_ROOT
export LANG=C LC_ALL=C
for N in 100000 500000; do
B=$(tools/gen_fastq $N /tmp/e2e.fastq); T=$((B/2))
echo "== $N reads, $B bases"
filtlong_amd/bin/filtlong --target_bases $T /tmp/e2e.fastq > /tmp/e2e.out 2> /dev/null
for MB in 64 128 256 512 1024; do
python - <<PY
import subprocess, time, os
best=1e9
for rep in range(3):
    t0=time.time()
    p=subprocess.run(["filtlong_amd/bin/filtlong","--target_bases","$T","/tmp/e2e.fastq"],stdout=open("/tmp/e2e.out","wb"),stderr=subprocess.DEVNULL,env=dict(os.environ,FLX_CLI_CHUNK_MB="$MB"))
    best=min(best,time.time()-t0)
print("chunk $MB MiB: %.3f s  (%.2f Gbases/s) rc %d" % (best, $B/best/1e9, p.returncode))
PY
done
sha256sum /tmp/e2e.out | cut -c1-16
done
rm -f /tmp/e2e.fastq /tmp/e2e.out
