#!/bin/bash
cd $GRAFT_REPO_ROOT
export LANG=C LC_ALL=C
N=500000
B=$(tools/gen_fastq $N /tmp/e2e.fastq); T=$((B/2))
echo "== $N reads, $B bases"
filtlong_amd/bin/filtlong --target_bases $T /tmp/e2e.fastq > /tmp/e2e.out 2> /dev/null
for TH in 16 32 64; do
python - <<PY
import subprocess, time, os
t0=time.time(); 
p=subprocess.run(["filtlong_amd/bin/filtlong","--target_bases","$T","/tmp/e2e.fastq"],stdout=open("/tmp/e2e.out","wb"),stderr=subprocess.PIPE,env=dict(os.environ,FLX_CLI_TIMING="1",FLX_CLI_THREADS="$TH"))
t1=time.time()
print("threads $TH: start %.3f end %.3f total %.3f" % (t0%100000, t1%100000, t1-t0))
for l in p.stderr.decode().replace("\r","\n").split("\n"):
    if "timing" in l: print("  "+l[l.index("[timing]"):][:62])
PY
done
rm -f /tmp/e2e.fastq /tmp/e2e.out
