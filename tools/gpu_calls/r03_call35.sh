#!/bin/bash
cd $GRAFT_REPO_ROOT
export LANG=C LC_ALL=C
timeout 900 python -m pytest tests/test_gpu_cli.py tests/test_gpu_stream.py tests/test_gpu_comm2.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | cut -c1-400 | tail -4
for N in 100000 500000; do
B=$(tools/gen_fastq $N /tmp/e2e.fastq); T=$((B/2))
python - <<PY
import subprocess, time, os, hashlib
best=1e9
for rep in range(4):
    if os.path.exists("/tmp/e2e.out"): os.unlink("/tmp/e2e.out")
    t0=time.time()
    p=subprocess.run(["filtlong_amd/bin/filtlong","--target_bases","$T","/tmp/e2e.fastq"],stdout=open("/tmp/e2e.out","wb"),stderr=subprocess.DEVNULL)
    best=min(best,time.time()-t0)
print("$N reads: %.3f s  (%.2f Gbases/s) rc %d sha %s" % (best, $B/best/1e9, p.returncode, hashlib.sha256(open("/tmp/e2e.out","rb").read()).hexdigest()[:16]))
PY
done
rm -f /tmp/e2e.fastq /tmp/e2e.out
