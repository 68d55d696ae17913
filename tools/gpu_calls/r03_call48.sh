#!/bin/bash
# parallel duplicate-name check: CLI suites + 10 GB / 20 GB timing
cd $GRAFT_REPO_ROOT
export LANG=C LC_ALL=C
timeout 1200 python -m pytest tests/test_gpu_cli.py tests/test_gpu_fuzz.py tests/test_gpu_ref_suite.py tests/test_gpu_comm2.py -q -m gpu -x 2>&1 | tail -5
for N in 500000 1000000; do
B=$(tools/gen_fastq $N /tmp/e2e.fastq); T=$((B/2))
for rep in 1 2 3; do rm -f /tmp/e2e.out; S=$(date +%s%N); filtlong_amd/bin/filtlong --target_bases $T /tmp/e2e.fastq > /tmp/e2e.out 2>/dev/null; E=$(date +%s%N); echo "$N reads run $rep: $(python -c "print(($E-$S)/1e9)") s"; done
rm -f /tmp/e2e.out; FLX_CLI_TIMING=1 filtlong_amd/bin/filtlong --target_bases $T /tmp/e2e.fastq 2>&1 >/tmp/e2e.out | tr '\r' '\n' | grep -E "timing\] (record|parse|pack|output)"
sha256sum /tmp/e2e.out | cut -c1-16
done
rm -f /tmp/e2e.fastq /tmp/e2e.out
