#!/bin/bash
# round 4, call 15: C4 at BASELINE size with the order-24 text: bench line + PMC passes (1e7 reads)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/final
timeout 600 python bench.py --config c4 --steps 2 --warmup 1 > gpurun_out/final/bench_c4.json 2> gpurun_out/final/bench_c4.err; tail -c 1500 gpurun_out/final/bench_c4.json | head -c 700; echo
rm -rf gpurun_out/prof_kmer; bash tools/prof_kmer.sh 10000000 "c4" light > gpurun_out/final/prof_kmer_c4.out 2>&1; grep -E "TCC_|cover|FETCH" gpurun_out/final/prof_kmer_c4.out | head
