#!/bin/bash
# round 4, call 3: PMC passes of the cover kernel with the locus path (C3, 1e6 reads) + the full k-mer tests at size
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
bash tools/prof_kmer.sh 1000000 "c3" > gpurun_out/r04_call3_prof.log 2>&1
tail -60 gpurun_out/r04_call3_prof.log
