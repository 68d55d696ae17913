#!/bin/bash
# round 4, call 8: PMC passes of the cover kernel, C3 and C4, 1e6 reads (requests by class, issue mix)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
bash tools/prof_kmer.sh 1000000 "c3 c4" > gpurun_out/r04_call8_prof.log 2>&1
grep -E "cover_w|FETCH|TCC" gpurun_out/r04_call8_prof.log | head -40
