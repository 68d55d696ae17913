#!/bin/bash
# round 4, call 7: the whole GPU suite (incl. C3 / C4 at BASELINE size with the locus path), smoke with the build record, XCD-slice microbenchmark
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --durations=12 2>&1 | tail -30 | tee gpurun_out/r04_call7.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee -a gpurun_out/r04_call7.log
./tools/xcdslice 2>&1 | tee gpurun_out/r04_xcdslice.txt
