cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kmer.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r06_call3_tests.log
bash tools/prof_kmer.sh 1000000 c3 > gpurun_out/r06_call3_prof_q.log 2>&1
mkdir -p gpurun_out/prof_kmer_q && mv gpurun_out/prof_kmer/*.txt gpurun_out/prof_kmer_q/
FLX_KMER_COVER=w bash tools/prof_kmer.sh 1000000 c3 > gpurun_out/r06_call3_prof_w.log 2>&1
mkdir -p gpurun_out/prof_kmer_w && mv gpurun_out/prof_kmer/*.txt gpurun_out/prof_kmer_w/
