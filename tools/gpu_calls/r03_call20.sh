#!/bin/bash
# one lane per child: parity (kmer tests + full-size) and C4 timing, per-child vs words
set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kmer.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/c20_kmer_tests.log
cat gpurun_out/c20_kmer_tests.log
timeout 600 python bench.py --config c4 --no-cpu-baseline > gpurun_out/c20_c4.json 2> gpurun_out/c20_c4.err; tail -3 gpurun_out/c20_c4.err; cat gpurun_out/c20_c4.json
FLX_KMER_FOLD=words timeout 600 python bench.py --config c4 --no-cpu-baseline > gpurun_out/c20_c4_words.json 2>> gpurun_out/c20_c4.err; cat gpurun_out/c20_c4_words.json
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/c20_full_tests.log
cat gpurun_out/c20_full_tests.log
