cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out filtlong_amd/lib/exp
F="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wall -Wno-unused-function"
python3 tools/exp/make_cover_queue_stats.py
/opt/rocm/bin/hipcc $F -Ifiltlong_amd/csrc -Iinclude -c -o filtlong_amd/lib/exp/cq_stats.o tools/exp/cover_queue_stats.hip
objs=$(ls filtlong_amd/lib/obj/*.o | grep -v "/cover_queue.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o filtlong_amd/lib/exp/libfiltlong_hip_cqstats.so $objs filtlong_amd/lib/exp/cq_stats.o -ldl -lpthread
FLX_LIB_PATH=$PWD/filtlong_amd/lib/exp/libfiltlong_hip_cqstats.so timeout 600 python tools/exp/cq_stats.py 1000000 1 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06_call32.log
for prof in 0 1; do echo "== profile $prof"; timeout 300 python tools/bench_kmer.py --reads 2000000 --steps 3 --profile $prof 2>&1 | grep -v "Warning\|amdgpu.ids"; done | tee -a gpurun_out/r06_call32.log
timeout 1500 python -m pytest tests/test_gpu_kmer.py "tests/test_gpu_fullsize.py::test_kmer_read_profiles_whole_population" -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r06_call32_tests.log
