cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out filtlong_amd/lib/exp
F="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $F -Ifiltlong_amd/csrc -Iinclude -c -o filtlong_amd/lib/exp/cq_old.o tools/exp/cover_queue_head.hip
objs=$(ls filtlong_amd/lib/obj/*.o | grep -v "/cover_queue.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o filtlong_amd/lib/exp/libfiltlong_hip_head.so $objs filtlong_amd/lib/exp/cq_old.o -ldl -lpthread
for rep in 1 2; do
for v in "" head; do
  if [ -n "$v" ]; then export FLX_LIB_PATH=$PWD/filtlong_amd/lib/exp/libfiltlong_hip_$v.so; else unset FLX_LIB_PATH; fi
  echo "== variant ${v:-new} profile 1 rep $rep"; timeout 300 python tools/bench_kmer.py --reads 2000000 --steps 3 --profile 1 2>&1 | grep -v "Warning\|amdgpu.ids"
done; done | tee gpurun_out/r06_call36.log
unset FLX_LIB_PATH
timeout 1500 python -m pytest tests/test_gpu_kmer.py "tests/test_gpu_fullsize.py::test_kmer_read_profiles_whole_population" -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r06_call36_tests.log
