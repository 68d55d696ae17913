cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for prof in 0 1; do
  echo "== C3 profile $prof q"; timeout 600 python tools/bench_kmer.py --reads 2000000 --steps 3 --profile $prof
done
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r06_call9.log
export TMPDIR=/tmp; R=$PWD; cd /tmp
B="python $R/tools/bench_kmer.py --reads 1000000 --steps 1 --profile 1"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p1_k -o p -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD -d $R/gpurun_out/p1_s -o p -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $R/gpurun_out/p1_t -o p -- $B > /dev/null 2>&1
for d in k s t; do python $R/tools/rocprof_summary.py $R/gpurun_out/p1_$d/p_results.db kmer > $R/gpurun_out/p1_$d.txt 2>&1; rm -rf $R/gpurun_out/p1_$d; done
