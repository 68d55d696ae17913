#!/bin/bash
# PMC passes of the k-mer cover kernel: far requests (FETCH_SIZE), L2 hit rate, issue mix.  usage: prof_kmer.sh [reads] [configs] [light]
# (light: FETCH_SIZE, TCC and kernel stats only — the passes the request model is made of)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_kmer
mkdir -p $OUT
cd /tmp
N=${1:-1000000}
CFGS=${2:-"c3 c4"}
LIGHT=${3:-}
for cfg in $CFGS; do
# (c3_indels / c3_unrelated: C3 on the read profiles of round 6 — filtlong_amd/synth.py: seq_read, profile 1 / 2)
case $cfg in
  *_indels) BCFG="${cfg%_indels} --read-profile 1" ;;
  *_unrelated) BCFG="${cfg%_unrelated} --read-profile 2" ;;
  *) BCFG=$cfg ;;
esac
B="python $R/bench.py --config $BCFG --reads $N --steps 1 --warmup 0 --no-cpu-baseline"
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/f_$cfg -o p -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $OUT/t_$cfg -o p -- $B > /dev/null 2>&1
if [ -z "$LIGHT" ]; then
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD -d $OUT/s_$cfg -o p -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -d $OUT/u_$cfg -o p -- $B > /dev/null 2>&1
fi
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/k_$cfg -o p -- $B > /dev/null 2>&1
for d in f t s u k; do [ -d $OUT/${d}_$cfg ] || continue; python $R/tools/rocprof_summary.py $OUT/${d}_$cfg/p_results.db kmer > $OUT/${d}_$cfg.txt 2>&1; rm -rf $OUT/${d}_$cfg; done
done
cat $OUT/*.txt | grep -v "^$"
