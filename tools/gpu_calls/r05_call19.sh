# round 5, call 19 (the last seconds of the budget): a gzip input with 8 forked ranks — streamed by every rank (default) against inflated whole
# into every rank's memory (FLX_CLI_RANK_STREAM=0): wall clock and the summed peak resident set of the job's processes
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export LANG=C LC_ALL=C
make -s -C tests/shim
export FLX_RCCL_LIB=$PWD/tests/shim/libloopback_rccl.so FLX_DEVICE=0
B=$PWD/filtlong_amd/bin/filtlong
{
BASES=$(tools/gen_fastq 60000 /tmp/g8.fastq); T=$((BASES / 2))
gzip -1 -c /tmp/g8.fastq > /tmp/g8.fastq.gz; echo "input: 60000 reads, $BASES bases, $(stat -c %s /tmp/g8.fastq) bytes of FASTQ, $(stat -c %s /tmp/g8.fastq.gz) bytes of gzip"
$B --target_bases $T /tmp/g8.fastq.gz > /tmp/g8_one.out 2>/dev/null; ONE=$(sha256sum < /tmp/g8_one.out | cut -c1-16)
for MODE in 1 0; do
  for rep in 1 2; do
    S=$(date +%s%N)
    FLX_CLI_RANK_STREAM=$MODE $B --gpus 8 --target_bases $T /tmp/g8.fastq.gz > /tmp/g8_r.out 2>/dev/null &
    PID=$!
    PEAK=0
    while kill -0 $PID 2>/dev/null; do
      R=$(ps -o rss= --ppid $PID -p $PID 2>/dev/null | awk '{s+=$1} END{print s+0}'); [ "$R" -gt "$PEAK" ] && PEAK=$R; sleep 0.05
    done
    wait $PID; RC=$?
    E=$(date +%s%N)
    echo "FLX_CLI_RANK_STREAM=$MODE run $rep: rc $RC, $(python3 -c "print(($E - $S) / 1e9)") s, summed resident set of the 8 ranks at its highest sample $((PEAK / 1024)) MiB, stdout identical to one rank: $([ "$(sha256sum < /tmp/g8_r.out | cut -c1-16)" = "$ONE" ] && echo yes || echo NO)"
  done
done
} 2>&1 | tee gpurun_out/r05_gz_ranks.log
rm -f /tmp/g8*
