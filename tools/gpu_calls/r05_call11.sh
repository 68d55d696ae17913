cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export LANG=C LC_ALL=C
make -s -C tests/shim
export FLX_RCCL_LIB=$PWD/tests/shim/libloopback_rccl.so FLX_DEVICE=0
B=filtlong_amd/bin/filtlong
BASES=$(tools/gen_fastq 2000000 /tmp/rr.fastq); T=$((BASES / 2))
for i in 1 2; do
FLX_CLI_RANK_RANGES=0 FLX_CLI_TIMING=1 $B --gpus 8 --target_bases $T /tmp/rr.fastq > /tmp/rr_0.out 2> /tmp/rr_0.err; echo "rc $?"
tr '\r' '\n' < /tmp/rr_0.err | grep -v "^\[timing\]" | tail -6
done
