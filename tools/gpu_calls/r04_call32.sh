# call 32: is the range path taken, and are stdout and stderr those of one rank?  (the last seconds of the budget)
cd "$GRAFT_REPO_ROOT"
make -s -C tests/shim
python - <<'PY'
import sys; sys.path.insert(0, "tests")
import _cases
open("/tmp/c1.fastq", "wb").write(_cases.c1_fastq_bytes())
PY
export FLX_RCCL_LIB=$PWD/tests/shim/libloopback_rccl.so FLX_DEVICE=0 LANG=C
B=filtlong_amd/bin/filtlong
$B --target_bases 20000000 /tmp/c1.fastq > /tmp/one.out 2> /tmp/one.err; echo "one rank rc $?"
FLX_CLI_RANK_RANGES=1 FLX_CLI_TIMING=1 $B --gpus 3 --target_bases 20000000 /tmp/c1.fastq 2>&1 >/dev/null | grep "rank ranges"
FLX_CLI_RANK_RANGES=1 $B --gpus 3 --target_bases 20000000 /tmp/c1.fastq > /tmp/three.out 2> /tmp/three.err; echo "three ranks rc $?"

cmp /tmp/one.out /tmp/three.out && echo "stdout identical ($(wc -c < /tmp/one.out) bytes)"
cmp /tmp/three.err /tmp/one.err && echo "stderr identical ($(wc -c < /tmp/one.err) bytes)"
