#!/bin/bash
# End to end on a FASTQ larger than 20 GB: filtlong-amd (streaming ingest) vs the reference binary, same box, same file.
# usage: tools/bench_e2e_big.sh [n_reads=1000000] [prefix=r03]   -> gpurun_out/${PFX}_e2e_big.json, gpurun_out/${PFX}_e2e_big.log
R=${GRAFT_REPO_ROOT:-$PWD}
N=${1:-1000000}
PFX=${2:-r03}
OUT=$R/gpurun_out
mkdir -p $OUT
export LANG=C LC_ALL=C
cd /tmp
BASES=$($R/tools/gen_fastq $N /tmp/big.fastq)
SIZE=$(stat -c %s /tmp/big.fastq)
TARGET=$((BASES / 2))
{
echo "file /tmp/big.fastq: $N reads, $BASES bases, $SIZE bytes; --target_bases $TARGET; host cores $(nproc)"
free -g | head -2
for rep in 1 2; do
  rm -f /tmp/amd.out; sync  # the shell would truncate the previous 10 GB output inside the timed window otherwise (1-2 s)
  S=$(date +%s%N)
  FLX_CLI_TIMING=1 $R/filtlong_amd/bin/filtlong --target_bases $TARGET /tmp/big.fastq > /tmp/amd.out 2> /tmp/amd.err
  E=$(date +%s%N)
  AMD_S=$(python -c "print(($E - $S) / 1e9)")
  echo "filtlong-amd run $rep: $AMD_S s"
done
tr '\r' '\n' < /tmp/amd.err | grep timing
for T in 8 32 64; do
  rm -f /tmp/amd_t.out
  S=$(date +%s%N)
  FLX_CLI_THREADS=$T FLX_CLI_TIMING=1 $R/filtlong_amd/bin/filtlong --target_bases $TARGET /tmp/big.fastq > /tmp/amd_t.out 2> /tmp/amd_t.err
  E=$(date +%s%N)
  echo "filtlong-amd with FLX_CLI_THREADS=$T: $(python -c "print(($E - $S) / 1e9)") s;$(tr '\r' '\n' < /tmp/amd_t.err | grep -E "pack|output" | sed -E 's/\[timing\] ([a-z +A-Z0-9()]+[a-z)]) +([0-9.]+) s.*/ \1 \2 s;/' | tr '\n' ' ')"
done
S=$(date +%s%N)
FLX_CLI_TIMING=1 $R/filtlong_amd/bin/filtlong --target_bases $TARGET /tmp/big.fastq 2> /tmp/amd_t.err | cat > /tmp/amd_pipe.out
E=$(date +%s%N)
echo "filtlong-amd into a pipe (ordered writes): $(python -c "print(($E - $S) / 1e9)") s; identical: $(cmp /tmp/amd_pipe.out /tmp/amd.out && echo yes)"
rm -f /tmp/amd_t.out /tmp/amd_pipe.out
S=$(date +%s%N)
$R/oracle/_ref/filtlong --target_bases $TARGET /tmp/big.fastq > /tmp/ref.out 2> /tmp/ref.err
E=$(date +%s%N)
REF_S=$(python -c "print(($E - $S) / 1e9)")
echo "reference: $REF_S s"
if cmp /tmp/ref.out /tmp/amd.out; then IDENT=true; echo "stdout identical ($(stat -c %s /tmp/amd.out) bytes)"; else IDENT=false; echo "STDOUT DIFFERS"; fi
grep -E "target|keeping" /tmp/ref.err /tmp/amd.err
} > $OUT/${PFX}_e2e_big.log 2>&1
ANON=$(tr '\r' '\n' < /tmp/amd.err | grep timing | sed -E 's/.*RssAnon +([0-9]+) MiB.*/\1/' | sort -n | tail -1)
python - <<PY
import json
amd, ref = float("$AMD_S"), float("$REF_S")
json.dump({"reads": $N, "bases": $BASES, "fastq_bytes": $SIZE, "target_bases": $TARGET, "filtlong_amd_s": amd, "reference_s": ref,
           "speedup": ref / amd, "e2e_gbases_per_s": $BASES / amd / 1e9, "stdout_identical": "$IDENT" == "true",
           "peak_rss_anon_mib_at_stage_ends": int("$ANON" or 0),
           "note": "file -> stdout(file), page cache warm on the second run; the streamed H2D moves 1 byte per base over PCIe"},
          open("$OUT/${PFX}_e2e_big.json", "w"), indent=1)
PY
cat $OUT/${PFX}_e2e_big.log | tail -30; cat $OUT/${PFX}_e2e_big.json
rm -f /tmp/big.fastq /tmp/amd.out /tmp/ref.out
