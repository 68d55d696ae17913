#!/bin/bash
# End to end on a gzip-compressed FASTQ (the usual form of real inputs): filtlong-amd streams it block by block, the
# in-memory path (FLX_CLI_NO_STREAM=1) and the reference binary beside it, same box, same file.
# usage: tools/bench_e2e_gz.sh [n_reads=150000] [prefix=r03]   -> gpurun_out/<prefix>_e2e_gz.json, gpurun_out/<prefix>_e2e_gz.log
R=${GRAFT_REPO_ROOT:-$PWD}
N=${1:-150000}
PFX=${2:-r03}
OUT=$R/gpurun_out
mkdir -p $OUT
export LANG=C LC_ALL=C
cd /tmp
BASES=$($R/tools/gen_fastq $N /tmp/gzin.fastq)
RAW=$(stat -c %s /tmp/gzin.fastq)
S=$(date +%s%N); gzip -1 -f /tmp/gzin.fastq; E=$(date +%s%N)
SIZE=$(stat -c %s /tmp/gzin.fastq.gz)
TARGET=$((BASES / 2))
t() { python -c "print(($2 - $1) / 1e9)"; }
{
echo "file /tmp/gzin.fastq.gz: $N reads, $BASES bases, $RAW bytes of FASTQ, $SIZE bytes compressed (gzip -1: $(t $S $E) s); --target_bases $TARGET"
S=$(date +%s%N); gzip -dc /tmp/gzin.fastq.gz > /dev/null; E=$(date +%s%N)
INFLATE_S=$(t $S $E)
echo "gzip -dc alone: $INFLATE_S s"
for rep in 1 2; do
  rm -f /tmp/amd.out
  S=$(date +%s%N)
  FLX_CLI_TIMING=1 $R/filtlong_amd/bin/filtlong --target_bases $TARGET /tmp/gzin.fastq.gz > /tmp/amd.out 2> /tmp/amd.err
  E=$(date +%s%N)
  AMD_S=$(t $S $E)
  echo "filtlong-amd (streamed) run $rep: $AMD_S s"
done
tr '\r' '\n' < /tmp/amd.err | grep timing
for TH in 1 16 32 64; do
  rm -f /tmp/amd.out
  S=$(date +%s%N); FLX_CLI_INFLATE_THREADS=$TH $R/filtlong_amd/bin/filtlong --target_bases $TARGET /tmp/gzin.fastq.gz > /tmp/amd.out 2> /dev/null; E=$(date +%s%N)
  echo "filtlong-amd (streamed), $TH inflate thread(s) in pass 1: $(t $S $E) s"
done
rm -f /tmp/amd.out
FLX_CLI_TIMING=1 $R/filtlong_amd/bin/filtlong --target_bases $TARGET /tmp/gzin.fastq.gz > /tmp/amd.out 2> /tmp/amd.err
ANON=$(tr '\r' '\n' < /tmp/amd.err | grep timing | sed -E 's/.*RssAnon +([0-9]+) MiB.*/\1/' | sort -n | tail -1)
S=$(date +%s%N)
FLX_CLI_NO_STREAM=1 FLX_CLI_TIMING=1 $R/filtlong_amd/bin/filtlong --target_bases $TARGET /tmp/gzin.fastq.gz > /tmp/amd2.out 2> /tmp/amd2.err
E=$(date +%s%N)
MEM_S=$(t $S $E)
echo "filtlong-amd (inflated into memory): $MEM_S s"
tr '\r' '\n' < /tmp/amd2.err | grep timing
ANON2=$(tr '\r' '\n' < /tmp/amd2.err | grep timing | sed -E 's/.*RssAnon +([0-9]+) MiB.*/\1/' | sort -n | tail -1)
S=$(date +%s%N)
$R/oracle/_ref/filtlong --target_bases $TARGET /tmp/gzin.fastq.gz > /tmp/ref.out 2> /tmp/ref.err
E=$(date +%s%N)
REF_S=$(t $S $E)
echo "reference: $REF_S s"
if cmp /tmp/ref.out /tmp/amd.out && cmp /tmp/ref.out /tmp/amd2.out; then IDENT=true; echo "stdout identical ($(stat -c %s /tmp/amd.out) bytes)"; else IDENT=false; echo "STDOUT DIFFERS"; fi
grep -E "target|keeping" /tmp/ref.err /tmp/amd.err
} > $OUT/${PFX}_e2e_gz.log 2>&1
python - <<PY
import json
amd, mem, ref = float("$AMD_S"), float("$MEM_S"), float("$REF_S")
json.dump({"reads": $N, "bases": $BASES, "fastq_bytes": $RAW, "gz_bytes": $SIZE, "target_bases": $TARGET,
           "gzip_dc_alone_s": float("$INFLATE_S"), "filtlong_amd_streamed_s": amd, "filtlong_amd_in_memory_s": mem, "reference_s": ref,
           "speedup": ref / amd, "stdout_identical": "$IDENT" == "true",
           "peak_rss_anon_mib_streamed": int("$ANON" or 0), "peak_rss_anon_mib_in_memory": int("$ANON2" or 0),
           "note": "pass 1 inflates block-parallel (cli/pinflate.h: block starts found by search, markers for the unknown window, zlib from the last boundary as fallback); the output pass inflates the record-aligned pieces between the access points pass 1 left; the reference inflates the file twice on one thread"},
          open("$OUT/${PFX}_e2e_gz.json", "w"), indent=1)
PY
tail -40 $OUT/${PFX}_e2e_gz.log; cat $OUT/${PFX}_e2e_gz.json
rm -f /tmp/gzin.fastq.gz /tmp/amd.out /tmp/amd2.out /tmp/ref.out
