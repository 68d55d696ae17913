#!/bin/bash
# C4 end to end through the command line (round-4 review, item 4): -1/-2 short-read reference streamed into the device set,
# --trim --split 500, filtlong-amd vs the reference binary on the same files; then a short-read set of >= 10 GB of bases
# (filtlong-amd alone: the reference would hash for hours) with the process's peak resident memory.
# usage: tools/bench_e2e_kmer.sh [pairs=1000000] [big_pairs=50000000] [prefix=r05]  -> gpurun_out/${PFX}_e2e_kmer.json / .log
R=${GRAFT_REPO_ROOT:-$PWD}
PAIRS=${1:-1000000}
BIG=${2:-50000000}
PFX=${3:-r05}
OUT=$R/gpurun_out
mkdir -p $OUT /tmp/e2ek
export LANG=C LC_ALL=C
B=$R/filtlong_amd/bin/filtlong
{
echo "host cores $(nproc)"; free -g | head -2
read LB SB < <($R/tools/gen_kmer_inputs /tmp/e2ek 3000 $PAIRS)
T=$((LB / 2))
ARGS="-1 /tmp/e2ek/sr_1.fastq -2 /tmp/e2ek/sr_2.fastq --trim --split 500 --target_bases $T /tmp/e2ek/reads.fastq"
echo "C4 pairs: $PAIRS pairs ($SB bases), 3000 long reads ($LB bases)"
for rep in 1 2; do
  python3 $R/tools/timed.py /tmp/e2ek/amd.time $B $ARGS > /tmp/e2ek/amd.out 2> /tmp/e2ek/amd.err
  echo "filtlong-amd run $rep: $(cat /tmp/e2ek/amd.time)"
done
AMD_S=$(cut -d' ' -f1 /tmp/e2ek/amd.time); AMD_RSS=$(sed -E 's/.*wall, ([0-9]+) KiB.*/\1/' /tmp/e2ek/amd.time)
FLX_CLI_TIMING=1 $B $ARGS 2>&1 >/dev/null | tr '\r' '\n' | grep timing
python3 $R/tools/timed.py /tmp/e2ek/ref.time $R/oracle/_ref/filtlong $ARGS > /tmp/e2ek/ref.out 2> /tmp/e2ek/ref.err
echo "reference: $(cat /tmp/e2ek/ref.time)"
REF_S=$(cut -d' ' -f1 /tmp/e2ek/ref.time); REF_RSS=$(sed -E 's/.*wall, ([0-9]+) KiB.*/\1/' /tmp/e2ek/ref.time)
if cmp /tmp/e2ek/ref.out /tmp/e2ek/amd.out; then IDENT=true; echo "stdout identical ($(stat -c %s /tmp/e2ek/amd.out) bytes)"; else IDENT=false; echo "STDOUT DIFFERS"; fi
if cmp /tmp/e2ek/ref.err /tmp/e2ek/amd.err; then ERRID=true; echo "stderr identical byte for byte ($(stat -c %s /tmp/e2ek/amd.err) bytes, $(tr -cd '\r' < /tmp/e2ek/amd.err | wc -c) progress updates)"; else ERRID=false; echo "STDERR DIFFERS"; fi
# ---- a short-read set of >= 10 GB of bases ----
read LB2 SB2 < <($R/tools/gen_kmer_inputs /tmp/e2ek 3000 $BIG)
echo "big set: $BIG pairs ($SB2 bases; files $(stat -c %s /tmp/e2ek/sr_1.fastq) + $(stat -c %s /tmp/e2ek/sr_2.fastq) bytes)"
python3 $R/tools/timed.py /tmp/e2ek/big.time env FLX_CLI_TIMING=1 $B $ARGS > /tmp/e2ek/big.out 2> /tmp/e2ek/big.err
BIG_RC=$?
echo "filtlong-amd, big set: rc $BIG_RC, $(cat /tmp/e2ek/big.time)"
tr '\r' '\n' < /tmp/e2ek/big.err | grep -E "timing|16-mers|Error" | tail -12
BIG_S=$(cut -d' ' -f1 /tmp/e2ek/big.time); BIG_RSS=$(sed -E 's/.*wall, ([0-9]+) KiB.*/\1/' /tmp/e2ek/big.time)
} > $OUT/${PFX}_e2e_kmer.log 2>&1
python - <<PY
import json
json.dump({"c4_pairs": {"short_read_pairs": $PAIRS, "short_read_bases": $SB, "long_reads": 3000, "long_read_bases": $LB,
                        "flags": "-1 sr_1.fastq -2 sr_2.fastq --trim --split 500 --target_bases <half>",
                        "filtlong_amd_s": float("$AMD_S"), "filtlong_amd_peak_rss_mib": int("$AMD_RSS") // 1024,
                        "reference_s": float("$REF_S"), "reference_peak_rss_mib": int("$REF_RSS") // 1024,
                        "speedup": float("$REF_S") / float("$AMD_S"), "stdout_identical": "$IDENT" == "true", "stderr_identical_raw_bytes": "$ERRID" == "true"},
           "big_short_read_set": {"short_read_pairs": $BIG, "short_read_bases": $SB2, "exit_code": $BIG_RC, "filtlong_amd_s": float("$BIG_S"),
                                  "filtlong_amd_peak_rss_mib": int("$BIG_RSS") // 1024,
                                  "note": "the two FASTQ files are streamed in blocks of 256 MiB and handed to the device set in batches of 256 MiB "
                                          "(cli/reference.h): peak host memory is O(block + batch), not O(file)"}},
          open("$OUT/${PFX}_e2e_kmer.json", "w"), indent=1)
PY
tail -40 $OUT/${PFX}_e2e_kmer.log; cat $OUT/${PFX}_e2e_kmer.json
rm -rf /tmp/e2ek
