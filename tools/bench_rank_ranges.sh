#!/bin/bash
# N ranks, one 20 GB FASTQ: every rank indexes only its byte range of the mapped file (cli/fastx.h: parse_rank_range; the default
# since round 5) against every rank parsing the whole file (FLX_CLI_RANK_RANGES=0).  Ranks forked by --gpus N on ONE GPU over the
# loopback communicator (tests/shim): what is measured is the host side — parse and record checks per rank — not the exchange.
# usage: tools/bench_rank_ranges.sh [n_reads=2000000] [ranks=8] [prefix=r05]  -> gpurun_out/${PFX}_rank_ranges.log / .json
R=${GRAFT_REPO_ROOT:-$PWD}
N=${1:-2000000}
W=${2:-8}
PFX=${3:-r05}
OUT=$R/gpurun_out
mkdir -p $OUT
export LANG=C LC_ALL=C
make -s -C $R/tests/shim
export FLX_RCCL_LIB=$R/tests/shim/libloopback_rccl.so FLX_DEVICE=0
B=$R/filtlong_amd/bin/filtlong
{
BASES=$($R/tools/gen_fastq $N /tmp/rr.fastq)
SIZE=$(stat -c %s /tmp/rr.fastq)
T=$((BASES / 2))
echo "file: $N reads, $BASES bases, $SIZE bytes; $W ranks; host cores $(nproc)"
$B --target_bases $T /tmp/rr.fastq > /tmp/rr_one.out 2> /tmp/rr_one.err; echo "one rank: rc $?"
ONE=$(sha256sum < /tmp/rr_one.out | cut -c1-16); rm -f /tmp/rr_one.out  # (the outputs are half the input each: /tmp does not hold three of them beside it)
for MODE in 1 0; do
  for rep in 1 2; do
    S=$(date +%s%N)
    FLX_CLI_RANK_RANGES=$MODE FLX_CLI_TIMING=1 $B --gpus $W --target_bases $T /tmp/rr.fastq > /tmp/rr_$MODE.out 2> /tmp/rr_$MODE.err
    RC=$?
    E=$(date +%s%N)
    echo "FLX_CLI_RANK_RANGES=$MODE run $rep: rc $RC, $(python3 -c "print(($E - $S) / 1e9)") s; stdout identical to one rank: $([ "$(sha256sum < /tmp/rr_$MODE.out | cut -c1-16)" = "$ONE" ] && echo yes || echo NO)"
    rm -f /tmp/rr_$MODE.out
  done
  tr '\r' '\n' < /tmp/rr_$MODE.err | grep -E "\[timing\] (parse|record checks|rank ranges|read input file|pack)" | sed 's/  RssAnon.*//'
done
} > $OUT/${PFX}_rank_ranges.log 2>&1
python3 - <<PY
import json, re
log = open("$OUT/${PFX}_rank_ranges.log").read()
def runs(mode):
    return [float(x) for x in re.findall(r"FLX_CLI_RANK_RANGES=%d run \d: rc 0, ([0-9.]+) s" % mode, log)]
def stage(mode, name):
    blk = log.split("FLX_CLI_RANK_RANGES=%d run 2" % mode)[1]
    m = re.search(r"\[timing\] %s\s+([0-9.]+) s" % name, blk)
    return float(m.group(1)) if m else None
m = re.search(r"rank ranges: (\d+) of (\d+) records indexed here", log)
json.dump({"reads": $N, "ranks": $W, "fastq_bytes": int(re.search(r"(\d+) bytes", log).group(1)),
           "ranges_seconds": runs(1), "whole_file_seconds": runs(0),
           "rank0_parse_s": {"ranges": stage(1, "parse"), "whole_file": stage(0, "parse")},
           "rank0_record_checks_s": {"ranges": stage(1, "record checks"), "whole_file": stage(0, "record checks")},
           "records_indexed_by_rank0": [int(m.group(1)), int(m.group(2))] if m else None,
           "stdout_identical": "NO" not in log,
           "note": "ranks forked on one GPU over the loopback communicator: the host side of N-rank ingest (rank 0's stage clocks)"},
          open("$OUT/${PFX}_rank_ranges.json", "w"), indent=1)
PY
cat $OUT/${PFX}_rank_ranges.log; cat $OUT/${PFX}_rank_ranges.json
rm -f /tmp/rr.fastq /tmp/rr_*.out
