#!/bin/bash
# experimental build of libfiltlong_hip.so with extra -D flags for one translation unit (FLX_VARIANT_SRC, default: the part of
# score_phred_regs.hip that holds ws 192..255): tools/build_variant.sh name -DFOO ...
# (select it with FLX_LIB_PATH=filtlong_amd/lib/exp/libfiltlong_hip_<name>.so)
set -e
name=$1; shift
cd "$(dirname "$0")/.."
F="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wall -Wno-unused-function"
SRC=${FLX_VARIANT_SRC:-score_phred_regs_p2}   # object to replace (score_phred_regs_p<k> = part k of score_phred_regs.hip, or e.g. score_kmer)
case $SRC in
  score_phred_regs_p*) /opt/rocm/bin/hipcc $F -DFLX_REGS_PART=${SRC##*_p} "$@" -c -o filtlong_amd/lib/exp/${SRC}_$name.o filtlong_amd/csrc/score_phred_regs.hip ;;
  *) /opt/rocm/bin/hipcc $F "$@" -c -o filtlong_amd/lib/exp/${SRC}_$name.o filtlong_amd/csrc/$SRC.hip ;;
esac
objs=$(ls filtlong_amd/lib/obj/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o filtlong_amd/lib/exp/libfiltlong_hip_$name.so $objs filtlong_amd/lib/exp/${SRC}_$name.o -ldl -lpthread
echo built filtlong_amd/lib/exp/libfiltlong_hip_$name.so
