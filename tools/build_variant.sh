#!/bin/bash
# experimental build of libfiltlong_hip.so with extra -D flags for score_phred_regs.hip: tools/build_variant.sh name -DFOO ...
# (select it with FLX_LIB_PATH=filtlong_amd/lib/exp/libfiltlong_hip_<name>.so)
set -e
name=$1; shift
cd "$(dirname "$0")/.."
F="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $F "$@" -c -o filtlong_amd/lib/exp/regs_$name.o filtlong_amd/csrc/score_phred_regs.hip
objs=$(ls filtlong_amd/lib/obj/*.o | grep -v score_phred_regs.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o filtlong_amd/lib/exp/libfiltlong_hip_$name.so $objs filtlong_amd/lib/exp/regs_$name.o
echo built filtlong_amd/lib/exp/libfiltlong_hip_$name.so
