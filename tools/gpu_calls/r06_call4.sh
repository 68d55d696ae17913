cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out filtlong_amd/lib/exp
export FLX_VARIANT_SRC=cover_queue
{
bash tools/build_variant.sh occ7 -DFLX_COVER_WAVES_PER_EU=7
bash tools/build_variant.sh occ6 -DFLX_COVER_WAVES_PER_EU=6
bash tools/build_variant.sh occ9 -DFLX_COVER_WAVES_PER_EU=9
bash tools/build_variant.sh occ10 -DFLX_COVER_WAVES_PER_EU=10
bash tools/build_variant.sh t128 -DFLX_COVER_THREADS=128
bash tools/build_variant.sh t64 -DFLX_COVER_THREADS=64
bash tools/build_variant.sh lag1 -DFLX_COVER_LAG=1
bash tools/build_variant.sh ring16 -DFLX_COVER_RING=16 -DFLX_COVER_LAG=12
bash tools/build_variant.sh noseed -DFLX_LOCUS_SEEDS=0
} 2>&1 | grep -v "^$" | tail -12
for v in "" occ7 occ6 occ9 occ10 t128 t64 lag1 ring16 noseed; do
  echo "== variant ${v:-default}"
  if [ -n "$v" ]; then export FLX_LIB_PATH=$PWD/filtlong_amd/lib/exp/libfiltlong_hip_$v.so; else unset FLX_LIB_PATH; fi
  timeout 300 python tools/bench_kmer.py --reads 2000000 --steps 3 2>&1 | grep -v "Warning\|amdgpu.ids"
done | tee gpurun_out/r06_call4.log
