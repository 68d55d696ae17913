# call 26: issue mix of the cover kernel as it stands (is it the VALU now?) — all PMC sets at 1e6 reads, C3
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_kmer
bash tools/prof_kmer.sh 1000000 "c3" > gpurun_out/c26_prof.out 2>&1
grep -E "cover" gpurun_out/c26_prof.out | cut -c1-200
