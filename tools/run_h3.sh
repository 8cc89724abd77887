cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for sh in 14 20 21; do
for cfg in "36 4608 512 512" "36 8960 256 256" "36 4608 512 256" "5 18176 512 1024"; do
timeout -k 5 60 tools/h3_gemm_bench $cfg 30 $sh || exit 1
done; done; done
for v in _NODMA _NOSTORE; do echo "== $v"; timeout -k 5 60 tools/h3_gemm_bench$v 36 4608 512 512 30 21;  timeout -k 5 60 tools/h3_gemm_bench$v 36 8960 256 256 30 21; done
