# fit time A/B: tools/fit_ab.sh ENVVAR "v1 v2"   (e.g. ABO_GEMM256 "0 1")
cd $GRAFT_REPO_ROOT
VAR=$1; VALS=${2:-"0 1"}
for n in "1024 4" "2048 4" "4096 8" "8192 8" "16384 16"; do
  for v in $VALS; do
    env $VAR=$v timeout -k 10 120 python tools/fit_only.py $n 10 2>/dev/null | cut -c1-300 | sed "s/^/$VAR=$v /"
  done
done
