cd $GRAFT_REPO_ROOT
for n in "1024 4" "2048 4" "4096 8" "8192 8" "16384 16"; do
  timeout -k 10 120 python tools/fit_only.py $n 10 2>/dev/null | cut -c1-280
done
