# super-strip width of the three-level Cholesky (ABO_CHOL_SUPER; 0 = the two-level schedule of rounds 1-2) x strip width
cd $GRAFT_REPO_ROOT
for ss in 0 1024 2048 4096; do
  for n in "2048 4" "4096 8" "8192 8" "16384 16"; do
    echo -n "ABO_CHOL_SUPER=$ss  "
    ABO_CHOL_SUPER=$ss timeout -k 10 120 python tools/fit_only.py $n 10 2>/dev/null | python3 -c "
import sys,re
l=sys.stdin.read()
m=re.search(r'(N=\d+ d=\d+: [0-9.]+ ms per fit)', l); c=re.search(r\"'fit_cholesky_ms': ([0-9.]+)\", l); i=re.search(r\"'fit_inverse_ms': ([0-9.]+)\", l)
print(m.group(1) if m else l[:100], 'cholesky', c.group(1) if c else '?', 'inverse', i.group(1) if i else '?')"
  done
done
