# configs[4] at half scale, four processes one after the other on the same box: is the first process on a fresh box slow whatever it runs?
for v in "X=1" "X=2" "FZP_TB_SERIAL=1" "X=3"; do
  echo "== $v"
  python3 tools/runs/cfg5_prof.py $v 2>&1 | tail -1
done
