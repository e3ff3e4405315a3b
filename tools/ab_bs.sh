# back-solve launch time of a workload under several environment settings: bash tools/ab_bs.sh <workload> "VAR=a" ... ("-" = defaults)
wl=$1; shift
for e in "$@"; do
  [ "$e" = "-" ] && e=""
  env $e timeout -s KILL 300 python tools/run_case.py $wl - 12 2>&1 | awk -v w=$wl -v e="$e" '/^[0-9]+ iters/{n++; if(n>2){it+=$3; ms+=$7; lp+=$9; c++}} /^backsolve/{bs=$0} END{printf "%s [%s] iters %.2f ms %.3f loop %.3f | %s\n", w, e, it/c, ms/c, lp/c, bs}'
done
