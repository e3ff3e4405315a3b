# per-step stats of a workload under several environment settings, same box:
#   bash tools/ab_env.sh <workload> <steps> "VAR=a VAR2=b" "VAR=c" ...     ("-" = defaults)
wl=$1; steps=$2; shift; shift
for e in "$@"; do
  [ "$e" = "-" ] && e=""
  env $e timeout -s KILL 600 python tools/run_case.py $wl - $steps 2>&1 | awk -v w=$wl -v e="$e" '/^[0-9]+ iters/{n++; if(n>2){it+=$3; ms+=$7; lp+=$9; fa+=$13; c++}} END{printf "%s [%s] steps %d iters %.2f ms %.3f loop %.3f factor %.3f\n", w, e, c, it/c, ms/c, lp/c, fa/c}'
done
