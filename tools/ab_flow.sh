# level launches vs the dataflow launch of the tile factorisation, per workload: bash tools/ab_flow.sh [env settings ...]
for wl in ${FLOW_WORKLOADS:-bar17K_twist bunny5K_LTSS monkey18K_stiff synbar:140x35x35:256}; do
  steps=20; [ "${wl#synbar}" != "$wl" ] && steps=6
  for f in 0 1; do
    env DOTMI_TILE_FLOW=$f "$@" timeout -s KILL 300 python tools/run_case.py $wl - $steps 2>&1 | awk -v w=$wl -v f=$f '/^[0-9]+ iters/{n++; if(n>2){it+=$3; ms+=$7; lp+=$9; fa+=$13; c++}} END{printf "%s flow=%s steps %d iters %.2f ms %.3f loop %.3f factor %.3f\n", w, f, c, it/c, ms/c, lp/c, fa/c}'
  done
done
